#!/usr/bin/env python3
"""Algorithm search driver (experiment; CPU only).  Evaluates members of the min-sum family of nms_family.c
against flooding sum-product (oracle) on identical noise.  Results are cached under /tmp/nms_search."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

L = C.CDLL(os.path.join(HERE, "libnmsf.so"))


class P(C.Structure):
    _fields_ = [("alpha", C.c_float * 46), ("c0", C.c_float), ("c1", C.c_float), ("beta", C.c_float),
                ("scale", C.c_int), ("msg_max", C.c_int), ("app_max", C.c_int), ("fp8", C.c_int)]


L.nmsf_decode.argtypes = [C.c_int] * 4 + [C.POINTER(P), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]

CASES = {  # name: bg, Z, Kp, E, n_layers, iters, snrs
    "headline": (1, 384, 8448, 25272, 46, 25, [-1.7, -1.6, -1.5, -1.4, -1.3, -1.2, -1.1, -1.0]),
    "bg2_r13": (2, 384, 3840, 11472, 22, 25, [-1.5, -1.4, -1.3, -1.2, -1.1, -1.0, -0.9, -0.8]),
    "bg2_r15": (2, 384, 3840, 19120, 42, 25, [-4.2, -4.1, -4.0, -3.9, -3.8, -3.7]),
    "bg2_r23": (2, 384, 3840, 5736, 7, 25, [2.4, 2.6, 2.8, 3.0, 3.2, 3.4]),
    "bg1_r89": (1, 384, 8448, 9478, 5, 25, [5.8, 6.0, 6.2, 6.4, 6.6]),
    "cfg1": (2, 20, 116, 300, 12, 10, [0.0, 1.0, 2.0, 3.0, 4.0]),
    "bg1_r12": (1, 384, 8448, 16896, 24, 25, [0.6, 0.8, 1.0, 1.2]),
    "bg1_r23": (1, 384, 8448, 12672, 13, 25, [2.6, 2.8, 3.0, 3.2]),
    "bg2_r12": (2, 384, 3840, 7680, 12, 25, [0.8, 1.0, 1.2, 1.4]),
    "bg2_r14": (2, 384, 3840, 15296, 32, 25, [-3.0, -2.8, -2.6, -2.4]),
    "bg1_z96": (1, 96, 2112, 6336, 46, 25, [-1.2, -1.0, -0.8, -0.6]),
    "bg2_z36": (2, 36, 360, 1080, 22, 25, [-0.5, 0.0, 0.5, 1.0]),
}
CACHE = "/tmp/nms_search"
os.makedirs(CACHE, exist_ok=True)


def make_llr(case, snr, nblk, seed=2026):
    bg, Z, Kp, E, nl, iters, _ = CASES[case]
    rows, cols, kb = O.BG_DIMS[bg]
    K = kb * Z
    rng = np.random.default_rng(seed)
    info = rng.integers(0, 2, (nblk, K), dtype=np.uint8)
    info[:, Kp:] = 0
    cw = O.encode(bg, Z, info)
    noise = rng.standard_normal(cw.shape)
    mu = 2 * 10 ** (snr / 10)
    llr = (1 - 2.0 * cw) * mu + np.sqrt(2 * mu) * noise
    llr[:, : 2 * Z] = 0
    llr[:, 2 * Z + E + (K - Kp):] = 0
    llr[:, Kp:K] = np.inf
    return info, llr


def bp_ref(case, snr, nblk, iters):
    f = os.path.join(CACHE, "bp_%s_%.2f_%d_%d.json" % (case, snr, nblk, iters))
    if os.path.exists(f):
        return json.load(open(f))
    bg, Z, Kp, E, nl, _, _ = CASES[case]
    info, llr = make_llr(case, snr, nblk)
    h, it = O.decode_bp_flood(bg, Z, llr, iters, n_layers=nl)
    r = {"bler": float((h[:, :Kp] != info[:, :Kp]).any(1).mean()), "it": float(it.mean())}
    json.dump(r, open(f, "w"))
    return r


def run(case, snr, nblk, alpha, c0=0.0, c1=0.0, beta=0.0, scale=8, msg_max=127, app_max=0, iters=None, fp8=0):
    bg, Z, Kp, E, nl, it0, _ = CASES[case]
    iters = iters or it0
    info, llr = make_llr(case, snr, nblk)
    p = P()
    al = np.broadcast_to(np.asarray(alpha, np.float32), (46,)) if np.ndim(alpha) == 0 else np.asarray(alpha, np.float32)
    for i in range(46):
        p.alpha[i] = float(al[i]) if i < len(al) else float(al[-1])
    p.c0, p.c1, p.beta, p.scale, p.msg_max, p.app_max, p.fp8 = c0, c1, beta, scale, msg_max, app_max, fp8
    rows, cols, kb = O.BG_DIMS[bg]
    hard = np.zeros((nblk, kb * Z), np.uint8)
    its = np.zeros(nblk, np.int32)
    llr = np.ascontiguousarray(llr)
    L.nmsf_decode(bg, Z, nl, iters, C.byref(p), llr.ctypes.data, nblk, hard.ctypes.data, its.ctypes.data)
    return float((hard[:, :Kp] != info[:, :Kp]).any(1).mean()), float(its.mean())


def alpha_vec(core, ext):
    return [core] * 4 + [ext] * 42


if __name__ == "__main__":
    case = sys.argv[1]
    nblk = int(sys.argv[2])
    variants = json.loads(sys.argv[3])  # list of dicts of run() kwargs (alpha may be [core, ext])
    snrs = CASES[case][6] if len(sys.argv) < 5 else json.loads(sys.argv[4])
    for v in variants:
        kw = dict(v)
        if isinstance(kw.get("alpha"), list) and len(kw["alpha"]) == 2:
            kw["alpha"] = alpha_vec(*kw["alpha"])
        row = []
        t0 = time.time()
        for s in snrs:
            b, it = run(case, s, nblk, **kw)
            row.append((s, b, round(it, 1)))
        print(json.dumps(v), " ".join("%.1f:%.4f(%.1f)" % r for r in row), "%.0fs" % (time.time() - t0), flush=True)
