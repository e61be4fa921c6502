#!/usr/bin/env python3
"""A/B of the interleaved block geometry (NRLDPC_Z64I_LIST) against the kernels that served the same sizes before: run once as it
is and once with NRLDPC_NO_ILV=1 (the switch is read once per process), each run writes one JSON line per (BG, Z, layers, stop).
python tools/ab_ilv.py <out.jsonl> [bg,Z[,nl] ...]   (default: every entry of the list, all rows)
Every size is first checked against the oracle (four small batches: fixed / parity stop, ragged last workgroup)."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import BG_DIMS, awgn_llr
import oracle as orc
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
bld = importlib.import_module("ldpc-3gpp-matlab_amd.build")
T = importlib.import_module("test_decode_gpu")
out = open(sys.argv[1], "a")
cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [(bg, z) for bg, z, _, _ in bld.Z64I]
check = not os.environ.get("NO_CHECK")
for case in cases:
    bg, Z = case[:2]
    nl = case[2] if len(case) > 2 else 0
    esn0 = -0.5 if bg == 1 else -1.0
    rng = np.random.default_rng(11)
    ok = True
    if check:
        try:
            for B, it, et in ((3, 6, False), (5, 12, True), (2, 25, False), (7, 9, True)):
                T.run_case(pkg, orc, rng, bg, Z, B, esn0 + 0.3, it, nl=nl, et=et, app=False)
        except AssertionError as e:
            ok = False
            print("PARITY FAIL BG%d Z=%d nl=%d: %s" % (bg, Z, nl, e), flush=True)
    rows, cols, kb = BG_DIMS[bg]
    if os.environ.get("WATERFALL"):  # the SNR where the parity stop takes about 10 of 25 iterations on average (bisection, 256 codewords)
        cwf = orc.encode(bg, Z, rng.integers(0, 2, (256, kb * Z), dtype=np.uint8))
        c = pkg.Codec(bg, Z, max_iter=25, early_term=True, n_layers=nl, llr_dtype=np.float16)
        lo, hi = -3.0, 9.0
        for _ in range(8):
            mid = 0.5 * (lo + hi)
            its = c.decode(awgn_llr(np.random.default_rng(5), cwf, mid, np.float16, Z), want_iters=True)[1]
            if its.mean() > 10: lo = mid
            else: hi = mid
        c.close()
        esn0 = 0.5 * (lo + hi) + float(os.environ.get("WF_OFFSET", "0"))  # WF_OFFSET: dB away from the waterfall
    B = max(4096, (4096 * 384 // Z) // 256 * 256)
    nb = min(B, max(256, 98304 // Z // 256 * 256))  # distinct noisy codewords (the parity stop's time depends on which ones share a workgroup)
    B = B // nb * nb
    info = rng.integers(0, 2, (nb, kb * Z), dtype=np.uint8)
    cw = orc.encode(bg, Z, info)
    llr = torch.from_numpy(awgn_llr(rng, cw, esn0, np.float16, Z)).cuda().repeat(B // nb, 1).contiguous()
    hard = torch.empty((B, kb * Z), device="cuda", dtype=torch.uint8)
    for et in (0, 1):
        c = pkg.Codec(bg, Z, max_iter=25, early_term=bool(et), n_layers=nl, llr_dtype=np.float16)
        c.set_timing(True)
        ms = []
        for i in range(8):
            c.decode_dev(llr.data_ptr(), B, hard.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
            ms.append(c.last_kernel_ms())
        c.close()
        rec = {"bg": bg, "Z": Z, "nl": nl, "et": et, "batch": B, "ms": float(np.median(ms[2:])), "parity_ok": ok, "esn0": esn0,
               "no_ilv": bool(os.environ.get("NRLDPC_NO_ILV"))}
        out.write(json.dumps(rec) + "\n"); out.flush()
        print(rec, flush=True)
