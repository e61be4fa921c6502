#!/usr/bin/env python3
"""Generate tests/golden/chain_golden.npz: seeded inputs and expected outputs of the stages either side of the
decoder core -- CRC attachment + segmentation, encoding, rate matching, rate
recovery.  Expected values come from the host mirror of NRLDPCEncoder.step (numpy, itself checked against the
literal restatement of the reference's loops) and from oracle/nrldpc_chain_oracle.c; like nmsq_golden.npz they
pin the build (regression + GPU parity), the reference holds no vectors for these stages.
Run from the repo root:  python tests/golden/make_chain_golden.py
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
CASES = {
    "bg2_a100": dict(BG=2, A=100, G=300, Q_m=2),
    "bg1_a5000_rv3_16qam": dict(BG=1, A=5000, G=6000, Q_m=4, rv_id=3),
    "bg2_a3842_c2_lbrm_rv2": dict(BG=2, A=3842, G=11526, Q_m=2, I_LBRM=1, TBS_LBRM=6000, rv_id=2),
    "bg2_a500_rep_64qam": dict(BG=2, A=500, G=5004, Q_m=6),
}


def main():
    out = {}
    rng = np.random.default_rng(20260930)
    for name, kw in CASES.items():
        enc = pkg.NRLDPCEncoder(**kw)
        enc.validate()
        n_tb = 2
        a = rng.integers(0, 2, (n_tb, enc.A), dtype=np.uint8)
        c = enc.code_block_segmentation(enc.crc_calculation(a))          # [n_tb][C][K], fillers as 0
        cw = np.stack([O.encode(enc.BG, enc.Z_c, c[t]) for t in range(n_tb)])   # [n_tb][C][ncols*Z]
        g = enc.rate_match(cw[:, :, 2 * enc.Z_c:])                        # [n_tb][G]; no GPU needed here
        g_tilde = (4 * rng.standard_normal((n_tb, enc.G))).astype(np.float32)
        rr = O.rate_recover(enc.Z_c, enc.C, enc.K, int(enc.K_prime), enc.N, enc.N_cb, enc.k_0, enc.Q_m, enc.G, enc.E_r,
                            g_tilde, None)
        out[name + "/kw"] = np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8)
        out[name + "/a"] = a
        out[name + "/c_packed"] = np.packbits(c.reshape(n_tb * enc.C, enc.K), axis=1)
        out[name + "/cw_packed"] = np.packbits(cw.reshape(n_tb * enc.C, -1), axis=1)
        out[name + "/g_packed"] = np.packbits(g, axis=1)
        out[name + "/g_tilde"] = g_tilde
        out[name + "/rate_recovered"] = rr.astype(np.float32)             # +inf at filler positions
        enc.release()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chain_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
