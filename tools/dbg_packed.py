#!/usr/bin/env python3
"""Where do a kernel's hard decisions leave the oracle's?  CASES="bg,Z,batch,iterations ..." (fixed iteration counts, hard
output, SNR 0.5 dB: far from convergence after one iteration, so a single stale LDS word shows); prints the mismatching
codewords, columns and rows.  NRLDPC_LIB / NRLDPC_FORCE_GENERIC / NRLDPC_NO_PACKED select the kernel under test.
Found the exec-mask literal defect of the packed kernels."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import BG_DIMS, awgn_llr
import oracle as orc
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
T = importlib.import_module("test_decode_gpu")
rng = np.random.default_rng(5)
import os
CASES = [tuple(int(x) for x in c.split(',')) for c in os.environ.get('CASES','1,24,5,1').split()]
for bg, Z, B, it in CASES:
    kb = BG_DIMS[bg][2]
    info = rng.integers(0, 2, (B, kb * Z), dtype=np.uint8)
    cw = orc.encode(bg, Z, info)
    llr = awgn_llr(rng, cw, 0.5, np.float16, Z)
    c = pkg.Codec(bg, Z, max_iter=it, early_term=False, llr_dtype=np.float16)
    out = c.decode(llr, want_iters=True)
    ref = orc.decode_nmsq(bg, Z, llr.astype(np.float64), it, n_layers=0, early_term=False, scale=8, want_app=False, **T.rule_kw(c, 8))
    d = (out[0] != ref[0])
    print(bg, Z, B, it, "mismatches:", int(d.sum()))
    for b in range(B):
        idx = np.nonzero(d[b])[0]
        if len(idx): print("  cw", b, "n", len(idx), "cols", sorted(set((idx // Z).tolist()))[:30], "z", sorted(set((idx % Z).tolist()))[:30])
    c.close()
