#!/usr/bin/env python3
"""Kernel experiment: check one library variant (NRLDPC_LIB) against the oracle on a (BG, Z), then time it.
python tools/exp_check.py <bg> <Z> [nl]  -- prints one line per timing (fixed 25 iterations; parity stop at a waterfall point)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import BG_DIMS, awgn_llr
import oracle as orc
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
T = importlib.import_module("test_decode_gpu")
bg, Z = int(sys.argv[1]), int(sys.argv[2])
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
esn0 = float(os.environ.get("ESN0", "-0.5" if bg == 1 else "-1.0"))
tag = os.path.basename(os.environ.get("NRLDPC_LIB", "default"))
rng = np.random.default_rng(11)
ok = True
try:
    for B, it, et in ((3, 6, False), (5, 12, True), (2, 25, False), (4, 9, True)):
        T.run_case(pkg, orc, rng, bg, Z, B, esn0 + 0.3, it, nl=nl, et=et, app=False)
except AssertionError as e:
    ok = False
    print(tag, "PARITY FAIL:", e, flush=True)
rows, cols, kb = BG_DIMS[bg]
B = max(4096, (4096 * 384 // Z) // 256 * 256)
info = rng.integers(0, 2, (64, kb * Z), dtype=np.uint8)
cw = orc.encode(bg, Z, info)
llr = torch.from_numpy(awgn_llr(rng, np.tile(cw, (B // 64, 1)), esn0, np.float16, Z)).cuda()
hard = torch.empty((B, kb * Z), device="cuda", dtype=torch.uint8)
for et in (0, 1):
    c = pkg.Codec(bg, Z, max_iter=25, early_term=bool(et), n_layers=nl, llr_dtype=np.float16)
    c.set_timing(True)
    ms = []
    for i in range(8):
        c.decode_dev(llr.data_ptr(), B, hard.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
        ms.append(c.last_kernel_ms())
    t = float(np.median(ms[2:]))
    print("%-28s BG%d Z=%3d nl=%d batch %5d et=%d: %.3f ms (min %.3f)  %.2f Gbit/s  parity %s" % (tag, bg, Z, nl, B, et, t, min(ms[2:]), B * kb * Z / t / 1e6, "ok" if ok else "FAIL"), flush=True)
    c.close()
