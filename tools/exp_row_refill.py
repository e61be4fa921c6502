#!/usr/bin/env python3
"""Experiment (round 5): slot refill in the block geometry's one-thread-per-row kernels that hold 2-4 codewords per workgroup under the parity
stop.  Parity against the oracle with the workgroups capped (NRLDPC_REFILL_GRID), then the stop's time at the size's waterfall.
NRLDPC_LIB selects the library.  python tools/exp_row_refill.py bg,Z ..."""
import importlib, os, sys, json
os.environ.setdefault("NRLDPC_TEST_HOOKS", "1")  # NRLDPC_REFILL_GRID is read only under this (nrldpc_decode_z64p.h)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import BG_DIMS, awgn_llr
import oracle as orc
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
T = importlib.import_module("test_decode_gpu")
for arg in sys.argv[1:]:
    bg, Z = (int(x) for x in arg.split(","))
    rows, cols, kb = BG_DIMS[bg]
    rng = np.random.default_rng(3)
    ok = True
    try:
        for grid in (1, 2):
            os.environ["NRLDPC_REFILL_GRID"] = str(grid)
            for nl in (0, 17):
                T.run_case(pkg, orc, rng, bg, Z, 23, T._waterfall_esn0(bg, nl or rows) + 0.3, 12, nl=nl, et=True, app=False)
        del os.environ["NRLDPC_REFILL_GRID"]
        T.run_case(pkg, orc, rng, bg, Z, 5, 0.5, 9, nl=0, et=True, app=False)
    except AssertionError as e:
        ok = False
        print("PARITY FAIL", bg, Z, e, flush=True)
    os.environ.pop("NRLDPC_REFILL_GRID", None)
    B = max(4096, (4096 * 384 // Z) // 256 * 256) * 4
    nb = 1024
    info = rng.integers(0, 2, (nb, kb * Z), dtype=np.uint8)
    cw = orc.encode(bg, Z, info)
    c = pkg.Codec(bg, Z, max_iter=25, early_term=True, llr_dtype=np.float16)
    lo, hi = -5.0, 6.0
    for _ in range(8):
        mid = 0.5 * (lo + hi)
        its = c.decode(awgn_llr(np.random.default_rng(5), cw[:256], mid, np.float16, Z), want_iters=True)[1]
        if its.mean() > 10: lo = mid
        else: hi = mid
    esn0 = 0.5 * (lo + hi)
    llr = torch.from_numpy(awgn_llr(rng, cw, esn0, np.float16, Z)).cuda().repeat(B // nb, 1).contiguous()
    hard = torch.empty((llr.shape[0], kb * Z), device="cuda", dtype=torch.uint8)
    it = torch.zeros(llr.shape[0], device="cuda", dtype=torch.int32)
    c.set_timing(True)
    ms = []
    for i in range(7):
        c.decode_dev(llr.data_ptr(), llr.shape[0], hard.data_ptr(), it.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        ms.append(c.last_kernel_ms())
    c.close()
    print(json.dumps({"lib": os.path.basename(os.environ.get("NRLDPC_LIB", "default")), "bg": bg, "Z": Z, "parity_ok": ok, "esn0": esn0, "batch": int(llr.shape[0]),
                      "mean_iters": float(it.float().mean().item()), "ms": float(np.median(ms[2:]))}), flush=True)
