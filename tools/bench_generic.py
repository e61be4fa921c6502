#!/usr/bin/env python3
"""Kernel time of the generic (runtime-Z) decoder for a few lifting sizes; used for A/B of kernel builds
(NRLDPC_LIB selects the library)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
DIMS = {1: (46, 68, 22), 2: (42, 52, 10)}
for bg, Z, B in ((2, 320, 4096), (2, 256, 4096), (2, 192, 4096), (2, 96, 8192), (2, 20, 16384), (1, 320, 4096), (1, 256, 4096), (1, 192, 4096), (1, 88, 8192), (1, 24, 16384)):
    rows, cols, kb = DIMS[bg]
    c = pkg.Codec(bg, Z, max_iter=25, early_term=False, llr_dtype=np.float16)
    llr = (torch.randn((B, cols * Z), device="cuda") * 2 + 1.5).half()
    hard = torch.empty((B, kb * Z), device="cuda", dtype=torch.uint8)
    c.set_timing(True)
    ms = []
    for i in range(4):
        c.decode_dev(llr.data_ptr(), B, hard.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
        ms.append(c.last_kernel_ms())
    c.close()
    t = min(ms[1:])
    print("BG%d Z=%3d batch %5d: %.3f ms  %.2f Gbit/s info" % (bg, Z, B, t, B * kb * Z / t / 1e6), flush=True)
