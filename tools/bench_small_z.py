#!/usr/bin/env python3
"""Decoder kernel time on the small lifting sizes (A/B of kernel builds via NRLDPC_LIB; NRLDPC_FORCE_GENERIC=1 = the run-time-Z
kernel everywhere; ZS="2 3 4" picks the sizes)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
DIMS = {1: (46, 68, 22), 2: (42, 52, 10)}
tag = os.path.basename(os.environ.get("NRLDPC_LIB", "default")) + (" generic" if os.environ.get("NRLDPC_FORCE_GENERIC") else "")
ZS = [int(x) for x in os.environ.get("ZS", "2 4 8 16 20 32 36 48 72 80 160").split()]
for bg in (1, 2):
    rows, cols, kb = DIMS[bg]
    for Z in ZS:
        B = max(4096, min(262144, (4096 * 384 // Z) // 256 * 256))
        for et in (0, 1):
            c = pkg.Codec(bg, Z, max_iter=25, early_term=bool(et), llr_dtype=np.float16)
            llr = (torch.randn((B, cols * Z), device="cuda") * 2 + 1.5).half()
            hard = torch.empty((B, kb * Z), device="cuda", dtype=torch.uint8)
            c.set_timing(True)
            ms = []
            for i in range(4):
                c.decode_dev(llr.data_ptr(), B, hard.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
                ms.append(c.last_kernel_ms())
            c.close()
            print("%-22s BG%d Z=%3d et=%d: %.3f ms  %.2f Gbit/s" % (tag, bg, Z, et, min(ms[1:]), B * kb * Z / min(ms[1:]) / 1e6), flush=True)
