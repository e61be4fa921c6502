#!/usr/bin/env python3
"""Kernel time of one (BG, Z) decoder configuration: python tools/bench_one.py <bg> <Z> [batch] [early_term] [n_layers] [max_iter].
NRLDPC_LIB selects the library (A/B of kernel builds)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
DIMS = {1: (46, 68, 22), 2: (42, 52, 10)}
bg, Z = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else max(4096, (4096 * 384 // Z) // 256 * 256)
et = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nl = int(sys.argv[5]) if len(sys.argv) > 5 else 0
mi = int(sys.argv[6]) if len(sys.argv) > 6 else 25
rows, cols, kb = DIMS[bg]
c = pkg.Codec(bg, Z, max_iter=mi, early_term=bool(et), n_layers=nl, llr_dtype=np.float16)
llr = (torch.randn((B, cols * Z), device="cuda") * 2 + 1.5).half()
hard = torch.empty((B, kb * Z), device="cuda", dtype=torch.uint8)
c.set_timing(True)
ms = []
for i in range(6):
    c.decode_dev(llr.data_ptr(), B, hard.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
    ms.append(c.last_kernel_ms())
t = min(ms[1:])
print("%s BG%d Z=%3d batch %5d et=%d nl=%d it=%d: %.3f ms  %.2f Gbit/s info" % (os.path.basename(os.environ.get("NRLDPC_LIB", "default")), bg, Z, B, et, nl, mi, t, B * kb * Z / t / 1e6), flush=True)
