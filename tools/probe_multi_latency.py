#!/usr/bin/env python3
"""cfg4 diagnosis: the latency of ONE workgroup of the run-time-Z kernel per (BG, Z) -- a mixed-batch call (nrldpc_decode_multi_dev) that holds a single
bucket of n codewords, all workgroups resident at once, parity stop at Es/N0 = 3 dB (BASELINE configs[3]'s operating point) -- and the same bucket through
the bucket's own kernel (nrldpc_decode_dev).  Wall clock around call + synchronize, minimum of 10."""
import importlib, sys, time, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
import bench_configs as bc
s0 = torch.cuda.current_stream().cuda_stream
def wall(f):
    ts = []
    for _ in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts[2:])
empty = wall(lambda: None)
print("empty synchronize: %.3f ms" % empty)
for bg in (1, 2):
    rows, cols, kb = bc.DIMS[bg]
    for Z in (384, 320, 256, 192, 128, 96, 64, 32, 16, 8, 2):
        for n in (1, 80):
            codec = pkg.Codec(bg, Z, max_iter=25, early_term=True, alpha=0.625, llr_dtype=np.float16)
            info, llr = bc.synth(codec, bg, Z, n, cols * Z, 3.0, Z)
            hard = torch.empty((n, kb * Z), device="cuda", dtype=torch.uint8)
            iters = torch.zeros(n, device="cuda", dtype=torch.int32)
            call = pkg.MultiCall([codec], [llr.data_ptr()], [n], [hard.data_ptr()], [iters.data_ptr()])
            t_multi = wall(lambda: call(s0))
            t_own = wall(lambda: codec.decode_dev(llr.data_ptr(), n, hard.data_ptr(), iters.data_ptr(), None, s0))
            print("BG%d Z=%3d n=%2d: shared-kernel call %.3f ms, own kernel %.3f ms, iterations mean %.2f max %d" % (bg, Z, n, t_multi, t_own, iters.float().mean().item(), iters.max().item()), flush=True)
            codec.close()
