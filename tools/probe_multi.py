#!/usr/bin/env python3
"""Where the wall time of one mixed-batch call (BASELINE configs[3], nrldpc_decode_multi_dev) goes: when the call returns to the host,\nwhen the device is done, the iteration counts behind it.  NRLDPC_MULTI_Z64_MIN_ROWS / NRLDPC_MULTI_ONE_STREAM are the A/B switches."""
import importlib, sys, time, os
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
import bench_configs as bc
rng = np.random.default_rng(4)
B = 8192
draws = [(int(rng.integers(1, 3)), int(rng.choice(bc.ALL_Z))) for _ in range(B)]
buckets = {}
for key in draws: buckets[key] = buckets.get(key, 0) + 1
work = []
for (bg, Z), n in sorted(buckets.items()):
    rows, cols, kb = bc.DIMS[bg]
    codec = pkg.Codec(bg, Z, max_iter=25, early_term=True, alpha=0.625, llr_dtype=np.float16)
    info, llr = bc.synth(codec, bg, Z, n, cols * Z, 3.0, Z)
    hard = torch.empty((n, kb * Z), device="cuda", dtype=torch.uint8)
    iters = torch.zeros(n, device="cuda", dtype=torch.int32)
    work.append((codec, llr, hard, n, kb * Z, iters, bg, Z))
s0 = torch.cuda.current_stream().cuda_stream
call = pkg.MultiCall([w[0] for w in work], [w[1].data_ptr() for w in work], [w[3] for w in work], [w[2].data_ptr() for w in work], [w[5].data_ptr() for w in work])
for rep in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    call(s0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("call returns after %.3f ms, all done after %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
tot = 0; cnt = 0; strag = []
for w in work:
    it = w[5].cpu().numpy(); tot += it.sum(); cnt += it.size
    if (it >= 25).any(): strag.append((w[6], w[7], int((it >= 25).sum()), w[3]))
print("mean iterations %.2f over %d codewords; buckets with codewords at the cap:" % (tot / cnt, cnt), strag)
mx = sorted(((int(w[5].max().item()), w[6], w[7], w[3]) for w in work), reverse=True)
print("largest iteration counts (iterations, BG, Z, codewords in the bucket):", mx[:10])
