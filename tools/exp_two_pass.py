#!/usr/bin/env python3
"""Measured lead, not product: the parity stop in TWO passes for the sizes whose workgroups hold several codewords.  Pass 1 decodes
the whole batch with a cap of I1 iterations; the codewords that reach the cap are gathered (torch index_select on the LLR rows),
decoded again from scratch with the full cap, and scattered back -- the same hard decisions and iteration counts as one pass
(the decoder is deterministic; checked here), but a workgroup no longer runs 25 iterations because ONE of its codewords does.
python tools/exp_two_pass.py [bg,Z ...]   prints one-pass / two-pass device time (torch events around everything, gathers included)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
DIMS = {1: (46, 68, 22), 2: (42, 52, 10)}
cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1, 4), (1, 8), (1, 16), (1, 32), (1, 64), (1, 128), (1, 384), (2, 8), (2, 32), (2, 128)]
I1 = int(os.environ.get("I1", "10"))
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(5)
for bg, Z in cases:
    rows, cols, kb = DIMS[bg]
    B = max(4096, min(262144, (4096 * 384 // Z) // 256 * 256))
    nb = min(B, max(256, 98304 // Z // 256 * 256)); B = B // nb * nb
    full = pkg.Codec(bg, Z, max_iter=25, early_term=True, llr_dtype=np.float16)
    first = pkg.Codec(bg, Z, max_iter=I1, early_term=True, llr_dtype=np.float16)
    info = np.random.default_rng(Z).integers(0, 2, (nb, kb * Z), dtype=np.uint8)
    sgn = 1.0 - 2.0 * torch.from_numpy(full.encode(info).astype(np.float32)).cuda()
    noise = torch.randn(sgn.shape, device="cuda", generator=g)
    def llr_at(esn0):
        mu = 2.0 * 10.0 ** (esn0 / 10.0)
        x = sgn * mu + (2 * mu) ** 0.5 * noise
        x[:, : 2 * Z] = 0
        return x.half().contiguous()
    hard = torch.empty((B, kb * Z), device="cuda", dtype=torch.uint8)
    iters = torch.zeros(B, device="cuda", dtype=torch.int32)
    lo, hi = -4.0, 4.0
    for _ in range(8):  # the waterfall: mean 10 of 25 iterations
        mid = 0.5 * (lo + hi)
        x = llr_at(mid)
        full.decode_dev(x.data_ptr(), nb, hard.data_ptr(), iters.data_ptr(), None, st); torch.cuda.synchronize()
        if iters[:nb].float().mean().item() > 10: lo = mid
        else: hi = mid
    llr = llr_at(0.5 * (lo + hi)).repeat(B // nb, 1).contiguous()
    def one_pass():
        full.decode_dev(llr.data_ptr(), B, hard.data_ptr(), iters.data_ptr(), None, st)
    hard2 = torch.empty_like(hard); iters2 = torch.zeros_like(iters)
    def two_pass():
        first.decode_dev(llr.data_ptr(), B, hard2.data_ptr(), iters2.data_ptr(), None, st)
        idx = torch.nonzero(iters2 >= I1).squeeze(1)
        n = int(idx.numel())  # (a host round trip: the product form would size the second launch on the device)
        if n:
            sub = llr.index_select(0, idx)
            h = torch.empty((n, kb * Z), device="cuda", dtype=torch.uint8); it = torch.empty(n, device="cuda", dtype=torch.int32)
            full.decode_dev(sub.data_ptr(), n, h.data_ptr(), it.data_ptr(), None, st)
            hard2.index_copy_(0, idx, h); iters2.index_copy_(0, idx, it)
        return n
    def timed(f):
        ts = []
        for i in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); r = f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return min(ts[1:]), r
    t1, _ = timed(one_pass)
    t2, n = timed(two_pass)
    same = bool((hard == hard2).all().item() and (iters == iters2).all().item())
    print("BG%d Z=%3d batch %6d: one pass %.3f ms, two passes (cap %d, %4.1f %% decoded again) %.3f ms (%+.0f %%)  identical: %s" % (
        bg, Z, B, t1, I1, 100.0 * n / B, t2, 100 * (t2 / t1 - 1), same), flush=True)
    full.close(); first.close()
