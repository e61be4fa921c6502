"""BASELINE.json's full sizes through size-independent properties (the oracle would need minutes):
encode -> noise -> decode round trips on 4096-codeword batches, agreement between the host and the
device entry points, oracle spot checks on a sample, and the mixed-(BG,Z) configuration."""
import os

import numpy as np
import pytest

from conftest import rule_kw, ALL_Z, BG_DIMS, awgn_llr

pytestmark = pytest.mark.gpu


def _roundtrip(pkg, orc, bg, Z, B, nl, E, esn0, iters, et, seed):
    rng = np.random.default_rng(seed)
    kb = BG_DIMS[bg][2]
    c = pkg.Codec(bg, Z, max_iter=iters, n_layers=nl, early_term=et, llr_dtype=np.float16)
    info = rng.integers(0, 2, (B, kb * Z), dtype=np.uint8)
    cw = c.encode(info)
    llr = awgn_llr(rng, cw, esn0, np.float16, Z, E=E)
    hard, it = c.decode(llr, want_iters=True)
    c.close()
    bler = float((hard != info).any(1).mean())
    sample = rng.choice(B, 6, replace=False)
    ho, io = orc.decode_nmsq(bg, Z, llr[sample].astype(np.float64), iters, n_layers=nl, early_term=et,
                             **rule_kw(c))  # the library's rule for this rate (cfg.alpha = 0)
    assert (hard[sample] == ho).all() and (it[sample] == io).all()
    assert all(orc.syndrome_weight(bg, Z, cw[b]) == 0 for b in sample)
    return bler, it


def test_cfg2_bg1_z384_r13_batch4096(pkg, orc):
    bler, it = _roundtrip(pkg, orc, 1, 384, 4096, 0, 25344, -0.6, 25, False, 1)
    assert bler < 0.02 and (it == 25).all()


def test_headline_batch_every_codeword_against_the_oracle(pkg, orc):
    """The benchmarked call itself -- nrldpc_decode_dev, fp16 LLRs resident in HBM, batch 4096, BG1 Z=384, 25 fixed iterations
    (bench.py's timed step) -- and its parity-stop twin, with EVERY one of the 4096 codewords compared with the oracle: hard
    decisions, and iteration counts for the parity stop (the full-size checks used to sample 6 codewords)."""
    import torch
    rng = np.random.default_rng(4096)
    bg, Z, B, K = 1, 384, 4096, 22 * 384
    enc = pkg.Codec(bg, Z, max_iter=1, llr_dtype=np.float16)
    info = rng.integers(0, 2, (B, K), dtype=np.uint8)
    llr = awgn_llr(rng, enc.encode(info), -1.3, np.float16, Z, E=25344)  # in the waterfall: converged and unconverged codewords
    enc.close()
    d_llr = torch.from_numpy(llr).cuda()
    d_hard = torch.empty((B, K), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(B, dtype=torch.int32, device="cuda")
    nthr = 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        nthr = max(1, int(q) // int(per)) if q != "max" else (os.cpu_count() or 1)
    except (OSError, ValueError):
        nthr = min(16, os.cpu_count() or 1)
    orc.lib().orc_set_threads(nthr)
    for et in (False, True):
        c = pkg.Codec(bg, Z, max_iter=25, early_term=et, llr_dtype=np.float16)
        c.decode_dev(d_llr.data_ptr(), B, d_hard.data_ptr(), d_it.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        c.close()
        ho, io = orc.decode_nmsq(bg, Z, llr.astype(np.float64), 25, early_term=et, **rule_kw(c))
        assert (d_hard.cpu().numpy() == ho).all(), "hard decisions differ (early_term=%s)" % et
        assert (d_it.cpu().numpy() == io).all(), "iteration counts differ (early_term=%s)" % et
        assert 0.0 < (ho != info).any(1).mean() < 0.9, (ho != info).any(1).mean()  # the batch straddles the waterfall


def test_cfg3_bg2_z384_rate_sweep_batch4096(pkg, orc):
    # G for R = 1/5 ... 2/3 -> active layers; Es/N0 about 1 dB above each waterfall
    for E, nl, esn0 in ((19120, 42, -3.0), (15296, 32, -2.0), (11472, 22, -0.5), (9560, 17, 0.5), (7648, 12, 2.0),
                        (6374, 9, 3.2), (5736, 7, 4.5)):  # all seven rates of BASELINE configs[2]
        bler, it = _roundtrip(pkg, orc, 2, 384, 4096, nl, E, esn0, 25, True, E)
        assert bler < 0.05, (E, bler)
        assert it.min() >= 1 and it.mean() < 20


def test_cfg5_bg1_r89_early_termination_shard(pkg, orc):
    """One GPU's shard (8192 codewords) of the 65536-codeword R=8/9 configuration."""
    bler, it = _roundtrip(pkg, orc, 1, 384, 8192, 5, 9478, 7.5, 25, True, 5)
    assert bler < 0.02 and it.mean() < 8


def test_cfg4_mixed_bg_and_Z(pkg, orc):
    """Mixed BG1/BG2, Z drawn per codeword: host buckets by (BG, Z), one launch per bucket."""
    rng = np.random.default_rng(4)
    draws = [(int(rng.integers(1, 3)), int(rng.choice(ALL_Z))) for _ in range(8192)]
    buckets = {}
    for i, key in enumerate(draws):
        buckets.setdefault(key, []).append(i)
    n_ok = n = 0
    for (bg, Z), idx in sorted(buckets.items()):
        kb = BG_DIMS[bg][2]
        c = pkg.Codec(bg, Z, max_iter=25, early_term=True, llr_dtype=np.float16, alpha=0.625)
        info = rng.integers(0, 2, (len(idx), kb * Z), dtype=np.uint8)
        llr = awgn_llr(rng, c.encode(info), 3.0, np.float16, Z)
        hard, it = c.decode(llr, want_iters=True)
        c.close()
        if Z >= 32:  # tiny codes have a real error rate at any SNR; count only the rest
            n_ok += int((hard == info).all(1).sum())
            n += len(idx)
        j = int(rng.integers(0, len(idx)))
        ho, io = orc.decode_nmsq(bg, Z, llr[j:j + 1].astype(np.float64), 25, early_term=True, alpha=0.625)
        assert (hard[j] == ho[0]).all() and it[j] == io[0]
    assert len(buckets) >= 90 and n_ok / n > 0.97


def test_host_and_device_entry_points_agree(pkg, orc):
    import torch
    rng = np.random.default_rng(12)
    c = pkg.Codec(2, 384, max_iter=25, n_layers=22, early_term=True, llr_dtype=np.float16)
    info = rng.integers(0, 2, (4096, c.K), dtype=np.uint8)
    llr = awgn_llr(rng, c.encode(info), 0.0, np.float16, 384, E=11472)
    h1, it1 = c.decode(llr, want_iters=True)
    d_llr = torch.from_numpy(llr).cuda()
    d_h = torch.empty((4096, c.K), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(4096, dtype=torch.int32, device="cuda")
    c.decode_dev(d_llr.data_ptr(), 4096, d_h.data_ptr(), d_it.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c.close()
    assert (d_h.cpu().numpy() == h1).all() and (d_it.cpu().numpy() == it1).all()


@pytest.mark.parametrize("dt,B,bg,Z", [(np.float64, 1237, 1, 384), (np.float32, 1001, 1, 384), (np.float16, 2049, 1, 384),
                                        (np.float32, 9001, 2, 20), (np.float64, 3001, 1, 36)])  # ... and two run-time-Z sizes
def test_pipelined_host_path_equals_one_device_launch(pkg, orc, dt, B, bg, Z):
    """Batches above 32 MB take the chunked, double-buffered host path of nrldpc_decode (pinned staging, copy
    threads, two streams); it must return exactly what one device launch on the same LLRs returns, for a
    batch that is not a multiple of the chunk size and for every boundary dtype (MATLAB doubles included)."""
    import torch
    rng = np.random.default_rng(B)
    c = pkg.Codec(bg, Z, max_iter=12, early_term=True, alpha=0.625, llr_dtype=dt)
    info = rng.integers(0, 2, (B, c.K), dtype=np.uint8)
    llr = awgn_llr(rng, c.encode(info), -0.3 if Z == 384 else 1.5, dt, Z)
    assert llr.nbytes >= (8 << 20) * (2 if dt == np.float64 else 1)  # the pipelined path starts at 8 MB of device-format input
    # The copy threads quantise to int8 on the way into the pinned slots (nrldpc_quantise_llr): the values that could
    # tell the host's arithmetic from the kernel's, in core and extension columns of a few codewords -- rounding ties,
    # the clamp, NaN, +inf fillers (NRLDPCDecoder.m:264), and one -inf, whose chunk must travel in its own format.
    odd = np.array([0.0625, -0.0625, 0.1875, -0.3125, 15.875, 15.9375, -15.9375, 1e4, -1e4, np.nan, np.inf, 1e-3, -0.0,
                    0.06250001, 0.31249999], dtype=np.float64).astype(dt)
    for b in (0, 3, B // 2, B - 1):
        llr[b, 2 * Z + 5: 2 * Z + 5 + odd.size] = odd
        llr[b, 30 * Z + 7: 30 * Z + 7 + odd.size] = odd
        llr[b, 8 * Z:8 * Z + 11] = np.inf
    llr[B // 3, 9 * Z + 3] = -np.inf
    llr[B // 3, 40 * Z + 1] = -np.inf
    h1, it1 = c.decode(llr, want_iters=True)
    h2 = c.decode(llr)                                       # second call reuses the pinned slots
    dev_dt = np.float32 if dt == np.float64 else dt
    cd = pkg.Codec(bg, Z, max_iter=12, early_term=True, alpha=0.625, llr_dtype=dev_dt)
    d_llr = torch.from_numpy(llr.astype(dev_dt)).cuda()
    d_h = torch.empty((B, c.K), dtype=torch.uint8, device="cuda")
    d_it = torch.empty(B, dtype=torch.int32, device="cuda")
    cd.decode_dev(d_llr.data_ptr(), B, d_h.data_ptr(), d_it.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c.close(); cd.close()
    assert (d_h.cpu().numpy() == h1).all() and (d_it.cpu().numpy() == it1).all() and (h2 == h1).all()
    assert (h1 != info).any(1).mean() < (0.05 if Z == 384 else 0.5) and it1.min() < 12
    os.environ["NRLDPC_HOST_I8"] = "0"                       # ... and what the native-format host path returns
    try:
        c = pkg.Codec(bg, Z, max_iter=12, early_term=True, alpha=0.625, llr_dtype=dt)
        h3, it3 = c.decode(llr, want_iters=True)
        c.close()
    finally:
        del os.environ["NRLDPC_HOST_I8"]
    assert (h3 == h1).all() and (it3 == it1).all()


def test_cfg4_mixed_batch_in_one_call_matches_the_oracle(pkg, orc):
    """nrldpc_decode_multi_dev on the mixed batch of BASELINE configuration 4 plus two large buckets: small buckets
    share one launch of the run-time-Z kernel per base graph, buckets that fill the chip go to their compile-time-Z
    kernels.  Checked against the ORACLE (two codewords of every bucket, all of the single-codeword bucket) and
    against one nrldpc_decode_dev launch per bucket (everything): hard bits and iteration counts, including pruned
    layer counts, differing iteration caps, fixed-iteration and early-termination configurations."""
    import torch
    rng = np.random.default_rng(44)
    draws = [(int(rng.integers(1, 3)), int(rng.choice(ALL_Z))) for _ in range(8192)]
    buckets = {}
    for key in draws:
        buckets[key] = buckets.get(key, 0) + 1
    buckets[(1, 7)] = 1                                            # a single-codeword bucket
    buckets[(1, 384)] = 600                                        # >= 512*384 rows: routed to the z64 kernels
    buckets[(2, 256)] = 800
    codecs, llrs, ns, ref_h, ref_i, host_llr = [], [], [], [], [], []
    s = torch.cuda.current_stream().cuda_stream
    for k, ((bg, Z), n) in enumerate(sorted(buckets.items())):
        rows, cols, kb = BG_DIMS[bg]
        nl = 0 if k % 3 else max(4, rows - (k % 11))               # some pruned layer counts
        dt = np.float16 if k % 3 else np.float32                   # both LLR types in one call: four launch groups
        c = pkg.Codec(bg, Z, max_iter=8 + (k % 5), n_layers=nl, early_term=bool(k % 4), llr_dtype=dt)
        info = rng.integers(0, 2, (n, kb * Z), dtype=np.uint8)
        x = awgn_llr(rng, c.encode(info), 2.0, dt, Z)
        llr = torch.from_numpy(x).cuda()
        h = torch.empty((n, kb * Z), dtype=torch.uint8, device="cuda")
        it = torch.empty(n, dtype=torch.int32, device="cuda")
        c.decode_dev(llr.data_ptr(), n, h.data_ptr(), it.data_ptr(), None, s)
        codecs.append(c); llrs.append(llr); ns.append(n); ref_h.append(h); ref_i.append(it); host_llr.append(x[:2])
    out_h = [torch.zeros_like(h) for h in ref_h]
    out_i = [torch.zeros_like(i) for i in ref_i]
    out_h2 = [torch.zeros_like(h) for h in ref_h]
    torch.cuda.synchronize()  # the zero fills run on the current stream; s2 below does not wait for that stream
    pkg.decode_multi_dev(codecs, [x.data_ptr() for x in llrs], ns, [x.data_ptr() for x in out_h],
                         [x.data_ptr() for x in out_i], s)
    # a second call on another stream while the first may still be running: each call has its own table slot
    s2 = torch.cuda.Stream()
    pkg.decode_multi_dev(codecs, [x.data_ptr() for x in llrs], ns, [x.data_ptr() for x in out_h2], None, s2.cuda_stream)
    torch.cuda.synchronize()
    for k, c in enumerate(codecs):
        assert (out_h[k] == ref_h[k]).all() and (out_i[k] == ref_i[k]).all(), (k, c.bg, c.Z)
        assert (out_h2[k] == ref_h[k]).all(), (k, c.bg, c.Z)
        m = min(2, ns[k])
        ho, io = orc.decode_nmsq(c.bg, c.Z, host_llr[k][:m].astype(np.float64), 8 + (k % 5), n_layers=c.n_layers,
                                 early_term=bool(k % 4), **rule_kw(c))
        assert (out_h[k][:m].cpu().numpy() == ho).all() and (out_i[k][:m].cpu().numpy() == io).all(), (k, c.bg, c.Z)
    # no iteration counts requested, and an empty configuration in the middle
    out2 = [torch.zeros_like(h) for h in ref_h]
    ns2 = list(ns); ns2[3] = 0
    pkg.decode_multi_dev(codecs, [x.data_ptr() for x in llrs], ns2, [x.data_ptr() for x in out2], None, s)
    torch.cuda.synchronize()
    assert all((out2[k] == ref_h[k]).all() for k in range(len(codecs)) if k != 3) and int(out2[3].sum()) == 0
    for c in codecs:
        c.close()


def test_pool_of_logical_shards_equals_one_launch(pkg, orc):
    """nrldpc_pool_*: N handles + N host threads + a chunk queue (the in-library multi-GPU dispatcher of SURVEY 8e),
    exercised on one GPU with four logical shards: hard bits and iteration counts of the pooled decode equal one
    plain decode of the same batch, every codeword is decoded exactly once, and with early termination the work
    split is whatever the queue gave (cfg5-like R = 8/9 code, half of the codewords hopeless so shards run unevenly)."""
    rng = np.random.default_rng(8)
    bg, Z, B = 1, 384, 3001
    c = pkg.Codec(bg, Z, max_iter=25, n_layers=5, early_term=True, llr_dtype=np.float32)
    info = rng.integers(0, 2, (B, c.K), dtype=np.uint8)
    llr = awgn_llr(rng, c.encode(info), 7.5, np.float32, Z, E=9478)
    llr[B // 2:] *= 0.05                                      # second half: never converges, 25 iterations each
    ref_h, ref_i = c.decode(llr, want_iters=True)
    c.close()
    pool = pkg.CodecPool(bg, Z, [0, 0, 0, 0], chunks_per_device=4, max_iter=25, n_layers=5, early_term=True,
                         llr_dtype=np.float32)
    for _ in range(2):                                        # second call: the pool is reusable
        h, it = pool.decode(llr, want_iters=True)
        assert (h == ref_h).all() and (it == ref_i).all()
        split = pool.last_split()
        assert sum(split) == B and len(split) == 4 and min(split) > 0
    assert pool.decode(llr[:1]).shape == (1, 22 * Z) and pool.decode(llr[:0]).shape == (0, 22 * Z)
    pool.close()
    with pytest.raises(pkg.NRLDPCError):
        pkg.CodecPool(bg, Z, [0, 99])                         # a device that does not exist: the whole pool fails


def test_abi_revision_6_through_the_python_binding(pkg, orc):
    """What ABI revision 6 added, through ldpc-3gpp-matlab_amd/_capi.py (tests/abi_caller covers the C side): the layer count as an argument
    of ONE call (nothing sticks to the handle), the phase times of a chunked host-pointer call, per-shard kernel times of a pool."""
    rng = np.random.default_rng(66)
    bg, Z, B, act = 1, 384, 1200, 9
    c = pkg.Codec(bg, Z, max_iter=12, n_layers=0, early_term=True, llr_dtype=np.float32)
    info = rng.integers(0, 2, (B, c.K), dtype=np.uint8)
    llr = awgn_llr(rng, c.encode(info), 7.5, np.float32, Z, E=(22 + act - 2) * Z)  # nothing transmitted above `act` rows
    assert c.last_host_phases() is None                                   # no chunked call yet
    full, it_full = c.decode_packed(llr, want_iters=True)                 # the handle's own setting: every row
    assert c.last_layers() == 46
    ph = c.last_host_phases()
    assert ph is not None and ph["chunks"] >= 1 and ph["layers"] == 46 and ph["copy_quantise_ms"] > 0 and ph["copy_out_ms"] >= 0
    per_call, it_call = c.decode_packed(llr, want_iters=True, n_layers=act)
    assert c.last_layers() == act and c.last_host_phases()["layers"] == act
    again = c.decode_packed(llr)                                          # ... and the next call without a count: every row again
    assert c.last_layers() == 46 and (again == full).all()
    c.set_layers(act)
    sticky, it_sticky = c.decode_packed(llr, want_iters=True)
    assert (sticky == per_call).all() and (it_sticky == it_call).all()    # the per-call count == nrldpc_set_layers(act)
    auto = c.decode_packed(llr, n_layers=pkg._capi.LAYERS_AUTO)
    assert c.last_layers() == act and (auto == per_call).all()
    with pytest.raises(pkg.UnsupportedParameters):
        c.decode_packed(llr, n_layers=3)
    c.close()
    bits = np.unpackbits(per_call, axis=1, bitorder="little")[:, : 22 * Z]
    assert (bits == info).mean() > 0.999
    pool = pkg.CodecPool(bg, Z, [0, 0], chunks_per_device=2, max_iter=12, n_layers=act, early_term=True, llr_dtype=np.float32)
    with pytest.raises(pkg.NRLDPCError):
        pool.last_kernel_ms()                                             # timing not enabled
    pool.set_timing(True)
    h = pool.decode(llr)
    ms = pool.last_kernel_ms()
    assert len(ms) == 2 and all(m >= 0 for m in ms) and sum(ms) > 0 and (np.packbits(h, axis=1, bitorder="little") == per_call).all()
    pool.close()


def test_pool_decode_dev_shards_device_resident_batches(pkg, orc):
    """nrldpc_pool_decode_dev: per-shard DEVICE pointers -- data that is already on the GPUs goes
    through the pool without touching the host; one launching thread and stream per shard.  Three logical shards on this
    box's GPU with unequal slices (one empty) against one plain decode; unmeasured on more than one GPU."""
    import torch
    rng = np.random.default_rng(81)
    bg, Z, B = 2, 384, 1200
    c = pkg.Codec(bg, Z, max_iter=20, early_term=True, llr_dtype=np.float16)
    info = rng.integers(0, 2, (B, c.K), dtype=np.uint8)
    llr = awgn_llr(rng, c.encode(info), -2.5, np.float16, Z)
    ref_h, ref_i = c.decode(llr, want_iters=True)
    c.close()
    pool = pkg.CodecPool(bg, Z, [0, 0, 0, 0], max_iter=20, early_term=True, llr_dtype=np.float16)
    cuts = [0, 500, 500, 1100, 1200]                          # shard 1 has nothing to do
    d_llr = [torch.from_numpy(llr[cuts[i]:cuts[i + 1]]).cuda() for i in range(4)]
    d_hard = [torch.full((cuts[i + 1] - cuts[i], c.K), 7, dtype=torch.uint8, device="cuda") for i in range(4)]
    d_it = [torch.zeros(cuts[i + 1] - cuts[i], dtype=torch.int32, device="cuda") for i in range(4)]
    torch.cuda.synchronize()
    for _ in range(2):
        pool.decode_dev([t.data_ptr() for t in d_llr], [t.shape[0] for t in d_llr], [t.data_ptr() for t in d_hard],
                        [t.data_ptr() for t in d_it])
        got_h = torch.cat(d_hard).cpu().numpy()
        got_i = torch.cat(d_it).cpu().numpy()
        assert (got_h == ref_h).all() and (got_i == ref_i).all()
        assert pool.last_split() == [500, 0, 600, 100]
    pool.close()


def test_bench_two_ranks_sharing_one_gpu():
    """bench.py launched the way the driver launches the scaling run (torch.distributed.run, one rank per 'GPU'), with
    both ranks on this box's single GPU and gloo for the barrier / max-over-ranks (RCCL refuses two ranks on one
    device): the real decode path under the multi-rank protocol.  Weak scaling: 2 x batch codewords in the job."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--batch", "1024", "--backend", "gloo", "--share-gpu", "--cfg5-total", "4096", "--cfg5-steps", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 1.0 and rec["bler"] < 0.05 and "cpu_baseline" not in rec
    assert len(rec["roofline"]["per_gpu"]) == 2 and all(g["kernel_ms"] > 0 for g in rec["roofline"]["per_gpu"])
    # the strong-scaled leg of BASELINE configs[4]: the job's 4096 codewords cut over the two ranks, every one decoded
    leg = rec["cfg5_strong"]
    assert leg["scaling"] == "strong" and leg["n_gpus"] == 2 and leg["value"] > 1.0
    assert [g["codewords"] for g in leg["per_gpu"]] == [2048, 2048]
    assert all(g["bler"] < 0.05 and 1.0 <= g["mean_iterations"] < 10.0 and 0.0 < g["hbm_frac"] < 1.0 for g in leg["per_gpu"])
    assert rec["comm"]["backend"] == "gloo" and rec["comm"]["world_size"] == 2 and rec["comm"]["fallback"] is None
    # ... and launched PLAINLY: `python bench.py --gpus 2` starts its two ranks itself
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "512",
                        "--backend", "gloo", "--share-gpu", "--cfg5-total", "2048", "--cfg5-steps", "1"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["comm"]["world_size"] == 2 and rec["value"] > 1.0 and len(rec["roofline"]["per_gpu"]) == 2
    # without --share-gpu a box with one GPU cannot run two ranks: refused with the device count, no line
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=600, cwd=root, env=env)
    import torch
    if torch.cuda.device_count() < 2:
        assert p.returncode != 0 and "HIP device(s) visible" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_bench_survives_an_rccl_that_does_not_come_up_and_has_a_one_process_route():
    """VERDICT r5 item 9.  (a) The driver's form -- default `--backend nccl` -- with two ranks on this box's ONE GPU: RCCL cannot build a
    communicator for two ranks on one device, which is exactly the kind of bring-up failure a first 8-GPU run could meet; the probe
    fails (or times out) on every rank, the barrier and the max-over-ranks go over gloo, the real decode path runs, and the line says
    so.  (b) `--gpus 2 --in-process`: one process, nrldpc_pool_decode_dev, no process group at all -- the same line by another route,
    per-GPU kernel times from the library's event pairs on each shard's own stream."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "512",
                        "--share-gpu", "--cfg5-total", "2048", "--cfg5-steps", "1"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 1.0 and rec["bler"] < 0.05 and len(rec["roofline"]["per_gpu"]) == 2
    assert rec["comm"]["requested"] == "nccl" and rec["comm"]["world_size"] == 2
    assert rec["comm"]["backend"] in ("gloo", "nccl")  # gloo with a reason on a one-GPU box; nccl if this RCCL does accept it
    if rec["comm"]["backend"] == "gloo":
        assert rec["comm"]["fallback"]
    assert [g["codewords"] for g in rec["cfg5_strong"]["per_gpu"]] == [1024, 1024]
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--in-process", "--share-gpu", "--steps", "4",
                        "--warmup", "1", "--batch", "1024"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 1.0 and rec["bler"] < 0.05 and rec["scaling"] == "weak"
    assert rec["comm"]["in_process"] is True and rec["comm"]["shards"] == 2 and rec["comm"]["backend"] is None
    assert len(rec["roofline"]["per_gpu"]) == 2 and all(g["kernel_ms"] > 0 for g in rec["roofline"]["per_gpu"])
    assert rec["ms_per_step"] > 0 and rec["roofline"]["kernel_ms"] > 0


def test_bench_over_a_probed_rccl_group_with_one_rank():
    """The branch the driver's 8-GPU run takes when RCCL comes up: gloo default group, RCCL probed in a thread (second process group + one
    all-reduce on the rank's GPU), then the barrier, the max-over-ranks and the per-GPU gathers over that group.  A one-GPU box cannot hold
    two RCCL ranks, but it can hold ONE: BENCH_COMM_WORLD1=1 makes a single rank under the launcher go through the whole protocol, so every
    torch.distributed call of that branch has run on the real backend before the first multi-GPU job does."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}, BENCH_COMM_WORLD1="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "512",
           "--cpu-sample", "0", "--no-e2e", "--no-early-term", "--cfg5", "--cfg5-total", "1024", "--cfg5-steps", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 1.0 and rec["comm"]["requested"] == "nccl" and rec["comm"]["world_size"] == 1
    assert rec["comm"]["backend"] == "nccl" and rec["comm"]["fallback"] is None and rec["comm"]["rccl_version"], rec["comm"]
    assert rec["cfg5_strong"]["per_gpu"][0]["codewords"] == 1024 and len(rec["roofline"]["per_gpu"]) == 1


def test_cfg5_full_batch_65536_through_eight_shards(pkg, orc):
    """BASELINE configs[4] at its full size: 65536 BG1 Z=384 R=8/9 codewords with early termination, cut over EIGHT
    shards (the pool dispatcher with eight handles and host threads; on this one-GPU box all eight sit on device 0).
    Size-independent properties: every codeword decoded exactly once and back to its payload (BLER < 2 %), iteration
    counts in range, a sample equal to the oracle, the work split covers the batch."""
    rng = np.random.default_rng(65536)
    bg, Z, B, E = 1, 384, 65536, 9478
    enc = pkg.Codec(bg, Z, max_iter=1, llr_dtype=np.float16)
    pool = pkg.CodecPool(bg, Z, [0] * 8, chunks_per_device=3, max_iter=25, n_layers=5, early_term=True, llr_dtype=np.float16)
    info = rng.integers(0, 2, (B, enc.K), dtype=np.uint8)
    llr = np.empty((B, enc.N_cw), np.float16)
    for lo in range(0, B, 8192):                       # chunked: the float noise of the whole batch would be 7 GB
        llr[lo:lo + 8192] = awgn_llr(rng, enc.encode(info[lo:lo + 8192]), 7.5, np.float16, Z, E=E)
    hard, it = pool.decode(llr, want_iters=True)
    split = pool.last_split()
    pool.close(); enc.close()
    assert sum(split) == B and len(split) == 8 and min(split) > 0
    assert (hard != info).any(1).mean() < 0.02 and it.min() >= 1 and it.max() <= 25 and it.mean() < 8
    a, b = pkg.default_rule(bg, 5)
    sample = rng.choice(B, 6, replace=False)
    ho, io = orc.decode_nmsq(bg, Z, llr[sample].astype(np.float64), 25, n_layers=5, early_term=True, alpha=a, beta=b * 8)
    assert (hard[sample] == ho).all() and (it[sample] == io).all()
