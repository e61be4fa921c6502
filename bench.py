#!/usr/bin/env python3
"""Headline benchmark: decoded information Gbit/s, BG1 Z=384 (K=8448) R=1/3, 25 iterations.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (the decode kernel behind nrldpc_decode_dev, i.e. the
replacement of step(hLDPCDecoder, cw_tilde) at NRLDPCDecoder.m:265) over one batch of 4096 synthetic
codewords per GPU, LLRs already resident in HBM (fp16).  Codeword batches shard across GPUs with no
data-path collective (weak scaling: 4096 codewords per GPU); torch.distributed is used only for the
barrier and the max-over-ranks timing.  Rank 0 prints ONE JSON line.

Timing: ONE loop.  The K timed launches are bracketed by barrier + synchronize (wall clock -> value,
ms_per_step) and each launch is additionally bracketed by a HIP event pair recorded on the launch
stream (-> roofline.kernel_ms = mean of the K event durations of the same K launches).

roofline leads with the bound that binds: VALU issue.  achieved = wave64 VALU instructions per launch
(rocprofv3 PMC summary under profiles/, used only when its nrldpc_kernel_id equals the loaded library's)
/ kernel_ms, peak = 1024 SIMDs x 2.4 GHz / 2 cycles per op, frac <= 1; `cycle_weighted` prices every
opcode of the iteration loop at its measured issue interval (half of them take 4 cycles on gfx950);
`lds` is the LDS array's busy fraction.  roofline.context keeps what the north star's HBM framing
gives: the SURVEY 8(d) streaming-model bytes at the kernel's own storage width (s = 1 byte per
message: int8) against 8 TB/s -- above 1, because a codeword stays in LDS/VGPRs for all iterations --
and the MEASURED HBM bytes per launch (`traffic`), which equal the compulsory input + output.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BG, Z, ITERS, BATCH = 1, 384, 25, 4096
KB, NCOLS, NNZ = 22, 68, 316
K = KB * Z                     # 8448 information bits per codeword
N_CW = NCOLS * Z               # 26112 LLRs per codeword
E_TX = 25344                   # transmitted bits at R = 1/3 (rv0, no repetition): all 66*Z of N
ESN0_DB = -0.5                 # QPSK/AWGN operating point (plot_BLER_vs_SNR.m:105-106)
S_BYTES = 1                    # storage bytes per message in the algorithmic-bytes model = the kernel's int8
ALG_BYTES_PER_CW = ITERS * 4 * S_BYTES * NNZ * Z + N_CW * S_BYTES + K // 8  # 12 161 568
HBM_PEAK_GBS = 8000.0
VALU_PEAK_WAVE_INSTS = 1024 * 2.4e9 / 2  # 256 CUs x 4 SIMDs, one wave64 VALU op per 2 cycles (MI355X_MICROARCH.md)
# rocprofv3 summaries of this very command (tools/profile_gpu.sh + tools/summarise_profile.py); newest round first
PROFILE_TAGS = ("r06", "r05", "r04", "r03", "r02", "r01")


def _profile(kernel_id):
    """(tag, pmc summary, traffic, isa mix) of the newest committed profile of this bench whose kernels are the loaded
    library's (nrldpc_kernel_id); (None, {}, {}, None) when there is none -- then no instruction-count-based fraction is
    reported rather than one of another build."""
    for tag in PROFILE_TAGS:
        p = os.path.join(ROOT, "profiles", tag + "_bench_pmc_summary.json")
        if not os.path.exists(p):
            continue
        try:
            pmc = json.load(open(p))
            if pmc.get("_nrldpc_kernel_id") != kernel_id:
                continue
            tf = os.path.join(ROOT, "profiles", tag + "_traffic_bytes_per_launch.json")
            traffic = json.load(open(tf)) if os.path.exists(tf) else {}
            mixp = os.path.join(ROOT, "profiles", tag + "_headline_isa_mix.json")
            mix = json.load(open(mixp)) if os.path.exists(mixp) else None
            if mix and mix.get("nrldpc_kernel_id") != kernel_id:
                mix = None
            return tag, pmc, traffic, mix
        except Exception:
            pass
    return None, {}, {}, None


def usable_cpus():
    """CPUs this process may really use: the cgroup quota when there is one (the MI355X boxes show 256 CPUs and grant 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def bler_match():
    """Second half of the metric ('BLER match vs MATLAB ref'; the decoder arithmetic of the reference is closed source, so
    this is a match to a restatement of its documented algorithm -- PARITY UNPINNED): the dB gap to
    flooding sum-product at equal iteration caps at BLER 0.1, 0.01 and 0.001, and -- recorded, not bounded -- to the reference's
    default of 50 sweeps (NRLDPCDecoder.m:41); measured by tests/test_bler_gap_gpu.py on identical noise, committed under
    profiles/."""
    for tag in PROFILE_TAGS:
        p = os.path.join(ROOT, "profiles", tag + "_bler_gap.json")
        if os.path.exists(p):
            try:
                d = json.load(open(p))
                head = next(v for k, v in d.items() if "headline" in k and "gap_dB" in v)
                out = {"parity": "unpinned (closed-source reference decoder; compared with a restatement of its documented algorithm)",
                       "headline_gap_dB": head["gap_dB"], "EsN0_at_bler_0.1_gpu": head["EsN0_at_bler_0.1_gpu"],
                       "EsN0_at_bler_0.1_sum_product": head["EsN0_at_bler_0.1_sum_product"], "blocks": head["blocks"],
                       "worst_gap_dB_all_configs": max(v["gap_dB"] for k, v in d.items() if k.startswith("cfg") and v.get("gap_dB") is not None),
                       "bound_dB": head["bound_dB"],
                       "source": "profiles/%s_bler_gap.json (tests/test_bler_gap_gpu.py)" % tag}
                for k in ("gap_dB_at_bler_0.01", "bound_dB_at_bler_0.01", "blocks_at_bler_0.01",
                          "gap_dB_vs_50_sum_product_sweeps_at_bler_0.01",
                          "gap_dB_at_bler_0.001", "bound_dB_at_bler_0.001", "blocks_at_bler_0.001"):  # 1e-3: plot_BLER_vs_SNR.m:38
                    if k in head:
                        out["headline_" + k] = head[k]
                # round 4: the reference's own operating points (tests/test_bler_gap_gpu.py, same file)
                at = {}
                for k, v in d.items():
                    if k.endswith("50it") and v.get("gap_dB_at_bler_0.01") is not None:  # 50 iterations against 50 sweeps (NRLDPCDecoder.m:41)
                        at["50_vs_50_iterations: " + k] = {"gap_dB_at_bler_0.01": v["gap_dB_at_bler_0.01"], "bound_dB": v.get("bound_dB")}
                    if k.startswith("demo") and v.get("gap_dB") is not None:  # plot_BLER_vs_SNR.m:29-41 through the harness, 8 vs 8
                        at[k] = {"gap_dB_at_bler_0.1": v["gap_dB"], "bound_dB": v.get("bound_dB")}
                    if "grid_cost" in v:
                        at["cost_of_the_8_bit_grid: " + k] = {"dB_at_bler_0.01": v["grid_cost"].get("cost_dB_of_the_8_bit_grid_at_bler_0.01")}
                if at:
                    out["at_the_reference_settings"] = at
                return out
            except Exception:
                pass
    return None


def synth_llr(torch, codec, batch, seed, dev):
    """Random payloads -> GPU encoder -> QPSK/AWGN LLRs, fp16, first 2Z columns punctured (=0)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    info = torch.randint(0, 2, (batch, K), generator=g, device=dev, dtype=torch.uint8)
    cw = torch.empty((batch, N_CW), device=dev, dtype=torch.uint8)
    codec.encode_dev(info.data_ptr(), batch, cw.data_ptr(), torch.cuda.current_stream().cuda_stream)
    mu = 2.0 * 10.0 ** (ESN0_DB / 10.0)  # LLR mean for unit-power QPSK, N0 = 10^(-EsN0/10)
    noise = torch.randn((batch, N_CW), generator=g, device=dev, dtype=torch.float32)
    llr = (1.0 - 2.0 * cw.to(torch.float32)) * mu + (2.0 * mu) ** 0.5 * noise
    llr[:, : 2 * Z] = 0.0
    llr[:, 2 * Z + E_TX:] = 0.0
    torch.cuda.synchronize()
    return info, llr.to(torch.float16).contiguous()


CFG5_TOTAL, CFG5_LAYERS, CFG5_E, CFG5_ESN0 = 65536, 5, 9478, 7.5  # BASELINE.json configs[4]: BG1 Z=384 R=8/9, early termination


def cfg5_strong_leg(torch, nrldpc, comm, args, world, rank, local_rank, dev, valu_insts_per_edge_iter):
    """BASELINE.json configs[4] next to the headline when the job has more than one rank (or --cfg5): 65536 BG1 Z=384 R=8/9
    codewords (5 active rows, 27 of the 68 columns transmitted), early termination, STRONG-scaled -- the total is fixed and
    rank r decodes the contiguous slice [r*total/N, (r+1)*total/N) on its own GPU with its own handle, no data-path
    collective.  Timed like the headline: barrier + synchronize on both sides, max over ranks; every rank's
    kernel time (event pairs on the launch stream) is gathered so that the line carries a per-GPU figure.  Returns the
    leg's record on rank 0, None elsewhere."""
    total = args.cfg5_total
    lo, hi = importlib.import_module("ldpc-3gpp-matlab_amd.shard").shard_range(total, rank, world)
    n = hi - lo
    codec = nrldpc.Codec(BG, Z, max_iter=ITERS, n_layers=CFG5_LAYERS, early_term=True, llr_dtype=np.float16, device_id=local_rank)
    g = torch.Generator(device=dev)
    llr = torch.empty((n, N_CW), device=dev, dtype=torch.float16)
    info = torch.empty((n, K), device=dev, dtype=torch.uint8)
    mu = 2.0 * 10.0 ** (CFG5_ESN0 / 10.0)
    for c0 in range(0, n, 4096):  # chunked: the fp32 noise of a whole shard would be several GB
        c1 = min(n, c0 + 4096)
        g.manual_seed(0xC0DE + 5 + (lo + c0))  # a function of the GLOBAL codeword index: the same job for every N
        info[c0:c1] = torch.randint(0, 2, (c1 - c0, K), generator=g, device=dev, dtype=torch.uint8)
        cw = torch.empty((c1 - c0, N_CW), device=dev, dtype=torch.uint8)
        codec.encode_dev(info[c0:c1].data_ptr(), c1 - c0, cw.data_ptr(), torch.cuda.current_stream().cuda_stream)
        x = (1.0 - 2.0 * cw.to(torch.float32)) * mu + (2.0 * mu) ** 0.5 * torch.randn((c1 - c0, N_CW), generator=g, device=dev)
        x[:, : 2 * Z] = 0.0
        x[:, 2 * Z + CFG5_E:] = 0.0
        llr[c0:c1] = x.to(torch.float16)
    hard = torch.empty((n, K), device=dev, dtype=torch.uint8)
    its = torch.empty(n, device=dev, dtype=torch.int32)
    tstream = torch.cuda.current_stream()

    def step():
        if n:
            codec.decode_dev(llr.data_ptr(), n, hard.data_ptr(), its.data_ptr(), None, tstream.cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        comm.barrier()
        torch.cuda.synchronize()
    steps = max(1, args.cfg5_steps)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for _ in range(2):
        step()
    barrier()
    t0 = time.perf_counter()
    for e0, e1 in ev:
        e0.record(tstream)
        step()
        e1.record(tstream)
    barrier()
    elapsed = time.perf_counter() - t0
    kms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
    mean_it = float(its.float().mean().item()) if n else 0.0
    bler = float((hard != info).any(dim=1).float().mean().item()) if n else 0.0
    rows = comm.gather([elapsed, kms, mean_it, bler, float(n)])  # five numbers per rank: the leg's bookkeeping, not its data path
    codec.close()
    if rank != 0:
        return None
    elapsed = max(r[0] for r in rows)
    edges5 = 79  # base-graph edges of BG1's rows 0..4
    cols_in = KB + 4 + (CFG5_LAYERS - 4)  # columns the 5 active rows reach
    per_gpu = []
    for r in rows:
        ni, ms = int(r[4]), r[1]
        io_bytes = ni * (cols_in * Z * 2 + K)  # fp16 LLRs of the columns read + one byte per hard bit
        rec = {"codewords": ni, "kernel_ms": ms, "mean_iterations": r[2], "bler": r[3],
               "value": ni * K / ms / 1e6 if ms > 0 else None, "unit": "Gbit/s",
               "hbm_frac": (io_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else None}
        if valu_insts_per_edge_iter and ms > 0:
            insts = ni * r[2] * edges5 * (Z / 64.0) * valu_insts_per_edge_iter
            rec["valu_issue_frac_estimate"] = insts / (ms * 1e-3) / VALU_PEAK_WAVE_INSTS
        per_gpu.append(rec)
    return {"workload": "BASELINE configs[4]: BG1 Z=384 R=8/9 (5 active rows, E=%d), <=%d iterations with the parity-check stop, "
                        "%d codewords in the job, QPSK/AWGN Es/N0=%.1f dB" % (CFG5_E, ITERS, total, CFG5_ESN0),
            "scaling": "strong", "n_gpus": world, "steps": steps, "ms_per_step": elapsed / steps * 1e3,
            "value": total * steps * K / elapsed / 1e9, "unit": "Gbit/s", "per_gpu": per_gpu,
            "note": "contiguous slices of the job per rank, no data-path collective; value = whole job / max-over-ranks wall time; "
                    "per_gpu.hbm_frac = compulsory input + output bytes / kernel time / 8 TB/s; valu_issue_frac_estimate prices "
                    "the EXECUTED iterations at the headline kernel's measured VALU instructions per edge and iteration (the parity "
                    "pass and the prologue are not in it: an under-estimate)"}


def cpu_baseline(llr_host_f64, info_host, rule):
    """Reference-semantics CPU path (flooding sum-product, double, parity-check early stop, the
    comm.LDPCDecoder configuration of NRLDPCDecoder.m:120) restated in oracle/.  `value` is the single-thread
    figure (MATLAB's one-codeword step()); BASELINE.md section 2's other rows ride along: the same decoder on
    all host cores (cpu_ref_bp_mt) and the build's own algorithm on all cores (cpu_nms_mt).  Bounded samples."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    cores = usable_cpus()  # threads actually used = what the cgroup grants, not what the host shows
    n1 = min(llr_host_f64.shape[0], 256)
    O.lib().orc_set_threads(1)
    t0 = time.perf_counter()
    hard, iters = O.decode_bp_flood(BG, Z, llr_host_f64[:n1], ITERS, nthreads=1)
    dt = time.perf_counter() - t0
    ok = int((hard == info_host[:n1]).all(axis=1).sum())
    n = llr_host_f64.shape[0]
    t0 = time.perf_counter()
    hard_mt, iters_mt = O.decode_bp_flood(BG, Z, llr_host_f64, ITERS, nthreads=cores)
    dt_mt = time.perf_counter() - t0
    assert (hard_mt[:n1] == hard).all()
    O.lib().orc_set_threads(cores)
    t1 = time.perf_counter()
    O.decode_nmsq(BG, Z, llr_host_f64, ITERS, early_term=False, alpha=rule[0], beta=rule[1] * 8)
    dt_nms = time.perf_counter() - t1
    return {
        "value": n1 * K / dt / 1e9, "unit": "Gbit/s", "cores": 1, "kind": "port",
        "sample": "%d codewords of the same workload, flooding BP double, <=%d sweeps with parity-check stop "
                  "(mean %.1f sweeps), %d/%d blocks correct, %.1f s" % (n1, ITERS, float(iters.mean()), ok, n1, dt),
        "host_cores_usable": cores, "host_cores_visible": os.cpu_count(),
        "cpu_ref_bp_mt": {"value": n * K / dt_mt / 1e9, "unit": "Gbit/s", "cores": cores,
                          "sample": "%d codewords, same decoder, one codeword per thread, %.2f s" % (n, dt_mt)},
        "cpu_nms_mt": {"value": n * K / dt_nms / 1e9, "unit": "Gbit/s", "cores": cores,
                       "sample": "%d codewords, the build's layered offset min-sum (oracle), %d iterations, no early "
                                 "stop, %.2f s" % (n, ITERS, dt_nms)},
    }


def e2e_host_path(nrldpc, info_host, llr_host_f16, rule, reps=7):
    """PCIe-inclusive rate through the host-pointer entry points (pageable arrays as a MEX gateway would hand them over), same
    codewords, 25 iterations, no early stop; median of `reps` calls per boundary dtype after ONE untimed call of the same size
    (the first call of a size allocates the pinned slots and device staging).  `f16` / `f32_matlab_single` / `f64_matlab_double`:
    nrldpc_decode_packed (bit-packed hard decisions, what matlab/nrldpc_mex.cpp calls); `*_byte_per_bit`: nrldpc_decode.  The
    output array is allocated once per leg, as the gateway's is (a std::vector of the call) and as a caller that decodes batch
    after batch does; `f16_byte_per_bit_fresh_output_array` is round 4's way -- a new 35 MB numpy array every call, mmap'd,
    page-faulted by the copy threads and munmap'd each time -- which is where that round's "25-35 ms stall in every second call"
    came from (profiles/r05_host_stall.txt).  Never `value`."""
    out = {}
    legs = (("f16", np.float16, True, True), ("f32_matlab_single", np.float32, True, True), ("f64_matlab_double", np.float64, True, True),
            ("f16_byte_per_bit", np.float16, False, True), ("f64_byte_per_bit", np.float64, False, True),
            ("f16_byte_per_bit_fresh_output_array", np.float16, False, False))
    for name, dt, packed, reuse in legs:
        c = nrldpc.Codec(BG, Z, max_iter=ITERS, n_layers=0, early_term=False, llr_dtype=dt, alpha=rule[0], beta=rule[1])
        x = llr_host_f16.astype(dt)
        call = c.decode_packed if packed else c.decode
        buf = np.empty((x.shape[0], (K + 7) // 8 if packed else K), np.uint8) if reuse else None
        if buf is not None:
            buf[:] = 0  # touched once, like any array a caller has used before
        call(x, out=buf)
        ts, phases = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            h = call(x, out=buf)
            ts.append(time.perf_counter() - t0)
            phases.append(c.last_host_phases())  # where the caller's thread spent this call (nrldpc_last_host_phases)
        c.close()
        med_i = int(np.argsort(ts)[len(ts) // 2])
        if packed:
            h = np.unpackbits(h, axis=1, bitorder="little")[:, :K]
        assert (h == info_host).all(axis=1).mean() > 0.99
        ts.sort()
        n = x.shape[0]
        out[name] = {"ms_median": ts[len(ts) // 2] * 1e3, "ms_min": ts[0] * 1e3, "ms_max": ts[-1] * 1e3,
                     "value": n * K / ts[len(ts) // 2] / 1e9, "unit": "Gbit/s", "runs": reps,
                     "host_bytes_in": int(x.nbytes), "host_bytes_out": int(n * ((K + 7) // 8 if packed else K)),
                     "phases_of_the_median_call": phases[med_i], "leg_order": len(out)}
    out["r89_active_layers"] = e2e_active_layers(nrldpc, reps)
    out["host_dram_read"] = host_dram_read(llr_host_f16.astype(np.float64))
    d = out["host_dram_read"]
    out["f64_matlab_double"]["read_bound_ms"] = out["f64_matlab_double"]["host_bytes_in"] / (d["GB_per_s"] * 1e9) * 1e3
    return out


def host_dram_read(x, threads=15):
    """What THIS box's host memory gives the copy threads: the 856 MB double array of the f64 leg summed by `threads` threads
    (numpy releases the GIL inside sum), best of 3.  The f64 leg cannot be faster than bytes / this rate -- the boxes the builder saw
    ranged from 90 to 250 GB/s, which is the whole spread of that leg (4.0 - 10 ms)."""
    from concurrent.futures import ThreadPoolExecutor
    flat = x.reshape(-1)
    cuts = [len(flat) * i // threads for i in range(threads + 1)]
    best = None
    with ThreadPoolExecutor(threads) as ex:
        for _ in range(3):
            t0 = time.perf_counter()
            list(ex.map(lambda i: float(flat[cuts[i]:cuts[i + 1]].sum()), range(threads)))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    return {"bytes": int(flat.nbytes), "threads": threads, "seconds": best, "GB_per_s": flat.nbytes / best / 1e9}


def e2e_active_layers(nrldpc, reps=7, n=4096):
    """What ABI revision 5 gives the reference's seam (NRLDPCDecoder.m:265: the decoder sees cw_tilde and nothing else): BG1 Z=384
    at R = 8/9 (BASELINE configs[4]: G = 9478, 27 of 68 columns transmitted, 5 of 46 rows active), MATLAB doubles through
    nrldpc_decode_packed with the parity stop, (a) every row of H as the reference decodes it and as the gateway did until
    revision 4, (b) NRLDPC_LAYERS_AUTO -- the count read off the LLRs, what `nrldpc_mex('create', BG, Z_c, iterations)` now asks
    for, (c) the count given explicitly.  Same hard decisions in all three (asserted); PCIe-inclusive times."""
    E, esn0 = 9478, 7.5
    c = nrldpc.Codec(BG, Z, max_iter=ITERS, n_layers=0, early_term=True, llr_dtype=np.float64)
    rng = np.random.default_rng(89)
    info = rng.integers(0, 2, (n, K), dtype=np.uint8)
    cw = c.encode(info)
    mu = 2 * 10 ** (esn0 / 10)
    x = (1 - 2.0 * cw) * mu + np.sqrt(2 * mu) * rng.standard_normal(cw.shape)
    x[:, : 2 * Z] = 0
    x[:, 2 * Z + E:] = 0
    buf = np.zeros((n, (K + 7) // 8), np.uint8)
    res, ref = {}, None
    # what the patched NRLDPCDecoder.LDPC_coding hands the gateway with every call (matlab/ldpc-3gpp-matlab.patch: active_layers from
    # E_r, k_0, N_cb; here through the Python mirror of the same arithmetic) -- the handle itself stays under AUTO, as 'create' leaves it
    per_call = nrldpc.NRLDPC(BG=BG, A=8424, G=E, Q_m=2).active_layers()
    for name, nl, call_nl in (("all_rows_as_the_reference", 0, None), ("auto", -1, None), ("explicit_5_rows", 5, None),
                              ("gateway_default", -1, per_call)):
        c.set_layers(nl)
        c.decode_packed(x, out=buf, n_layers=call_nl)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            c.decode_packed(x, out=buf, n_layers=call_nl)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        h = np.unpackbits(buf, axis=1, bitorder="little")[:, :K]
        if ref is None:
            ref = h.copy()
        res[name] = {"ms_median": ts[len(ts) // 2] * 1e3, "ms_min": ts[0] * 1e3, "ms_max": ts[-1] * 1e3, "layers": c.last_layers(),
                     "value": n * K / ts[len(ts) // 2] / 1e9, "unit": "Gbit/s", "block_errors": int((h != info).any(1).sum()),
                     "same_bits_as_all_rows": bool((h == ref).all())}
    c.close()
    res["speedup_auto_over_all_rows"] = res["all_rows_as_the_reference"]["ms_median"] / res["auto"]["ms_median"]
    res["gateway_default"]["per_call_count"] = per_call
    res["gateway_default"]["note"] = ("nrldpc_decode_packed_layers with the count the patched System object derives from its own parameters "
                                     "(ABI revision 6): no scan of the array, nothing sticks to the handle")
    res["speedup_gateway_default_over_all_rows"] = res["all_rows_as_the_reference"]["ms_median"] / res["gateway_default"]["ms_median"]
    res["config"] = "BG1 Z=384 R=8/9 (G=9478), %d codewords of MATLAB doubles, parity stop, Es/N0 %.1f dB" % (n, esn0)
    return res


class Comm:
    """What the ranks of a job say to each other: a barrier, the maximum of a number, a few numbers per rank.  The DATA path has no
    collective (codeword batches shard), so nothing here may be able to lose the run: the default process group is
    always gloo (TCP on 127.0.0.1), and RCCL -- `--backend nccl`, the default -- is a second group that is PROBED (communicator
    init + one all-reduce on the rank's GPU, in a thread with a time limit); only when every rank's probe succeeded do the
    barrier and the reductions go over it.  Otherwise the line says so: comm.backend "gloo", comm.requested "nccl",
    comm.fallback = the reason (RCCL has never run with more than one rank on this pool's boxes)."""

    def __init__(self, torch, backend, world, rank, local_rank, use_gpu, probe_timeout=90.0):
        self.torch, self.world, self.rank, self.dist, self.pg, self.dev = torch, world, rank, None, None, None
        self.stuck = False  # a probe thread that never came back: leave through os._exit
        self.info = {"backend": None, "world_size": world}
        # BENCH_COMM_WORLD1=1 (test aid, under a launcher with one rank): run the whole protocol -- gloo group, RCCL probe, barrier and
        # gathers over the probed group -- with a single rank, so that the branch an 8-GPU run takes when RCCL DOES come up is executed
        # on a one-GPU box too (tests/test_full_size_gpu.py)
        if world == 1 and not os.environ.get("BENCH_COMM_WORLD1"):
            return
        import datetime
        import torch.distributed as dist
        self.dist = dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=max(600.0, 4 * probe_timeout)))
        self.info = {"backend": "gloo", "world_size": dist.get_world_size(), "requested": backend, "fallback": None}
        if backend != "nccl":
            return
        ok, why = self._probe_rccl(local_rank, use_gpu, probe_timeout)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # over gloo: every rank learns whether EVERY rank's probe succeeded
        if int(flag.item()) == 1:
            self.pg, self.dev = self._pg, torch.device("cuda", local_rank)
            self.info.update({"backend": "nccl", "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())})
        else:
            self.info["fallback"] = why or "the RCCL probe failed on another rank"

    def _probe_rccl(self, local_rank, use_gpu, limit):
        import threading
        torch, dist = self.torch, self.dist
        if not use_gpu or not torch.cuda.is_available():
            return False, "no HIP device in this process"
        # a collective that times out must come back to THIS code as an error (or not at all: the join below has its own limit), not as
        # the watchdog's default reaction -- tearing the whole process down -- or the fall back could never run
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        res = {}

        def run():
            try:
                import datetime
                torch.cuda.set_device(local_rank)  # the current device is per thread
                pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=limit))
                t = torch.ones(1, device=torch.device("cuda", local_rank))
                dist.all_reduce(t, group=pg)
                torch.cuda.synchronize()
                if int(t.item()) != self.world:
                    raise RuntimeError("all-reduce of ones gave %r for %d ranks" % (t.item(), self.world))
                res["pg"] = pg
            except BaseException as e:  # noqa: BLE001 -- whatever RCCL raises, the bench goes on over gloo
                res["err"] = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(limit + 15.0)
        if th.is_alive():
            self.stuck = True
            return False, "the RCCL probe did not return within %.0f s" % (limit + 15.0)
        if "pg" in res:
            self._pg = res["pg"]
            return True, None
        return False, res.get("err", "unknown RCCL failure")

    def barrier(self):
        if self.dist is None:
            return
        if self.pg is not None:
            self.dist.barrier(group=self.pg, device_ids=[self.dev.index])
        else:
            self.dist.barrier()

    def gather(self, values):
        """values: a few floats of this rank -> [[...] per rank] on every rank (bookkeeping, never the data path)."""
        if self.dist is None:
            return [[float(v) for v in values]]
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=self.dev if self.pg is not None else "cpu")
        allr = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(allr, t, group=self.pg)
        return [[float(v) for v in r.tolist()] for r in allr]

    def max(self, x):
        return max(r[0] for r in self.gather([x]))

    def close(self):
        if self.dist is None:
            return
        self.barrier()
        if self.stuck:  # a thread is still inside RCCL: a normal interpreter exit would wait for it
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        self.dist.destroy_process_group()


def dry_run(args, torch):
    """The multi-rank protocol of main() with the GPU work replaced by a sleep: exercised by the CPU test suite
    (tests/test_dist_cpu.py) so that the torch.distributed branch of this file -- Comm, with `--backend nccl` its probe and the
    fall back to gloo -- is executed somewhere other than the driver's 8-GPU run.  Prints a line marked dry_run; no measurement."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    comm = Comm(torch, args.backend, world, rank, int(os.environ.get("LOCAL_RANK", "0")), use_gpu=False)
    assert comm.info["world_size"] == args.gpus
    for _ in range(args.warmup):
        time.sleep(0.001)
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))
    comm.barrier()
    elapsed = comm.max(time.perf_counter() - t0)
    # the per-GPU bookkeeping of the cfg5_strong leg: a few numbers per rank
    leg = None
    if world > 1:
        lo, hi = importlib.import_module("ldpc-3gpp-matlab_amd.shard").shard_range(args.cfg5_total, rank, world)
        comm.barrier()
        t0 = time.perf_counter()
        time.sleep(0.001 * (rank + 1))
        comm.barrier()
        rows = comm.gather([time.perf_counter() - t0, float(hi - lo)])
        leg = {"dry_run": True, "scaling": "strong", "n_gpus": world,
               "per_gpu": [{"codewords": int(r[1]), "seconds": r[0]} for r in rows]}
    if rank == 0:
        print(json.dumps({"metric": "decoded info Gbit/s @ BG1 Z=384 R=1/3, 25 iters; BLER match vs MATLAB ref",
                          "dry_run": True, "value": None, "unit": "Gbit/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "scaling": "weak",
                          "comm": comm.info, "cfg5_strong": leg}), flush=True)
    comm.close()


def check_or_start_ranks(args):
    """`--gpus N` means N ranks, one per GPU (plot_BLER_vs_SNR.m:23-27: "parallel instances ... aggregated").  Launched the
    driver's way (torch.distributed.run sets WORLD_SIZE) the rank count must BE N: a line that says n_gpus: 1 for --gpus 8 is
    refused, not printed.  Launched plainly -- `python bench.py --gpus 8`, WORLD_SIZE unset -- this process starts the N ranks
    itself (re-exec under torch.distributed.run on 127.0.0.1, a free port) and hands their output and exit code through."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (python -m torch.distributed.run "
                             "--nproc-per-node %d ... bench.py --gpus %d), or run `python bench.py --gpus %d` without a launcher "
                             "and it starts the ranks itself" % (args.gpus, env_world, args.gpus, args.gpus, args.gpus))
        return
    if args.gpus == 1:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: starting %d ranks: %s" % (args.gpus, args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd))


def build_line(nrldpc, args, n_gpus, batch, elapsed, kms, kms_all, rule, bler, early, cfg5, comm_info):
    """The JSON line of rank 0 (or of the one process of --in-process): value from the wall clock of the timed loop, the
    roofline object from the event-pair kernel times of the same launches and the committed rocprofv3 summary of this build."""
    kernel_ms = float(np.mean(kms))
    kid = nrldpc.load().nrldpc_kernel_id().decode()
    bid = nrldpc.load().nrldpc_build_id().decode()
    tag, pmc, tr, mix = _profile(kid)
    value = n_gpus * batch * args.steps * K / elapsed / 1e9
    traffic = tr.get("hbm_bytes_per_launch")
    scale = batch / float(BATCH)  # the committed profile is of the default batch
    if traffic is not None:
        traffic = traffic * scale

    def c(name):
        v = pmc.get(name, {}).get("mean_per_launch")
        return None if v is None else v * scale
    compulsory = batch * (N_CW * 2 + K)
    alg_gbs = batch * ALG_BYTES_PER_CW / (kernel_ms * 1e-3) / 1e9
    roof = {"bound": "valu_issue", "achieved": None, "peak": VALU_PEAK_WAVE_INSTS, "unit": "wave64 VALU instructions/s",
            "frac": None, "traffic": traffic, "kernel": pmc.get("_kernel"), "kernel_ms": kernel_ms,
            "kernel_ms_median": float(np.median(kms)), "kernel_ms_min": float(np.min(kms)),
            # the same kernel's average under rocprofv3 --kernel-trace (first launch left out), from the committed profile
            # of this build: tracing adds a few per cent; kernel_ms above is what every fraction of this line divides by
            "kernel_ms_rocprofv3_avg": (pmc.get("_kernel_trace", {}).get("avg_ns_without_first_launch") or 0.0) * 1e-6 * (batch / float(BATCH)) or None}
    if c("SQ_INSTS_VALU"):
        insts = c("SQ_INSTS_VALU")
        rate = insts / (kernel_ms * 1e-3)
        wc = c("SQ_WAVE_CYCLES")
        cyc = c("GRBM_GUI_ACTIVE") / 8.0 if c("GRBM_GUI_ACTIVE") else None  # the counter is summed over the 8 XCDs
        roof.update({
            "achieved": rate, "frac": rate / VALU_PEAK_WAVE_INSTS, "insts_per_launch": insts,
            "valu_insts_per_edge_iteration": insts / (batch * ITERS * NNZ * Z / 64.0),
            "wave_cycle_split": None if not wc else {
                "issuing": c("SQ_ACTIVE_INST_ANY") / wc, "issue_stalled": c("SQ_WAIT_INST_ANY") / wc,
                "parked_at_waitcnt_or_barrier": c("SQ_WAIT_ANY") / wc},
            "lds": None if not (c("SQ_LDS_IDX_ACTIVE") and cyc) else {
                "busy_frac": c("SQ_LDS_IDX_ACTIVE") / (256 * cyc), "bank_conflict_cycles": c("SQ_LDS_BANK_CONFLICT"),
                "note": "SQ_LDS_IDX_ACTIVE / (256 CUs x busy cycles): the LDS array is not the bound either"},
            "note": "peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU op (MI355X_MICROARCH.md); instruction counts per "
                    "launch are data-independent without early termination; half of this kernel's opcodes issue at 4 cycles "
                    "on gfx950 (profiles/r03_ubench_valu_rates.txt), which cycle_weighted accounts for"})
    if mix and mix.get("valu_ns_per_iteration_all_waves_of_a_row"):
        # static: disassembly of the loaded kernels x measured per-opcode issue intervals (tools/isa_mix.py): the time the
        # launch needs if the VALU pipes never idle = VALU-ns of one iteration of one codeword (ns per row-wave set x Z/64
        # waves), x iterations x codewords, spread over the chip's 1024 SIMDs -- independent of how many codewords a CU holds
        valu_ms = mix["valu_ns_per_iteration_all_waves_of_a_row"] * (Z / 64.0) * ITERS * batch / 1024.0 * 1e-6
        roof["cycle_weighted"] = {"valu_bound_ms_per_launch": valu_ms, "frac": valu_ms / kernel_ms,
                                  "valu_ns_per_iteration_all_waves_of_a_row": mix["valu_ns_per_iteration_all_waves_of_a_row"],
                                  "source": "profiles/%s_headline_isa_mix.json" % tag}
    if roof.get("insts_per_launch"):  # every rank decodes the same number of codewords for the same 25 iterations
        roof["per_gpu"] = [{"rank": r, "kernel_ms": ms, "frac": roof["insts_per_launch"] / (ms * 1e-3) / VALU_PEAK_WAVE_INSTS}
                           for r, ms in enumerate(kms_all)]
    else:
        roof["per_gpu"] = [{"rank": r, "kernel_ms": ms, "frac": None} for r, ms in enumerate(kms_all)]
    roof["profile"] = {"tag": tag, "nrldpc_kernel_id": kid, "nrldpc_build_id": bid,
                       "matches_loaded_library": tag is not None,
                       "source": None if tag is None else "profiles/%s_bench_pmc_summary.json, profiles/%s_bench_kernel_stats.csv "
                                 "(rocprofv3 --kernel-trace --stats / --pmc, separate passes, of this command)" % (tag, tag)}
    roof["context"] = {
        "hbm_streaming_model": {
            "bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_codeword": ALG_BYTES_PER_CW, "storage_bytes_per_message": S_BYTES,
            "note": "SURVEY 8(d): bytes a decoder that streams a-posteriori values and messages through HBM every layer "
                    "would move, at this kernel's storage width (int8); above 1 because a codeword stays in LDS/VGPRs for "
                    "all %d iterations -- not a fraction of anything this kernel is bound by" % ITERS},
        "hbm_measured": None if not traffic else {
            "bytes_per_launch": traffic, "achieved": traffic / (kernel_ms * 1e-3) / 1e9, "unit": "GB/s",
            "frac_of_peak": traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "ratio_to_compulsory": traffic / compulsory, "compulsory_bytes_per_launch": compulsory,
            "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate --pmc passes: every LLR read once, every "
                    "hard bit written once, nothing else"}}
    return {
        "metric": "decoded info Gbit/s @ BG1 Z=384 R=1/3, 25 iters; BLER match vs MATLAB ref",
        "value": value, "unit": "Gbit/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "comm": comm_info,
        "vs_baseline": None, "dtype": "i8 messages / integer-valued f32 a-posteriori (fp16 LLR input)",
        "data": "synthetic",
        "config": {"workload": "BG1 Z=384 (K=8448) R=1/3, 25 layered min-sum iterations, no early termination, "
                               "batch=%d codewords per GPU, QPSK/AWGN Es/N0=%.1f dB" % (batch, ESN0_DB),
                   "bg": BG, "Z": Z, "iterations": ITERS, "batch_per_gpu": batch, "n_layers": 46,
                   "check_node_rule": {"alpha": rule[0], "beta_llr": rule[1], "source": "nrldpc_default_rule (cfg.alpha = 0)"},
                   "sharding": "codeword batches per GPU, no collective"},
        "roofline": roof,
        "bler": bler,
        "bler_match": bler_match(),
        "early_term": early,
        "cfg5_strong": cfg5,
    }


def valu_insts_per_edge_iteration(nrldpc, batch):
    kid = nrldpc.load().nrldpc_kernel_id().decode()
    _, pmc, _, _ = _profile(kid)
    insts = (pmc.get("SQ_INSTS_VALU", {}).get("mean_per_launch") or 0.0) * (batch / float(BATCH))
    return insts / (batch * ITERS * NNZ * Z / 64.0) if insts else None


def in_process(args, torch, nrldpc):
    """`--gpus N --in-process`: ONE process drives the N devices through nrldpc_pool_decode_dev (one handle, one host thread and
    one private stream per device inside the library; plot_BLER_vs_SNR.m:23-27 runs "parallel instances" by hand) -- no
    launcher, no process group, nothing that can fail to initialise.  Same workload per GPU, same timed region (K pool calls
    bracketed by a synchronize of every device), same JSON line; per-GPU kernel times are the library's event pairs on each
    shard's own launch stream (nrldpc_pool_last_kernel_ms).  --share-gpu: every shard on device 0 (the 1-GPU test aid)."""
    n = args.gpus
    ids = [0] * n if args.share_gpu else list(range(n))
    if not args.share_gpu and torch.cuda.device_count() < n:
        raise SystemExit("bench.py: --gpus %d --in-process but only %d HIP device(s) visible" % (n, torch.cuda.device_count()))
    batch = args.batch
    pool = nrldpc.CodecPool(BG, Z, ids, chunks_per_device=1, max_iter=ITERS, n_layers=0, early_term=False, llr_dtype=np.float16)
    pool.set_timing(True)
    infos, llrs, hards, rule = [], [], [], None
    for i, d in enumerate(ids):
        with torch.cuda.device(d):
            dev = torch.device("cuda", d)
            enc = nrldpc.Codec(BG, Z, max_iter=ITERS, n_layers=0, early_term=False, llr_dtype=np.float16, device_id=d)
            rule = (enc.alpha, enc.beta)
            info, llr = synth_llr(torch, enc, batch, 0xC0DE + 1 + i, dev)  # the seeds of ranks 0..N-1 of the launcher form
            enc.close()
            infos.append(info); llrs.append(llr); hards.append(torch.empty((batch, K), device=dev, dtype=torch.uint8))
    d_llr, d_hard, counts = [x.data_ptr() for x in llrs], [x.data_ptr() for x in hards], [batch] * n

    def sync_all():
        for d in set(ids):
            torch.cuda.synchronize(d)
    for _ in range(args.warmup):
        pool.decode_dev(d_llr, counts, d_hard)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pool.decode_dev(d_llr, counts, d_hard)  # returns when every shard's stream is idle
    sync_all()
    elapsed = time.perf_counter() - t0
    # event-pair kernel times: of the last timed call per shard, plus a second, untimed pass that reads them after every call
    kms_rows = []
    for _ in range(args.steps):
        pool.decode_dev(d_llr, counts, d_hard)
        kms_rows.append(pool.last_kernel_ms())
    kms_all = [float(np.mean([r[i] for r in kms_rows])) for i in range(n)]
    kms = [float(np.mean(r)) for r in kms_rows]  # per call: mean over the shards
    bler = float(np.mean([float((h != i_).any(dim=1).float().mean().item()) for h, i_ in zip(hards, infos)]))
    pool.close()
    info = {"backend": None, "world_size": 1, "in_process": True, "shards": n, "devices": ids,
            "note": "one process, nrldpc_pool_decode_dev: one host thread + stream per device inside the library, no process group; "
                    "value = all shards' codewords / wall clock of the K pool calls; roofline.kernel_ms = event pairs on each shard's "
                    "own stream (a second pass of K calls: reading them costs a host synchronisation per call)"}
    out = build_line(nrldpc, args, n, batch, elapsed, kms, kms_all, rule, bler, None, None, info)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH, help="codewords per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="codewords for the all-core CPU baselines (0 = skip)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive host-path leg")
    ap.add_argument("--no-early-term", action="store_true", help="skip the extra early_term leg (reference semantics)")
    ap.add_argument("--cfg5", action="store_true", help="also run the cfg5_strong leg at N = 1 (it always runs with more than one rank)")
    ap.add_argument("--cfg5-total", type=int, default=CFG5_TOTAL, help="codewords in the WHOLE job of the cfg5_strong leg")
    ap.add_argument("--cfg5-steps", type=int, default=5, help="timed passes of the cfg5_strong leg")
    ap.add_argument("--backend", default="nccl", help="what carries the barrier / max-time of a multi-rank job: nccl (= RCCL; probed, "
                    "with gloo as the fall back: the data path needs no collective) or gloo")
    ap.add_argument("--in-process", action="store_true", help="drive the N GPUs from THIS process through nrldpc_pool_decode_dev: "
                    "no launcher, no process group (a second route to the same line)")
    ap.add_argument("--share-gpu", action="store_true", help="test aid for 1-GPU boxes: every rank / shard uses device 0")
    ap.add_argument("--dry-run", action="store_true", help="test aid: run the rank protocol (init, barrier, timed loop "
                    "bracket, max-over-ranks, one JSON line) without touching a GPU; the line says dry_run and is no result")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.in_process and os.environ.get("WORLD_SIZE") not in (None, "1"):
        raise SystemExit("bench.py: --in-process is the one-process form; do not start it under a launcher (WORLD_SIZE=%s)" % os.environ["WORLD_SIZE"])
    if not args.in_process:
        check_or_start_ranks(args)

    import torch
    torch.set_num_threads(4)  # host-side tensor copies only; idle OpenMP workers would spend the container's CPU quota
    if args.dry_run:
        return dry_run(args, torch)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and this framework has no CPU path")
    nrldpc = importlib.import_module("ldpc-3gpp-matlab_amd")
    if args.in_process:
        return in_process(args, torch, nrldpc)
    if not args.share_gpu and torch.cuda.device_count() < args.gpus:
        raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) visible (one rank per GPU; --share-gpu is the "
                         "1-GPU test aid)" % (args.gpus, torch.cuda.device_count()))

    world = int(os.environ.get("WORLD_SIZE", "1"))  # == args.gpus: check_or_start_ranks
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the barrier and two small reductions: RCCL when it comes up on every rank, gloo otherwise (the line says which)
    comm = Comm(torch, args.backend, world, rank, local_rank, use_gpu=True)
    if comm.info["world_size"] != args.gpus:
        raise SystemExit("bench.py: the process group holds %d ranks, --gpus says %d" % (comm.info["world_size"], args.gpus))

    batch = args.batch
    # alpha = 0: the C ABI's own check-node rule for this rate (nrldpc_default_rule), what a MEX gateway gets
    codec = nrldpc.Codec(BG, Z, max_iter=ITERS, n_layers=0, early_term=False, llr_dtype=np.float16,
                         device_id=local_rank)
    rule = (codec.alpha, codec.beta)
    info, llr = synth_llr(torch, codec, batch, 0xC0DE + 1 + rank, dev)
    hard = torch.empty((batch, K), device=dev, dtype=torch.uint8)
    tstream = torch.cuda.current_stream()
    stream = tstream.cuda_stream

    def step():
        codec.decode_dev(llr.data_ptr(), batch, hard.data_ptr(), None, None, stream)

    def barrier():
        torch.cuda.synchronize()
        comm.barrier()
        torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for e0, e1 in ev:  # events go onto the launch stream (torch's current stream is the one handed to the library)
        e0.record(tstream)
        step()
        e1.record(tstream)
    barrier()
    elapsed = comm.max(time.perf_counter() - t0)
    kms = [e0.elapsed_time(e1) for e0, e1 in ev]
    kernel_ms = float(np.mean(kms))

    bler = float((hard != info).any(dim=1).float().mean().item())
    # every rank's kernel time, for the per-GPU roofline fractions of the line (bookkeeping: one number per rank)
    kms_all = [r[0] for r in comm.gather([kernel_ms])]

    # Extra leg, outside the timed region: the reference's own mode -- 'Parity check satisfied' (NRLDPCDecoder.m:120) --
    # on the same LLRs (rank 0 only; never `value`, whose workload is the fixed-25 configuration BASELINE.json names)
    early = None
    if rank == 0 and not args.no_early_term:
        c_et = nrldpc.Codec(BG, Z, max_iter=ITERS, n_layers=0, early_term=True, llr_dtype=np.float16, device_id=local_rank)
        its = torch.empty(batch, device=dev, dtype=torch.int32)
        hard_et = torch.empty_like(hard)
        c_et.set_timing(True)
        ms = []
        for _ in range(7):
            c_et.decode_dev(llr.data_ptr(), batch, hard_et.data_ptr(), its.data_ptr(), None, stream)
            ms.append(c_et.last_kernel_ms())
        c_et.close()
        t_et = float(np.median(ms[2:]))
        early = {"kernel_ms": t_et, "value": batch * K / t_et / 1e6, "unit": "Gbit/s", "mean_iterations": float(its.float().mean().item()),
                 "max_iterations": ITERS, "EsN0_dB": ESN0_DB, "bler": float((hard_et != info).any(dim=1).float().mean().item()),
                 "note": "parity-check stop per codeword (the reference's only mode), same LLRs, median of 5 launches after 2"}

    # BASELINE configs[4], strong-scaled over the ranks of the job (every rank takes part: it has a barrier of its own)
    cfg5 = None
    if world > 1 or args.cfg5:
        cfg5 = cfg5_strong_leg(torch, nrldpc, comm, args, world, rank, local_rank, dev, valu_insts_per_edge_iteration(nrldpc, batch))

    if rank == 0:
        out = build_line(nrldpc, args, world, batch, elapsed, kms, kms_all, rule, bler, early, cfg5, comm.info)
        if world == 1:  # CPU baseline and host-path legs at N = 1 only
            # the host-path leg first: the all-core CPU baselines spend the process's CPU quota (the MI355X boxes grant 16
            # CPUs of the 256 they show) and a throttled process measures the throttle, not the path
            if not args.no_e2e:
                e2e = e2e_host_path(nrldpc, info.cpu().numpy(), llr.cpu().numpy(), rule)
                e2e["note"] = "host pointers in and out (PCIe + host copies included); never `value`"
            if args.cpu_sample > 0:
                n = min(args.cpu_sample, batch)
                out["cpu_baseline"] = cpu_baseline(llr[:n].double().cpu().numpy(), info[:n].cpu().numpy(), rule)
            if not args.no_e2e:
                out["e2e"] = e2e
        print(json.dumps(out), flush=True)
    codec.close()
    comm.close()


if __name__ == "__main__":
    main()
