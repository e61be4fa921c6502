#!/usr/bin/env python3
"""Headline benchmark: decoded information Gbit/s, BG1 Z=384 (K=8448) R=1/3, 25 iterations.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (the decode kernel behind nrldpc_decode_dev, i.e. the
replacement of step(hLDPCDecoder, cw_tilde) at NRLDPCDecoder.m:265) over one batch of 4096 synthetic
codewords per GPU, LLRs already resident in HBM (fp16).  Codeword batches shard across GPUs with no
data-path collective (weak scaling: 4096 codewords per GPU); torch.distributed is used only for the
barrier and the max-over-ranks timing.  Rank 0 prints ONE JSON line.

roofline.achieved uses the ALGORITHMIC bytes of SURVEY.md section 8(d) (a streaming layered decoder
with s = 2 byte messages: 24 322 080 B per codeword) divided by the decode kernel's average duration
measured with HIP events on the launch stream inside the library.  Because this decoder keeps a
codeword on-chip for all 25 iterations, that figure may exceed the 8 TB/s HBM peak; `traffic` is the
HBM byte count per launch measured with rocprofv3 PMC counters (profiles/), null if not collected.
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
S_BYTES = 2                    # storage bytes per LLR/message in the algorithmic-bytes model
ALG_BYTES_PER_CW = ITERS * 4 * S_BYTES * NNZ * Z + N_CW * S_BYTES + K // 8  # 24 322 080
HBM_PEAK_GBS = 8000.0
# measured by tools/collect_traffic.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 corrections
# of MI355X_MICROARCH.md applied); bytes per decode launch of BATCH codewords, or None.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_bytes_per_launch.json")


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


def cpu_baseline(llr_host_f64, info_host):
    """Reference-semantics CPU path (flooding sum-product, double, parity-check early stop, the
    comm.LDPCDecoder configuration of NRLDPCDecoder.m:120) restated in oracle/, single thread like
    MATLAB's one-codeword step().  Bounded sample so the default run stays within minutes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.lib().orc_set_threads(1)
    n = llr_host_f64.shape[0]
    t0 = time.perf_counter()
    hard, iters = O.decode_bp_flood(BG, Z, llr_host_f64, ITERS, nthreads=1)
    dt = time.perf_counter() - t0
    ok = int((hard == info_host).all(axis=1).sum())
    cores = os.cpu_count() or 1
    O.lib().orc_set_threads(cores)
    t1 = time.perf_counter()
    O.decode_nmsq(BG, Z, llr_host_f64, ITERS, early_term=False)
    dt_nms = time.perf_counter() - t1
    return {
        "value": n * K / dt / 1e9, "unit": "Gbit/s", "cores": 1, "kind": "port",
        "sample": "%d codewords of the same workload, flooding BP double, <=%d sweeps with parity-check stop "
                  "(mean %.1f sweeps), %d/%d blocks correct, %.1f s" % (n, ITERS, float(iters.mean()), ok, n, dt),
        "host_cores_available": cores,
        "nms_oracle_all_cores": {"value": n * K / dt_nms / 1e9, "unit": "Gbit/s", "cores": cores,
                                 "sample": "%d codewords, layered NMS-Q oracle, 25 iterations, %.2f s" % (n, dt_nms)},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="codewords per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=384, help="codewords for the CPU baseline (0 = skip)")
    args = ap.parse_args()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and this framework has no CPU path")
    nrldpc = importlib.import_module("ldpc-3gpp-matlab_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    batch = args.batch
    codec = nrldpc.Codec(BG, Z, max_iter=ITERS, n_layers=0, early_term=False, llr_dtype=np.float16,
                         device_id=local_rank)
    info, llr = synth_llr(torch, codec, batch, 0xC0DE + 1 + rank, dev)
    hard = torch.empty((batch, K), device=dev, dtype=torch.uint8)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        codec.decode_dev(llr.data_ptr(), batch, hard.data_ptr(), None, None, stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    bler = float((hard != info).any(dim=1).float().mean().item())

    # per-launch kernel duration from HIP events recorded by the library on the launch stream
    codec.set_timing(True)
    kms = []
    for _ in range(args.steps):
        step()
        kms.append(codec.last_kernel_ms())
    codec.set_timing(False)
    kernel_ms = float(np.mean(kms))

    if rank == 0:
        value = world * batch * args.steps * K / elapsed / 1e9
        achieved = batch * ALG_BYTES_PER_CW / (kernel_ms * 1e-3) / 1e9
        traffic = None
        if os.path.exists(TRAFFIC_FILE):
            try:
                traffic = json.load(open(TRAFFIC_FILE)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "decoded info Gbit/s @ BG1 Z=384 R=1/3, 25 iters; BLER match vs MATLAB ref",
            "value": value, "unit": "Gbit/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i8 messages / integer-valued f32 a-posteriori (fp16 LLR input)",
            "data": "synthetic",
            "config": {"workload": "BG1 Z=384 (K=8448) R=1/3, 25 layered NMS iterations, no early termination, "
                                   "batch=%d codewords per GPU, QPSK/AWGN Es/N0=%.1f dB" % (batch, ESN0_DB),
                       "bg": BG, "Z": Z, "iterations": ITERS, "batch_per_gpu": batch, "n_layers": 46,
                       "sharding": "codeword batches per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "nrldpc::nrldpc_decode_z64_kernel<1, 384, 2, true, true, false>", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_codeword": ALG_BYTES_PER_CW,
                         "hbm_achieved_GBs_from_traffic": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
                         "note": "algorithmic = streaming-model bytes (SURVEY 8d, s=2); codewords stay in "
                                 "LDS/VGPRs for all iterations so frac may exceed 1; compulsory HBM I/O is "
                                 "%d B/codeword" % (N_CW * S_BYTES + K)},
            "bler": bler,
        }
        if args.cpu_sample > 0 and world == 1:  # the CPU baseline is timed at N = 1 only
            n = min(args.cpu_sample, batch)
            out["cpu_baseline"] = cpu_baseline(llr[:n].double().cpu().numpy(), info[:n].cpu().numpy())
        print(json.dumps(out), flush=True)
    codec.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
