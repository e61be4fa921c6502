#!/usr/bin/env python3
"""Per-stage timing of the device chains either side of the decoder core
on one MI355X, with each stage's algorithmic HBM bytes and its fraction of the 8 TB/s roofline.
Writes gpurun_out/bench_chain.json.  Stage kernels run on torch's current stream, so torch events see them.

Algorithmic bytes (every stage is a pure gather / scatter, HBM-bound):
  rate_recover  read G f32 LLRs + write C*(N+2Z) fp16 core inputs          (+ C*N_cb f32 read+write with HARQ)
  crc_check     read C*K hard bits (bytes) + write B bytes + flags
  crc_attach    read A bytes + write C*K bytes
  encode        read C*K bytes + write C*(N+2Z) bytes
  rate_match    read G selected code bits + write G bytes
  awgn_llr      read G bytes + write G f32 LLRs (noise is generated in registers)
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
capi = importlib.import_module("ldpc-3gpp-matlab_amd._capi")
PEAK = 8.0e12


def timed(fn, reps=10, inner=8):
    """Device time per call: `inner` calls queued back to back between one event pair (so that the ~20 us a Python ->
    ctypes -> hipLaunchKernel round trip takes is hidden behind the previous kernel, as it is in a pipeline), median over
    `reps` such groups.  (Round 2 timed single calls: the short stage kernels read 2x their kernel-trace durations.)"""
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); e1.synchronize()
        ms.append(e0.elapsed_time(e1) / inner)
    return float(np.median(ms))


def run(name, n_tb, harq, **props):
    p = pkg.NRLDPC(**props)
    p.validate()
    t = capi.tb_params(p)
    C, K, Z, N, G, A, B = p.C, p.K, p.Z_c, p.N, p.G, p.A, p.B
    ncwz = N + 2 * Z
    s = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    a = torch.randint(0, 2, (n_tb, A), generator=gen, device="cuda", dtype=torch.uint8)
    c = torch.empty((n_tb * C, K), device="cuda", dtype=torch.uint8)
    cw = torch.empty((n_tb * C, ncwz), device="cuda", dtype=torch.uint8)
    g = torch.empty((n_tb, G), device="cuda", dtype=torch.uint8)
    codec = pkg.Codec(p.BG, Z, max_iter=1, llr_dtype=np.float16)
    out = []

    def rec(stage, ms, nbytes):
        r = {"config": name, "stage": stage, "n_tb": n_tb, "C": C, "Z": Z, "G": G, "ms": ms,
             "algorithmic_bytes": int(nbytes), "GB_s": nbytes / ms / 1e6, "frac_of_8TBs": nbytes / ms / 1e6 / 8000.0,
             "tb_per_s": n_tb / ms * 1e3, "payload_Gbit_s": n_tb * A / ms / 1e6}
        print(r, flush=True)
        out.append(r)

    rec("crc_attach", timed(lambda: capi.crc_attach_dev(t, a.data_ptr(), n_tb, c.data_ptr(), s)), n_tb * (A + C * K))
    rec("encode", timed(lambda: codec.encode_dev(c.data_ptr(), n_tb * C, cw.data_ptr(), s)), n_tb * C * (K + ncwz))
    rec("rate_match", timed(lambda: capi.rate_match_dev(t, cw.data_ptr(), n_tb, g.data_ptr(), s)), n_tb * 2 * G)
    g_ch = torch.empty((n_tb, G), device="cuda", dtype=torch.float32)
    rec("awgn_llr (modulate + AWGN + exact LLR)", timed(lambda: capi.awgn_llr_dev(g.data_ptr(), n_tb * G, props["Q_m"], 3.0, 11, 0,
                                                                            g_ch.data_ptr(), s)), n_tb * 5 * G)
    g_tilde = (1.0 - 2.0 * g.float()) * 4.0
    llr = torch.empty((n_tb * C, ncwz), device="cuda", dtype=torch.float16)
    rec("rate_recover", timed(lambda: capi.rate_recover_dev(t, g_tilde.data_ptr(), n_tb, None, llr.data_ptr(),
                                                            capi.LLR_F16, s)), n_tb * (4 * G + 2 * C * ncwz))
    if harq:
        hb = torch.zeros((n_tb, C, p.N_cb), device="cuda", dtype=torch.float32)
        rec("rate_recover+harq", timed(lambda: capi.rate_recover_dev(t, g_tilde.data_ptr(), n_tb, hb.data_ptr(),
                                                                     llr.data_ptr(), capi.LLR_F16, s)),
            n_tb * (4 * G + 2 * C * ncwz + 8 * C * p.N_cb))
    b_hat = torch.empty((n_tb, B), device="cuda", dtype=torch.uint8)
    ok = torch.empty(n_tb, device="cuda", dtype=torch.int32)
    c_hat = c  # noise-free hard decisions
    rec("crc_check", timed(lambda: capi.crc_check_dev(t, c_hat.data_ptr(), n_tb, b_hat.data_ptr(), ok.data_ptr(),
                                                      None, s)), n_tb * (C * K + B + 4))
    assert bool((ok != 0).all()) and bool((b_hat[:, :A] == a).all()), "chain round trip failed"
    codec.close()
    # whole receive chain as NRLDPCDecoder.step runs it: g_tilde -> rate recovery -> decode (parity-check stop) -> CRC
    DC = importlib.import_module("ldpc-3gpp-matlab_amd.device_chain")
    esn0 = 13.0 if props["Q_m"] == 6 else (-0.5 if props["BG"] == 1 else (0.0 if Z > 64 else 2.5))
    H = importlib.import_module("ldpc-3gpp-matlab_amd.harness")
    N0 = 10 ** (-esn0 / 10)
    tx = H.modulate_t(g, props["Q_m"])
    rx = tx + (N0 / 2) ** 0.5 * torch.view_as_complex(torch.randn(tx.shape + (2,), device="cuda", dtype=torch.float64, generator=gen))
    g_noisy = H.demodulate_llr_t(rx, props["Q_m"], N0).float().contiguous()
    chain = DC.DeviceDecodeChain(p, iterations=25, llr_dtype=np.float16)
    a_hat, okc, iters = chain.step(g_noisy)
    ms = timed(lambda: chain.step(g_noisy), reps=5)
    r = {"config": name, "stage": "receive chain (rate recovery + decode with parity-check stop + CRC)", "n_tb": n_tb, "C": C, "Z": Z,
         "G": G, "ms": ms, "EsN0_dB": esn0, "tb_ok_fraction": float(okc.float().mean()), "mean_iters": float(iters.float().mean()),
         "tb_per_s": n_tb / ms * 1e3, "payload_Gbit_s": n_tb * A / ms / 1e6}
    print(r, flush=True)
    out.append(r)
    chain.close()
    return out


def main():
    res = []
    res += run("cfg2 BG1 A=8424 R=1/3 QPSK, 4096 transport blocks", 4096, True, BG=1, A=8424, G=25272, Q_m=2)
    res += run("cfg3 BG2 A=3824 R=1/3 QPSK, 4096 transport blocks", 4096, False, BG=2, A=3824, G=11472, Q_m=2)
    res += run("BG1 A=25344 C=4 64QAM R~1/2, 1024 transport blocks", 1024, True, BG=1, A=25344, G=50688 + 12, Q_m=6)
    res += run("cfg1 BG2 A=100 R=1/3 QPSK, 65536 transport blocks", 65536, False, BG=2, A=100, G=300, Q_m=2)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_chain%s.json" % os.environ.get("OUT_SUFFIX", "")), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
