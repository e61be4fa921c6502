#!/usr/bin/env python3
"""Throughput of the on-device Monte-Carlo loop (harness.simulate_point_device): transport blocks per
second through payload RNG -> CRC attach -> encode -> rate match -> channel -> rate recovery -> decode (parity-check stop)
-> CRC -> error count, for the headline code.  Writes gpurun_out/bench_montecarlo.json."""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ldpc-3gpp-matlab_amd")
H = importlib.import_module("ldpc-3gpp-matlab_amd.harness")
DC = importlib.import_module("ldpc-3gpp-matlab_amd.device_chain")

def run(name, batch, esn0, reps=12, **props):
    p = pkg.NRLDPC(**props); p.validate()
    tx = DC.DeviceEncodeChain(p); rx = DC.DeviceDecodeChain(p, iterations=25, I_HARQ=0)
    H.simulate_point_device([(tx, rx)], p.Q_m, esn0, (0,), batch, 1234, 0)
    torch.cuda.synchronize()
    ts, errs = [], 0
    for i in range(reps):
        t0 = time.perf_counter()
        ok = H.simulate_point_device([(tx, rx)], p.Q_m, esn0, (0,), batch, 1234, (i + 1) * batch)
        ts.append(time.perf_counter() - t0); errs += int((~ok).sum())
    tx.close(); rx.close()
    ts.sort(); med = ts[len(ts) // 2]
    rec = {"config": name, "batch": batch, "EsN0_dB": esn0, "ms_median": med * 1e3, "ms_min": ts[0] * 1e3, "ms_max": ts[-1] * 1e3,
           "transport_blocks_per_s": batch / med, "payload_Gbit_s": batch * p.A / med / 1e9, "block_errors": errs, "blocks": batch * reps}
    print(json.dumps(rec), flush=True)
    return rec

if __name__ == "__main__":
    out = [run("cfg2 BG1 A=8424 R=1/3 QPSK", 4096, -0.5, BG=1, A=8424, G=25272, Q_m=2),
           run("cfg2 BG1 A=8424 R=1/3 QPSK, batch 16384", 16384, -0.5, reps=6, BG=1, A=8424, G=25272, Q_m=2),
           run("cfg1 BG2 A=100 R=1/3 QPSK", 65536, 3.0, BG=2, A=100, G=300, Q_m=2)]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_montecarlo.json"), "w"), indent=1)
