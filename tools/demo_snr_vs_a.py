#!/usr/bin/env python3
"""The reference's second top-level script on the GPU: harness.plot_SNR_vs_A (plot_SNR_vs_A.m:1-194, no figure) with every stage on the device, at
the script's own defaults (A = 1000:1000:8000, R = 1/3, BG1, QPSK, 50 iterations, 100 block errors per point, target BLER 1e-2, from -2 dB in steps
of 0.1 dB, seed 0; :38-48), then the same sweep for BG2's range and for three rates.  Prints the result files (the reference's format:
A <tab> Es/N0 per line) and the wall time of each sweep."""
import importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = importlib.import_module("ldpc-3gpp-matlab_amd.harness")
for name, kw in (("the script's defaults", dict()),
                 ("BG2, A = 500:500:3500", dict(A=list(range(500, 4000, 500)), BG=2, EsN0_start=-3.0)),
                 ("BG1, R = 1/2, 2/3, 5/6", dict(R=[1 / 2, 2 / 3, 5 / 6], EsN0_start=0.0))):
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        out = H.plot_SNR_vs_A(results_dir=d, device=True, batch=8192, **kw)
        dt = time.perf_counter() - t0
        print("== %s: %d (A, R) pairs, %.1f s" % (name, sum(len(v) for v in out.values()), dt))
        for f in sorted(os.listdir(d)):
            print("-- results/%s" % f)
            sys.stdout.write(open(os.path.join(d, f)).read())
