#!/usr/bin/env python3
"""The reference's top-level script on the GPU: harness.plot_BLER_vs_SNR (plot_BLER_vs_SNR.m:1-171, no figure) with every stage on the
device, at the script's own defaults (A = 3842, R = 1/3, BG2, QPSK, 8 iterations, target BLER 1e-3; :29-41) but from -1.5 dB on in steps of 0.25 dB and with 100 block errors
per point instead of 3, and for the headline code (A = 8424, R = 1/3, BG1, 25 iterations).  Prints the result files (the reference's
format: Es/N0 <tab> BLER per line) and the wall time of each sweep."""
import importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = importlib.import_module("ldpc-3gpp-matlab_amd.harness")
for name, kw in (("the script's defaults, 100 errors per point", dict(target_block_errors=100, EsN0_start=-1.5, EsN0_delta=0.25, batch=16384)),
                 ("headline code: A = 8424, R = 1/3, BG1, 25 iterations, 100 errors per point",
                  dict(A=8424, BG=1, iterations=25, target_block_errors=100, EsN0_start=-2.0, EsN0_delta=0.1, batch=16384))):
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        curves = H.plot_BLER_vs_SNR(results_dir=d, device=True, **kw)
        dt = time.perf_counter() - t0
        (key, pts), = curves.items()
        print("== %s: %d points, %d transport blocks, %.1f s" % (name, len(pts), sum(p[2] for p in pts), dt))
        for f in sorted(os.listdir(d)):
            print("-- results/%s" % f)
            sys.stdout.write(open(os.path.join(d, f)).read())
