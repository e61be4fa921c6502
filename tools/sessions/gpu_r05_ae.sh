#!/bin/bash
mkdir -p gpurun_out/r05ae; cd /root/repo
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05ae/sweep_profile.txt
import cProfile, pstats, importlib, sys, tempfile, time, io
sys.path.insert(0, "/root/repo")
H = importlib.import_module("ldpc-3gpp-matlab_amd.harness")
kw = dict(target_block_errors=100, EsN0_start=-1.5, EsN0_delta=0.25, batch=16384)
with tempfile.TemporaryDirectory() as d:
    H.plot_BLER_vs_SNR(results_dir=d, device=True, **kw)   # warm
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter(); H.plot_BLER_vs_SNR(results_dir=d, device=True, **kw); dt = time.perf_counter() - t0
    pr.disable()
print("sweep %.2f s" % dt)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
PY
