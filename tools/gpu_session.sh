cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_chain_gpu.py tests/test_harness_gpu.py tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/s9_tests.txt
cat gpurun_out/s9_tests.txt
python tools/bench_one.py 1 384 4096 0 0 25 2>&1 | grep -v amdgpu
python tools/bench_host_path.py --sweep > gpurun_out/s9_host_sweep.txt 2>&1
grep -h batch gpurun_out/s9_host_sweep.txt | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['llr_dtype'], 'th',d['host_threads'],'mb',d['chunk_mb'], 'min %.2f med %.2f max %.2f first %.1f'%(d['ms_min'],d['ms_median'],d['ms_max'],d['ms_first_call']))"
