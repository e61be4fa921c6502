cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)" > gpurun_out/s14_host.txt
for i in 1 2 3 4; do
  for th in 4 8 16; do
  NRLDPC_HOST_THREADS=$th python tools/bench_host_path.py --big-only 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('PIN   run$i th$th', d['llr_dtype'], 'min %.2f med %.2f max %.2f'%(d['ms_min'],d['ms_median'],d['ms_max']))" >> gpurun_out/s14_host.txt
  done
  NRLDPC_HOST_THREADS=8 NRLDPC_HOST_NO_PIN=1 python tools/bench_host_path.py --big-only 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('NOPIN run$i th8', d['llr_dtype'], 'min %.2f med %.2f max %.2f'%(d['ms_min'],d['ms_median'],d['ms_max']))" >> gpurun_out/s14_host.txt
done
cat gpurun_out/s14_host.txt
