cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/gputests.log; cat gpurun_out/gputests.log
python tools/bench_host_path.py > gpurun_out/hostpath.log 2>&1
{ echo "# nrldpc_decode host path, BG1 Z=384 batch 4096, early termination on, 12 calls per line, four separate processes (round 2, later session: int8 on the wire, copy threads on the NUMA node that holds the caller's array; MI355X host: 2 x EPYC 9575F, 2 NUMA nodes).  ms min / median / max";
  for r in 1 2 3 4; do python tools/bench_host_path.py --big-only 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('run$r %-8s min %.2f med %.2f max %.2f' % (r['llr_dtype'], r['ms_min'], r['ms_median'], r['ms_max']))"; done;
  echo "# the same with the native format on the wire (NRLDPC_HOST_I8=0)";
  for r in 1 2; do NRLDPC_HOST_I8=0 python tools/bench_host_path.py --big-only 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('run$r %-8s min %.2f med %.2f max %.2f' % (r['llr_dtype'], r['ms_min'], r['ms_median'], r['ms_max']))"; done; } > gpurun_out/host_path_variance.txt 2>&1
cat gpurun_out/host_path_variance.txt
python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_line.json
python tools/bench_configs.py > gpurun_out/cfg.log 2>&1
python tools/bench_chain.py > gpurun_out/chain.log 2>&1
bash tools/profile_gpu.sh r02b > gpurun_out/profile.log 2>&1
ls gpurun_out/prof_r02b | head
