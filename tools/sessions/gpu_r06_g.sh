#!/bin/bash
# round 6, call g: the copy threads poll 300 us between the jobs of a call (in-call only).  The default bench three times in a row on one box
# (as profiles/r05_e2e_run_to_run.txt), then three times with NRLDPC_HOST_SPIN_US=0 (round 5's behaviour), interleaved; host-path tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06g; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_layers_gpu.py tests/test_abi_caller_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2 3; do for sp in 300 0; do
  NRLDPC_HOST_SPIN_US=$sp python bench.py --cpu-sample 0 --no-early-term --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/line_${sp}_$rep.json
  python - <<PY
import json
d=json.load(open('$O/line_${sp}_$rep.json'))
e=d['e2e']
print('spin $sp rep $rep value %.2f' % d['value'], ' '.join('%s %.2f/%.2f(q%.2f)' % (k[:12], v['ms_median'], v['ms_max'], v['phases_of_the_median_call']['copy_quantise_ms']) for k,v in e.items() if isinstance(v,dict) and 'ms_median' in v), 'dram %.0f' % e['host_dram_read']['GB_per_s'], 'r89', ' '.join('%.2f' % v['ms_median'] for v in e['r89_active_layers'].values() if isinstance(v,dict) and 'ms_median' in v))
PY
done; done 2>&1 | tee $O/run_to_run.txt
