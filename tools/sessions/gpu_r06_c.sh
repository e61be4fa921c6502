#!/bin/bash
# round 6, call c: input-driven rate recovery (A/B), parity-pass read batching (A/B), row-form slot refill (experiment units), f64 leg probe
mkdir -p gpurun_out/r06c; cd /root/repo; O=gpurun_out/r06c
timeout 1500 python -m pytest tests/test_chain_gpu.py tests/test_testbench_gpu.py tests/test_harness_gpu.py tests/test_decode_gpu.py tests/test_refill_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee $O/tests.txt
for v in 1 0 1 0; do NRLDPC_RR_SCATTER=$v OUT_SUFFIX=_scatter$v python tools/bench_chain.py > $O/chain_$v.log 2>&1; python - <<PY
import json
d=json.load(open('gpurun_out/bench_chain_scatter$v.json'))
for r in d:
    if 'rate_recover' in r['stage'] or 'receive chain' in r['stage']: print('scatter=$v', r['config'][:34], r['stage'][:22], round(r['ms'],4), round(r.get('frac_of_8TBs',0),3))
PY
done 2>&1 | tee $O/chain_ab.txt
timeout 600 python tools/probe_f64_leg.py 2>&1 | grep "^{" | tee $O/f64_probe.txt
for lib in "" exp_libs/lib_pb0.so "" exp_libs/lib_pb0.so; do
  NRLDPC_LIB=${lib:+/root/repo/$lib} timeout 600 python tools/exp_row_refill.py 2,384 2,288 2,192 2,144 1,320 1,384 2>&1 | grep "^{\|PARITY\|Error\|error"
done | tee $O/parity_batch_ab.txt
# last: the experiment that could hang (barrier counts of the refilling row-form kernels)
for lib in exp_libs/lib_rowrefill.so exp_libs/lib_rowrefill.so; do
  NRLDPC_LIB=/root/repo/$lib timeout 300 python tools/exp_row_refill.py 2,384 2,288 2,192 2,144 1,320 2>&1 | grep "^{\|PARITY\|Error\|error"
done | tee $O/row_refill.txt
