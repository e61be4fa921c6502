#!/bin/bash
# round 6, call d: the split form against the row form under the parity stop AT THE WATERFALL for the five sizes that hold 2-4 codewords per
# workgroup in the row form (A/B units built with -DNRLDPC_Z64_AB: NRLDPC_SPLIT=0/1); chain tests with the shipped rate-recovery dispatch
mkdir -p gpurun_out/r06d; cd /root/repo; O=gpurun_out/r06d
for sp in 0 1 0 1; do
  NRLDPC_SPLIT=$sp NRLDPC_LIB=/root/repo/exp_libs/lib_ab.so timeout 600 python tools/exp_row_refill.py 2,384 2,288 2,192 2,144 1,320 2>&1 | grep "^{\|PARITY\|rror" | sed "s/^{/{\"split\": $sp, /"
done | tee $O/split_vs_row_stop.txt
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_testbench_gpu.py tests/test_harness_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
python tools/bench_chain.py 2>&1 | grep "rate_recover\|receive chain" | cut -c1-200 | tee $O/chain.txt
