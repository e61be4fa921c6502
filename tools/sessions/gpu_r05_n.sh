#!/bin/bash
# cfg4: the shared launches grouped by workgroup class (NRLDPC_MULTI_CLASSES) against one launch per base graph
mkdir -p gpurun_out/r05n; cd /root/repo
O=gpurun_out/r05n/multi_classes.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout 300 python tools/probe_multi.py 2>&1 | tail -7 >> $O; }
run NRLDPC_MULTI_CLASSES=0
run NRLDPC_MULTI_CLASSES=1
run NRLDPC_MULTI_CLASSES=0
run NRLDPC_MULTI_CLASSES=1
run NRLDPC_MULTI_CLASSES=1 NRLDPC_MULTI_STREAMS=3
run NRLDPC_MULTI_CLASSES=1 NRLDPC_MULTI_STREAMS=7 GPU_MAX_HW_QUEUES=8
run NRLDPC_MULTI_CLASSES=1 NRLDPC_MULTI_ONE_STREAM=1
cat $O
export TMPDIR=/tmp
for c in 0 1; do
  (cd /tmp && NRLDPC_MULTI_CLASSES=$c rocprofv3 --kernel-trace -d /tmp/tl$c -o tl -- python /root/repo/tools/probe_multi.py > /dev/null 2>&1)
  python tools/multi_timeline.py /tmp/tl$c > gpurun_out/r05n/timeline_classes$c.txt 2>&1
  cat gpurun_out/r05n/timeline_classes$c.txt
done
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_decode_gpu.py tests/test_layers_gpu.py -m gpu -x -q -k "multi or mixed or cfg4" 2>&1 | tail -5 | tee gpurun_out/r05n/tests.txt
python tools/bench_configs.py --only-mixed 2>&1 | tail -2 | tee gpurun_out/r05n/mixed.txt
