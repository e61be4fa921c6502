#!/bin/bash
# cfg4: class edges of the shared launches; kernel timeline of one call
mkdir -p gpurun_out/r05o; cd /root/repo
O=gpurun_out/r05o/multi_edges.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout 300 python tools/probe_multi.py 2>&1 | grep "^call" | tail -5 >> $O; }
run NRLDPC_MULTI_CLASSES=0
run A=1
run NRLDPC_MULTI_CLASS_EDGES=128,256,512
run NRLDPC_MULTI_CLASS_EDGES=192,256,512
run NRLDPC_MULTI_CLASS_EDGES=192,256,384,512
run NRLDPC_MULTI_CLASS_EDGES=192,256,320,512
run NRLDPC_MULTI_CLASS_EDGES=256,320,512
run NRLDPC_MULTI_CLASS_EDGES=256,448
run NRLDPC_MULTI_CLASS_EDGES=256
run NRLDPC_MULTI_CLASS_EDGES=192,256,384,512 NRLDPC_MULTI_STREAMS=7 GPU_MAX_HW_QUEUES=8
run NRLDPC_MULTI_Z64_MIN_ROWS=24576
run NRLDPC_MULTI_Z64_MIN_ROWS=49152
run A=1
cat $O
export TMPDIR=/tmp
for c in 0 1; do
  (cd /tmp && NRLDPC_MULTI_CLASSES=$c rocprofv3 --kernel-trace --output-format csv -d /tmp/tl$c -o tl -- python /root/repo/tools/probe_multi.py > /tmp/tl$c.log 2>&1)
  python tools/multi_timeline.py /tmp/tl$c > gpurun_out/r05o/timeline_classes$c.txt 2>&1
  cat gpurun_out/r05o/timeline_classes$c.txt
done
