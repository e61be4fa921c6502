#!/bin/bash
# round 5, session b: (1) ABI revision 5 tests (layers per call / auto), abi_caller; (2) the byte-per-bit stall in the Python context
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05b; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$PWD/ldpc-3gpp-matlab_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 1500 python -m pytest tests/test_layers_gpu.py tests/test_abi_caller_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; tail -15 $O/tests.txt
for v in fresh reuse fresh+del; do
  NRLDPC_HOST_TRACE=1 timeout 200 python tools/host_stall/py_probe.py $v f16 10 > $O/py_$v.txt 2>&1
done
NRLDPC_HOST_TRACE=1 timeout 200 python tools/host_stall/py_probe.py fresh f16 10 notorch > $O/py_fresh_notorch.txt 2>&1
NRLDPC_HOST_TRACE=1 MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=4294967296 timeout 200 python tools/host_stall/py_probe.py fresh f16 10 > $O/py_fresh_nommap.txt 2>&1
NRLDPC_HOST_TRACE=1 timeout 200 python tools/host_stall/py_probe.py fresh f64 8 > $O/py_fresh_f64.txt 2>&1
grep -h "^call" $O/py_*.txt | head -80
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/tools/host_stall/py_probe.py fresh f16 8 > $GRAFT_REPO_ROOT/$O/trace_run.txt 2>&1
cd $GRAFT_REPO_ROOT; du -sh $O
