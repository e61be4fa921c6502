#!/bin/bash
# round 5, session a: take apart the alternating 25-35 ms wait of nrldpc_decode (byte-per-bit output) -- DESIGN.md section 7
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05a; mkdir -p $O; rm -rf $O/*
export LD_LIBRARY_PATH=$PWD/ldpc-3gpp-matlab_amd:/opt/rocm/lib:$LD_LIBRARY_PATH
g++ -O2 -std=c++17 -I include tools/host_stall/stall_probe.cpp -L ldpc-3gpp-matlab_amd -lnrldpc_hip -o $O/stall_probe || exit 1
cat /sys/fs/cgroup/cpu.max > $O/cpu_max.txt 2>&1; nproc >> $O/cpu_max.txt
P=$O/stall_probe
run() { name=$1; shift; echo "== $name: $*" >> $O/runs.txt; ( env "$@" NRLDPC_HOST_TRACE=1 timeout 120 $P $ARGS ) >> $O/runs.txt 2>&1; }
ARGS="f16 0 12"; run base_f16_bytes X=1
ARGS="f64 0 12"; run base_f64_bytes X=1
ARGS="f16 1 12"; run base_f16_packed X=1
ARGS="f64 1 12"; run base_f64_packed X=1
ARGS="f16 0 12"
run threads4 NRLDPC_HOST_THREADS=4
run threads8 NRLDPC_HOST_THREADS=8
run no_sdma HSA_ENABLE_SDMA=0
run chunk8mb NRLDPC_HOST_CHUNK_MB=8
run chunk64mb NRLDPC_HOST_CHUNK_MB=64
run nopin NRLDPC_HOST_NO_PIN=1
run no_i8 NRLDPC_HOST_I8=0
run spin200 NRLDPC_HOST_SPIN_US=200
run hwq8 GPU_MAX_HW_QUEUES=8
run blocking_sync HIP_FORCE_SYNC_COPY=0 AMD_DIRECT_DISPATCH=0
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- $GRAFT_REPO_ROOT/$P f16 0 8 > $GRAFT_REPO_ROOT/$O/trace_run.txt 2>&1
cd $GRAFT_REPO_ROOT
find $O/trace -name '*.csv' | head; du -sh $O
tail -5 $O/runs.txt
