#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05l; mkdir -p $O; rm -rf $O/*
AUTO=1 timeout 1500 python tools/fuzz_decode.py 600 71 > $O/fuzz_auto.log 2>&1; tail -2 $O/fuzz_auto.log
