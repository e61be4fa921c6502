#!/bin/bash
# round 5, session f: the whole GPU suite with durations (the driver's limit is 1200 s), and where test_interleaved_block_geometry spends its time
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05f; mkdir -p $O; rm -rf $O/*
( time timeout 3000 python -m pytest tests -q -m gpu --durations=30 2>&1 | tail -50 ) > $O/gputests.log 2>&1; tail -45 $O/gputests.log
timeout 900 python -m cProfile -o $O/ilv.prof -m pytest tests/test_decode_gpu.py -q -m gpu -k "interleaved_block_geometry and 1" > $O/ilv_prof.log 2>&1
python - <<'PY' > gpurun_out/r05f/ilv_prof.txt 2>&1
import pstats
p = pstats.Stats("gpurun_out/r05f/ilv.prof"); p.sort_stats("cumulative").print_stats(35)
PY
head -70 $O/ilv_prof.txt | cut -c1-160
