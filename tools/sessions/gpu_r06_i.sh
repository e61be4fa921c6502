#!/bin/bash
# round 6, call i: the whole GPU suite and the default bench line on the tree as committed (copy threads polling, Comm test aid)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r06i; mkdir -p $O; rm -rf $O/*
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > $O/gputests.log 2>&1; cat $O/gputests.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 > $O/bench_line.json; cut -c1-300 $O/bench_line.json
