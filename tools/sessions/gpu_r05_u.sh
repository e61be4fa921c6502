#!/bin/bash
# end-of-round check on the committed tree: smoke(), the whole GPU suite, the default bench run with its wall time
mkdir -p gpurun_out/r05u; cd /root/repo
( time python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" ) 2>&1 | tail -5 | tee gpurun_out/r05u/smoke.txt
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | tee gpurun_out/r05u/gputests.txt
( time python bench.py > gpurun_out/r05u/bench_line.json 2> gpurun_out/r05u/bench_err.txt ) 2>&1 | tail -4 | tee gpurun_out/r05u/bench_time.txt
cut -c1-300 gpurun_out/r05u/bench_line.json
