#!/bin/bash
# round 6, call k: the two multi-GPU routes of bench.py at EIGHT ranks / shards on the box's one GPU (protocol check: values are one GPU's)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r06k; mkdir -p $O; rm -rf $O/*
( time timeout 900 python bench.py --gpus 8 --share-gpu --steps 5 --warmup 2 --batch 1024 --cfg5-total 8192 --cfg5-steps 2 2>$O/ranks8.err | tail -1 > $O/ranks8.json ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06k/ranks8.json')); print('8 ranks:', d['n_gpus'], round(d['value'],2), d['comm'], [round(g['kernel_ms'],2) for g in d['roofline']['per_gpu']], [g['codewords'] for g in d['cfg5_strong']['per_gpu']])
PY
( time timeout 900 python bench.py --gpus 8 --in-process --share-gpu --steps 5 --warmup 2 --batch 1024 2>$O/inproc8.err | tail -1 > $O/inproc8.json ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06k/inproc8.json')); print('8 shards in one process:', d['n_gpus'], round(d['value'],2), d['comm']['shards'], [round(g['kernel_ms'],2) for g in d['roofline']['per_gpu']])
PY
( time timeout 600 python bench.py --gpus 1 --in-process --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/inproc1.json ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06k/inproc1.json')); print('1 shard in process:', d['n_gpus'], round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['roofline']['frac'])
PY
python tools/bench_chain.py > $O/chain.log 2>&1; cp gpurun_out/bench_chain.json $O/
