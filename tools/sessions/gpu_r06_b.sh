#!/bin/bash
# round 6, call b: ABI revision 6 (per-call layer count, pool timing), bench.py Comm fall-back + --in-process, refill ring events
mkdir -p gpurun_out/r06b; cd /root/repo
timeout 2400 python -m pytest tests/test_abi_caller_gpu.py tests/test_full_size_gpu.py tests/test_refill_gpu.py tests/test_layers_gpu.py tests/test_system_objects_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r06b/tests.txt
python bench.py --gpus 2 --in-process --share-gpu --steps 10 --warmup 3 2>gpurun_out/r06b/inproc.err | tee gpurun_out/r06b/inproc.json | cut -c1-400
