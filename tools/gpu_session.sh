cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 1200 python tools/fuzz_decode.py 800 99 2>&1 | tail -1
python tools/bench_one.py 1 384 4096 0 0 25 2>&1 | grep -v amdgpu
python tools/bench_one.py 1 96 8192 0 0 25 2>&1 | grep -v amdgpu
python tools/bench_one.py 2 384 4096 0 30 25 2>&1 | grep -v amdgpu
