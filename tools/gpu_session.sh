cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/s4_tests.txt
cat gpurun_out/s4_tests.txt
python tools/bench_one.py 1 384 8192 0 5 25 2>&1 | grep -v amdgpu
python tools/bench_one.py 1 384 8192 1 5 25 2>&1 | grep -v amdgpu
