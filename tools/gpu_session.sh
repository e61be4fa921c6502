cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/fuzz_decode.py 1200 20260930 > gpurun_out/s12_fuzz.txt 2>&1; tail -3 gpurun_out/s12_fuzz.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
