set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s1_tests.txt
tools/ubench/valu_rate > gpurun_out/s1_ubench1.txt 2>&1
tools/ubench/valu_rate2.bin > gpurun_out/s1_ubench2.txt 2>&1
for i in 1 2; do
python tools/bench_one.py 1 384 4096 0 0 25 >> gpurun_out/s1_ab.txt 2>&1
NRLDPC_LIB=$PWD/exp_libs/lib_selcmp.so python tools/bench_one.py 1 384 4096 0 0 25 >> gpurun_out/s1_ab.txt 2>&1
done
python tools/bench_one.py 2 384 4096 0 0 25 >> gpurun_out/s1_ab.txt 2>&1
python tools/bench_one.py 1 256 4096 0 0 25 >> gpurun_out/s1_ab.txt 2>&1
python tools/bench_one.py 1 384 4096 1 0 25 >> gpurun_out/s1_ab.txt 2>&1
python tools/bench_one.py 1 96 8192 0 0 25 >> gpurun_out/s1_ab.txt 2>&1
python bench.py --steps 10 --warmup 2 > gpurun_out/s1_bench.txt 2>&1
cat gpurun_out/s1_tests.txt gpurun_out/s1_ab.txt
tail -c 3000 gpurun_out/s1_bench.txt
