cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_gpu.sh r02 > gpurun_out/s15_profile.log 2>&1
python bench.py > gpurun_out/s15_bench.txt 2>gpurun_out/s15_bench.err
python tools/bench_configs.py > gpurun_out/s15_cfg.txt 2>&1
python tools/bench_all_z.py > gpurun_out/s15_allz.txt 2>&1
tail -c 200 gpurun_out/s15_bench.txt
