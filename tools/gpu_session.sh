cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_gpu.sh r02 > gpurun_out/s10_profile.log 2>&1
python bench.py > gpurun_out/s10_bench.txt 2>gpurun_out/s10_bench.err
python tools/bench_configs.py > gpurun_out/s10_cfg.txt 2>&1
python tools/bench_all_z.py > gpurun_out/s10_allz.txt 2>&1
python tools/bench_chain.py > gpurun_out/s10_chain.txt 2>&1
python tools/bench_host_path.py > gpurun_out/s10_host.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_chain -o chain -- python $GRAFT_REPO_ROOT/tools/bench_chain.py > $GRAFT_REPO_ROOT/gpurun_out/s10_chain_prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -c 600 gpurun_out/s10_bench.txt; tail -2 gpurun_out/s10_chain.txt | cut -c1-300; tail -3 gpurun_out/s10_host.txt
ls gpurun_out/prof_chain
