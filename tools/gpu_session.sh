set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/bler_gap.json
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s2_tests.txt
python tools/bench_configs.py > gpurun_out/s2_cfg_new.txt 2>&1
cp gpurun_out/bench_configs.json gpurun_out/s2_bench_configs_new.json
NRLDPC_NO_PRUNED_PIPELINE=1 python tools/bench_configs.py > gpurun_out/s2_cfg_old.txt 2>&1
cp gpurun_out/bench_configs.json gpurun_out/s2_bench_configs_general.json
cat gpurun_out/s2_tests.txt
grep -h "info_Gbit_s" gpurun_out/s2_cfg_new.txt | python -c "
import sys,ast
for l in sys.stdin:
    d=ast.literal_eval(l); print('NEW', d['config'][:60], d.get('kernel_ms', d.get('wall_ms')), round(d['info_Gbit_s'],2))"
grep -h "info_Gbit_s" gpurun_out/s2_cfg_old.txt | python -c "
import sys,ast
for l in sys.stdin:
    d=ast.literal_eval(l); print('OLD', d['config'][:60], d.get('kernel_ms', d.get('wall_ms')), round(d['info_Gbit_s'],2))"
