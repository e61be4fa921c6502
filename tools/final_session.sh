#!/bin/bash
# The measurement set behind profiles/r06_*: GPU tests, bench line, every BASELINE configuration, all lifting sizes, chain stages
# (+ their kernel trace), Monte-Carlo loop, host path, rocprofv3 kernel trace + PMC passes of the bench command.
# TAG=r06 bash tools/final_session.sh ; then on the build box: python tools/collect_profiles.py r06
TAG=${TAG:-r06}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/gputests.log 2>&1; cat gpurun_out/gputests.log
fi
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_line_noprofile.json
python tools/bench_configs.py > gpurun_out/cfg.log 2>&1
python tools/bench_chain.py > gpurun_out/chain.log 2>&1
python tools/bench_montecarlo.py > gpurun_out/mc.log 2>&1
python tools/bench_all_z.py > gpurun_out/allz.log 2>&1
# round 5: the same landscape under the reference's only mode (parity stop at each size's waterfall), with and without slot refill
STOP=1 python tools/bench_all_z.py > gpurun_out/allz_stop.log 2>&1
STOP=1 OUT_SUFFIX=_norefill NRLDPC_NO_REFILL=1 python tools/bench_all_z.py > gpurun_out/allz_stop_norefill.log 2>&1
if [ -z "$SKIP_HOST" ]; then python tools/bench_host_path.py > gpurun_out/hostpath.log 2>&1; fi
# round 4: pruned layer counts (three routes), CRC-aided stop, BASELINE configs[4] as the bench leg at N = 1, host-path phase trace
python tools/bench_nl.py default > gpurun_out/nl_default.log 2>&1
NRLDPC_NO_PRUNED_PIPELINE=1 python tools/bench_nl.py rt > gpurun_out/nl_rt.log 2>&1
NRLDPC_NO_PRUNED_PIPELINE=1 NRLDPC_NO_RT=1 python tools/bench_nl.py general > gpurun_out/nl_general.log 2>&1
python tools/bench_nl.py --merge > gpurun_out/bench_nl.txt 2>&1
python tools/bench_crc_stop.py > gpurun_out/crc_stop.log 2>&1
python bench.py --cfg5 --steps 10 --warmup 3 --cpu-sample 0 --no-e2e --no-early-term 2>/dev/null | tail -1 > gpurun_out/bench_cfg5_n1.json
NRLDPC_HOST_TRACE=1 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-early-term 2> gpurun_out/host_trace.err | tail -1 > gpurun_out/bench_host_trace_line.json
grep "host path" gpurun_out/host_trace.err > gpurun_out/host_trace.txt
bash tools/profile_gpu.sh $TAG > gpurun_out/profile.log 2>&1
# (the profiled runs write bench_*_under_rocprof.json: through round 5's third session they overwrote the unprofiled files above, so
# that the committed bench_configs / bench_chain figures carried the profiler's launch and synchronisation overhead -- cfg4 0.52 for 0.46 ms)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_chain -o chain -- env OUT_SUFFIX=_under_rocprof python $GRAFT_REPO_ROOT/tools/bench_chain.py > $GRAFT_REPO_ROOT/gpurun_out/prof_chain.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg -o cfg -- env OUT_SUFFIX=_under_rocprof python $GRAFT_REPO_ROOT/tools/bench_configs.py > $GRAFT_REPO_ROOT/gpurun_out/prof_cfg.log 2>&1 )
# the bench line again, now that a profile of this very build exists on the box: summarise in place so that roofline.frac is filled
python tools/summarise_profile.py $TAG > gpurun_out/summarise.log 2>&1
python tools/isa_mix.py --form split --json profiles/${TAG}_headline_isa_mix.json > profiles/${TAG}_headline_isa_mix.txt 2>gpurun_out/isa_mix.err
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_line.json
mkdir -p gpurun_out/profiles_$TAG; cp profiles/${TAG}_* gpurun_out/profiles_$TAG/ 2>/dev/null
cut -c1-400 gpurun_out/bench_line.json; tail -3 gpurun_out/summarise.log
