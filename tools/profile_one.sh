#!/bin/bash
# rocprofv3 kernel trace + two PMC passes of ONE (BG, Z) decoder configuration (tools/bench_one.py), on the GPU box:
#   tools/profile_one.sh <bg> <Z>   -> gpurun_out/prof_one_<bg>_<Z>/{stats,sqA,sqB}/...  and a summary line per kernel
# (separate runs with --kernel-trace only, never combined with sys/hip traces)
set -u
BG=$1; Z=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_one_${BG}_${Z}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_one.py $BG $Z"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT/sqA -o sqA -- $CMD > $OUT/sqA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sqB -o sqB -- $CMD > $OUT/sqB.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o grbm -- $CMD > $OUT/grbm.log 2>&1
python - <<PY
import csv, glob, collections, json
out = {"bg": $BG, "Z": $Z}
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nrldpc_decode" in r["Name"]:
            out["kernel"] = r["Name"]; out["avg_ns"] = float(r["AverageNs"]); out["calls"] = int(r["Calls"])
for name in ("sqA", "sqB", "grbm"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if "nrldpc_decode" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
if "SQ_INSTS_VALU" in out and "avg_ns" in out:
    out["valu_issue_frac_2cycle"] = out["SQ_INSTS_VALU"] / (out["avg_ns"] * 1e-9) / 1.2288e12
if "SQ_LDS_IDX_ACTIVE" in out and "GRBM_GUI_ACTIVE" in out:  # as bench.py: the GRBM counter is summed over the 8 XCDs
    cyc = out["GRBM_GUI_ACTIVE"] / 8.0
    out["lds_busy_frac"] = out["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
    if "SQ_ACTIVE_INST_VALU" in out:
        out["valu_pipe_busy_at_4_cycles_per_op"] = 4.0 * out["SQ_ACTIVE_INST_VALU"] / (1024 * cyc)
if "SQ_WAVE_CYCLES" in out and "SQ_WAIT_ANY" in out:
    out["parked_frac_of_wave_cycles"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
print(json.dumps(out))
open("$ROOT/gpurun_out/prof_one_${BG}_${Z}.json", "w").write(json.dumps(out, indent=1))
PY
