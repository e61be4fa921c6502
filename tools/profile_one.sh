#!/bin/bash
# rocprofv3 kernel trace + two PMC passes of ONE (BG, Z) decoder configuration (tools/bench_one.py), on the GPU box:
#   tools/profile_one.sh <bg> <Z> [n_layers [early_term]]  -> gpurun_out/prof_one_<bg>_<Z>[_nl<n>]/{stats,sqA,sqB}/...  and a summary line per kernel
# (a layer count profiles the pruned call; NRLDPC_NO_PRUNED_PIPELINE=1 in the environment sends it to the run-time-prefix build)
# (separate runs with --kernel-trace only, never combined with sys/hip traces)
set -u
BG=$1; Z=$2; NL=${3:-0}; ET=${4:-0}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAGN=${BG}_${Z}$([ "$NL" != "0" ] && echo _nl$NL)$([ "$ET" != "0" ] && echo _et)${TAGX:-}
OUT=$ROOT/gpurun_out/prof_one_${TAGN}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$(python -c "print(max(4096, (4096 * 384 // $Z) // 256 * 256))")
CMD="python $ROOT/tools/bench_one.py $BG $Z $B $ET $NL"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT/sqA -o sqA -- $CMD > $OUT/sqA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sqB -o sqB -- $CMD > $OUT/sqB.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o grbm -- $CMD > $OUT/grbm.log 2>&1
python - <<PY
import csv, glob, collections, json
out = {"bg": $BG, "Z": $Z, "n_layers": $NL, "batch": $B, "iterations": 25}
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
if "SQ_WAVE_CYCLES" in out and "SQ_WAIT_ANY" in out:
    out["parked_frac_of_wave_cycles"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
print(json.dumps(out))
open("$ROOT/gpurun_out/prof_one_${TAGN}.json", "w").write(json.dumps(out, indent=1))
PY
