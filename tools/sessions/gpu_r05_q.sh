#!/bin/bash
# cfg4 (one nrldpc_decode_multi_dev call): SQ counters of the shared launches -- how busy the SIMDs are while the call runs
mkdir -p gpurun_out/r05q; cd /tmp; export TMPDIR=/tmp
R=/root/repo
pmc() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/q_$name -o $name -- python $R/tools/probe_multi.py > /tmp/q_$name.log 2>&1; }
pmc sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY
pmc sqB SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
python - <<'PY' > $R/gpurun_out/r05q/multi_pmc.txt 2>&1
import csv, glob, collections
for name in ("sqA", "sqB", "grbm"):
    f = glob.glob("/tmp/q_%s/**/*counter_collection.csv" % name, recursive=True)
    if not f: print(name, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "multi_kernel" not in k: continue
        key = k[k.find("nrldpc_decode_multi"):][:40] + " wg%s grid%s" % (r.get("Workgroup_Size", "?"), r.get("Grid_Size", "?"))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if (key, r["Dispatch_Id"]) not in seen: seen.add((key, r["Dispatch_Id"])); n[key] += 1
    for key in sorted(acc):
        print(name, key, "dispatches", n[key], {c: round(v / n[key]) for c, v in sorted(acc[key].items())})
PY
cat $R/gpurun_out/r05q/multi_pmc.txt
