#!/bin/bash
# the default bench run three times on one box: how much the host-pointer legs move from run to run
mkdir -p gpurun_out/r05aa; cd /root/repo
for i in 1 2 3; do python bench.py 2>/dev/null | tail -1 > gpurun_out/r05aa/bench_line_$i.json; done
python - <<'PY'
import json
for i in (1, 2, 3):
    l = json.load(open("gpurun_out/r05aa/bench_line_%d.json" % i)); e = l["e2e"]
    print(i, round(l["value"], 3), round(l["ms_per_step"], 4), {k: (round(e[k]["ms_median"], 2), round(e[k]["ms_max"], 2)) for k in ("f16", "f32_matlab_single", "f64_matlab_double", "f16_byte_per_bit", "f64_byte_per_bit")}, round(e["host_dram_read"]["GB_per_s"]))
PY
