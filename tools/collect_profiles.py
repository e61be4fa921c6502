#!/usr/bin/env python3
"""Copy what tools/final_session.sh left under gpurun_out/ into the tracked profiles/<tag>_* files.
    python tools/collect_profiles.py r03"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = [("bench_line.json", "_bench_line.json"), ("bench_configs.json", "_bench_configs.json"), ("bench_all_z.json", "_bench_all_z.json"),
         ("bench_chain.json", "_bench_chain.json"), ("bench_montecarlo.json", "_bench_montecarlo.json"),
         ("bench_host_path.json", "_bench_host_path.json"), ("bler_gap.json", "_bler_gap.json"),
         ("prof_chain/chain_kernel_stats.csv", "_chain_kernel_stats.csv"), ("prof_cfg/cfg_kernel_stats.csv", "_configs_kernel_stats.csv"),
         ("gputests.log", "_gputests.txt"), ("bench_nl.txt", "_bench_nl.txt"), ("bench_crc_stop.json", "_crc_stop.json"),
         ("bench_cfg5_n1.json", "_bench_cfg5_n1.json"), ("host_trace.txt", "_host_trace.txt"),
         ("bench_all_z_stop.json", "_bench_all_z_stop.json"), ("bench_all_z_stop_norefill.json", "_bench_all_z_stop_norefill.json")]
# what the box summarised in place (PMC summary, traffic, kernel stats, instruction mix) first; the session's own outputs then
# overwrite any stale copy of themselves that travelled to the box inside profiles/
d = os.path.join(G, "profiles_" + tag)
if os.path.isdir(d):
    for f in sorted(os.listdir(d)):
        shutil.copy(os.path.join(d, f), os.path.join(P, f))
        print("copied", f)
for src, dst in pairs:
    s = os.path.join(G, src)
    if os.path.exists(s):
        shutil.copy(s, os.path.join(P, tag + dst))
        print("copied", src, "->", tag + dst)
    else:
        print("missing", src)
# the kernel-trace tables carry the build they belong to (a trace older than the kernels beside it)
ids = {}
try:
    ids = json.load(open(os.path.join(G, "prof_" + tag, "ids.json")))
except (OSError, ValueError):
    pass
for f in (tag + "_chain_kernel_stats.csv", tag + "_configs_kernel_stats.csv", tag + "_bench_kernel_stats.csv"):
    p = os.path.join(P, f)
    if os.path.exists(p) and ids:
        txt = open(p).read()
        if not txt.startswith("#"):
            open(p, "w").write("# nrldpc_build_id %s nrldpc_kernel_id %s\n" % (ids.get("nrldpc_build_id"), ids.get("nrldpc_kernel_id")) + txt)
