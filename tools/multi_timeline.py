import csv, glob, sys
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "nrldpc_decode" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last call: kernels within 1 ms of the last start
    last = int(rows[-1]["Start_Timestamp"])
    sel = [r for r in rows if last - int(r["Start_Timestamp"]) < 900000]
    t0 = int(sel[0]["Start_Timestamp"])
    print(d, len(sel), "kernels in the last call; span %.3f ms" % ((max(int(r["End_Timestamp"]) for r in sel) - t0) / 1e6))
    for r in sel:
        n = r["Kernel_Name"]; n = n[n.find("nrldpc_decode"):][:60]
        print("  %-62s start %7.1f us  dur %7.1f us  grid %s wg %s lds %s" % (n, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("LDS_Block_Size", "?")))
