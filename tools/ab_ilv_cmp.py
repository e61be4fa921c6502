#!/usr/bin/env python3
"""Compare tools/ab_ilv.py runs: python tools/ab_ilv_cmp.py new.jsonl old.jsonl [new2.jsonl old2.jsonl ...]
One line per (BG, Z): per layer count, fixed-25 and parity-stop times old -> new."""
import json, sys
def load(ps):
    d = {}
    for p in ps:
        for l in open(p):
            r = json.loads(l); d[(r["bg"], r["Z"], r["nl"], r["et"], "wf" if p.find("7_") >= 0 or p.find("wf") >= 0 else "fx")] = r
    return d
new, old = load(sys.argv[1::2]), load(sys.argv[2::2])
E = {1: 316, 2: 197}
for bg in (1, 2):
    for Z in sorted({k[1] for k in new if k[0] == bg}):
        s = "BG%d Z=%3d" % (bg, Z)
        for tag in ("fx", "wf"):
            for nl in sorted({k[2] for k in new if k[0] == bg and k[1] == Z and k[4] == tag}):
                k0, k1 = (bg, Z, nl, 0, tag), (bg, Z, nl, 1, tag)
                if k0 not in old or k1 not in old: continue
                f = lambda a, b: "%.3f>%.3f(%+3.0f%%)" % (a["ms"], b["ms"], 100 * (b["ms"] / a["ms"] - 1))
                eu = ""
                if nl == 0 and tag == "fx":
                    eu = " eu/ns %4.0f>%4.0f" % tuple(r["batch"] * 25 * E[bg] * Z / r["ms"] / 1e6 for r in (old[k0], new[k0]))
                s += " | %s nl=%2d F %s S %s%s" % (tag, nl, f(old[k0], new[k0]), f(old[k1], new[k1]), eu)
        print(s)
