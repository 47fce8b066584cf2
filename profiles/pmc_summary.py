#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes written by profiles/pmc_run.sh: per kernel, mean of every counter
over the dispatches of the non-counting build.  Usage: profiles/pmc_summary.py <dir> [> summary.txt]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pass*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_"): continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:32s} mean {sum(v)/len(v):16.1f}   n={len(v)}")

# traffic.json: HBM bytes per launch = 2 x FETCH_SIZE KiB (gfx950 reports half of the bytes of wide reads,
# MI355X_MICROARCH.md "HBM"; our 16-B-per-lane record gathers are dwordx4 loads, taken as wide) + WRITE_SIZE KiB
if len(sys.argv) > 3:
    import json
    key, out = sys.argv[2], sys.argv[3]
    t = json.load(open(out)) if os.path.exists(out) else {}
    t[key] = {}
    for k in acc:
        if "<true" in k: continue
        name = k.split("<")[0]
        f = acc[k].get("FETCH_SIZE"); w = acc[k].get("WRITE_SIZE")
        if f and w:
            t[key][name] = int(2 * 1024 * sum(f) / len(f) + 1024 * sum(w) / len(w))
        # VALU issue: wave-instructions per launch and the launch's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
        v = acc[k].get("SQ_INSTS_VALU"); c = acc[k].get("GRBM_GUI_ACTIVE"); th = acc[k].get("SQ_THREAD_CYCLES_VALU")
        if v and c:
            t[key].setdefault("_valu", {})[name] = {"insts": int(sum(v) / len(v)), "cycles": int(sum(c) / len(c) / 8),
                                                    "lanes_active": round(sum(th) / len(th) / (sum(v) / len(v)), 1) if th else None}
    json.dump(t, open(out, "w"), indent=1, sort_keys=True)
