#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc outputs (…_counter_collection.csv found under the given directories) into
per-kernel, per-launch means: {kernel short name: {counter: mean over launches, "launches": n, meta}}.
usage: python3 tools/prof_collect.py out.json dir [dir ...]   (run where the CSVs are; no GPU needed)"""
import csv, glob, json, os, sys
from collections import defaultdict


def short(name):
    for k in ("pipeline_kernel", "render_kernel", "resume_team_kernel", "resume_kernel", "block_var_kernel", "order_tiles_kernel",
              "stats_reduce_kernel"):
        if k in name:
            return k
    return None


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(lambda: defaultdict(float))
            names = {}
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k is None:
                    continue
                key = (f, r["Dispatch_Id"])
                names[key] = k
                per_dispatch[key][r["Counter_Name"]] += float(r["Counter_Value"])   # summed over XCDs / instances
                per_dispatch[key]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                meta.setdefault(k, {m: r[m] for m in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                                          "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")})
            for key, ctrs in per_dispatch.items():
                for c, v in ctrs.items():
                    acc[names[key]][c].append(v)
    res = {}
    for k, ctrs in acc.items():
        res[k] = {c: sum(v) / len(v) for c, v in ctrs.items() if c != "_ns"}
        res[k]["launches_per_counter_pass"] = min(len(v) for c, v in ctrs.items())
        res[k]["duration_ms_under_pmc"] = sum(ctrs["_ns"]) / len(ctrs["_ns"]) / 1e6
        res[k]["launch_meta"] = meta[k]
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {"launches": v["launches_per_counter_pass"], "ms": round(v["duration_ms_under_pmc"], 3)} for k, v in res.items()}))


if __name__ == "__main__":
    main()
