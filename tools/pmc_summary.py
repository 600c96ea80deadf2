#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch.

usage: pmc_summary.py <counter_collection.csv> [out.json]
Kernel names are cut at the first '(' and templates are kept (harris_kernel<30>)."""
import collections
import csv
import json
import sys


def main():
    src = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    with open(src, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(")[0].replace("okvfe::", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[name] = {"grid": int(row["Grid_Size"]), "wg": int(row["Workgroup_Size"]),
                          "lds": int(row["LDS_Block_Size"]), "vgpr": int(row["VGPR_Count"]),
                          "scratch": int(row["Scratch_Size"])}
    out = {}
    for name, ctrs in sorted(acc.items()):
        out[name] = dict(meta[name])
        out[name]["dispatches"] = max(len(v) for v in ctrs.values())
        # drop the first dispatch of each kernel when there are several (cold caches)
        out[name]["mean_per_dispatch"] = {
            c: sum(v[1:] if len(v) > 1 else v) / max(1, len(v) - 1 if len(v) > 1 else 1)
            for c, v in sorted(ctrs.items())}
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
