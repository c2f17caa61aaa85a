#!/usr/bin/env python3
"""Per-kernel table of what profiles/pmc_kernel.sh collected: average duration (stats pass) and the mean per launch of every
counter (each pass separately), plus derived ratios.   usage: python profiles/pmc_table.py gpurun_out/pmc_<tag> [--json out.json]"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    root = sys.argv[1]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    res = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            res[k]["calls"] = int(r["Calls"])
            res[k]["avg_ms"] = float(r["AverageNs"]) / 1e6
            res[k]["min_ms"] = float(r["MinNs"]) / 1e6
    for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
        per = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
            res[short(r["Kernel_Name"])].setdefault("vgpr", r.get("VGPR_Count") or r.get("Arch_VGPR_Count"))
            res[short(r["Kernel_Name"])].setdefault("lds", r.get("LDS_Block_Size"))
            res[short(r["Kernel_Name"])].setdefault("wg", r.get("Workgroup_Size"))
            res[short(r["Kernel_Name"])].setdefault("grid", r.get("Grid_Size"))
        agg = collections.defaultdict(list)
        for (d, c), v in per.items():
            agg[(names[d], c)].append(v)
        for (k, c), vals in agg.items():
            res[k][c] = sum(vals) / len(vals)
    for k, d in res.items():
        if not any(c.startswith(("SQ_", "GRBM", "FETCH", "WRITE")) for c in d):
            continue
        print(f"### `{k}`\n")
        print("| item | value |\n|---|---|")
        for c in sorted(d):
            v = d[c]
            print(f"| {c} | {v:.6g} |" if isinstance(v, float) else f"| {c} | {v} |")
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                if c in d:
                    print(f"| {c} / SQ_WAVE_CYCLES | {d[c] / wc:.3f} |")
        g = d.get("GRBM_GUI_ACTIVE")
        if g:
            cyc = g / 8.0       # per XCD
            if "SQ_ACTIVE_INST_VALU" in d:
                print(f"| VALU busy (SQ_ACTIVE_INST_VALU x4 / 1024 SIMDs / cycles) | {d['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cyc:.3f} |")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
                print(f"| MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles) | {d['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:.3f} |")
            if "avg_ms" in d:
                print(f"| effective clock under the counter pass (GRBM_GUI_ACTIVE/8 / stats avg) GHz | {cyc / (d['avg_ms'] * 1e6):.3f} |")
        if "FETCH_SIZE" in d:
            fb = d["FETCH_SIZE"] * 1024 * 2
            wb = d.get("WRITE_SIZE", 0.0) * 1024
            print(f"| HBM bytes per launch: FETCH_SIZE x1024 x2 (gfx950 correction) + WRITE_SIZE x1024 | {fb + wb:.6g} |")
        print()
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
