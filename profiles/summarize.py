#!/usr/bin/env python3
"""Condense raw rocprofv3 CSV output (profiles/run_profiles.sh) into a markdown summary + pmc_traffic.json.
   usage: python profiles/summarize.py gpurun_out/prof_<tag> [--json profiles/pmc_traffic.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def find(root, sub, suffix):
    hits = glob.glob(os.path.join(root, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def main():
    root = sys.argv[1]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    print(f"# rocprofv3 summary for `{os.path.basename(root)}`\n")
    st = find(root, "stats", "kernel_stats.csv")
    if st:
        print("## kernel-trace --stats (un-instrumented timing pass)\n")
        print("| kernel | calls | total ms | avg ms | min ms | max ms | % |")
        print("|---|---|---|---|---|---|---|")
        for r in rows(st):
            name = r.get("Name", "")[:90]
            g = lambda k: float(r.get(k, 0) or 0)
            print(f"| `{name}` | {r.get('Calls')} | {g('TotalDurationNs')/1e6:.3f} | {g('AverageNs')/1e6:.4f} | "
                  f"{g('MinNs')/1e6:.4f} | {g('MaxNs')/1e6:.4f} | {r.get('Percentage')} |")
        print()
    tr = find(root, "stats", "kernel_trace.csv")
    if tr:
        for r in rows(tr):
            if "fused_fc" in r.get("Kernel_Name", ""):
                print("fused kernel dispatch: grid", r.get("Grid_Size_X", r.get("Grid_Size")), "wg", r.get("Workgroup_Size_X", r.get("Workgroup_Size")),
                      "VGPR", r.get("VGPR_Count"), "accum VGPR", r.get("Accum_VGPR_Count"), "SGPR", r.get("SGPR_Count"),
                      "LDS", r.get("LDS_Block_Size"), "scratch", r.get("Scratch_Size"))
                break
        print()
    counters = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [values per dispatch]
    for sub in sorted(os.listdir(root)):
        cc = find(root, sub, "counter_collection.csv")
        if not cc or not sub.startswith("pmc"):
            continue
        per_dispatch = defaultdict(float)
        meta = {}
        for r in rows(cc):
            key = (r.get("Dispatch_Id"), r.get("Counter_Name"))
            per_dispatch[key] += float(r.get("Counter_Value", 0) or 0)
            meta[r.get("Dispatch_Id")] = r.get("Kernel_Name", "")
        for (did, cname), v in per_dispatch.items():
            k = meta[did]
            short = "fused_fc_kernel" if "fused_fc" in k else k.split("(")[0][:40]
            counters[short][cname].append(v)
    if counters:
        print("## PMC counters (each block collected in its own pass; per-dispatch mean over the bench's launches)\n")
        for k, cs in counters.items():
            if "fused_fc" not in k:
                continue
            print(f"### `{k}`\n")
            print("| counter | mean per launch | launches |")
            print("|---|---|---|")
            for c, vals in sorted(cs.items()):
                print(f"| {c} | {sum(vals)/len(vals):.6g} | {len(vals)} |")
            print()
            f = cs.get("FETCH_SIZE")
            w = cs.get("WRITE_SIZE")
            if f:
                fetch_kb = sum(f) / len(f)
                write_kb = sum(w) / len(w) if w else 0.0
                # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB-ish units of 1024 B; on gfx950 FETCH_SIZE
                # reports exactly half the bytes of a 16 B/lane coalesced streaming read -> x2 (the image stream is
                # read with 16 B/lane global_load_lds_dwordx4 / global_load_dwordx4).
                fetch_b = fetch_kb * 1024 * 2
                write_b = write_kb * 1024
                print(f"HBM read  (FETCH_SIZE x 1024 x 2, gfx950 correction): {fetch_b/1e9:.3f} GB per launch")
                print(f"HBM write (WRITE_SIZE x 1024, uncalibrated):          {write_b/1e9:.3f} GB per launch")
                print(f"total: {(fetch_b+write_b)/1e9:.3f} GB per launch\n")
                if out_json:
                    json.dump({"hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes": fetch_b, "write_bytes": write_b,
                               "source": os.path.basename(root), "note": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (MI355X_MICROARCH.md HBM section)"},
                              open(out_json, "w"))
            if "GRBM_GUI_ACTIVE" in cs and "SQ_BUSY_CYCLES" in cs:
                g = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
                print(f"GRBM_GUI_ACTIVE (cycles per launch): {g:.4g}")
            if "SQ_WAVE_CYCLES" in cs:
                wc = sum(cs["SQ_WAVE_CYCLES"]) / len(cs["SQ_WAVE_CYCLES"])
                for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                    if c in cs:
                        print(f"{c} / SQ_WAVE_CYCLES = {sum(cs[c])/len(cs[c])/wc:.3f}")
                print()


if __name__ == "__main__":
    main()
