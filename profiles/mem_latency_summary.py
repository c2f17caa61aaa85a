#!/usr/bin/env python3
"""Condense profiles/mem_latency_probe.sh output: per kernel, mean counter value per dispatch and the derived
average latencies / occupancies."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    vals = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if not any(t in k for t in ("fused_fc", "diag_stream")):
                continue
            short = k.split("(")[0].replace("void ", "")[:60]
            per[(short, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (short, _, c), v in per.items():
            vals[short][c].append(v)
    print("| kernel | counter | mean per dispatch | dispatches |")
    print("|---|---|---|---|")
    for k in sorted(vals):
        for c in sorted(vals[k]):
            v = vals[k][c]
            print(f"| `{k}` | {c} | {sum(v)/len(v):.6g} | {len(v)} |")
    print()
    for k in sorted(vals):
        cs = {c: sum(v) / len(v) for c, v in vals[k].items()}
        line = [f"`{k}`:"]
        if cs.get("TCC_EA0_RDREQ_sum"):
            line.append(f"avg EA read latency {cs.get('TCC_EA0_RDREQ_LEVEL_sum', 0) / cs['TCC_EA0_RDREQ_sum']:.0f} TCC cycles")
            line.append(f"DRAM-credit stall cycles per EA read {cs.get('TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum', 0) / cs['TCC_EA0_RDREQ_sum']:.3f}")
            line.append(f"tag stall cycles per EA read {cs.get('TCC_TAG_STALL_sum', 0) / cs['TCC_EA0_RDREQ_sum']:.3f}")
        if cs.get("TCP_TCC_READ_REQ_sum"):
            line.append(f"avg TCP->TCC read latency {cs.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / cs['TCP_TCC_READ_REQ_sum']:.0f} cycles")
        if cs.get("TCC_CYCLE_sum"):
            line.append(f"TCC busy {cs.get('TCC_BUSY_sum', 0) / cs['TCC_CYCLE_sum']:.3f}")
        if cs.get("TCC_EA0_RDREQ_LEVEL_sum") and cs.get("TCC_CYCLE_sum"):
            pass
        print(" ".join(line))


if __name__ == "__main__":
    main()
