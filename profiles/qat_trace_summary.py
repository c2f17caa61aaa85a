#!/usr/bin/env python3
"""The whole-model QAT forward under rocprofv3 --kernel-trace: per-call durations of its two kernels (weight preparation + the model
kernel), warm-ups excluded, next to the HIP-event numbers profiles/qat_model_bench.py printed in the SAME run, and the HBM roofline
from the rocprof durations (1,024 B read + 4 B x classes written per row).
  usage: qat_trace_summary.py <dir with *kernel_trace.csv> <qat_model_bench stdout log> <timed steps>"""
import csv
import glob
import json
import os
import statistics
import sys


def main():
    root, log, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    hits = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not hits:
        sys.exit("no kernel_trace.csv under " + root)
    rows = list(csv.DictReader(open(hits[0], newline="")))
    bench = json.loads([l for l in open(log) if l.startswith("{")][-1])
    out = {}
    for prefix in ("qat_model_prep_kernel", "void qat_fc_model_fwd_kernel"):
        mine = sorted((r for r in rows if r.get("Kernel_Name", "").startswith(prefix)), key=lambda r: int(r["Start_Timestamp"]))
        ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in mine][-steps:]
        out[prefix] = (mine[0]["Kernel_Name"] if mine else prefix, ms)
    fwd_name, fwd = out["void qat_fc_model_fwd_kernel"]
    _, prep = out["qat_model_prep_kernel"]
    n, bpr = bench["rows"], bench["bytes_per_row"]
    print(f"# `{fwd_name.split('(')[0]}` + `qat_model_prep_kernel` under rocprofv3 --kernel-trace: {n} rows per call, widths {bench['widths']}, {bench['quant']} / {bench['norm']}\n")
    print("| kernel | timed launches | median ms | min ms | mean ms | max ms |")
    print("|---|---|---|---|---|---|")
    for name, v in (("qat_fc_model_fwd_kernel", fwd), ("qat_model_prep_kernel", prep)):
        print(f"| {name} | {len(v)} | {statistics.median(v):.4f} | {min(v):.4f} | {statistics.fmean(v):.4f} | {max(v):.4f} |")
    both = statistics.fmean(fwd) + statistics.fmean(prep)
    alg = n * bpr
    print(f"\nqat_model_bench.py in the same run (HIP events around the whole call, both launches + the gap between them): median "
          f"{bench['ms_median']:.4f} ms, min {bench['ms_min']:.4f} ms; {bench['rows_per_s']:.4e} rows/s; HBM fraction {bench['hbm_frac']:.4f}")
    print(f"\nroofline from the rocprof durations of this run: {alg / 1e9:.3f} GB ({bpr} B per row) / {statistics.fmean(fwd):.4f} ms / 8000 GB/s = "
          f"**{alg / (statistics.fmean(fwd) * 1e-3) / 8e12:.4f}** (the model kernel alone, mean), {alg / (both * 1e-3) / 8e12:.4f} (both kernels), "
          f"{alg / (bench['ms_median'] * 1e-3) / 8e12:.4f} (the whole call by HIP events); kernel time / call time = {both / bench['ms_median']:.3f}")


if __name__ == "__main__":
    main()
