#!/usr/bin/env python3
"""Per-launch durations of one kernel out of a rocprofv3 --kernel-trace CSV, warm-up launches excluded, next to the HIP-event
numbers bench.py printed in the SAME run (the JSON line on its stdout).
  usage: kernel_trace_summary.py <dir with *kernel_trace.csv> <bench stdout log> <kernel name prefix> [--per-step L]
Prints a markdown summary: all launches, the timed ones (= bench.py's steps x L launches per step, warm-ups excluded): median /
min / mean / max per launch, the per-STEP sums (a step of a chunked path is several launches of unequal size: only their sum is
comparable with ms_per_step and only the sum yields a roofline), and the ratio to the run's own ms_per_step."""
import csv
import glob
import json
import os
import statistics
import sys


def main():
    root, log, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
    per_step = int(sys.argv[sys.argv.index("--per-step") + 1]) if "--per-step" in sys.argv else 1
    hits = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not hits:
        sys.exit("no kernel_trace.csv under " + root)
    rows = list(csv.DictReader(open(hits[0], newline="")))
    mine = [r for r in rows if r.get("Kernel_Name", "").startswith(prefix)]
    mine.sort(key=lambda r: int(r["Start_Timestamp"]))
    ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in mine]
    line = [l for l in open(log) if l.startswith("{")]
    bench = json.loads(line[-1]) if line else {}
    steps = bench.get("steps") or max(1, len(ms) // per_step)
    last = ms[-steps * per_step:]
    step_ms = [sum(last[k * per_step:(k + 1) * per_step]) for k in range(len(last) // per_step)]
    print(f"# `{mine[0]['Kernel_Name'] if mine else prefix}` under rocprofv3 --kernel-trace\n")
    print(f"launches in the trace: {len(ms)} (bench.py: {bench.get('warmup')} warm-up + {bench.get('steps')} timed)\n")
    print("| set | launches | median ms | min ms | mean ms | max ms |")
    print("|---|---|---|---|---|---|")
    sets = [("all launches", ms), (f"the {len(last)} timed launches (warm-ups excluded)", last)]
    if per_step > 1:
        sets.append((f"per step: sums of {per_step} consecutive timed launches", step_ms))
    for name, v in sets:
        if v:
            print(f"| {name} | {len(v)} | {statistics.median(v):.4f} | {min(v):.4f} | {statistics.fmean(v):.4f} | {max(v):.4f} |")
    if bench:
        r = bench.get("roofline", {})
        print(f"\nbench.py in the same run (HIP events on the launch stream): ms_per_step {bench['ms_per_step']:.4f}, "
              f"avg_launch_ms {r.get('avg_launch_ms', 0):.4f}, median {r.get('median_launch_ms', 0):.4f}, min {r.get('min_launch_ms', 0):.4f}; "
              f"value {bench['value']:.4e} {bench['unit']}; roofline.frac {r.get('frac', 0):.4f}")
        if step_ms:
            print(f"\nrocprof mean per step (kernel time only) / bench ms_per_step = {statistics.fmean(step_ms) / bench['ms_per_step']:.4f}; "
                  f"rocprof median per step / bench median = {statistics.median(step_ms) / r.get('median_launch_ms', 1):.4f}")
            alg = r.get("algorithmic_bytes_per_launch")      # (bench.py: the algorithmic bytes of one STEP)
            if alg:
                print(f"\nroofline from the rocprof durations of this run: {alg / 1e9:.3f} GB / {statistics.fmean(step_ms):.4f} ms / 8000 GB/s = "
                      f"**{alg / (statistics.fmean(step_ms) * 1e-3) / 8e12:.4f}** (mean), {alg / (statistics.median(step_ms) * 1e-3) / 8e12:.4f} (median), "
                      f"{alg / (min(step_ms) * 1e-3) / 8e12:.4f} (fastest step)")


if __name__ == "__main__":
    main()
