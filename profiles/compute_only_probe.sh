#!/bin/bash
# Runs ON THE GPU BOX with the DIAGNOSTIC library (python bitnetmcu_amd/build.py --diag before gpurun):
# the fused kernel with a cache-resident source (BNM_DIAG_SRC_WRAP, honoured by that library only) under the SQ counters, to
# compare its CYCLE count with the HBM-streaming run (profiles/run_profiles.sh pass sq1): same cycles at a higher
# clock => the kernel is issue-bound and HBM only costs clock (power); fewer cycles => memory stalls matter.
set -u
TAG=${1:-probe}
REPO=$(pwd)
OUT=$REPO/gpurun_out/conly_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export BNM_LIBRARY=$REPO/bitnetmcu_amd/libbitnetmcu_hip_diag.so
cd /tmp
# counter set: default = timing/utilisation; "stall" = where the extra wave-cycles go (second argument)
CTR="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
if [ "${2:-}" = "stall" ]; then
  CTR="SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL GRBM_GUI_ACTIVE"
fi
for wrap in 256 0; do
  BNM_DIAG_SRC_WRAP=$wrap rocprofv3 --kernel-trace --pmc $CTR -f csv -d "$OUT/wrap$wrap" -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-verify --no-extra > "$OUT/wrap$wrap.log" 2>&1
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, collections, sys
for w in (256, 0):
    f = glob.glob(f"{sys.argv[1]}/wrap{w}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("wrap", w, ": no counter file (see the log)"); continue
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if "fused_fc" in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    agg = collections.defaultdict(list)
    for (d, c), v in per.items():
        agg[c].append(v)
    print("== BNM_DIAG_SRC_WRAP =", w)
    for c in sorted(agg):
        print(f"   {c:32s} {sum(agg[c]) / len(agg[c]):.5g}")
PY
for wrap in 256 0; do grep -o '"ms_per_step": [0-9.]*' "$OUT/wrap$wrap.log"; done
