#!/bin/bash
# rocprofv3 stats + SQ counters for the ternary ALU and CNN front-end kernels (run on the GPU box from the repo root)
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
for spec in "tern tern_96 3 20000000" "cnn cnn_64 0 5000000"; do
  set -- $spec; tag=$1; model=$2; path=$3; n=$4
  OUT=$REPO/gpurun_out/prof_$tag; mkdir -p $OUT
  B="python $REPO/bench.py --model $model --path $path --images $n --steps 3 --warmup 1 --no-cpu --no-verify"
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- $B > $OUT/stats.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq1 -o pmc -- $B > $OUT/pmc_sq1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -f csv -d $OUT/pmc_sq2 -o pmc -- $B > $OUT/pmc_sq2.log 2>&1
done
cd $REPO
python - <<'PY'
import csv,glob,collections
for tag,key in (("tern","ternary_alu"),("cnn","cnn_front")):
    print(f"## {key}_kernel")
    for f in glob.glob(f"gpurun_out/prof_{tag}/stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Name"] or "fused_fc" in r["Name"]:
                print("stats:", r["Name"][:70], "calls", r["Calls"], "avg ms", round(float(r["AverageNs"])/1e6,4))
    for f in sorted(glob.glob(f"gpurun_out/prof_{tag}/pmc*/**/*counter_collection.csv", recursive=True)):
        acc=collections.defaultdict(float); ids=set()
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); ids.add(r["Dispatch_Id"])
        for c,v in sorted(acc.items()): print(f"pmc: {c} = {v/len(ids):.6g} per launch")
PY
