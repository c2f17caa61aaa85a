nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" ; 
python bench.py --model mcu_cnn_16 --images 2000000 --steps 3 --warmup 1 --no-cpu 2>&1 | tail -5
for t in 8 32 64 128 256; do ./oracle/cpu_bench oracle/_ref/fc_4bitsym_64/Bitnet_inf_O3.dll $t 3 0 8192; done
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/prof_tern/stats -o stats -- python /root/repo/bench.py --model tern_96 --path 3 --images 20000000 --steps 3 --warmup 1 --no-cpu --no-verify > /root/repo/gpurun_out/prof_tern.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -f csv -d /root/repo/gpurun_out/prof_tern/pmc_sq1 -o pmc -- python /root/repo/bench.py --model tern_96 --path 3 --images 20000000 --steps 3 --warmup 1 --no-cpu --no-verify >> /root/repo/gpurun_out/prof_tern.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -f csv -d /root/repo/gpurun_out/prof_tern/pmc_sq2 -o pmc -- python /root/repo/bench.py --model tern_96 --path 3 --images 20000000 --steps 3 --warmup 1 --no-cpu --no-verify >> /root/repo/gpurun_out/prof_tern.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/prof_cnn/stats -o stats -- python /root/repo/bench.py --model cnn_64 --images 5000000 --steps 3 --warmup 1 --no-cpu --no-verify > /root/repo/gpurun_out/prof_cnn.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -f csv -d /root/repo/gpurun_out/prof_cnn/pmc_sq1 -o pmc -- python /root/repo/bench.py --model cnn_64 --images 5000000 --steps 3 --warmup 1 --no-cpu --no-verify >> /root/repo/gpurun_out/prof_cnn.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA -f csv -d /root/repo/gpurun_out/prof_cnn/pmc_sq2 -o pmc -- python /root/repo/bench.py --model cnn_64 --images 5000000 --steps 3 --warmup 1 --no-cpu --no-verify >> /root/repo/gpurun_out/prof_cnn.log 2>&1
cd /root/repo
python - <<'PY'
import csv,glob,collections
for tag in ("tern","cnn"):
    for f in glob.glob(f"gpurun_out/prof_{tag}/stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            print(tag, r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6, "ms")
    for f in glob.glob(f"gpurun_out/prof_{tag}/pmc*/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:30]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
        for k in acc:
            if "ternary" in k or "cnn_front" in k:
                print(tag,k,{c:v/len(cnt[k]) for c,v in acc[k].items()})
PY
