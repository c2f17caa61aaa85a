#!/bin/bash
# BASELINE configs[4] in one command, ON A MULTI-GPU NODE (never measured so far: every box this repository has seen had one GPU):
# the headline model on N = 1, 2, 4, 8 GPUs (as many as the node has), one rank per GPU over RCCL,
#   weak   scaling: 10^8 images PER GPU (each rank's shard is the single-GPU workload)
#   strong scaling: 10^8 images IN TOTAL, split into contiguous shards
# and the single-process C entry point (one host thread + one RCCL communicator per GPU, bnm_run_synth_multi_gpu).
# Output: one bench.py JSON line per run in gpurun_out/scale/{weak,strong}_N.json and a table on stdout; per-rank times are in
# every line (per_rank_ms_per_step), so a straggler GPU shows.  usage: profiles/run_scale.sh [steps] [warmup] [gpus needed]
# Exit status: 0 = every run printed its line; 4 = fewer GPUs visible than the curve needs (default 2: a one-GPU box measures no
# scaling - one line on stderr says so, nothing runs); 3 = some run gave up (its watchdog's one-line reason is repeated on stderr).
set -u
STEPS=${1:-20}; WARM=${2:-3}; NEED=${3:-2}
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=$REPO/gpurun_out/scale; mkdir -p "$OUT"; cd "$REPO"
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "GPUs visible: $NGPU"
if [ "$NGPU" -lt "$NEED" ]; then
  echo "run_scale: a scaling curve over $NEED or more GPUs was asked for and $NGPU device(s) are visible on this node - nothing measured" >&2
  exit 4
fi
PORT=29517
FAILED=0
for N in 1 2 4 8; do
  [ "$N" -le "$NGPU" ] || continue
  for MODE in weak strong; do
    if [ "$N" -eq 1 ]; then
      python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --scaling $MODE --no-extra --no-cpu > "$OUT/${MODE}_$N.json" 2> "$OUT/${MODE}_$N.err"
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $PORT bench.py \
        --gpus "$N" --steps "$STEPS" --warmup "$WARM" --scaling $MODE > "$OUT/${MODE}_$N.json" 2> "$OUT/${MODE}_$N.err"
      PORT=$((PORT + 1))
    fi
    if ! grep -q '^{' "$OUT/${MODE}_$N.json"; then      # no JSON line: the run gave up - say why in one line (bitnetmcu_amd/dist.py's watchdog, or the last stderr line)
      FAILED=3
      REASON=$(grep -m1 '^bitnetmcu_amd.dist:' "$OUT/${MODE}_$N.err" || tail -n 1 "$OUT/${MODE}_$N.err")
      echo "run_scale: $MODE scaling on $N GPU(s) printed no line: $REASON" >&2
    fi
  done
done
python - "$OUT" <<'PY'
import glob, json, os, sys
rows = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*_*.json"))):
    lines = [l for l in open(f) if l.startswith("{")]
    if not lines:
        print("no JSON line in", f); continue
    d = json.loads(lines[-1])
    rows[(d["scaling"], d["n_gpus"])] = d
ORACLE_DIGEST_1E8 = "0x81b56c9fafee6636"      # the oracle's digest of the 1e8 class ids of global images [0, 1e8) (tests/test_gpu_fullsize.py)
print("| scaling | GPUs | RCCL ranks | inferences/s | ms per step (slowest rank) | per-rank ms per step | per-rank kernel ms | vs N x 1 GPU | of 8 TB/s per GPU | sample vs oracle | all-reduced digest |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for (mode, n), d in sorted(rows.items()):
    one = rows.get((mode, 1))
    eff = d["value"] / (one["value"] * n) if one else float("nan")
    pr = d.get("per_rank_ms_per_step") or [d["ms_per_step"]]
    pk = d.get("per_rank_kernel_ms") or [d["roofline"]["avg_launch_ms"]]
    # strong scaling covers exactly global images [0, 1e8): the all-reduced digest must be the oracle's constant; weak scaling at
    # N > 1 covers [0, N x 1e8), for which no host-side constant exists (the sample check still runs)
    dig = d.get("digest")
    dig_ok = (dig == ORACLE_DIGEST_1E8) if d["config"]["global_images"] == 100_000_000 else None
    print(f"| {mode} | {n} | {d['config'].get('rccl_ranks')} | {d['value']:.4g} | {d['ms_per_step']:.3f} | {' '.join('%.3f' % x for x in pr)} | "
          f"{' '.join('%.3f' % x for x in pk)} | {eff:.3f} | {d['value'] * 260 / n / 8e12:.3f} | {d.get('verified_vs_oracle')} | "
          f"{dig} {'== oracle' if dig_ok else '!= ORACLE ' + ORACLE_DIGEST_1E8 if dig_ok is False else '(no constant for this range)'} |")
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*_*.err"))):      # a run that gave up says why in one line (bitnetmcu_amd/dist.py)
    for l in open(f):
        if l.startswith("bitnetmcu_amd.dist:"):
            print(os.path.basename(f), l.strip())
PY
# the C host's entry point on all GPUs (RCCL bound at run time)
python - <<'PY'
import ctypes as C
import bitnetmcu_amd as b
L = b.load()
m = b.Model.from_zoo("fc_4bitsym_64")
out = (C.c_uint64 * 11)(); secs = C.c_double()
used = L.bnm_run_synth_multi_gpu(m._h, 100_000_000, 0, 0, b.SEED_DIST_U, out, 10, C.byref(secs))
print(f"bnm_run_synth_multi_gpu: {used} GPU(s), transport {L.bnm_multi_gpu_transport().decode()}, 1e8 images in {secs.value * 1e3:.3f} ms "
      f"= {1e8 / secs.value:.4g} inferences/s, digest {hex(out[0])} (oracle: 0x81b56c9fafee6636)")
PY
exit $FAILED
