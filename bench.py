#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json's metric — 16x16 int8 inferences/s on MI355X for the 4bitsym width-64 FC
model (configs[1]: 100M synthetic images per GPU, resident in HBM), bit-exact vs the C reference.

  python bench.py --gpus 1 --steps K --warmup W                 (N>1: launched by torch.distributed.run)

A "step" is one pass of the whole hot path (packed-weight FC x4 + ReLUNorm x4, one fused kernel launch)
over the rank's resident image shard.  Weak scaling: every GPU owns --images images; no data-path
collective (images are independent); one RCCL broadcast of the ~13 KB model blob at setup.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md chip table); 6290 measured copy ceiling
BYTES_PER_INFERENCE = 260      # 256 B image read + 4 B class id written (SURVEY.md §8d / DESIGN.md)
BYTES_PER_INFERENCE_LOGITS = 300


def cpu_baseline(model_name, dist, seconds):
    """The reference's UNMODIFIED BitMnistInference (oracle/_ref/<model>/Bitnet_inf_O3.dll) on all host cores
    over a bounded sample of the same synthetic workload.  Reported next to the GPU number; not a target."""
    cores = len(os.sched_getaffinity(0))
    try:   # container CPU quota (cgroup v2): "max" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per) + 0.5)))
    except Exception:
        pass
    dll = os.path.join(REPO, "oracle", "_ref", model_name, "Bitnet_inf_O3.dll")
    exe = os.path.join(REPO, "oracle", "cpu_bench")
    out = None
    if os.path.isfile(dll) and os.path.isfile(exe):
        out = subprocess.run([exe, dll, str(cores), str(seconds), str(dist), "8192"], capture_output=True, text=True,
                             timeout=seconds * 6 + 60)
    # (the reference's own x86 wrapper smashes its stack for CNN headers with MAX_N_ACTIVATIONS < 256 — DESIGN.md §7 —
    # in which case the port below is timed instead)
    if out is not None and out.returncode == 0 and out.stdout.strip():
        r = json.loads(out.stdout.strip().splitlines()[-1])
        return {"value": r["inf_per_s"], "unit": "inferences/s", "cores": cores, "kind": "reference",
                "sample": f"{r['inferences']} inferences in {r['seconds']:.1f} s: reference BitMnistInference "
                          f"(gcc -O3 -march=x86-64-v3) on {cores} threads, 8192 resident synthetic images per thread, "
                          f"dist {'U' if dist == 0 else 'M'}"}
    # port fallback (fresh clone without oracle/_ref): the C restatement, one Python thread per core
    import threading
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import util
    import bitnetmcu_amd as b
    om = util.OracleModel(util.load_golden_model(model_name))
    x = b.synth.images(0, 8192, dist)
    done = [0] * cores
    t_end = time.time() + seconds

    def work(i):
        while time.time() < t_end:
            om.infer(x)
            done[i] += len(x)
    t0 = time.time()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    el = time.time() - t0
    return {"value": sum(done) / el, "unit": "inferences/s", "cores": cores, "kind": "port",
            "sample": f"{sum(done)} inferences in {el:.1f} s: oracle/bitnet_oracle.c on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=100_000_000, help="images per GPU (configs[1]: 1e8)")
    ap.add_argument("--model", default="fc_4bitsym_64")
    ap.add_argument("--dist", type=int, default=0, help="0 = Dist-U (headline), 1 = Dist-M")
    ap.add_argument("--logits", action="store_true", help="also write the 10 int32 logits (300 B/inference)")
    ap.add_argument("--variant", type=int, default=-1, help="fused kernel variant (-1 = default)")
    ap.add_argument("--grid", type=int, default=0, help="workgroups (0 = default)")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 fused MFMA, 2 layer-wise ALU, 3 ternary ALU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    a = ap.parse_args()

    import numpy as np
    import torch
    import bitnetmcu_amd as b

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (even with one rank): go through the process group, so the N>1 code path
    # (RCCL init, model broadcast, barriers, MAX-reduced time, all-reduced digest) is the one that runs
    distributed = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if distributed:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if distributed:
        td.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- model: rank 0 reads the blob, RCCL-broadcasts it (~13 KB) ------------------------------------
    model = None
    if rank == 0:
        with open(os.path.join(REPO, "tests", "golden", "models", a.model + ".bnm"), "rb") as f:
            model = b.Model.from_blob(f.read())
    if distributed:
        model = b.dist.broadcast_model(model, src=0, device=dev)
    ctx = b.Context(model, device=local_rank)
    if a.path:
        ctx.set_path(a.path)
    if a.variant >= 0 or a.grid > 0:
        ctx.set_tuning(a.variant, a.grid)

    # ---- resident workload: this rank's shard of the global synthetic image stream --------------------
    n = a.images
    first = rank * n                      # weak scaling: rank r owns global images [r*n, (r+1)*n)
    images = torch.empty((n, 256), dtype=torch.int8, device=dev)
    cls = torch.empty(n, dtype=torch.int32, device=dev)
    logits = torch.empty((n, model.num_classes), dtype=torch.int32, device=dev) if a.logits else None
    b.synth.fill_device(images, first=first, dist=a.dist)
    torch.cuda.synchronize()

    def step():
        ctx.infer_device(images, cls, logits)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        td.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for k in range(a.steps):
        step()
        evs[k + 1].record()            # same stream as the launches
    torch.cuda.synchronize()
    if distributed:
        td.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
    launch_ms = [evs[k].elapsed_time(evs[k + 1]) for k in range(a.steps)]

    # ---- outside the timed region: verification ----------------------------------------------------------
    verified = None
    digest = b.synth.digest_device(cls, first=first, n_bins=model.num_classes)
    if distributed:
        digest = b.dist.allreduce_digest(digest)
    torch.cuda.synchronize()
    hist = digest[1:].cpu().numpy().astype(np.int64)
    if rank == 0 and not a.no_verify:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import util
        om = util.OracleModel(model)      # the checker
        idx = np.concatenate([np.arange(0, 4096), np.arange(n - 2048, n),
                              np.linspace(0, n - 1, 2048).astype(np.int64)])
        idx = np.unique(idx[(idx >= 0) & (idx < n)])
        sample = images[torch.from_numpy(idx).to(dev)].cpu().numpy()
        want = om.infer(sample)
        got = cls[torch.from_numpy(idx).to(dev)].cpu().numpy().astype(np.uint32)
        verified = bool(np.array_equal(want, got)) and int(hist.sum()) == n * world

    if rank == 0:
        total = n * world * a.steps
        bpi = BYTES_PER_INFERENCE_LOGITS if a.logits else BYTES_PER_INFERENCE
        avg_ms = float(np.mean(launch_ms))
        achieved = n * bpi / (avg_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(REPO, "profiles", "pmc_traffic.json")   # written by profiles/collect_pmc.py from a rocprofv3 --pmc pass
        if os.path.isfile(tf):
            try:
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": ("16x16 int8 MNIST inferences/s, FC 4bitsym 64-64-64 (BitNetMCU_model_fc.h), bit-exact vs C reference"
                       if a.model == "fc_4bitsym_64" else f"16x16 int8 inferences/s, model {a.model}, bit-exact vs C reference"),
            "value": total / elapsed,
            "unit": "inferences/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i8",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[1]: {a.model}, {n} synthetic 16x16 int8 images per GPU resident in HBM "
                            f"(dist {'U' if a.dist == 0 else 'M'}), class ids{' + logits' if a.logits else ''} written",
                "images_per_gpu": n,
                "global_images": n * world,
                "path": ctx.path,
                "parallelism": f"dp{world} image-shard, no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": {1: "fused_fc_dual_kernel" if ctx.variant == 3 else "fused_fc_kernel", 2: "fc_layer_bitserial_kernel+relunorm_kernel", 3: "ternary_alu_kernel"}.get(ctx.path, "?")
                          + ("+cnn_front_kernel" if model.kind == b.KIND_CNN else ""),
                "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": n * bpi,
                "fused_variant": ctx.variant,
            },
            "verified_vs_oracle": verified,
            "class_histogram": hist.tolist(),
        }
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(a.model, a.dist, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if distributed:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
