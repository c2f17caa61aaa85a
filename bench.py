#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json's metric — 16x16 int8 inferences/s on MI355X for the 4bitsym width-64 FC
model (configs[1]: 100M synthetic images per GPU, resident in HBM), bit-exact vs the C reference.

  python bench.py --gpus 1 --steps K --warmup W                 (N>1: launched by torch.distributed.run)

A "step" is one pass of the whole hot path (packed-weight FC x4 + ReLUNorm x4, one fused kernel launch)
over the rank's resident image shard.  --scaling weak (default): every GPU owns --images images; --scaling strong:
--images images in total, split contiguously over the ranks.  No data-path collective
(images are independent); one RCCL broadcast of the ~13 KB model blob at setup.
Prints ONE JSON line on rank 0.  At N = 1 the line also carries "extra_configs": BASELINE configs[2] (ternary, ALU
kernel, VALU-issue roofline), configs[3] (CNN), the logits variant, Dist-M and the same model through the generic
fused kernel — each timed on the same resident image set and checked against the oracle on a sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md chip table)
HBM_MEASURED_CEILING_GBS = 6290.0   # same table: measured float4-copy ceiling
BYTES_PER_INFERENCE = 260      # 256 B image read + 4 B class id written (SURVEY.md §8d / DESIGN.md)
BYTES_PER_INFERENCE_LOGITS = 300
# VALU issue roofline: one wave64 VALU instruction occupies its SIMD for 4 cycles (measured: SQ_ACTIVE_INST_VALU =
# 4.0 cycles per instruction, profiles/r01/rocprof_r01f_ternary_cnn.md); 256 CUs x 4 SIMDs at the 2.4 GHz peak clock
VALU_PEAK_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 4.0
# the oracle's digest / histogram of ALL 1e8 class ids of the headline workload (fc_4bitsym_64, Dist-U, first image 0),
# computed on the host cores by tests/test_gpu_fullsize.py::test_full_1e8_digest_and_histogram_equal_the_oracle
ORACLE_DIGEST_1E8 = 0x81b56c9fafee6636


def host_cores():
    cores = len(os.sched_getaffinity(0))
    try:   # container CPU quota (cgroup v2): "max" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per) + 0.5)))
    except Exception:
        pass
    return cores


def cpu_baseline(model_name, dist, seconds):
    """The reference's UNMODIFIED BitMnistInference (oracle/_ref/<model>/Bitnet_inf_O3.dll) on all host cores
    over a bounded sample of the same synthetic workload.  Reported next to the GPU number; not a target."""
    cores = host_cores()
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


def load_counters():
    """Per-kernel constants measured with rocprofv3 --pmc in separate passes (profiles/collect_counters.sh ->
    profiles/pmc_counters.json): HBM bytes per launch, VALU instructions per image, MFMA busy fraction.  They are properties
    of the kernel binaries; bench.py REPLAYS them next to its live timing and says so in every field it fills from here."""
    out = {}
    for name in ("pmc_traffic.json", "pmc_counters.json"):
        p = os.path.join(REPO, "profiles", name)
        if os.path.isfile(p):
            try:
                out[name] = json.load(open(p))
            except Exception:
                pass
    return out


def plain_stream_reference():
    """Replayed, not measured by this run: profiles/ceiling_ab.py's same-box, same-process comparison of the default kernel with
    plain 16 B/lane streaming loads of the same 25.6 GB (needs the diagnostic library, which bench.py never loads)."""
    p = os.path.join(REPO, "profiles", "r02", "ceiling_ab_r02_final5.json")
    try:
        t = open(p).read()
        d = json.loads(t[t.index("{"):])
        return {"plain_stream_ms": d["plain_stream"]["median_ms"], "kernel_ms_same_box": d["kernel"]["median_ms"],
                "kernel_over_plain_stream_bytes_per_s": d["kernel_over_plain_stream_bytes_per_s"],
                "source": "replayed from profiles/r02/ceiling_ab_r02_final5.json (profiles/ceiling_ab.py on another box; not measured by this run)"}
    except Exception:
        return None


def kernel_name(b, ctx, model):
    v = ctx.variant
    fused = {3: "fused_fc_dual_kernel", 5: "fused_fc_dual_kernel", 6: "fused_fc_dual_kernel", 4: "fused_fc_generic_kernel"}.get(v, "fused_fc_kernel")
    tern = "ternary_stream_kernel" if getattr(ctx, "ternary_variant", 2) else "ternary_alu_kernel"
    k = {1: fused, 2: "fc_layer_bitserial_kernel+relunorm_kernel", 3: tern}.get(ctx.path, "?")
    cnn = "cnn_front_mfma_kernel" if getattr(ctx, "cnn_variant", 1) else "cnn_front_kernel"
    return k + ("+" + cnn if model.kind == b.KIND_CNN else "")


def load_model_through_the_text_parser(b, util, name, header=None):
    """The product's model path is exporter TEXT -> run-time parser -> GPU.  The reference's headers do not travel to the GPU
    box (only their weight data does, as tests/golden/models/*.bnm), so the text is re-emitted in the exporter's dialect from
    the committed blob (tests/headerwriter.py), parsed by the library like any BitNetMCU_model.h, and must reproduce the blob."""
    if header:
        return b.Model.from_header(header), "header text: " + header
    import headerwriter
    blob_model = util.load_golden_model(name)
    model = b.Model.from_header_text(headerwriter.write_header(blob_model))
    if model.to_blob() != blob_model.to_blob():
        raise RuntimeError(f"{name}: header text -> parser does not reproduce the committed blob")
    return model, f"exporter-dialect header text of tests/golden/models/{name}.bnm through the run-time parser"


def timed_steps(torch, step, steps, warmup, barrier=None):
    """warmup untimed steps, then `steps` steps bracketed by barrier + synchronize on both sides.
    Returns (wall seconds, [per-launch ms from HIP events on the launch stream])."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for k in range(steps):
        step()
        evs[k + 1].record()            # same stream as the launches
    torch.cuda.synchronize()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    return elapsed, [evs[k].elapsed_time(evs[k + 1]) for k in range(steps)]


def verify_sample(np, torch, util, model, images, cls, logits, n, first=0):
    """head + tail + strided sample of the resident set against the oracle (class ids, and logits when written)"""
    om = util.OracleModel(model)      # the checker
    idx = np.concatenate([np.arange(0, 4096), np.arange(n - 2048, n), np.linspace(0, n - 1, 2048).astype(np.int64)])
    idx = np.unique(idx[(idx >= 0) & (idx < n)])
    ti = torch.from_numpy(idx).to(images.device)
    sample = images[ti].cpu().numpy()
    want, want_lg = om.infer(sample, logits=True)
    ok = bool(np.array_equal(want, cls[ti].cpu().numpy().astype(np.uint32)))
    if logits is not None:
        ok = ok and bool(np.array_equal(want_lg, logits[ti].cpu().numpy()))
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=100_000_000, help="images per GPU (weak) or in total (strong); configs[1]: 1e8")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--model", default="fc_4bitsym_64")
    ap.add_argument("--dist", type=int, default=0, help="0 = Dist-U (headline), 1 = Dist-M")
    ap.add_argument("--logits", action="store_true", help="also write the 10 int32 logits (300 B/inference)")
    ap.add_argument("--variant", type=int, default=-1, help="fused kernel variant (-1 = default; 4 = generic kernel)")
    ap.add_argument("--grid", type=int, default=0, help="workgroups (0 = default)")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 fused MFMA, 2 layer-wise ALU, 3 ternary ALU")
    ap.add_argument("--cnn-variant", type=int, default=-1, help="CNN front end: 1 conv1 on MFMA (default), 0 all-VALU kernel of round 1")
    ap.add_argument("--ternary-variant", type=int, default=-1,
                    help="ternary ALU kernel: 2 streamed weights, two images per lane (default), 1 one image per lane, 0 round 1's kernel")
    ap.add_argument("--work-batch", type=int, default=0, help="generic fused kernel: tiles per take from the work counter (0 = default)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs section (N = 1 only)")
    ap.add_argument("--header", default=None, help="load the model from this exporter-written BitNetMCU_model.h (text parser) "
                                                  "instead of tests/golden/models/<model>.bnm")
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
    barrier = td.barrier if distributed else None

    sys.path.insert(0, os.path.join(REPO, "tests"))
    import util

    # ---- model: rank 0 reads it, RCCL-broadcasts the blob (~13 KB) ------------------------------------
    model = None
    model_source = None
    if rank == 0:
        model, model_source = load_model_through_the_text_parser(b, util, a.model, a.header)
    if distributed:
        model = b.dist.broadcast_model(model, src=0, device=dev)
    ctx = b.Context(model, device=local_rank)
    if a.path:
        ctx.set_path(a.path)
    if a.variant >= 0 or a.grid > 0:
        ctx.set_tuning(a.variant, a.grid)
    if a.cnn_variant >= 0:
        ctx.set_cnn_variant(a.cnn_variant)
    if a.ternary_variant >= 0:
        ctx.set_ternary_variant(a.ternary_variant)
    if a.work_batch > 0:
        ctx.set_work_batch(a.work_batch)

    # ---- resident workload: this rank's shard of the global synthetic image stream --------------------
    if a.scaling == "weak":
        n = a.images
        first = rank * n                  # rank r owns global images [r*n, (r+1)*n)
        n_global = n * world
    else:
        first, last = b.dist.shard_range(a.images, rank, world)
        n = last - first
        n_global = a.images
    images = torch.empty((max(n, 1), 256), dtype=torch.int8, device=dev)[:n]
    cls = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    logits = torch.empty((n, model.num_classes), dtype=torch.int32, device=dev) if a.logits else None
    b.synth.fill_device(images, first=first, dist=a.dist)
    torch.cuda.synchronize()

    elapsed, launch_ms = timed_steps(torch, lambda: ctx.infer_device(images, cls, logits), a.steps, a.warmup, barrier)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- outside the timed region: verification ----------------------------------------------------------
    verified = None
    digest = b.synth.digest_device(cls, first=first, n_bins=model.num_classes)
    if distributed:
        digest = b.dist.allreduce_digest(digest)
    torch.cuda.synchronize()
    dg = digest.cpu().numpy()
    digest_hex = hex(int(dg[0].astype(np.uint64)))
    hist = dg[1:].astype(np.int64)
    headline = a.model == "fc_4bitsym_64" and a.dist == 0 and n_global == 100_000_000 and not a.header
    if rank == 0 and not a.no_verify:
        verified = verify_sample(np, torch, util, model, images, cls, logits, n) and int(hist.sum()) == n_global
        if headline:   # every one of the 1e8 class ids, through the order-independent digest the oracle produced on host cores
            verified = verified and int(dg[0].astype(np.uint64)) == ORACLE_DIGEST_1E8

    if rank == 0:
        counters = load_counters()
        total = n_global * a.steps
        bpi = BYTES_PER_INFERENCE_LOGITS if a.logits else BYTES_PER_INFERENCE
        avg_ms = float(np.mean(launch_ms))
        achieved = n * bpi / (avg_ms * 1e-3) / 1e9
        kname = kernel_name(b, ctx, model)
        traffic, traffic_source, mfma_busy = None, None, None
        tj = counters.get("pmc_traffic.json")
        if tj and headline and world == 1 and not a.logits and kname == tj.get("kernel", "fused_fc_dual_kernel"):
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_source = f"replayed from profiles/pmc_traffic.json (rocprofv3 --pmc pass {tj.get('source')}; not measured by this run)"
        cj = counters.get("pmc_counters.json", {}).get(kname.split("+")[0])
        if cj and "mfma_busy_frac" in cj:
            mfma_busy = {"value": cj["mfma_busy_frac"], "source": f"replayed from profiles/pmc_counters.json (pass {cj.get('source')})"}
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
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "i8",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[1]: {a.model}, {n} synthetic 16x16 int8 images per GPU resident in HBM "
                            f"(dist {'U' if a.dist == 0 else 'M'}), class ids{' + logits' if a.logits else ''} written",
                "images_per_gpu": n,
                "global_images": n_global,
                "path": ctx.path,
                "model_source": model_source,
                "parallelism": f"dp{world} image-shard ({a.scaling} scaling), no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_of_measured_ceiling": achieved / HBM_MEASURED_CEILING_GBS,
                "measured_ceiling": HBM_MEASURED_CEILING_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "mfma_busy_frac": mfma_busy,
                "plain_stream_reference": plain_stream_reference(),
                "kernel": kname,
                "avg_launch_ms": avg_ms,
                "min_launch_ms": float(np.min(launch_ms)),
                "algorithmic_bytes_per_launch": n * bpi,
                "fused_variant": ctx.variant,
            },
            "verified_vs_oracle": verified,
            "digest": digest_hex,
            "digest_expected": hex(ORACLE_DIGEST_1E8) if headline else None,
            "class_histogram": hist.tolist(),
        }
        if world == 1 and not a.no_extra:
            out["extra_configs"] = extra_configs(a, np, torch, b, util, dev, images, cls, n, counters)
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(a.model, a.dist, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if distributed:
        td.destroy_process_group()


def extra_configs(a, np, torch, b, util, dev, images, cls, n, counters):
    """The other BASELINE configs and variants on the SAME resident image set, each checked against the oracle on a sample.
    Every entry: inferences/s from HIP events over `steps` launches after `warmup`, the roofline that binds it."""
    cj = counters.get("pmc_counters.json", {})
    res = {}

    def hbm(rate, bpi):
        g = rate * bpi / 1e9
        return {"bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
                "frac_of_measured_ceiling": g / HBM_MEASURED_CEILING_GBS}

    def valu(rate, kernel, bpi):
        c = cj.get(kernel)
        r = {"bound": "valu", "unit": "wave64 VALU instructions/s", "peak": VALU_PEAK_WAVE_INSTR_PER_S,
             "hbm_frac": rate * bpi / 1e9 / HBM_PEAK_GBS}
        if c and "valu_per_image" in c:
            r.update({"achieved": rate * c["valu_per_image"], "frac": rate * c["valu_per_image"] / VALU_PEAK_WAVE_INSTR_PER_S,
                      "valu_per_image": c["valu_per_image"],
                      "source": f"instruction count replayed from profiles/pmc_counters.json (SQ_INSTS_VALU, pass {c.get('source')}); "
                                "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction"})
        else:
            r.update({"achieved": None, "frac": None})
        return r

    def run(name, model_name, count, steps, warmup, dist=0, want_logits=False, variant=-1, path=0, note=None):
        model, _ = load_model_through_the_text_parser(b, util, model_name)
        ctx = b.Context(model, device=dev.index)
        if path:
            ctx.set_path(path)
        if variant >= 0:
            ctx.set_tuning(variant, 0)
        x = images[:count]
        c = cls[:count]
        lg = torch.empty((count, model.num_classes), dtype=torch.int32, device=dev) if want_logits else None
        _, ms = timed_steps(torch, lambda: ctx.infer_device(x, c, lg), steps, warmup)
        rate = count / (float(np.mean(ms)) * 1e-3)
        ok = None if a.no_verify else verify_sample(np, torch, util, model, x, c, lg, count)
        res[name] = {"model": model_name, "images": count, "dist": "U" if dist == 0 else "M", "steps": steps, "warmup": warmup,
                     "value": rate, "unit": "inferences/s", "avg_launch_ms": float(np.mean(ms)), "min_launch_ms": float(np.min(ms)),
                     "kernel": kernel_name(b, ctx, model), "path": ctx.path, "fused_variant": ctx.variant, "verified_vs_oracle": ok}
        if note:
            res[name]["note"] = note
        ctx.close()
        del lg
        return rate

    n_cnn = min(n, 10_000_000)
    # configs[2]: ternary 96-96-96, bit-unpack / sign-accumulate ALU kernel, no MFMA — bound by the VALU issue rate
    r = run("ternary_alu", "tern_96", n, 3, 1, path=b.PATH_TERNARY_ALU, note="BASELINE configs[2]")
    res["ternary_alu"]["roofline"] = valu(r, "ternary_stream_kernel", BYTES_PER_INFERENCE)
    # the same model through the generic MFMA kernel (option; no spills since round 2)
    r = run("ternary_mfma_generic", "tern_96", n, 5, 1, path=b.PATH_FUSED_MFMA)
    res["ternary_mfma_generic"]["roofline"] = hbm(r, BYTES_PER_INFERENCE)
    # configs[3]: CNN 64-wide
    r = run("cnn_64", "cnn_64", n_cnn, 3, 1, note="BASELINE configs[3]")
    res["cnn_64"]["roofline"] = valu(r, "cnn_front_mfma_kernel", BYTES_PER_INFERENCE)
    # headline model through the generic kernel (what any non-zoo 64-wide export would get)
    r = run("fc_generic_kernel", "fc_4bitsym_64", n, 5, 2, variant=4)
    res["fc_generic_kernel"]["roofline"] = hbm(r, BYTES_PER_INFERENCE)
    # headline model, class ids + logits (300 B per inference)
    if n <= 100_000_000:
        r = run("fc_logits", "fc_4bitsym_64", n, 5, 2, want_logits=True)
        res["fc_logits"]["roofline"] = hbm(r, BYTES_PER_INFERENCE_LOGITS)
    # headline model on Dist-M (MNIST-like value statistics): refill the resident set in place
    if a.dist == 0:
        b.synth.fill_device(images, first=0, dist=1)
        torch.cuda.synchronize()
        r = run("fc_dist_m", "fc_4bitsym_64", n, 5, 2, dist=1)
        res["fc_dist_m"]["roofline"] = hbm(r, BYTES_PER_INFERENCE)
    return res


if __name__ == "__main__":
    main()
