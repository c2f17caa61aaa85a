#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json's metric — 16x16 int8 inferences/s on MI355X for the 4bitsym width-64 FC
model (configs[1]: 100M synthetic images per GPU, resident in HBM), bit-exact vs the C reference.

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: bench.py starts its N ranks itself (re-executes under `python -m torch.distributed.run
--nproc-per-node N`, one rank per GPU over RCCL); launched BY torch.distributed.run it takes RANK / LOCAL_RANK / WORLD_SIZE
from the environment and insists that WORLD_SIZE == --gpus.  `--dist-backend gloo --ranks-share-device` puts all ranks on
GPU 0 with gloo collectives: the N > 1 code path (model broadcast, shards, barriers, MAX-reduced time, all-reduced digest)
on a box that has ONE GPU - a functional check of the sharding, not a scaling measurement.

A "step" is one pass of the whole hot path (packed-weight FC x4 + ReLUNorm x4, one fused kernel launch)
over the rank's resident image shard.  --scaling weak (default): every GPU owns --images images; --scaling strong:
--images images in total, split contiguously over the ranks.  No data-path collective
(images are independent); one broadcast of the ~13 KB model blob at setup.
Prints ONE JSON line on rank 0.  At N = 1 the line also carries "extra_configs": BASELINE configs[2] (ternary, ALU
kernel, VALU-issue roofline), configs[3] (CNN), the logits variant, Dist-M, the same model through the generic fused kernel
and the reference's documented 12 KB model family (docs/documentation.md:169-183) — each timed on the same resident image set
and checked against the oracle on a sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md chip table)
BYTES_PER_INFERENCE = 260      # 256 B image read + 4 B class id written (SURVEY.md §8d / DESIGN.md)
BYTES_PER_INFERENCE_LOGITS_10 = 300
BYTES_PER_INFERENCE_FLOAT = 1028   # float-input row: 256 float32 read + 4 B class id written (fused)
# VALU issue roofline: one wave64 VALU instruction occupies its SIMD for 4 cycles (measured: SQ_ACTIVE_INST_VALU =
# 4.0 cycles per instruction, profiles/r01/rocprof_r01f_ternary_cnn.md); 256 CUs x 4 SIMDs at the 2.4 GHz peak clock
VALU_PEAK_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 4.0
# dense int8 matrix-core peak in units of the instruction the kernels use: one v_mfma_i32_32x32x32_i8 (65,536 int8 ops) occupies
# a SIMD's matrix core for 8 passes = 32 clocks -> 1024 SIMDs x 2.4 GHz / 32 = 7.68e10 per second = 5.03 Pop/s, the ~5 PF dense
# figure of MI355X_MICROARCH.md's MFMA table (I8 runs at twice the BF16 rate); profiles/energy_probe.py measured 7.5e10/s
MFMA_I8_32X32X32_PEAK_PER_S = 1024 * 2.4e9 / 32.0
MACS_PER_WAVE_DOT4 = 4 * 64    # one wave64 v_dot4_i32_i8: 4 MACs per lane
# the oracle's digest / histogram of ALL 1e8 class ids of the headline workload (fc_4bitsym_64, Dist-U, first image 0),
# computed on the host cores by tests/test_gpu_fullsize.py::test_full_1e8_digest_and_histogram_equal_the_oracle
ORACLE_DIGEST_1E8 = 0x81b56c9fafee6636


LINE_LIMIT = 4096      # the driver recovers the LAST stdout line; round 5's 37 KB line did not survive its buffer


def sig(x, digits=5):
    """floats to `digits` significant digits (the compact line is for reading and for the driver's consistency check, the full
    precision is in bench_full.json)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{digits}g}")


def compact_line(out):
    """The ONE line the driver parses: the contract keys, the dominant kernel's roofline, the CPU baseline and every other row as
    [inferences/s, binding roofline, fraction of it, verified vs oracle].  Everything else of `out` (definitions, sources, per-row
    detail) is in bench_full.json and on the '# ' stdout lines before this one.  Pure function of `out`: tests/test_bench_cpu.py
    runs it over committed full records."""
    cfg = out["config"]
    src = cfg.get("model_source") or ""
    c = {"workload": cfg["workload"].replace(" synthetic 16x16 int8 images per GPU resident in HBM", " synthetic images/GPU in HBM"),
         "images_per_gpu": cfg["images_per_gpu"], "global_images": cfg["global_images"],
         "model_source": ("reference's own header bytes" if "the reference's own" in src else
                          "reference exporter's header" if "exportquant.py" in src else
                          "header file" if src.startswith("exporter-written header file") else "blob re-emitted as header text")
                         + ", run-time parser",
         "parallelism": cfg["parallelism"].split(",")[0]}
    for k in ("dist_backend", "rccl_ranks"):
        if cfg.get(k) is not None:
            c[k] = cfg[k]
    if cfg.get("ranks_share_device"):
        c["ranks_share_device"] = True
    rf = out["roofline"]
    r = {k: sig(rf[k]) for k in ("bound", "achieved", "peak", "unit", "frac")}
    r["traffic"] = sig(rf["traffic"], 6) if isinstance(rf.get("traffic"), float) else rf.get("traffic")
    ts = rf.get("traffic_source")
    r["traffic_source"] = None if ts is None else ("pmc replay " + (ts.split("--pmc pass ")[1].split(";")[0] if "--pmc pass " in ts else ""))[:60]
    r["kernel"] = rf["kernel"]
    if rf.get("launched") and rf["launched"] != rf["kernel"]:
        r["launched"] = rf["launched"]
    for k in ("avg_launch_ms", "median_launch_ms", "min_launch_ms"):
        r[k] = sig(rf[k])
    r["algorithmic_bytes_per_launch"] = rf["algorithmic_bytes_per_launch"]
    if rf.get("stream_read"):
        r["stream_read"] = {"GB/s": sig(rf["stream_read"]["GB/s"])}
        r["time_vs_stream_read"] = sig(rf["time_vs_stream_read"], 4)
    if rf.get("mfma"):
        r["mfma"] = {"per_image": rf["mfma"]["per_image"], "frac": sig(rf["mfma"]["frac"], 4)}
        if "busy_frac" in rf["mfma"]:
            r["mfma"]["busy_frac"] = sig(rf["mfma"]["busy_frac"], 4)
    if rf.get("counters_dropped"):
        r["counters_dropped"] = len(rf["counters_dropped"])
    # (the headline pair at full precision: whoever checks value against ms_per_step must not meet a rounding of the line's own making)
    line = {k: (out[k] if k in ("value", "ms_per_step") else sig(out[k])) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                                                   "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["metric"] = line["metric"].replace(" (BitNetMCU_model_fc.h)", "")
    line["config"] = c
    line["roofline"] = r
    if out.get("per_rank_ms_per_step"):
        line["per_rank_ms_per_step"] = [sig(x, 4) for x in out["per_rank_ms_per_step"]]
    if out.get("cpu_baseline"):
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = {"value": sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb["sample"].replace("reference BitMnistInference (gcc -O3 -march=x86-64-v3)", "ref gcc -O3")
                                                      .replace(", 8192 resident synthetic images per thread", "")[:110]}
    line["verified_vs_oracle"] = out["verified_vs_oracle"]
    line["digest"] = out["digest"]
    if out.get("digest_expected"):
        line["digest_ok"] = out["digest"] == out["digest_expected"]
    ex = out.get("extra_configs")
    if ex:
        line["rows_format"] = "[inf/s, bound, frac, verified]"
        line["rows"] = {k: [sig(v["value"], 4), v["roofline"]["bound"], sig(float(v["roofline"]["frac"]), 4), v["verified_vs_oracle"]]
                        for k, v in ex.items() if "roofline" in v}
        for k in ("ternary_alu", "cnn_64"):           # configs[2] / configs[3]: the reference's CPU rate on the same host cores
            if k in ex and "cpu_baseline" in ex[k]:
                line.setdefault("rows_cpu", {})[k] = sig(ex[k]["cpu_baseline"]["value"], 4)
        line["full"] = "bench_full.json"
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, f"bench line is {len(text)} bytes (limit {LINE_LIMIT}): move detail to bench_full.json"
    return text


def emit(out, full_path):
    """bench_full.json beside the script + '# ' lines on stdout (every detail, one line per row; no line but the last starts with
    '{'), then the compact line as the LAST stdout line."""
    line = compact_line(out)
    try:
        with open(full_path, "w") as f:
            json.dump(out, f, indent=1)
    except OSError as e:
        print(f"# {full_path} not written: {e}", flush=True)
    for k, v in (out.get("extra_configs") or {}).items():
        print(f"# row {k} " + json.dumps(v), flush=True)
    print("# full " + json.dumps({k: v for k, v in out.items() if k != "extra_configs"}), flush=True)
    print(line, flush=True)


def checker():
    """The oracle-backed checker (oracle/checker.py): imported only for verification outside the timed region and for the
    cpu_baseline leg.  The product package never imports it."""
    p = os.path.join(REPO, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import checker as ck
    return ck


def host_cores():
    cores = len(os.sched_getaffinity(0))
    try:   # container CPU quota (cgroup v2): "max" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per) + 0.5)))
    except Exception:
        pass
    return cores


def cpu_baseline(b, model_name, dist, seconds):
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
    om = checker().OracleModel(b.Model.from_zoo(model_name))
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


def load_counters(lib_path=None):
    """Per-kernel constants measured with rocprofv3 --pmc in separate passes (profiles/pmc_kernel.sh ->
    profiles/pmc_counters.json, pmc_traffic.json): HBM bytes per launch, VALU instructions per image, matrix-core busy share.
    They are properties of ONE kernel binary; bench.py REPLAYS them next to its live timing and says so in every field it fills
    from here.  Every entry carries the machine-code hash of its kernel at measurement time ("code_sha1", written by
    profiles/make_counters_json.py from bitnetmcu_amd/codeobj.py); an entry whose kernel is absent from the loaded library or
    hashes differently there describes another binary and is DROPPED (listed under out["dropped"])."""
    out = {"dropped": []}
    hashes = None
    if lib_path and os.path.isfile(lib_path):
        try:
            from bitnetmcu_amd import codeobj
            hashes = codeobj.kernel_hashes(lib_path)
        except Exception as e:      # an unreadable library image: nothing can be validated, nothing is replayed
            out["dropped"].append(f"all entries: cannot fingerprint {lib_path}: {e}")
            return out

    def valid(key, e):
        if hashes is None:
            out["dropped"].append(f"{key}: no library to validate against")
            return False
        want, name = e.get("code_sha1"), e.get("mangled")
        if not want or not name:
            out["dropped"].append(f"{key}: entry carries no code hash (measured before round 4)")
            return False
        if hashes.get(name) != want:
            out["dropped"].append(f"{key}: kernel {e.get('kernel')} is {'absent from' if name not in hashes else 'a different binary in'} the loaded library")
            return False
        return True

    for name in ("pmc_traffic.json", "pmc_counters.json"):
        p = os.path.join(REPO, "profiles", name)
        if not os.path.isfile(p):
            continue
        try:
            j = json.load(open(p))
        except Exception:
            continue
        if name == "pmc_traffic.json":
            if valid(name, j):
                out[name] = j
        else:
            out[name] = {k: e for k, e in j.items() if valid(f"{name}:{k}", e)}
    return out


def model_mfmas_per_image(b, model):
    """v_mfma_i32_32x32x32_i8 instructions per image of the fused MFMA kernels, from the model's shape alone: a layer of
    ceil(n_out / 32) row tiles over ceil(K / 32) K-steps (K = the previous layer's outputs; 256, or 4 x channels behind the CNN
    front end, for the first) is tiles x K-steps MFMAs per tile of 32 images - twice that when the model holds an FP1.3.0 +128
    (second weight plane).  FC stack only: the CNN front end's conv1 MFMAs are not in this count."""
    k = 4 * model.layer(0).out_channels if model.kind == b.KIND_CNN else 256
    total, planes = 0, 1
    for i, li in model.fc_layers():
        total += ((li.n_output + 31) // 32) * ((k + 31) // 32)
        k = li.n_output
        if li.bits_per_weight == 20:
            w = model.layer_weights(i)
            if any((((w >> (4 * nib)) & 15) == 7).any() for nib in range(8)):
                planes = 2
    return planes * total / 32.0


def mfma_roofline(b, ctx, model, rate, kname, counters, model_key=None):
    """north_star: "for the MFMA path, int8 MFMA utilisation vs gfx950 peak".  Static instruction count x measured rate against
    the dense int8 peak, next to the matrix cores' busy share from the counter pass of the same kernel binary (replayed, tagged).
    None when the timed path is not an MFMA kernel."""
    if ctx.path != b.PATH_FUSED_MFMA:
        return None
    per_image = model_mfmas_per_image(b, model)
    r = {"instruction": "v_mfma_i32_32x32x32_i8", "per_image": per_image, "achieved_per_s": rate * per_image,
         "peak_per_s": MFMA_I8_32X32X32_PEAK_PER_S, "frac": rate * per_image / MFMA_I8_32X32X32_PEAK_PER_S,
         "int8_ops_per_s": rate * per_image * 65536.0,
         "definition": "MFMAs per image (from the model's tile counts) x inferences/s / (1024 SIMDs x 2.4 GHz / 32 clocks per MFMA)"}
    cj = counters.get("pmc_counters.json", {})
    base = kname.split("+")[0]
    c = (cj.get(f"{base}@{model_key}") if model_key else None) or cj.get(base)
    if c and "mfma_busy_frac" in c:
        r["busy_frac"] = c["mfma_busy_frac"]
        r["busy_frac_source"] = (f"SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of {c.get('kernel')}, replayed from "
                                 f"profiles/pmc_counters.json (rocprofv3 --pmc pass {c.get('source')}, at that pass's clock; same kernel binary: "
                                 f"code_sha1 {c.get('code_sha1', '')[:12]}); not measured by this run")
    return r


def kernel_name(b, ctx, model, count=None, cnn_variant_named=False):
    """count / cnn_variant_named: a context nobody named a CNN front end on gives calls of fewer than 2 C^2 images to the channel
    kernel whatever it reports (bnm_capi_infer.cpp)."""
    v = ctx.variant
    fused = {3: "fused_fc_dual_kernel", 5: "fused_fc_dual_kernel", 6: "fused_fc_dual_kernel", 4: "fused_fc_generic_kernel", 7: "fused_fc_generic_kernel",
             8: "fused_fc_generic_kernel", 9: "fused_fc_regw_kernel"}.get(v, "fused_fc_kernel")
    # the streamed kernel is the default for every shape of the ALU table (two images per lane for 96-96-96, one for the others);
    # variant 0 selects the plain ALU kernel
    tern = "ternary_stream_kernel" if getattr(ctx, "ternary_variant", 2) % 10 else "ternary_alu_kernel"
    k = {1: fused, 2: "fc_layer_bitserial_kernel+relunorm_kernel", 3: tern}.get(ctx.path, "?")
    cnn = {0: "cnn_front_kernel", 3: "cnn_li_kernel"}.get(getattr(ctx, "cnn_variant", 1), "cnn_front_mfma_kernel")
    if model.kind == b.KIND_CNN and cnn == "cnn_li_kernel" and count is not None and not cnn_variant_named:
        if count < 2 * model.layers()[0].out_channels ** 2:
            cnn = "cnn_front_mfma_kernel"
    if model.kind == b.KIND_CNN and cnn == "cnn_li_kernel" and ctx.path == b.PATH_FUSED_MFMA and ctx.cnn_tail_fused:
        return "cnn_li_fused_pipe_kernel" if ctx.cnn_pipelined else "cnn_li_fused_kernel"        # front end + FC tail in one kernel: the only launch of the call
    return k + ("+" + cnn if model.kind == b.KIND_CNN else "")


def model_macs(b, model):
    """multiply-accumulates of one inference: FC layers over their REAL inputs (a ternary layer's padded trits are not work),
    conv layers 9 per output position (BitNetMCU_MNIST_dll.c:66-80: 14x14, 12x12, 4x4 positions per channel)"""
    macs, width = 0, 256
    layers = model.layers()
    if model.kind == b.KIND_CNN:
        c = layers[0].out_channels
        macs += c * 9 * (14 * 14 + 12 * 12 + 4 * 4)
        width = 4 * c
    for li in layers:
        if li.type == 1:      # BNM_LAYER_FC
            macs += min(li.n_input, width) * li.n_output
            width = li.n_output
    return macs


def load_model_through_the_text_parser(b, name, header=None):
    """The product's model path is exporter TEXT -> run-time parser -> GPU.  Sources, in this order: --header; the header the
    REFERENCE'S OWN EXPORTER wrote in this repository (tests/golden/headers/, committed: the ternary 96-96-96 models and the
    documented 12 KB family); the REFERENCE'S OWN FILE (BitNetMCU_model_fc.h, BitNetMCU_model_cnn.h, mcu/*.h), a byte-identical
    copy staged by oracle/build_oracle.py under tests/golden/_ref_headers/ where /root/reference exists and shipped to the GPU box
    (sha256 in its MANIFEST.json).  Whatever was parsed must equal the committed weight blob (bitnetmcu_amd/zoo/<name>.bnm).  Only
    a tree that has neither (a fresh clone that never saw the reference) re-emits the blob as text with the package's own writer."""
    if header:
        return b.Model.from_header(header), "exporter-written header file " + header + " through the run-time parser"
    blob_model = b.Model.from_zoo(name)
    exported = os.path.join(REPO, "tests", "golden", "headers", name + ".h")
    staged = os.path.join(REPO, "tests", "golden", "_ref_headers", name + ".h")
    for path, what in ((exported, "written by the reference's exportquant.py in this repository"), (staged, None)):
        if not os.path.isfile(path):
            continue
        if what is None:
            import hashlib
            man = json.load(open(os.path.join(os.path.dirname(staged), "MANIFEST.json")))[name]
            sha = hashlib.sha256(open(path, "rb").read()).hexdigest()
            if sha != man["sha256"]:
                raise RuntimeError(f"{path}: not the bytes that were staged from {man['origin']}")
            what = f"the reference's own {man['origin'][len('/root/reference/'):]}, byte-identical staged copy, sha256 {sha[:16]}"
        model = b.Model.from_header(path)
        if model.to_blob() != blob_model.to_blob():
            raise RuntimeError(f"{name}: {path} does not parse to the committed blob")
        return model, f"{os.path.relpath(path, REPO)} ({what}) through the run-time parser"
    model = b.Model.from_header_text(blob_model.to_header_text())
    if model.to_blob() != blob_model.to_blob():
        raise RuntimeError(f"{name}: header text -> parser does not reproduce the committed blob")
    return model, (f"bitnetmcu_amd/zoo/{name}.bnm re-emitted as header text by bitnetmcu_amd/headerwriter.py (exporter dialect; "
                   "not the exporter's own bytes: no staged reference header in this tree) and parsed by the run-time parser")


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


def relaunch_with_ranks(a):
    """--gpus N > 1 and no launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node (N > 1 without a launcher: bench.py starts them itself)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=100_000_000, help="images per GPU (weak) or in total (strong); configs[1]: 1e8")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl", help="nccl = RCCL (one rank per GPU); gloo for --ranks-share-device")
    ap.add_argument("--init-timeout", type=float, default=60.0, help="seconds the process group may take to come up before the run gives up with a reason")
    ap.add_argument("--ranks-share-device", action="store_true",
                    help="all ranks on GPU 0 (needs --dist-backend gloo): the N > 1 code path on a one-GPU box; not a scaling measurement")
    ap.add_argument("--model", default="fc_4bitsym_64")
    ap.add_argument("--dist", type=int, default=0, help="0 = Dist-U (headline), 1 = Dist-M")
    ap.add_argument("--logits", action="store_true", help="also write the int32 logits (300 B/inference for 10 classes)")
    ap.add_argument("--variant", type=int, default=-1, help="fused kernel variant (-1 = default; 4 = generic kernel)")
    ap.add_argument("--grid", type=int, default=0, help="workgroups (0 = default)")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 fused MFMA, 2 layer-wise ALU, 3 ternary ALU")
    ap.add_argument("--cnn-variant", type=int, default=-1, help="CNN front end: 3 the lane = image kernel (the default up to 170 channels; calls of fewer than 2 C^2 images then go to 1), "
                                                                  "300 + g: the same with g tiles per take, 1 a lane = a channel with conv1 on the matrix cores, 0 round 1's all-VALU kernel")
    ap.add_argument("--ternary-variant", type=int, default=-1,
                    help="ternary ALU kernel: 2 streamed weights, two images per lane (default), 1 one image per lane, 0 round 1's kernel")
    ap.add_argument("--work-batch", type=int, default=0, help="fused kernels: tiles / pairs per take from the work counter (0 = default)")
    ap.add_argument("--input", choices=("int8", "float"), default="int8",
                    help="float: the timed call is bnm_infer_float_device on float32 images (the synthetic int8 images x 1/127) - the "
                         "fc_float_input row as the main workload (profiling runs)")
    ap.add_argument("--float-mode", type=int, default=0, help="--input float: 0 the fused float-input kernel where it exists, 2 quantise + infer")
    ap.add_argument("--float-groups", type=int, default=0, help="--input float: 8-image groups in flight per wave of the fused kernel (0 default, 2, 4)")
    ap.add_argument("--float-images", type=int, default=100_000_000, help="images of the float-input rows (102.4 GB of float32 at 1e8; 0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs section (N = 1 only)")
    ap.add_argument("--full-json", default=os.path.join(REPO, "bench_full.json"), help="where rank 0 writes the full record (every row in detail)")
    ap.add_argument("--header", default=None, help="load the model from this exporter-written BitNetMCU_model.h (text parser) "
                                                  "instead of bitnetmcu_amd/zoo/<model>.bnm")
    a = ap.parse_args()
    if a.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if a.ranks_share_device and a.dist_backend != "gloo":
        sys.exit("--ranks-share-device needs --dist-backend gloo (RCCL wants one device per rank)")

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if a.gpus > 1 and not launched:
        relaunch_with_ranks(a)          # does not return

    import numpy as np
    import torch
    import bitnetmcu_amd as b

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py --gpus {a.gpus} was launched with WORLD_SIZE={world}: the two must agree "
                 f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus}, or plain `python bench.py --gpus {a.gpus}`)")
    # launched by torch.distributed.run (even with one rank): go through the process group, so the N>1 code path
    # (RCCL init, model broadcast, barriers, MAX-reduced time, all-reduced digest) is the one that runs
    distributed = launched
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    dev_index = 0 if a.ranks_share_device else local_rank
    if dev_index >= torch.cuda.device_count():
        sys.exit(f"rank {rank}: GPU {dev_index} requested but only {torch.cuda.device_count()} visible "
                 "(one rank per GPU; --dist-backend gloo --ranks-share-device runs all ranks on GPU 0)")
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    td = None
    if distributed:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a group that does not come up within --init-timeout ends the run with one line on stderr (bitnetmcu_amd/dist.py), not a hang
        b.dist.init_process_group_or_exit(a.dist_backend, rank, world, device=dev if a.dist_backend == "nccl" else None,
                                          timeout_s=a.init_timeout)
    coll_dev = dev if a.dist_backend == "nccl" else torch.device("cpu")     # where the (tiny) collectives' tensors live
    barrier = td.barrier if distributed else None

    # ---- model: rank 0 reads it, the blob (~13 KB) is broadcast ---------------------------------------
    model = None
    model_source = None
    if rank == 0:
        model, model_source = load_model_through_the_text_parser(b, a.model, a.header)
    if distributed:
        model = b.dist.broadcast_model(model, src=0, device=coll_dev)
    ctx = b.Context(model, device=dev_index)
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

    xf = None
    if a.input == "float":
        ctx.set_float_mode(a.float_mode, a.float_groups)
        xf = b.synth.float_images_device(images)
        torch.cuda.synchronize()
        step = lambda: ctx.infer_float_device(xf, cls, logits)
    else:
        step = lambda: ctx.infer_device(images, cls, logits)
    elapsed, launch_ms = timed_steps(torch, step, a.steps, a.warmup, barrier)
    if distributed:
        # every rank's own time for the K steps, gathered: the line reports the MAX (the contract) and the list, so that a
        # straggler GPU is visible in the one line
        mine = torch.tensor([elapsed, float(np.mean(launch_ms))], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        td.all_gather(every, mine)
        per_rank = [[float(x[0].item()), float(x[1].item())] for x in every]
        elapsed = max(x[0] for x in per_rank)

    # ---- outside the timed region: verification ----------------------------------------------------------
    verified = None
    digest = b.synth.digest_device(cls, first=first, n_bins=model.num_classes)
    if distributed:
        digest = b.dist.allreduce_digest(digest.to(coll_dev))
    torch.cuda.synchronize()
    dg = digest.cpu().numpy()
    digest_hex = hex(int(dg[0].astype(np.uint64)))
    hist = dg[1:].astype(np.int64)
    headline = a.model == "fc_4bitsym_64" and a.dist == 0 and n_global == 100_000_000 and not a.header and a.input == "int8"
    if rank == 0 and not a.no_verify:
        if a.input == "float":
            verified = checker().verify_float_sample(torch, model, xf, cls, logits, n) and int(hist.sum()) == n_global
        else:
            verified = checker().verify_sample(torch, model, images, cls, logits, n) and int(hist.sum()) == n_global
        if headline:   # every one of the 1e8 class ids, through the order-independent digest the oracle produced on host cores
            verified = verified and int(dg[0].astype(np.uint64)) == ORACLE_DIGEST_1E8

    if rank == 0:
        counters = load_counters(b.LIB_PATH)
        total = n_global * a.steps
        bpi = (1024 if a.input == "float" else 256) + 4 + (4 * model.num_classes if a.logits else 0)
        avg_ms = float(np.mean(launch_ms))
        achieved = n * bpi / (avg_ms * 1e-3) / 1e9
        kname = kernel_name(b, ctx, model, n, a.cnn_variant >= 0)
        if a.input == "float":      # what the library says the timed call ran (CNNs: cnn_li_fused_pipe_kernel<float>; short calls: two kernels)
            kname = ctx.last_kernel
        # the box's plain read rate over the same resident images, same process, same stream (bnm_stream_read_device)
        sink = torch.zeros(1, dtype=torch.int32, device=dev)
        src_t = xf if a.input == "float" else images
        _, rd_ms = timed_steps(torch, lambda: b.synth.stream_read_device(src_t, sink), 5, 2)
        rd = float(np.median(rd_ms))
        in_bytes = 1024 if a.input == "float" else 256
        stream_read = {"ms": rd, "bytes": n * in_bytes, "GB/s": n * in_bytes / (rd * 1e-3) / 1e9,
                       "what": "plain nontemporal 16 B/lane loads of this rank's resident images, nothing written; median of 5 launches after 2"}
        traffic, traffic_source = None, None
        tj = counters.get("pmc_traffic.json")
        if tj and headline and world == 1 and not a.logits and tj.get("kernel", "").startswith(kname):
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_source = f"replayed from profiles/pmc_traffic.json (rocprofv3 --pmc pass {tj.get('source')}; not measured by this run)"
        out = {
            "metric": ("16x16 int8 MNIST inferences/s, FC 4bitsym 64-64-64 (BitNetMCU_model_fc.h), bit-exact vs C reference"
                       if a.model == "fc_4bitsym_64" else f"16x16 int8 inferences/s, model {a.model}, bit-exact vs C reference"),
            "value": total / elapsed,
            "unit": "inferences/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "per_rank_ms_per_step": ([x[0] / a.steps * 1e3 for x in per_rank] if distributed else None),
            "per_rank_kernel_ms": ([x[1] for x in per_rank] if distributed else None),
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "i8",
            "data": "synthetic",
            "config": {
                "workload": (f"BASELINE configs[{1 if world == 1 else 4}]: {a.model}, {n} synthetic 16x16 int8 images per GPU resident in HBM "
                             f"(dist {'U' if a.dist == 0 else 'M'}), class ids{' + logits' if a.logits else ''} written") if a.input == "int8" else
                            (f"SURVEY 8(f) row 1: {a.model}, {n} synthetic 16x16 FLOAT32 images per GPU resident in HBM (int8 dist "
                             f"{'U' if a.dist == 0 else 'M'} x 1/127), quantised as test_inference.py:140-141 inside the kernel, class ids written"),
                "images_per_gpu": n,
                "global_images": n_global,
                "path": ctx.path,
                "model_source": model_source,
                "parallelism": f"dp{world} image-shard ({a.scaling} scaling), no data-path collective"
                               + (" - ALL RANKS ON GPU 0 (functional check of the N > 1 path, not a scaling measurement)" if a.ranks_share_device else ""),
                "dist_backend": (a.dist_backend if distributed else None),
                "rccl_ranks": (td.get_world_size() if distributed and a.dist_backend == "nccl" else None),
                "ranks_share_device": bool(a.ranks_share_device),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": kname,
                "launched": ctx.last_kernel,      # bnm_ctx_last_kernel: what the library says the last call ran
                "avg_launch_ms": avg_ms,
                "median_launch_ms": float(np.median(launch_ms)),
                "min_launch_ms": float(np.min(launch_ms)),
                "algorithmic_bytes_per_launch": n * bpi,
                "fused_variant": ctx.variant,
                "mfma": mfma_roofline(b, ctx, model, total / elapsed, kname, counters),
                "counters_dropped": counters.get("dropped") or None,
                "stream_read": stream_read,
                # > 1: the kernel takes longer than merely reading its input on this box
                "time_vs_stream_read": float(np.median(launch_ms)) / rd,
            },
            "verified_vs_oracle": verified,
            "digest": digest_hex,
            "digest_expected": hex(ORACLE_DIGEST_1E8) if headline else None,
            "class_histogram": hist.tolist(),
        }
        if world == 1 and not a.no_extra and a.input == "int8":
            out["extra_configs"] = extra_configs(a, np, torch, b, dev, images, cls, n, counters, stream_read)
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(b, a.model, a.dist, a.cpu_seconds)
            if "extra_configs" in out and a.model == "fc_4bitsym_64":
                # configs[2] / configs[3]: the reference's CPU path on the same host cores, a quarter of the headline's sample each
                for row, name in (("ternary_alu", "tern_96"), ("cnn_64", "cnn_64")):
                    out["extra_configs"][row]["cpu_baseline"] = cpu_baseline(b, name, 0, max(1.0, a.cpu_seconds / 4))
        emit(out, a.full_json)
    if distributed:
        td.barrier()
        td.destroy_process_group()


def extra_configs(a, np, torch, b, dev, images, cls, n, counters, stream_read):
    """The other BASELINE configs and variants on the SAME resident image set, each checked against the oracle on a sample.
    Every entry: inferences/s from HIP events over `steps` launches after `warmup`, the roofline that binds it."""
    cj = counters.get("pmc_counters.json", {})
    res = {}
    ck = None if a.no_verify else checker()
    rd_rate = stream_read["GB/s"]

    def valu(rate, kernel, bpi, model, model_key=None):
        """VALU-bound kernels.  frac = ALGORITHMIC fraction of the VALU issue roofline: the model's multiply-accumulates at 4 per
        lane of a v_dot4 (the densest integer VALU form), i.e. MACs / 256 wave instructions per image - it falls when the kernel
        spends instructions on anything else.  pipe_utilisation = the kernel's own instruction count x rate / peak (how busy the
        pipe is, whatever it is busy with)."""
        macs = model_macs(b, model)
        alg = macs / MACS_PER_WAVE_DOT4
        r = {"bound": "valu", "unit": "wave64 VALU instructions/s", "peak": VALU_PEAK_WAVE_INSTR_PER_S,
             "macs_per_image": macs, "algorithmic_valu_per_image": alg, "achieved": rate * alg,
             "frac": rate * alg / VALU_PEAK_WAVE_INSTR_PER_S,
             "definition": "frac = MACs per image / (4 MACs x 64 lanes per wave64 v_dot4) x inferences/s / (1024 SIMDs x 2.4 GHz / 4 cycles)",
             "hbm_frac": rate * bpi / 1e9 / HBM_PEAK_GBS}
        c = cj.get(f"{kernel}@{model_key}") if model_key else cj.get(kernel)
        if c and "valu_per_image" in c:
            r.update({"valu_per_image": c["valu_per_image"],
                      "pipe_utilisation": rate * c["valu_per_image"] / VALU_PEAK_WAVE_INSTR_PER_S,
                      "valu_per_image_source": f"instruction count replayed from profiles/pmc_counters.json (SQ_INSTS_VALU, which counts MFMA instructions too; pass {c.get('source')})"})
        return r

    def run(name, model_name, count, steps, warmup, dist=0, want_logits=False, variant=-1, path=0, note=None, cnn_variant=-1, want_planes=False):
        model, src = load_model_through_the_text_parser(b, model_name)
        ctx = b.Context(model, device=dev.index)
        if cnn_variant >= 0:
            ctx.set_cnn_variant(cnn_variant)
        if path:
            ctx.set_path(path)
        if variant >= 0:
            ctx.set_tuning(variant, 0)
        x = images[:count]
        c = cls[:count]
        lg = torch.empty((count, model.num_classes), dtype=torch.int32, device=dev) if want_logits else None
        _, ms = timed_steps(torch, lambda: ctx.infer_device(x, c, lg), steps, warmup)
        rate = count / (float(np.mean(ms)) * 1e-3)
        ok = None if a.no_verify else ck.verify_sample(torch, model, x, c, lg, count)
        res[name] = {"model": model_name, "model_source": src, "images": count, "dist": "U" if dist == 0 else "M", "steps": steps, "warmup": warmup,
                     "value": rate, "unit": "inferences/s", "avg_launch_ms": float(np.mean(ms)), "median_launch_ms": float(np.median(ms)),
                     "min_launch_ms": float(np.min(ms)),
                     "kernel": kernel_name(b, ctx, model, count, cnn_variant >= 0), "launched": ctx.last_kernel, "path": ctx.path,
                     "fused_variant": ctx.variant, "verified_vs_oracle": ok,
                     "mfma_per_image": model_mfmas_per_image(b, model) if ctx.path == b.PATH_FUSED_MFMA else None}
        if note:
            res[name]["note"] = note
        planes = ctx.cnn_planes if want_planes else None      # conv3 operand planes of the lane = image kernels (bnm_ctx_cnn_planes)
        ctx.close()
        del lg
        return (rate, model, planes) if want_planes else (rate, model)

    def hbm_entry(name, rate, bpi):
        g = rate * bpi / 1e9
        res[name]["roofline"] = {"bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
                                 "time_vs_stream_read": res[name]["avg_launch_ms"] / (res[name]["images"] * 256 / rd_rate / 1e6)}
        if res[name].get("mfma_per_image"):
            per = res[name].pop("mfma_per_image")
            res[name]["roofline"]["mfma"] = {"per_image": per, "achieved_per_s": rate * per, "peak_per_s": MFMA_I8_32X32X32_PEAK_PER_S,
                                             "frac": rate * per / MFMA_I8_32X32X32_PEAK_PER_S}

    def float_row(name, model_name, count, steps, warmup, mode, note):
        """Float images in, class ids out (bnm_infer_float_device).  mode 0: the library's choice - ONE kernel where the fused
        float-input kernel exists; mode 2: bnm_quantize_input_device + the model's kernels (int8 round trip through HBM)."""
        model, src = load_model_through_the_text_parser(b, model_name)
        ctx = b.Context(model, device=dev.index)
        ctx.set_float_mode(mode)
        if model.kind == b.KIND_CNN:
            ctx.set_cnn_variant(3)
        xf = b.synth.float_images_device(images[:count])
        c = cls[:count]
        torch.cuda.synchronize()
        _, ms = timed_steps(torch, lambda: ctx.infer_float_device(xf, c), steps, warmup)
        rate = count / (float(np.mean(ms)) * 1e-3)
        ok = None if a.no_verify else ck.verify_float_sample(torch, model, xf, c, None, count)
        # the box's plain read of the same float bytes, same process
        sink = torch.zeros(1, dtype=torch.int32, device=dev)
        _, rd_ms = timed_steps(torch, lambda: b.synth.stream_read_device(xf, sink), 3, 1)
        rd = float(np.median(rd_ms))
        fused = ctx.float_fused
        fused_name = ("cnn_li_fused_pipe_kernel<float>" if ctx.cnn_pipelined else "cnn_li_fused_kernel<float>") if model.kind == b.KIND_CNN else "fused_fc_f32_kernel"
        bpi = BYTES_PER_INFERENCE_FLOAT
        g = rate * bpi / 1e9
        res[name] = {"model": model_name, "model_source": src, "images": count, "dist": "U", "steps": steps, "warmup": warmup,
                     "input": "float32 [n][256] resident in HBM: the synthetic int8 images x 1/127 (bitnetmcu_amd/synth.py float_images)",
                     "value": rate, "unit": "inferences/s", "avg_launch_ms": float(np.mean(ms)), "median_launch_ms": float(np.median(ms)),
                     "min_launch_ms": float(np.min(ms)),
                     "kernel": fused_name if fused else "quantize_input_kernel+" + kernel_name(b, ctx, model, count, True),
                     "launched": ctx.last_kernel,
                     "launches_per_step": 1 if fused else 2 * ((count + (1 << 22) - 1) >> 22), "path": ctx.path, "verified_vs_oracle": ok,
                     "note": note,
                     "roofline": {"bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
                                  "algorithmic_bytes_per_inference": bpi,
                                  "moved_bytes_per_inference": bpi if fused else bpi + 512,
                                  "stream_read": {"ms": rd, "bytes": count * 1024, "GB/s": count * 1024 / (rd * 1e-3) / 1e9},
                                  "time_vs_stream_read": float(np.median(ms)) / rd}}
        per = model_mfmas_per_image(b, model)
        res[name]["roofline"]["mfma"] = {"per_image": per, "achieved_per_s": rate * per, "peak_per_s": MFMA_I8_32X32X32_PEAK_PER_S,
                                         "frac": rate * per / MFMA_I8_32X32X32_PEAK_PER_S}
        if model.kind == b.KIND_CNN:
            # VALU-bound like the int8 form: the row's frac is the ALGORITHMIC VALU fraction (MACs at 4 per lane of a v_dot4), labelled;
            # the HBM fraction of the 1,028 B per inference is kept beside it
            macs = model_macs(b, model)
            hb = res[name]["roofline"]
            res[name]["roofline"] = {"bound": "valu", "unit": "wave64 VALU-pipe instructions/s", "peak": VALU_PEAK_WAVE_INSTR_PER_S,
                                     "achieved": rate * macs / MACS_PER_WAVE_DOT4, "frac": rate * macs / MACS_PER_WAVE_DOT4 / VALU_PEAK_WAVE_INSTR_PER_S,
                                     "frac_is": "ALGORITHMIC (MACs / 256 per image): no counter pass of this kernel binary is replayed",
                                     "macs_per_image": macs, "hbm_frac": hb["frac"], "algorithmic_bytes_per_inference": bpi,
                                     "moved_bytes_per_inference": hb["moved_bytes_per_inference"], "stream_read": hb["stream_read"],
                                     "time_vs_stream_read": hb["time_vs_stream_read"]}
        ctx.close()
        del xf
        torch.cuda.empty_cache()

    # the reference's Python-side input quantisation (test_inference.py:140-141) fused into the FC kernel
    n_f = min(n, a.float_images)
    if n_f > 0:
        float_row("fc_float_input", "fc_4bitsym_64", n_f, 10, 3, 0, "float images -> class ids, ONE kernel (bnm_fused_f32_kernel.hpp): HBM roofline on 1,028 B per inference")
        float_row("fc_float_input_two_kernels", "fc_4bitsym_64", n_f, 3, 1, 2, "the same call as quantise + infer (1,540 B moved per inference): what the fused kernel replaces")
        float_row("tern_float_input", "tern_96", n_f, 5, 2, 0, "the 4-tile class of the fused float-input kernel (ternary 96-96-96)")
        float_row("binary160_float_input", "doc12k_binary", n_f, 5, 2, 0, "the 6-tile class of the fused float-input kernel (binary 160-160-160: one landing group per wave)")
        # ... and the CNN: the one-kernel form with the quantisation in front of its convolution operands (VALU-bound like the int8 form:
        # the HBM fraction of its 1,028 B per inference is small; `value` is what to compare with the cnn_64 row)
        n_fc = min(n_f, 10_000_000)
        float_row("cnn_float_input", "cnn_64", n_fc, 3, 1, 0, "float images -> class ids through the one-kernel CNN (quantisation fused; three waves per SIMD)")
        float_row("cnn_float_input_two_kernels", "cnn_64", n_fc, 3, 1, 2, "the same call as quantise + the one-kernel CNN on int8")

    # the reference's training forward (models.py:70-90 FCMNIST over BitLinear, BitNetMCU.py:214-235)
    # as ONE kernel behind a weight-preparation launch - float32 rows in, float32 logits out
    def qat_row(name, rows, note):
        from bitnetmcu_amd import qat
        torch.manual_seed(20240324)
        widths = [256, 64, 64, 64, 10]
        ws = [torch.randn(widths[l + 1], widths[l], device=dev) * 0.08 for l in range(4)]
        ss = [w.abs().mean().reshape(1) / 0.25 for w in ws]          # update_clipping_scalar(..., 'prop', 0.25), BitNetMCU.py:107-110
        qts = ["4bitsym"] * 4
        # float rows with a continuous distribution (the int8 synthetic images x 1/127 put half of the rows on EXACT rounding ties
        # - 64 x 127/128 = 63.5 whenever a row holds -128 -, which no two float evaluation orders resolve alike)
        gen = torch.Generator(device=dev).manual_seed(20240324)
        xq = torch.randn(rows, 256, device=dev, generator=gen) * (torch.rand(rows, 1, device=dev, generator=gen) * 2 + 0.05)
        torch.cuda.synchronize()
        out = [None]
        def step():
            out[0] = qat.fc_model_forward(xq, ws, ss, qts, "RMS")
        _, ms = timed_steps(torch, step, 20, 3)
        rate = rows / (float(np.median(ms)) * 1e-3)
        ok = None
        if not a.no_verify:
            # a floating-point op: against the restated reference formula on torch's own fp32 kernels (bitnetmcu_amd/qat.py
            # fc_model_reference, pinned bit for bit to the reference module by tests/test_qat_cpu.py), tolerances of
            # tests/test_gpu_qat_model.py: 90 % of the rows within 5e-4 of the row's largest logit, all within 6e-2
            m = min(rows, 50_000)
            want, _ = qat.fc_model_reference(xq[:m], ws, [t[0] for t in ss], qts, "RMS")
            err = (out[0][:m] - want).abs().max(dim=1).values / want.abs().max(dim=1).values
            ok = bool((err <= 5e-4).float().mean() >= 0.9 and err.max() <= 6e-2 and not torch.isnan(out[0]).any())
        bpr = 1024 + 4 * widths[-1]
        g = rate * bpr / 1e9
        res[name] = {"model": "FCMNIST 64-64-64 4bitsym RMS PerTensor (random weights, clipping scalars as training.py's 'prop')", "rows": rows,
                     "steps": 20, "warmup": 3, "value": rate, "unit": "rows/s", "median_call_ms": float(np.median(ms)), "min_call_ms": float(np.min(ms)),
                     "avg_call_ms": float(np.mean(ms)), "kernel": "qat_model_prep_kernel+qat_fc_model_fwd_kernel", "launches_per_step": 2,
                     "verified_vs_oracle": ok, "verified_against": "the restated reference formula in fp32 (floating-point op: tolerances, not bit-exact)",
                     "note": note,
                     "roofline": {"bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
                                  "algorithmic_bytes_per_row": bpr,
                                  "definition": "(1,024 B of float32 read + 4 B x classes written) x rows / the median time of a whole call (both launches)"}}
        c = cj.get("qat_fc_model_fwd_kernel")
        if c:      # a counter pass of the same kernel binary (code hash checked by load_counters), replayed
            res[name]["roofline"].update({"traffic_bytes_per_row": c.get("hbm_bytes_per_image"), "valu_per_row": c.get("valu_per_image"),
                                          "mfma_per_row": c.get("mfma_per_image"), "valu_busy_frac": c.get("valu_busy_frac"),
                                          "counters_source": f"replayed from profiles/pmc_counters.json (pass {c.get('source')}, 1e7 rows per call; not measured by this run)"})
        del xq
        torch.cuda.empty_cache()
    # ... and of the model trainingparameters.yaml names, CNNMNIST (models.py:93-139): the convolution front as ONE kernel
    # (csrc/bnm_qat_cnn.hip), the FC stack behind it as the kernel above - float32 images in, float32 logits out, four launches
    def qat_cnn_row(name, rows, note):
        from bitnetmcu_amd import qat
        torch.manual_seed(20240419)
        mod = qat.CNNMNIST(96, 64, 0, cnn_width=64, QuantType="4bitsym", WScale="PerTensor", NormType="RMS", num_classes=10).to(dev)
        gen = torch.Generator(device=dev).manual_seed(20240419)
        xq = torch.randn(rows, 1, 16, 16, device=dev, generator=gen) * (torch.rand(rows, 1, 1, 1, device=dev, generator=gen) * 2 + 0.05)
        torch.cuda.synchronize()
        out = [None]
        with torch.no_grad():
            fused = bool(mod.front_fused(xq) and mod.fused(xq))
            def step():
                out[0] = mod(xq)
            _, ms = timed_steps(torch, step, 10, 2)
            rate = rows / (float(np.median(ms)) * 1e-3)
            ok = None
            if not a.no_verify:
                # floating point: against the layer-by-layer path (qat.BitConv2d's and BitLinear's ops, each pinned to the reference's
                # layer by tests/test_gpu_qat.py) on a sample, the end-to-end tolerances of tests/test_gpu_qat_model.py
                m = min(rows, 4096)
                y = xq[:m]
                for k in list(mod.model):
                    y = k(y)
                want = mod.classifier(y)
                err = (out[0][:m] - want).abs().max(dim=1).values / want.abs().max(dim=1).values
                ok = bool(fused and (err <= 5e-4).float().mean() >= 0.9 and err.max() <= 6e-2 and torch.isfinite(out[0]).all())
        # the front's arithmetic: 64 channels x (196 + 144 + 16) outputs x 9 multiply-adds = 205,056 per image = 1,602 packed
        # wave64 instructions (v_pk_fma_f32: 128 multiply-adds each)
        alg = 205056 / 128.0
        res[name] = {"model": "CNNMNIST 64 channels, FC 256-96-64-10, 8bit convolutions / 2bitsym + 4bitsym FC, RMS (random weights)", "rows": rows,
                     "steps": 10, "warmup": 2, "value": rate, "unit": "images/s", "median_call_ms": float(np.median(ms)), "min_call_ms": float(np.min(ms)),
                     "kernel": "qat_cnn_prep_kernel+qat_cnn_front_kernel+qat_model_prep_kernel+qat_fc_model_fwd_kernel", "launches_per_step": 4,
                     "verified_vs_oracle": ok, "verified_against": "the layer-by-layer ops in fp32 (floating-point op: tolerances, not bit-exact)",
                     "note": note,
                     "roofline": {"bound": "valu", "unit": "wave64 VALU instructions/s", "peak": VALU_PEAK_WAVE_INSTR_PER_S, "achieved": rate * alg,
                                  "frac": rate * alg / VALU_PEAK_WAVE_INSTR_PER_S, "algorithmic_valu_per_image": alg,
                                  "definition": "the front's 205,056 multiply-adds per image as packed float32 instructions (128 per wave64 instruction) x images / the "
                                                "median time of a whole call (four launches), against 1,024 SIMDs x 2.4 GHz / 4 clocks"}}
        c = cj.get("qat_cnn_front_kernel")
        if c and "valu_per_image" in c:
            res[name]["roofline"].update({"valu_per_image": c["valu_per_image"], "pipe_utilisation": rate * c["valu_per_image"] / VALU_PEAK_WAVE_INSTR_PER_S,
                                          "traffic_bytes_per_image": c.get("hbm_bytes_per_image"), "valu_busy_frac": c.get("valu_busy_frac"),
                                          "counters_source": f"replayed from profiles/pmc_counters.json (pass {c.get('source')}; not measured by this run)"})
        del xq
        torch.cuda.empty_cache()
    if a.model == "fc_4bitsym_64" and n >= 1000:
        qat_cnn_row("qat_cnn_forward", min(n, 1_000_000), "QAT forward of the reference's CNNMNIST, 1e6 images per call (convolution front + FC stack: two kernels)")
        qat_row("qat_fc_forward", min(n, 1_000_000), "QAT forward of the whole FC model, 1e6 rows per call (VERDICT r05 next #4)")
        if n >= 10_000_000:
            qat_row("qat_fc_forward_1e7", 10_000_000, "the same at 1e7 rows per call")

    n_cnn = min(n, 10_000_000)
    # configs[2]: ternary 96-96-96, bit-unpack / sign-accumulate ALU kernel, no MFMA — bound by the VALU issue rate
    # (selected by name: the library's AUTO path runs ternary models on the MFMA kernels, 5x faster - next entry)
    r, m = run("ternary_alu", "tern_96", n, 3, 1, path=b.PATH_TERNARY_ALU, note="BASELINE configs[2]: the no-MFMA kernel, selected explicitly")
    res["ternary_alu"]["roofline"] = valu(r, "ternary_stream_kernel", BYTES_PER_INFERENCE, m)
    r, _ = run("ternary_mfma_generic", "tern_96", n, 10, 3, note="the same model on the library's default (AUTO) path")
    hbm_entry("ternary_mfma_generic", r, BYTES_PER_INFERENCE)
    # configs[3]: CNN 64-wide.  Default: ONE kernel per call - the lane = image front end (all three convolutions as Toeplitz products
    # on the matrix cores, 44 MFMAs per channel and 32-image tile) with the FC tail in the same wave.  What binds it is the VALU
    # issue rate of the convolutions' epilogues, so the row's `frac` is the kernel's OWN VALU + MFMA instruction count per image
    # (SQ_INSTS_VALU of the counter pass of this kernel binary, replayed) x the measured rate against the VALU issue peak; beside it
    # the ALGORITHMIC int8 operations (2 x MACs of the reference's loops) against the dense int8 matrix-core peak, the issued MFMAs
    # against the same peak, and the HBM fraction of the 260 algorithmic bytes (+ counter traffic where measured)
    def cnn_row(name, model_name, note, cnn_variant=-1):
        r, m, planes = run(name, model_name, n_cnn, 10, 2, note=note, cnn_variant=cnn_variant, want_planes=True)
        kern = res[name]["kernel"].split("+")[-1]
        li = kern in ("cnn_li_kernel", "cnn_li_fused_kernel", "cnn_li_fused_pipe_kernel")
        macs = model_macs(b, m)
        # (the three-plane A/B form of the one-kernel CNN is another binary than the one the replayed counters describe)
        c = None if cnn_variant == 5 else (cj.get(f"{kern}@{model_name}") or cj.get(kern))
        roof = {"bound": "valu", "unit": "wave64 VALU-pipe instructions/s", "peak": VALU_PEAK_WAVE_INSTR_PER_S,
                "definition": "frac = (VALU + MFMA instructions per image, counted by SQ_INSTS_VALU on this kernel binary) x inferences/s / "
                              "(1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction): the share of the VALU issue slots the kernel's own "
                              "instruction stream fills",
                "macs_per_image": macs}
        if c and "valu_per_image" in c:
            roof.update({"valu_per_image": c["valu_per_image"], "achieved": r * c["valu_per_image"],
                         "frac": r * c["valu_per_image"] / VALU_PEAK_WAVE_INSTR_PER_S,
                         "valu_per_image_source": f"replayed from profiles/pmc_counters.json (SQ_INSTS_VALU, pass {c.get('source')}, same kernel binary: "
                                                  f"code_sha1 {c.get('code_sha1', '')[:12]}); not measured by this run"})
            if "hbm_bytes_per_image" in c:
                roof["traffic_bytes_per_image"] = c["hbm_bytes_per_image"]
        else:       # no counter pass of this binary: the algorithmic fraction (MACs at 4 per lane of a v_dot4), labelled as such
            roof.update({"achieved": r * macs / MACS_PER_WAVE_DOT4, "frac": r * macs / MACS_PER_WAVE_DOT4 / VALU_PEAK_WAVE_INSTR_PER_S,
                         "frac_is": "ALGORITHMIC (MACs / 256 per image): no counter pass of this kernel binary to replay"})
        roof["int8_ops_algorithmic"] = {"per_image": 2 * macs, "achieved_per_s": r * 2 * macs, "peak_per_s": MFMA_I8_32X32X32_PEAK_PER_S * 65536.0,
                                        "frac": r * 2 * macs / (MFMA_I8_32X32X32_PEAK_PER_S * 65536.0),
                                        "definition": "2 x the reference loops' multiply-accumulates per image x inferences/s / dense int8 peak (5.03e15 op/s)"}
        roof["hbm_frac"] = r * BYTES_PER_INFERENCE / 1e9 / HBM_PEAK_GBS
        if li:
            per_tile = 14 + 24 + 2 * planes      # conv1 14, conv2 24 (two operand planes), conv3 2 per operand plane
            per = per_tile * m.layer(0).out_channels / 32.0 + model_mfmas_per_image(b, m)
            roof["mfma"] = {"per_image": per, "per_channel_tile": per_tile, "conv3_planes": planes, "achieved_per_s": r * per,
                            "peak_per_s": MFMA_I8_32X32X32_PEAK_PER_S, "frac": r * per / MFMA_I8_32X32X32_PEAK_PER_S,
                            "definition": f"{per_tile} v_mfma_i32_32x32x32_i8 per channel and 32-image tile (conv1 14, conv2 24, conv3 {2 * planes}: "
                                          "bnm_ctx_cnn_planes) + the FC tail's: the Toeplitz form issues ~12 x the algorithmic int8 operations"}
        res[name]["roofline"] = roof
    cnn_row("cnn_64", "cnn_64", "BASELINE configs[3]")
    cnn_row("cnn_64_four_waves", "cnn_64", "the one-kernel form at four waves per SIMD (round 5's first form: SDWA conv1 epilogue, a wave waits for its MFMA results; A/B)", cnn_variant=6)
    cnn_row("cnn_64_three_planes", "cnn_64", "the one-kernel form with conv3's third operand plane kept (the model's weights rule it out: bnm_cnn_li_tables; A/B)", cnn_variant=5)
    cnn_row("cnn_64_two_launches", "cnn_64", "the same front end with the FC tail as its own launch over act rows in HBM (round 4's form: 772 B moved per image)", cnn_variant=4)
    cnn_row("cnn_64_channel_kernel", "cnn_64", "the same model on round 3's front end (a lane = a channel, conv1 only on the matrix cores)", cnn_variant=1)
    # the reference's smaller published CNNs (mcu/BitNetMCU_model_cnn_16.h, _48.h)
    for nm in ("mcu_cnn_16", "mcu_cnn_48"):
        cnn_row(nm, nm, "reference's published CNN family")
    # headline model through the generic kernel (what any non-zoo 64-wide export would get)
    r, _ = run("fc_generic_kernel", "fc_4bitsym_64", n, 10, 3, variant=4)
    hbm_entry("fc_generic_kernel", r, BYTES_PER_INFERENCE)
    # the reference's documented 12 KB family (docs/documentation.md:169-183; its 4-bit member is the headline model): random
    # weights of those shapes and codecs from the reference's own exporter (tests/golden/make_doc12k_headers.py)
    for nm in ("doc12k_binary", "doc12k_ternary", "doc12k_2bit", "doc12k_8bit"):
        r, _ = run(nm, nm, n, 10, 3, note="reference docs' 12 KB model family")
        hbm_entry(nm, r, BYTES_PER_INFERENCE)
    # ... and its ternary member (128-128-112) on the no-MFMA kernel: streamed weights, one image per lane
    r, m = run("doc12k_ternary_alu", "doc12k_ternary", n, 3, 1, path=b.PATH_TERNARY_ALU,
               note="the documented 12 KB ternary shape on the no-MFMA kernel, selected explicitly")
    res["doc12k_ternary_alu"]["roofline"] = valu(r, "ternary_stream_kernel", BYTES_PER_INFERENCE, m, "doc12k_ternary")
    # headline model, class ids + logits (300 B per inference)
    if n <= 100_000_000:
        r, _ = run("fc_logits", "fc_4bitsym_64", n, 10, 3, want_logits=True)
        hbm_entry("fc_logits", r, BYTES_PER_INFERENCE_LOGITS_10)
        # what the same bytes cost with no arithmetic on this box: 32-image tiles read, 44 bytes per image written (an id + ten int32
        # logits, nontemporal 16 B/lane stores of whole tiles) - a stream that mixes 13 % writes into its reads is slower per byte
        # than the plain read (x 1.07 .. 1.13 of the byte-proportional time, profiles/r04/rw_sweep_r05e.log), and the row's distance
        # from THAT is what the kernel itself costs.  The fastest of three shapes of the probe (tiles per batch / waves per SIMD).
        out_probe = torch.empty(((n // 32) * 32 * 44 + 64) // 4, dtype=torch.int32, device=dev)
        probes = {}
        for mode, what in ((8 + 32 * 4, "8 tiles per batch, 6 waves per SIMD"), (8 + 32 * 2, "8 tiles per batch, 4 waves per SIMD"),
                           (4, "4 tiles per batch, 2 waves per SIMD")):
            _, rw_ms = timed_steps(torch, lambda: b.synth.stream_rw_device(images[:(n // 32) * 32], out_probe, 44, mode), 5, 2)
            probes[what] = float(np.median(rw_ms))
        best = min(probes, key=probes.get)
        rw = probes[best]
        res["fc_logits"]["roofline"]["stream_read_write"] = {
            "ms": rw, "bytes": (n // 32) * 32 * 300, "GB/s": (n // 32) * 32 * 300 / (rw * 1e-3) / 1e9, "shape": best, "ms_by_shape": probes,
            "vs_byte_proportional_read": rw / (stream_read["ms"] * 300 / 256) if n == stream_read["bytes"] // 256 else None,
            "what": "bnm_stream_rw_device: the row's 256 B read + 44 B written per image, no arithmetic; median of 5 launches after 2"}
        res["fc_logits"]["roofline"]["time_vs_stream_read_write"] = res["fc_logits"]["median_launch_ms"] / rw
        del out_probe
    # headline model on Dist-M (MNIST-like value statistics): refill the resident set in place
    if a.dist == 0:
        b.synth.fill_device(images, first=0, dist=1)
        torch.cuda.synchronize()
        r, _ = run("fc_dist_m", "fc_4bitsym_64", n, 10, 3, dist=1)
        hbm_entry("fc_dist_m", r, BYTES_PER_INFERENCE)
    for e in res.values():
        e.pop("mfma_per_image", None)      # (consumed by hbm_entry on the MFMA rows)
    return res


if __name__ == "__main__":
    main()
