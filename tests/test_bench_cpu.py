"""bench.py / __graft_entry__ on a machine without a GPU: they must say so, not fall back to a CPU path."""
import os
import subprocess
import sys

import pytest

from util import REPO


def test_bench_refuses_to_run_without_gpu(bnm):
    if bnm.bnm_device_count() > 0:
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--images", "1000", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs a GPU" in (out.stderr + out.stdout)
    assert "{" not in out.stdout          # no JSON line, no fabricated number


def test_bench_gpus_2_starts_two_ranks_itself(bnm):
    """`python bench.py --gpus 2` without a launcher re-executes under torch.distributed.run with two ranks; without a GPU each
    rank must refuse - and no JSON line appears (in particular no one-rank number labelled n_gpus 1)."""
    if bnm.bnm_device_count() > 0:
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--images", "1000", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600)
    text = out.stderr + out.stdout
    assert out.returncode != 0 and "needs a GPU" in text and "{\"metric\"" not in out.stdout
    assert "local_rank: 1" in text or "rank      : 1" in text or text.count("needs a GPU") >= 2, text[-1500:]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout) and "{" not in out.stdout


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under bitnetmcu_amd/ or include/ may reference it."""
    bad = []
    for root in ("bitnetmcu_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(REPO, root)):
            if "_build" in dirpath or "dlls" in dirpath:
                continue
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", ".c")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    for needle in ("libbnm_oracle", "bitnet_oracle", "oracle/_ref", "import oracle", "from oracle", "orc_"):
                        if needle in text:
                            bad.append((os.path.join(dirpath, f), needle))
    assert not bad, bad


def test_profile_scripts_compile():
    """The measurement scripts only run on the GPU box; a syntax error there costs a box visit."""
    import glob
    import py_compile
    scripts = sorted(glob.glob(os.path.join(REPO, "profiles", "*.py")))
    assert len(scripts) >= 15
    for p in scripts:
        py_compile.compile(p, doraise=True)
    for p in sorted(glob.glob(os.path.join(REPO, "profiles", "*.sh"))):
        r = subprocess.run(["bash", "-n", p], capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stderr)


def test_replayed_counters_are_tied_to_the_kernel_binaries_of_the_loaded_library(tmp_path, monkeypatch):
    """bench.py replays per-kernel constants from separate counter passes (profiles/pmc_*.json).  Every entry carries the
    machine-code hash of its kernel (bitnetmcu_amd/codeobj.py reads the gfx950 code objects out of the library image); an entry
    is replayed only when the loaded library holds that very binary - otherwise it is dropped and listed."""
    import json
    import bench
    from bitnetmcu_amd import codeobj, LIB_PATH
    hashes = codeobj.kernel_hashes(LIB_PATH)
    assert len(hashes) > 100 and all(len(h) == 40 for h in hashes.values())
    dual = codeobj.find_kernels(hashes, "fused_fc_dual_kernel<2, 2, 2, 1, true, 2, 4, true>")
    assert len(dual) == 1 and dual[0].startswith("_Z20fused_fc_dual_kernelILi2ELi2ELi2ELi1ELb1ELi2ELi4ELb1EE")
    assert codeobj.find_kernels(hashes, "no_such_kernel<1>") == []
    # the committed files against the in-tree library: the headline kernel's entries are live
    c = bench.load_counters(LIB_PATH)
    assert "fused_fc_dual_kernel" in c["pmc_counters.json"] and "pmc_traffic.json" in c
    assert c["pmc_counters.json"]["fused_fc_dual_kernel"]["code_sha1"] == hashes[dual[0]]
    # a tampered hash, a missing one and a kernel the library does not hold are dropped, each with its reason
    prof = tmp_path / "profiles"
    prof.mkdir()
    good = dict(c["pmc_counters.json"]["fused_fc_dual_kernel"])
    entries = {"good": good, "stale": dict(good, code_sha1="0" * 40), "unstamped": {k: v for k, v in good.items() if k not in ("code_sha1", "mangled")},
               "gone": dict(good, mangled="_Z9not_thereILi1EEvv")}
    (prof / "pmc_counters.json").write_text(json.dumps(entries))
    (prof / "pmc_traffic.json").write_text(json.dumps(dict(good, code_sha1="1" * 40)))
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    c = bench.load_counters(LIB_PATH)
    assert list(c["pmc_counters.json"]) == ["good"] and "pmc_traffic.json" not in c
    text = " | ".join(c["dropped"])
    assert "stale" in text and "unstamped" in text and "gone" in text and "pmc_traffic.json" in text and "different binary" in text
    assert bench.load_counters(None)["pmc_counters.json"] == {}


def test_bench_line_is_short_enough_for_the_driver():
    """round 5's line was 37 KB and the driver's record of the round held no parsed headline.  The LAST stdout
    line is bench.compact_line(full record): under 4 KB (asserted inside it too), contract keys + roofline + cpu_baseline + compact
    rows; checked here over committed full records of real runs and over an 8-rank record with twice the rows."""
    import copy
    import glob
    import json
    import bench
    recs = sorted(glob.glob(os.path.join(REPO, "profiles", "r05", "r05zz_bench*.json"))) + \
        sorted(glob.glob(os.path.join(REPO, "profiles", "r06", "r06*_bench_full.json")))      # (round 6's records carry the QAT rows as well)
    assert len(recs) >= 5
    for p in recs:
        full = json.load(open(p))
        text = bench.compact_line(full)
        assert len(text) < 4096 and "\n" not in text
        c = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline", "verified_vs_oracle", "digest", "rows"):
            assert k in c, k
        assert abs(c["value"] - full["value"]) <= 1e-4 * full["value"] and c["steps"] == full["steps"] and c["warmup"] == full["warmup"]
        assert abs(c["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-4 and abs(c["roofline"]["traffic"] - full["roofline"]["traffic"]) <= 1e-5 * full["roofline"]["traffic"]
        assert c["roofline"]["traffic_source"] is None or len(c["roofline"]["traffic_source"]) <= 60
        assert set(c["rows"]) == set(k for k, v in full["extra_configs"].items() if "roofline" in v)
        assert "workload" in c["config"] and "model" not in c["config"]
    # the --gpus N line: per-rank times, no rows
    full = json.load(open(recs[0]))
    multi = copy.deepcopy(full)
    multi.pop("extra_configs")
    multi.pop("cpu_baseline")
    multi.update(n_gpus=8, per_rank_ms_per_step=[4.0123456789] * 8, per_rank_kernel_ms=[4.0] * 8)
    multi["config"].update(dist_backend="nccl", rccl_ranks=8, parallelism="dp8 image-shard (weak scaling), no data-path collective")
    c = json.loads(bench.compact_line(multi))
    assert c["n_gpus"] == 8 and len(c["per_rank_ms_per_step"]) == 8 and c["config"]["rccl_ranks"] == 8 and "rows" not in c
    # half as many rows again still fit (round 6's line: 25 rows, 2.7 KB); a record that cannot fit fails loudly instead of printing
    # a line the driver drops
    big = json.load(open(recs[-1]))
    assert len(big["extra_configs"]) >= 25
    for k, v in list(big["extra_configs"].items())[:12]:
        big["extra_configs"][k + "_again"] = v
    assert len(bench.compact_line(big)) < 4096
    for i in range(200):
        big["extra_configs"][f"row_with_a_long_name_{i}"] = v
    with pytest.raises(AssertionError, match="bench line is"):
        bench.compact_line(big)


def test_build_rebuilds_when_the_command_changes(tmp_path, capsys):
    """an object is reused only when the files it includes are older than it AND it was built by the same command
    (flags, architecture, compiler version - the stamp beside it); a changed flag or a foreign stamp recompiles."""
    from bitnetmcu_amd import build
    src = (("bnm_capi_model.cpp", False),)
    d = str(tmp_path)

    def status(**kw):
        build.objects(obj_dir=d, sources=src, **kw)
        out = capsys.readouterr().out
        return "compiled" if "build: compiled" in out else "reused"
    assert status() == "compiled" and status() == "reused"
    assert status(extra_flags=("-DBNM_SOME_FLAG=1",)) == "compiled" and status(extra_flags=("-DBNM_SOME_FLAG=1",)) == "reused"
    assert status() == "compiled"                                       # back to the first command: another object again
    stamp = os.path.join(d, "bnm_capi_model.o.cmd")
    open(stamp, "w").write("an object built by another compiler\n")
    assert status() == "compiled" and status() == "reused"
    os.remove(stamp)                                                    # no stamp: nothing is known about the object
    assert status() == "compiled"
    # the compiler's -MD record with an escaped blank in a path
    with open(os.path.join(d, "x.d"), "w") as f:
        f.write("x.o: /tmp/a\\ dir/one.h \\\n  /tmp/two.h /opt/rocm/include/hip/hip_runtime.h\n")
    assert build.deps_of(os.path.join(d, "x.o")) == ["/tmp/a dir/one.h", "/tmp/two.h"]
