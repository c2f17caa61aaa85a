#!/usr/bin/env python3
"""Turn the per-kernel JSON tables of profiles/pmc_table.py (--json) into the two small files bench.py replays:
  profiles/pmc_traffic.json   HBM bytes per launch of the headline kernel (FETCH_SIZE x1024 x2 + WRITE_SIZE x1024)
  profiles/pmc_counters.json  per kernel: VALU instructions per image, MFMA busy fraction, wait shares
usage: python profiles/make_counters_json.py <tag> name[@model]=table.json:images_per_launch [...]
  e.g. make_counters_json.py r02k fused_fc_dual_kernel=gpurun_out/pmc_dual/table.json:100000000 ..."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bitnetmcu_amd import codeobj  # noqa: E402

# every entry is stamped with the machine-code hash of its kernel in the library the counters were collected from (the in-tree
# build, which is what travels to the GPU box): bench.py replays an entry only next to that same binary (bench.load_counters)
LIB = os.environ.get("BNM_LIBRARY") or os.path.join(REPO, "bitnetmcu_amd", "libbitnetmcu_hip.so")
HASHES = codeobj.kernel_hashes(LIB)


def stamp(e):
    ks = codeobj.find_kernels(HASHES, e["kernel"])
    if len(ks) == 1:
        e["mangled"], e["code_sha1"] = ks[0], HASHES[ks[0]]
    else:
        print("cannot stamp", e["kernel"], "->", ks)
    return e



def main():
    tag = sys.argv[1]
    out = {}
    pc = os.path.join(REPO, "profiles", "pmc_counters.json")
    if os.path.isfile(pc):
        out = json.load(open(pc))
    for spec in sys.argv[2:]:
        name, rest = spec.split("=")
        alias = name
        if "@" in name:                      # kernel@model: stored under that key (bench.py looks "kernel@model" up first)
            name = name.split("@")[0]
        path, images = rest.rsplit(":", 1)
        images = float(images)
        tab = json.load(open(path))
        key = [k for k in tab if k.startswith(name)]
        if not key:
            print("no kernel", name, "in", path)
            continue
        d = tab[key[0]]
        e = stamp({"source": tag, "kernel": key[0], "images_per_launch": images})
        if "SQ_INSTS_VALU" in d:
            e["valu_per_image"] = d["SQ_INSTS_VALU"] / images
        if "SQ_INSTS_MFMA" in d:
            e["mfma_per_image"] = d["SQ_INSTS_MFMA"] / images
        if "SQ_INSTS_SALU" in d:
            e["salu_per_image"] = d["SQ_INSTS_SALU"] / images
        if "GRBM_GUI_ACTIVE" in d:
            cyc = d["GRBM_GUI_ACTIVE"] / 8.0
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
                e["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
            if "SQ_ACTIVE_INST_VALU" in d:
                e["valu_busy_frac"] = d["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc
        if "SQ_WAVE_CYCLES" in d:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
                if c in d:
                    e[c.lower() + "_share"] = d[c] / d["SQ_WAVE_CYCLES"]
        if "avg_ms" in d:
            e["stats_avg_ms"] = d["avg_ms"]
        if "FETCH_SIZE" in d:
            fb = d["FETCH_SIZE"] * 1024 * 2
            wb = d.get("WRITE_SIZE", 0.0) * 1024
            e["hbm_bytes_per_launch"] = fb + wb
            e["hbm_bytes_per_image"] = (fb + wb) / images
            if name.startswith("fused_fc_dual"):
                json.dump({"hbm_bytes_per_launch": fb + wb, "fetch_bytes": fb, "write_bytes": wb, "source": tag, "kernel": key[0],
                           "mangled": e.get("mangled"), "code_sha1": e.get("code_sha1"),
                           "images_per_launch": images,
                           "note": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (MI355X_MICROARCH.md HBM section: gfx950 reports half the bytes of 16 B/lane reads)"},
                          open(os.path.join(REPO, "profiles", "pmc_traffic.json"), "w"))
        out[alias] = e
    json.dump(out, open(pc, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
