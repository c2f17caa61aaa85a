"""Per-kernel fingerprints of the gfx950 code objects inside the native library.

bench.py replays per-kernel constants that were measured in separate rocprofv3 counter passes (profiles/pmc_counters.json,
pmc_traffic.json).  A constant describes one kernel BINARY: it must not be printed next to a timing of a different one.  Each
entry of those files therefore carries the machine-code hash of its kernel at measurement time (profiles/make_counters_json.py),
and bench.py drops an entry whose hash differs from the library it has loaded.

The library is a host ELF with one clang offload bundle per translation unit in its .hip_fatbin section; each bundle holds a
gfx950 ELF code object whose symbol table lists the kernels (STT_FUNC, mangled names) with address and size inside .text.
Pure Python (struct): no tool of the ROCm installation is needed at run time.
"""
import hashlib
import re
import struct

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    for m in re.finditer(re.escape(_MAGIC), blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + 24)
        q = base + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            q += 24
            triple = blob[q:q + tl]
            q += tl
            if size and b"amdgcn" in triple:
                yield blob[base + off: base + off + size]


def _functions(elf):
    """(name, code bytes) of every STT_FUNC symbol of an ELF64 little-endian code object."""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    for typ_i, s in enumerate(secs):
        if s[1] != 2:                                   # SHT_SYMTAB
            continue
        _, _, _, _, off, size, link, _, _, entsize = s
        str_off = secs[link][4]
        for k in range(size // entsize):
            name_i, info, _, shndx, value, sz = struct.unpack_from("<IBBHQQ", elf, off + k * entsize)
            if (info & 0xF) != 2 or sz == 0 or shndx == 0 or shndx >= shnum:      # STT_FUNC, defined
                continue
            end = elf.index(b"\0", str_off + name_i)
            name = elf[str_off + name_i:end].decode()
            sec = secs[shndx]
            start = sec[4] + (value - sec[3])
            yield name, elf[start:start + sz]


def kernel_hashes(lib_path):
    """{mangled kernel name: sha1 hex of its machine code} over every gfx950 code object in the library."""
    blob = open(lib_path, "rb").read()
    out = {}
    for co in _code_objects(blob):
        for name, code in _functions(co):
            out[name] = hashlib.sha1(code).hexdigest()
    return out


def find_kernels(hashes, demangled):
    """Mangled names in `hashes` that belong to a demangled rocprof kernel name such as
    'fused_fc_dual_kernel<2, 2, 2, 1, true, 2, 4, true>': the Itanium mangling of the template arguments (Li2E, Lb1E, ...) is
    rebuilt from the printed ones, so no demangler is needed."""
    m = re.match(r"\s*(?:void\s+)?([A-Za-z_0-9]+)\s*(?:<(.*)>)?", demangled)
    if not m:
        return []
    base, args = m.group(1), m.group(2)
    want = f"{len(base)}{base}"
    if args is not None:
        enc = ""
        for a in [x.strip() for x in args.split(",")]:
            if a in ("true", "false"):
                enc += "Lb%dE" % (a == "true")
            elif re.fullmatch(r"-?\d+", a):
                enc += "Li%sE" % (a.replace("-", "n"))
            else:
                return [k for k in hashes if want in k]      # an argument this helper does not encode: all instantiations
        want += "I" + enc + "E"
    return [k for k in hashes if k.startswith("_Z" + want)]
