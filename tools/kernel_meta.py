"""Registers, spills and scratch of every gfx950 kernel in a built libkanvit.so, read from the code objects' metadata notes (no GPU, no ROCm tool).
    python tools/kernel_meta.py [regex] [--md]        (default library: kan-vit_amd/kanvit/libkanvit.so, or KANVIT_LIB)
Used by tests/test_abi_cpu.py: the kernels whose waits on LDS-DMA fills are explicit s_waitcnt counts must not touch scratch (a scratch access
counts in vmcnt), and DESIGN.md's "zero spills" statements are checked against the build instead of being copied from a compile log."""
import os
import re
import struct
import sys

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels(path):
    data = open(path, "rb").read()
    out = {}
    for m in re.finditer(MAGIC, data):
        p = m.start()
        n = struct.unpack_from("<Q", data, p + 24)[0]
        q = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            q += 24
            triple = data[q:q + tl].decode()
            q += tl
            if "gfx950" not in triple or size == 0:
                continue
            elf = data[p + off:p + off + size]
            if elf[:4] != b"\x7fELF":
                continue
            shoff = struct.unpack_from("<Q", elf, 0x28)[0]
            shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
            for i in range(shnum):
                sh = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
                if sh[1] != 7:          # SHT_NOTE
                    continue
                o, end = sh[4], sh[4] + sh[5]
                while o < end:
                    namesz, descsz, typ = struct.unpack_from("<III", elf, o)
                    o += 12
                    name = elf[o:o + namesz]
                    o += (namesz + 3) & ~3
                    desc = elf[o:o + descsz]
                    o += (descsz + 3) & ~3
                    if name.startswith(b"AMDGPU") and typ == 32:
                        for k in msgpack.unpackb(desc, raw=False).get("amdhsa.kernels", []):
                            out[k[".name"]] = k
    return out


def demangled_short(name):
    m = re.search(r"\d+([a-z]\w+?_kernel)(I.*E)?Ev", name)
    if not m:
        return name
    args = re.findall(r"L[ib](\d+)E", m.group(2) or "")
    return m.group(1) + ("<" + ", ".join(args) + ">" if args else "")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pat = re.compile(args[0]) if args else re.compile(".")
    lib = os.environ.get("KANVIT_LIB") or os.path.join(ROOT, "kan-vit_amd", "kanvit", "libkanvit.so")
    ks = kernels(lib)
    md = "--md" in sys.argv
    if md:
        print("| kernel | VGPRs (arch + acc) | AGPRs | SGPRs | vgpr spills | sgpr spills | scratch bytes | LDS (static) |\n|---|---|---|---|---|---|---|---|")
    for name in sorted(ks, key=demangled_short):
        k = ks[name]
        short = demangled_short(name)
        if not pat.search(short):
            continue
        row = (short, k[".vgpr_count"], k.get(".agpr_count", 0), k[".sgpr_count"], k[".vgpr_spill_count"], k[".sgpr_spill_count"],
               k[".private_segment_fixed_size"], k[".group_segment_fixed_size"])
        print(("| `%s` | %d | %d | %d | %d | %d | %d | %d |" if md else "%-60s vgpr %3d agpr %3d sgpr %3d  spills v %3d s %3d  scratch %5d  lds %6d") % row)
