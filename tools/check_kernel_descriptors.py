#!/usr/bin/env python3
"""Walk every gfx950 code object embedded in libparo_mi355x.so and report kernels whose DESCRIPTOR enables the dispatch-packet pointer
(or the queue pointer): such a kernel reads the AQL packet from the queue's memory at run time -- one scalar load measured at 4 .. 7 us
per launch (profiles/NOTES.md 4.5) -- typically because a private array was moved to LDS and is addressed by flat work-item id.
Also prints the largest scratch (private segment) sizes: a non-zero value means spills.
    python tools/check_kernel_descriptors.py [path/to/lib.so]          exit code 1 if any kernel uses the dispatch / queue pointer"""
import os, struct, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """Yield (triple, bytes) for every entry of every offload bundle in the file."""
    at = 0
    while True:
        at = blob.find(MAGIC, at)
        if at < 0:
            return
        n = struct.unpack_from("<Q", blob, at + 24)[0]
        p = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if size:
                yield triple, blob[at + off:at + off + size]
        at += 24


def kernels(elf):
    """(name, kernel_code_properties, private_segment_fixed_size) of every <name>.kd symbol of an AMDGPU ELF."""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    out = []
    for (name, typ, flags, addr, off, size, link, info, align, entsize) in secs:
        if typ not in (2, 11):        # SHT_SYMTAB, SHT_DYNSYM
            continue
        stroff = secs[link][4]
        for k in range(size // 24):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, off + k * 24)
            end = elf.index(b"\0", stroff + st_name)
            nm = elf[stroff + st_name:end].decode()
            if not nm.endswith(".kd") or st_shndx == 0 or st_shndx >= shnum:
                continue
            s = secs[st_shndx]
            kd = elf[s[4] + (st_value - s[3]):s[4] + (st_value - s[3]) + 64]
            priv, = struct.unpack_from("<I", kd, 4)
            props, = struct.unpack_from("<H", kd, 56)
            out.append((nm[:-3], props, priv))
        if out:
            break
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "paroquant_amd", "_lib", "libparo_mi355x.so")
    blob = open(lib, "rb").read()
    seen, bad, spills = 0, [], []
    for triple, co in code_objects(blob):
        if "gfx950" not in triple or co[:4] != b"\x7fELF":
            continue
        for name, props, priv in kernels(co):
            seen += 1
            if props & 0b110:          # bit 1: ENABLE_SGPR_DISPATCH_PTR, bit 2: ENABLE_SGPR_QUEUE_PTR
                bad.append(name)
            if priv:
                spills.append((priv, name))
    print(f"{seen} kernels in {os.path.basename(lib)}; dispatch / queue pointer enabled in {len(bad)}; scratch in {len(spills)}")
    for n in bad:
        print("  dispatch-ptr:", n)
    for priv, n in sorted(spills, reverse=True)[:10]:
        print(f"  scratch {priv:6d} B:", n[:140])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
