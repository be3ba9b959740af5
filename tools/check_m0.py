#!/usr/bin/env python
"""Build check for the LDS-DMA copies whose pieces share one M0 setting (csrc/x6t_engine.h: xt_copy_piece_seq / xt_dma16_keep).

`global_load_lds_*` writes to LDS address M0 + instruction offset.  The two-tile exact-operand kernels set M0 in piece 0 (and piece 4) of a
chunk's copies and issue the other pieces as bare instructions, relying on NOTHING ELSE writing M0 in between -- which holds only as long
as the compiler never writes M0 itself in those kernels (movrel, LDS-direct, sendmsg ... would).  This script disassembles the gfx950 code
object of a library and fails if any kernel that contains a `global_load_lds` holds an M0 write that is not the inline-assembly pattern
    s_mov_b32 m0, <sgpr>  ;  s_nop 0  ;  global_load_lds_dword[x4] ...
i.e. one the compiler put there.   usage: python tools/check_m0.py [lib.so ...]   (default: both in-tree libraries)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(lib):
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(lib))
        os.symlink(lib, tmp)
        subprocess.run([OBJDUMP, "--offloading", tmp], cwd=d, check=True, capture_output=True)
        cos = [os.path.join(d, f) for f in os.listdir(d) if "amdgcn" in f]
        assert cos, "no gfx950 code object in " + lib
        return "\n".join(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", c], check=True, capture_output=True, text=True).stdout for c in cos)


def check(lib):
    text = disassemble(lib)
    bad, n_kernels, n_writes = [], 0, 0
    for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n((?:(?!^[0-9a-f]+ <).*\n?)*)", text, re.M):
        name, body = m.group(1), m.group(2)
        if "global_load_lds" not in body:
            continue
        n_kernels += 1
        ins = [l.split("//")[0].strip() for l in body.splitlines() if l.strip() and not l.strip().endswith(":")]
        for i, op in enumerate(ins):
            w = re.match(r"(\S+)\s+m0\b", op)                       # first operand m0 = destination
            if not w or w.group(1).startswith(("s_cmp", "v_cmp", "s_bitcmp")):
                continue
            n_writes += 1
            ok = (re.match(r"s_mov_b32\s+m0,\s*(s\d+|vcc_lo|vcc_hi|ttmp\d+)\b", op) and i + 2 < len(ins) and ins[i + 1].startswith("s_nop")
                  and ins[i + 2].startswith("global_load_lds_dword"))
            if not ok:
                bad.append((name, op, ins[i + 1:i + 3]))
    return n_kernels, n_writes, bad


def main():
    libs = sys.argv[1:] or [os.path.join(ROOT, "robir_amd", n) for n in ("librobir_hip.so", "librobir_hip_legacy.so")]
    rc = 0
    for lib in libs:
        if not os.path.exists(lib):
            print(lib, "missing: skipped")
            continue
        k, w, bad = check(lib)
        print(f"{os.path.basename(lib)}: {k} kernels with LDS-DMA copies, {w} M0 writes, {len(bad)} not of the inline-assembly pattern")
        for b in bad[:20]:
            print("   ", b)
        rc |= 1 if bad else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
