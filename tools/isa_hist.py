#!/usr/bin/env python
"""Instruction-class histogram per basic block of one kernel in a hipcc -save-temps .s file:
`python tools/isa_hist.py file.s kernel_name_substring [min_mfma_per_block]`."""
import collections
import re
import sys


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    return "other"


def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    m = re.search(r"^(\S*%s\S*):[^\n]*\n" % re.escape(key), s, re.M)
    start = m.end()
    body = s[start:s.index("s_endpgm", start)]
    parts = re.split(r"\n(\.LBB\d+_\d+):", body)
    names = ["entry"] + parts[1::2]
    texts = [parts[0]] + parts[2::2]
    total = collections.Counter()
    for n, t in zip(names, texts):
        ops = [l.split()[0] for l in t.split("\n") if l.startswith("\t") and l.strip() and not l.startswith(("\t.", "\t;"))]
        c = collections.Counter(cls(o) for o in ops)
        total.update(c)
        if c["mfma"] >= floor:
            non = sum(v for k, v in c.items() if k != "mfma")
            print(f"{n:12s} mfma {c['mfma']:4d}  valu {c['valu']:4d} acc {c['acc']:4d} lane {c['lane']:3d} salu {c['salu']:4d} "
                  f"lds {c['lds']:3d} vmem {c['vmem']:3d} wait {c['wait']:3d}   non-MFMA per MFMA {non / c['mfma']:.2f}")
    print("whole kernel (static):", dict(total))


if __name__ == "__main__":
    main()
