"""Loops of one kernel in a disassembled gfx950 code object: every backward branch with the instruction mix of its body.
usage: python tools/isa_loops.py <disassembly.s> <mangled-kernel-name> [--dump START END]"""
import collections
import re
import sys


def kernel_lines(path, sym):
    out, on = [], False
    for l in open(path):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", l)
        if m:
            on = m.group(1) == sym
            continue
        if on:
            m = re.match(r"\s+(\S.*?)\s*// ([0-9A-F]+):", l)
            if m:
                out.append((int(m.group(2), 16), m.group(1)))
    return out


def kind(t):
    op = t.split()[0]
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("v_"):
        return "dpp" if ("row_" in t or "quad_perm" in t) else "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    ins = kernel_lines(sys.argv[1], sys.argv[2])
    print(len(ins), "instructions")
    if len(sys.argv) > 3 and sys.argv[3] == "--dump":
        a0, a1 = int(sys.argv[4], 16), int(sys.argv[5], 16)
        for a, t in ins:
            if a0 <= a <= a1:
                print("%x: %s" % (a, t))
        return
    for a, t in ins:
        m = re.match(r"s_c?branch\w* (\d+)", t)
        if not m:
            continue
        off = int(m.group(1))
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + off * 4
        if tgt < a:
            body = [x for x in ins if tgt <= x[0] <= a]
            c = collections.Counter(kind(x[1]) for x in body)
            print("loop %x..%x  %5d instr  %s" % (tgt, a, len(body), dict(c)))


if __name__ == "__main__":
    main()
