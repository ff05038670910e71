"""Registers / scratch of every kernel in a `hipcc -S --cuda-device-only` listing, and (with a kernel-name substring) the static
instruction mix of its hottest loop nest.  usage: kernel_resources.py file.s [substring]"""
import re
import subprocess
import sys


def kernels(src):
    out = []
    for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?"
                         r"\.vgpr_count:\s+(\d+)", src, re.S):
        out.append(dict(name=m.group(2), agpr=int(m.group(1)), scratch=int(m.group(3)), sgpr=int(m.group(4)), vgpr=int(m.group(5))))
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
        return r.stdout.strip().split("\n")
    except FileNotFoundError:
        return names


if __name__ == "__main__":
    src = open(sys.argv[1]).read()
    ks = kernels(src)
    dn = demangle([k["name"] for k in ks])
    for k, d in zip(ks, dn):
        if len(sys.argv) < 3 or sys.argv[2] in d:
            print("%-90s vgpr %3d agpr %3d scratch %5d" % (d[:90], k["vgpr"], k["agpr"], k["scratch"]))
