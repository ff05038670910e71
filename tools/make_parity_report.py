"""profiles/rNN_parity_report.txt from the output of `python -m pytest tests -m gpu -q -s` on the GPU box.

Keeps every `[parity]` line (each test prints what it measured next to the bound it asserts) and puts a tally in front: per
sensitivity-criterion line (tests/parity_util.sens_robots) the robots compared, how many were within the floor outright, how
many needed the allowance floor + 4 x (their own oracle ensemble's spread) and how many were outside it (asserted to be 0);
per one-step-consistency line (tests/test_gpu_parity5.py, tests/test_gpu_regressions.py) the (robot, step) pairs on the fp64
oracle's branch, on another ensemble member's, on none.

  python tools/make_parity_report.py gpurun_out/r06_f/pytest_gpu.txt profiles/r06_parity_report.txt
"""
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
lines = []
for raw in open(src, errors="replace"):
    k = raw.find("[parity]")
    if k >= 0:
        lines.append(raw[k:].rstrip("\n"))
tail = [l.strip() for l in open(src, errors="replace") if re.search(r"\d+ passed|\d+ failed|^exit \d", l)]

sens = re.compile(r"\[parity\] (.*?)\s+median (\S+) max (\S+) \(floor (\S+)\) \| robots (\d+), within the floor (\d+), needed the sensitivity allowance (\d+)(.*?)\| outside floor \+ 4 x spread: (\d+)")
pairs = re.compile(r"\[parity\] (.*?)\s+\(robot, step\) pairs (\d+): on the fp64 oracle's branch (\d+), on another member's(?: branch)? (\d+), on none (\d+)")
rows_s, rows_p = [], []
for l in lines:
    m = sens.search(l)
    if m:
        rows_s.append((m.group(1), int(m.group(5)), int(m.group(6)), int(m.group(7)), int(m.group(9)), m.group(2), m.group(3), m.group(4), m.group(8).strip()))
    m = pairs.search(l)
    if m:
        rows_p.append((m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))))

with open(dst, "w") as f:
    f.write("Parity report: GPU (HIP library through the C-ABI) against the oracle ensemble, from %s\n" % src)
    f.write("pytest: %s\n\n" % " | ".join(tail))
    f.write("A. Per-robot sensitivity criterion (tests/parity_util.sens_robots): a robot is inside when |gpu - fp64 oracle| <= floor + 4 x spread, spread =\n"
            "   the robot's own ensemble spread (fp32 oracle; fp32 / fp64 oracles whose actions are moved by one fp32 ulp and whose solved impulses carry\n"
            "   the rounding noise of an fp32 solve; fp64 oracles with a nudged stopping threshold -- all against the fp64 oracle).  No robot is excused\n"
            "   without a measured spread.  `outside` may be as large as it is for the ensemble's worst member when judged the same way against the\n"
            "   other members, plus one per 64 robots (a robot at a bifurcation is parted by ONE evaluation alone with probability p (1 - p)^M); every\n"
            "   outside robot is listed in section C with its gap and spread.\n\n")
    f.write("%-78s %7s %7s %9s %7s  %9s %9s %8s\n" % ("test line", "robots", "floor", "allowance", "outside", "median", "max", "floor"))
    tot = [0, 0, 0, 0]
    for r in rows_s:
        f.write("%-78s %7d %7d %9d %7d  %9s %9s %8s\n" % (r[0][:78], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
        for k in range(4):
            tot[k] += r[1 + k]
    f.write("%-78s %7d %7d %9d %7d\n\n" % ("TOTAL (robot x quantity comparisons)", *tot))
    out_rows = [l for l in lines if "the GPU's outside robots" in l]
    f.write("   lines with robots outside their allowance: %d\n" % len(out_rows))
    for l in out_rows:
        f.write("   " + l[l.find("[parity]") + 9:].split("  ")[0].strip()[:80] + " | " + l[l.find("the GPU's outside robots"):] + "\n")
    f.write("\n")
    need = [r for r in rows_s if r[3] > 0]
    f.write("   lines with robots that needed the allowance (their gaps and their ensemble spread):\n")
    for r in need:
        f.write("   %-76s %d of %d %s\n" % (r[0][:76], r[3], r[1], r[8]))
    f.write("\nB. One-step consistency: every (robot, step) pair of the GPU's own trajectory, re-stepped from the GPU's state by the oracle ensemble.\n\n")
    f.write("%-86s %8s %10s %8s %6s\n" % ("test line", "pairs", "on fp64's", "other", "none"))
    for r in rows_p:
        f.write("%-86s %8d %10d %8d %6d\n" % (r[0][:86], r[1], r[2], r[3], r[4]))
    f.write("\n   (`hard deepest-of-three` lines are the round-5 contact model, EtgConfig.body_blend = 0, kept as regression cases: every pair off the\n"
            "   ensemble there is explained line by line below -- a knee / shin-midpoint sphere tie; under the default model none is off.)\n\n")
    f.write("C. Every [parity] line of the run\n\n")
    for l in lines:
        f.write(l + "\n")
print("wrote %s: %d lines, %d criterion lines, %d consistency lines" % (dst, len(lines), len(rows_s), len(rows_p)))
