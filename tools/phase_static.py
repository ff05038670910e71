"""Static instruction mix between the s_memtime phase markers of a -DETG_PROFILE_PHASES build.
usage: phase_static.py <file.s> <kernel-symbol-substring>"""
import re, sys, collections
src, want = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if want in l.split(":")[0] and ":" in l and l.startswith("_Z") and not l.startswith(".L") and not l.startswith(";"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
seg, segs = collections.Counter(), []
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
    op = t.split()[0]
    if op == "s_memtime":
        segs.append(seg); seg = collections.Counter(); continue
    seg["all"] += 1
    if op.startswith("v_"):
        seg["valu"] += 1
        if "dpp" in t: seg["dpp"] += 1
        if op.startswith("v_mfma"): seg["mfma"] += 1
        if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)", op): seg["trans"] += 1
        if "accvgpr" in op: seg["acc"] += 1
    elif op == "s_nop":
        seg["nop"] += 1; seg["nopcyc"] += int(t.split()[1]) + 1
    elif op.startswith("s_waitcnt"): seg["wait"] += 1
    elif op.startswith("ds_"): seg["lds"] += 1
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"):
        seg["vmem"] += 1
        if op.startswith("scratch_"): seg["scratch"] += 1
    elif op.startswith("s_"): seg["salu"] += 1
segs.append(seg)
keys = ["all", "valu", "dpp", "mfma", "trans", "acc", "nop", "nopcyc", "wait", "lds", "vmem", "scratch", "salu"]
print("seg " + " ".join("%7s" % k for k in keys))
for i, s in enumerate(segs):
    print("%3d " % i + " ".join("%7d" % s[k] for k in keys))
