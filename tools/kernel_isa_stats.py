"""Static ISA statistics of the kernels inside a built libetgsim.so (no GPU needed): registers, scratch and the instruction mix
of a kernel, from the gfx950 code object llvm-objdump extracts.  The CPU suite pins the headline kernel's numbers with it
(tests/test_kernel_isa.py) so that a toolchain bump or an innocent edit that moves the hot loop's code generation is caught.

usage: python tools/kernel_isa_stats.py [libetgsim.so] [kernel-name-substring ...]"""
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(so):
    """extract the gfx950 code objects of the fat binary into a temp dir -> (tmpdir, [paths])"""
    tmp = tempfile.mkdtemp(prefix="etgisa_")
    local = os.path.join(tmp, "lib.so")
    shutil.copy(so, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
    return tmp, sorted(glob.glob(local + ".*gfx950*"))


def disassemble(co):
    r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True)
    return r.stdout


def _vregs(tok):
    m = re.match(r"^-?\|?v(\d+)\|?$", tok)
    if m:
        return [int(m.group(1))]
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    return list(range(int(m.group(1)), int(m.group(2)) + 1)) if m else []


def dpp_hazards(text):
    """gfx9 data hazard: a VALU write of a VGPR needs 2 wait states before a DPP instruction reads that VGPR as its DPP operand.
    The compiler pads its own code; the hand-written asm blocks of the sweeps (etg_kernels.hip) pad by construction, and the
    compiler does not see into them -- so the finished code is checked: in straight-line code (the window is dropped at every
    branch) no DPP operand may have been written in the 2 preceding wait states (an instruction = 1, s_nop N = N + 1), and no
    VALU instruction may read a transcendental's result in the very next slot (gfx940+: 1 wait state).
    -> (number of DPP instructions checked, [(kernel, writer, reader)])"""
    bad, sym, hist, n = [], None, [], 0
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            sym, hist = m.group(1), []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//", line)
        if not m:
            continue
        op, rest = m.group(1), m.group(2)
        toks = [t.strip() for t in re.split(r",\s*", re.split(r" (?:quad_perm|row_|wave_)", rest)[0])] if rest else []
        if op.startswith(("s_branch", "s_cbranch", "s_setpc", "s_endpgm", "s_barrier")):
            hist = []
            continue
        if op.startswith("v_") and re.search(r" (?:quad_perm:|row_\w+)", rest) and len(toks) >= 2:
            n += 1
            src, need, i = set(_vregs(toks[1])), 2, len(hist) - 1
            while need > 0 and i >= 0:
                slots, wr, l = hist[i][:3]
                if src & set(wr):
                    bad.append((sym, l.split("//")[0].strip(), line.split("//")[0].strip()))
                need -= slots
                i -= 1
        # second rule (gfx940+): the result of a transcendental needs 1 wait state before another VALU instruction reads it
        if op.startswith("v_") and hist and hist[-1][0] == 1 and hist[-1][3]:
            rd = set(r for t in toks[1:] for r in _vregs(t))
            if op.startswith(("v_fmac", "v_mac", "v_fmaak")) or op.endswith("_dpp") and "fmac" in op:
                rd |= set(_vregs(toks[0]))
            if rd & set(hist[-1][1]):
                bad.append((sym, hist[-1][2].split("//")[0].strip(), line.split("//")[0].strip()))
        slots, wr = 1, []
        if op == "s_nop":
            slots = int(toks[0]) + 1 if toks and toks[0].isdigit() else 1
        elif op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and toks:
            wr = _vregs(toks[0])
        trans = op.startswith(("v_rsq_", "v_rcp_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_"))
        hist = (hist + [(slots, wr, line, trans)])[-6:]
    return n, bad


def notes(co):
    r = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True)
    out = {}
    for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?"
                         r"\.vgpr_count:\s+(\d+)", r.stdout, re.S):
        out[m.group(2)] = dict(agpr=int(m.group(1)), scratch=int(m.group(3)), sgpr=int(m.group(4)), vgpr=int(m.group(5)))
    return out


def kernel_bodies(asm):
    """symbol -> list of instruction lines"""
    out, cur = {}, None
    for line in asm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        t = line.strip()
        if cur is None or not t or t.startswith("Disassembly"):
            continue
        t = re.sub(r"^[0-9a-f]+:\s*", "", t)
        out[cur].append(t)
    return out


def mix(lines):
    c = collections.Counter()
    for t in lines:
        t = t.split("//")[0].strip()
        if not t or t.startswith("<"):
            continue
        op = t.split()[0]
        c["all"] += 1
        if op.startswith("v_"):
            c["valu"] += 1
            if "dpp" in t or "row_" in t or "quad_perm" in t:
                c["dpp"] += 1
            if op.startswith("v_mfma"):
                c["mfma"] += 1
            if op.startswith("v_pk_"):
                c["pk"] += 1
            if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)", op):
                c["trans"] += 1
            if "accvgpr" in op:
                c["acc_moves"] += 1
        elif op == "s_nop":
            c["s_nop"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        elif op.startswith("scratch_"):
            c["scratch_ops"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            c["branch"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return dict(c)


def stats(so, want=()):
    tmp, cos = code_objects(so)
    try:
        res = {}
        for co in cos:
            meta = notes(co)
            bodies = kernel_bodies(disassemble(co))
            for sym, lines in bodies.items():
                if sym not in meta:
                    continue
                if want and not any(w in sym for w in want):
                    continue
                res[sym] = dict(meta[sym], **mix(lines))
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


BASELINE_KERNELS = (
    "_ZN3etg11k_rollout16ILb1ELb1ELb1EEEvNS_4KCfgENS_8DevStateEiPfNS_7StatOutE",   # k_rollout16<flat, body rows, plain>: the headline (round 5: body spheres collide by default)
    "_ZN3etg11k_rollout16ILb1ELb0ELb1EEEvNS_4KCfgENS_8DevStateEiPfNS_7StatOutE",   # k_rollout16<flat, toe spheres only, plain>: the headline of rounds 1-4
    "_ZN3etg11k_rollout16ILb0ELb1ELb1EEEvNS_4KCfgENS_8DevStateEiPfNS_7StatOutE",   # ... on the heightfield (configs[4])
    "_ZN3etg8k_step16ILb1ELb1ELb1EEEvNS_4KCfgENS_8DevStateEPKfPKhPfS7_PhS7_",      # k_step16<flat, body rows, plain>: env.step()
    "_ZN3etg8k_step16ILb1ELb0ELb1EEEvNS_4KCfgENS_8DevStateEPKfPKhPfS7_PhS7_",      # k_step16<flat, toe spheres only, plain>
    "_ZN3etg9k_rolloutILb1ELb0ELi1EEEvNS_4KCfgENS_8DevStateEiPfNS_7StatOutE",      # k_rollout<flat, all options, body rows>: the 4-lane mapping (> 4096 robots)
    "_ZN3etg19k_rollout_policy16wILb1ELb1ELb1EEEvNS_4KCfgENS_8DevStateENS_7PolicyWEifPf",   # closed loop (configs[2]) since round 6: one wave = 4 robots + their policy tile, fp32, body rows; no scratch
    "_ZN3etg18k_rollout_policy16ILb1ELb0ELb1ELb1EEEvNS_4KCfgENS_8DevStateENS_7PolicyWEifPf",   # the workgroup-tile closed loop of rounds 3-5 (kept for bf16 operands), body rows
)


def write_baseline(so, path):
    import json
    st = stats(so, BASELINE_KERNELS)
    missing = set(BASELINE_KERNELS) - set(st)
    if missing:
        raise SystemExit("kernels not found in %s: %s" % (so, sorted(missing)))
    hipcc = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.strip().split("\n")
    json.dump({"what": "static ISA statistics of the hot kernels in the library build the round-6 numbers were measured with "
                       "(tools/kernel_isa_stats.py; compared by tests/test_kernel_isa.py)",
               "toolchain": hipcc[:2], "kernels": st}, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if "--write-baseline" in sys.argv:
        write_baseline(os.path.join(root, "paddlerobotics_amd", "csrc", "libetgsim.so"), os.path.join(root, "profiles", "r06_isa_baseline.json"))
        sys.exit(0)
    so = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(root, "paddlerobotics_amd", "csrc", "libetgsim.so")
    want = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for sym, st in sorted(stats(so, want).items()):
        print(sym)
        print("   ", " ".join("%s=%s" % kv for kv in sorted(st.items())))
