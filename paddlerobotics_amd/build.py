"""Build the gfx950 shared library in-tree (paddlerobotics_amd/csrc/libetgsim.so).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["etg_kernels.hip", "policy_mlp.hip", "etg_fit.hip", "etg_replay.hip"]
HEADERS = ["etg_core.h", "etg_core16.h", "etg_layout.h", "policy_core.h", os.path.join("..", "..", "include", "etgsim.h")]
LIB = os.path.join(CSRC, "libetgsim.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built")


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-fno-slp-vectorize", "-fno-signed-zeros", "-ffinite-math-only", "-Wno-unused-value"]


def kernel_source_hash():
    """sha256 (16 hex digits) over the kernel sources, headers and compiler flags the library is built from: profile
    artefacts that describe a kernel (profiles/r0N_pmc.json) are stamped with it and bench.py marks them stale when the
    sources have moved on since."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(SOURCES + HEADERS):
        path = os.path.join(CSRC, f)
        if os.path.exists(path):
            h.update(f.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


TU_PARTS = 8          # etg_kernels.hip: part 0 = host side + small kernels, parts 1..7 = the tick kernels' instantiations
OBJ_DIR = os.path.join(CSRC, "build")


def compile_commands(extra_flags=(), obj_dir=None, parts=TU_PARTS):
    """[(object file, command)] of the library: etg_kernels.hip once per translation-unit part (the explicit-instantiation
    table at the end of its namespace says which part owns which kernel), the other sources once each."""
    obj_dir = obj_dir or OBJ_DIR
    cc = [hipcc_path()] + [f for f in FLAGS if f != "-shared"] + list(extra_flags)
    cmds = []
    for part in range(parts):
        o = os.path.join(obj_dir, "etg_kernels.%d.o" % part)
        cmds.append((o, cc + ["-DETG_TU_PARTS=%d" % parts, "-DETG_TU_PART=%d" % part, "-c", "-o", o, os.path.join(CSRC, "etg_kernels.hip")]))
    for f in SOURCES[1:]:
        if os.path.exists(os.path.join(CSRC, f)):
            o = os.path.join(obj_dir, f.replace(".hip", ".o"))
            cmds.append((o, cc + ["-c", "-o", o, os.path.join(CSRC, f)]))
    return cmds


def build(force=False, verbose=False, extra_flags=(), lib=None, jobs=None):
    """Compile the translation units in parallel (one hipcc per part, at most `jobs` at a time: default = the CPUs this
    process may use) and link them.  extra_flags / lib: an A/B variant of the kernels (tools/build_variant.sh)."""
    lib = lib or LIB
    if not force and lib == LIB and not is_stale():
        return lib
    # -fno-slp-vectorize: the SLP vectoriser otherwise packs the scalar 3-vector algebra into
    # v_pk_*_f32 pairs, which costs ~900 v_mov + ~450 accvgpr moves per tick and 500+ registers
    # (measured: 373 -> 282 registers, 45% fewer instructions; DESIGN.md "register pressure")
    # -fno-signed-zeros -ffinite-math-only: lets the compiler fold the x*0 / x*1 terms that the
    # structured link frames (axes (1,0,0), (0,c,s)) put into the generic vector algebra (-6 %
    # instructions); NaN guards use an exponent bit test, IK validity an explicit domain test.
    obj_dir = OBJ_DIR if lib == LIB else lib + ".obj"
    os.makedirs(obj_dir, exist_ok=True)
    cmds = compile_commands(extra_flags, obj_dir)
    jobs = jobs or (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    pending, running, failed = list(cmds), [], []
    while pending or running:
        while pending and len(running) < jobs:
            o, cmd = pending.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((o, subprocess.Popen(cmd)))
        o, proc = running.pop(0)
        if proc.wait() != 0:
            failed.append(o)
    if failed:
        raise RuntimeError("hipcc failed for %s" % ", ".join(failed))
    link = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [o for o, _ in cmds]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return lib


if __name__ == "__main__":
    import sys
    # python -m paddlerobotics_amd.build [--lib PATH] [-DFLAG ...]: extra flags build an A/B variant next to the product
    a = sys.argv[1:]
    out = a[a.index("--lib") + 1] if "--lib" in a else None
    flags = [x for i, x in enumerate(a) if x != "--lib" and (i == 0 or a[i - 1] != "--lib")]
    print(build(force=True, verbose=True, extra_flags=flags, lib=out))
