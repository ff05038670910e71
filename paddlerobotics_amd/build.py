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


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    # -fno-slp-vectorize: the SLP vectoriser otherwise packs the scalar 3-vector algebra into
    # v_pk_*_f32 pairs, which costs ~900 v_mov + ~450 accvgpr moves per tick and 500+ registers
    # (measured: 373 -> 282 registers, 45% fewer instructions; DESIGN.md "register pressure")
    # -fno-signed-zeros -ffinite-math-only: lets the compiler fold the x*0 / x*1 terms that the
    # structured link frames (axes (1,0,0), (0,c,s)) put into the generic vector algebra (-6 %
    # instructions); NaN guards use an exponent bit test, IK validity an explicit domain test.
    cmd = [hipcc_path()] + FLAGS + ["-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
