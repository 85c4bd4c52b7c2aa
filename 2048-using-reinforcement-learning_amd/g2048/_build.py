"""Builds csrc/libg2048_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m g2048._build            (from the package directory)

The library is built in-tree so that it travels with the source snapshot; it is
git-ignored. -ffp-contract=off is REQUIRED: the reward / heuristic kernels follow
the reference's f64 operation order (mul then add, never fma).
"""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
LIB = os.environ.get("G2048_LIB") or os.path.join(CSRC, "libg2048_hip.so")     # G2048_LIB: A/B builds only (tools/)
SOURCES = ["g2048_kernels.hip", "g2048_beam.hip", "g2048_rollout.hip"]
INCLUDE = os.path.join(CSRC, "..", "..", "include")
PUBLIC_HEADERS = [os.path.join(INCLUDE, "g2048.h"), os.path.join(INCLUDE, "g2048_testing.h")]
# -fvisibility=hidden: the export table is exactly what the two headers declare with G2048_API (tests/test_abi_and_host.py)
# -amdgpu-kernarg-preload-count: the first kernel-argument dwords arrive in SGPRs with the wavefront instead of through s_load +
# s_waitcnt at its head (gfx950 takes up to 14 next to the kernarg pointer; the short kernels order their arguments for it:
# 12.9 -> 12.55 us per 1 Mi-board step launch, profiles/r05_kernel_heads.txt)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden",
         "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "-Wl,--version-script=" + os.path.join(CSRC, "g2048_exports.map"), "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    # every file a source can include is a dependency: whatever lies in csrc/ (sources, headers, .inc) + the public header
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc", ".hpp", ".map"))] + PUBLIC_HEADERS
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    cmd = [hipcc] + FLAGS + ["-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
