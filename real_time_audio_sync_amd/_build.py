"""In-tree build of librtsync.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(_HERE, "librtsync.so")

# -ffp-contract=off: the kernels follow the oracle's exact float64 operation order; every fused
# multiply-add in them is an explicit fma().
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def needs_build():
    if not os.path.exists(SO_PATH):
        return True
    m = os.path.getmtime(SO_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(_HERE, "..", "include", "rtsync.h")]
    return any(os.path.getmtime(d) > m for d in deps if os.path.exists(d))


def build(force=False, verbose=False, extra_flags=()):
    """Compile every HIP/C++ source under csrc/ into real_time_audio_sync_amd/librtsync.so."""
    if not force and not needs_build():
        return SO_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + list(extra_flags) + ["-o", SO_PATH]
    for s in sources():
        cmd += (["-x", "hip", s] if s.endswith(".hip") else ["-x", "hip", s])
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO_PATH


if __name__ == "__main__":
    import sys
    print(build(force="-f" in sys.argv, verbose=True))
