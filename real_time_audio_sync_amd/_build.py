"""In-tree build of librtsync.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(_HERE, "librtsync.so")

# -ffp-contract=off: the kernels follow the oracle's exact float64 operation order; every fused
# multiply-add in them is an explicit fma().
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# Per-source additions.  The pacing waves of the OTW and DTW kernels work through dependent chains, and one wave issues
# a dependent VALU instruction only every 8.25 cycles against 4-5 for an independent one (tools/microbench/
# chain_latency.hip), so the order the compiler gives the instructions matters: LLVM's ILP-first scheduling strategy is
# worth 1.6 % on the OTW headline and 4-8 % on DTW pairs (same-call A/B, profiles/experiments/README.md), and costs the
# WTW kernels 4-22 % -- hence per file.  It changes the order of instructions, not their results.
PER_SOURCE_FLAGS = {
    "otw.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "dtw.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
}
OBJ_DIR = os.path.join(_HERE, "_obj")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def needs_build():
    if not os.path.exists(SO_PATH):
        return True
    m = os.path.getmtime(SO_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(_HERE, "..", "include", "rtsync.h")]
    return any(os.path.getmtime(d) > m for d in deps if os.path.exists(d))


def _compile(hipcc, src, extra_flags, verbose):
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
    cmd = [hipcc] + HIPCC_FLAGS + PER_SOURCE_FLAGS.get(os.path.basename(src), []) + list(extra_flags) + ["-c", "-x", "hip", src, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False, extra_flags=()):
    """Compile every HIP/C++ source under csrc/ (one hipcc per source, side by side) and link real_time_audio_sync_amd/
    librtsync.so.  Raises subprocess.CalledProcessError when a compilation fails: there is nothing to fall back to."""
    if not force and not needs_build():
        return SO_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(hipcc, s, extra_flags, verbose), srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO_PATH


if __name__ == "__main__":
    import sys
    print(build(force="-f" in sys.argv, verbose=True))
