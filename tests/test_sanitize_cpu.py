"""SURVEY 5, "race detection / sanitizers": the reference has none; the build's equivalent is AddressSanitizer +
UndefinedBehaviorSanitizer on everything that runs on the host (GPU sanitizers are not available on this pool).

  * the CPU oracle (oracle/rtsync_oracle.c) compiled with gcc -fsanitize=address,undefined together with
    tests/sanitize/oracle_driver.c: OTW / LiveNote / LiveNoteV2 (insert loop, set_live, frame-by-frame insert, overflow,
    stop), DTW and WTW cases; the driver's checksums must equal the regular liboracle.so's on the same inputs;
  * the HOST half of librtsync.so -- every argument-validation and early-error path of include/rtsync.h -- as a host-only
    sanitizer build of csrc/ (hipcc --cuda-host-only: seconds, the kernels are not compiled; the undefined fat-binary
    symbols are satisfied by empty stubs) driven by tests/sanitize/shim_driver.c, which asserts every return code.
Leak detection is on: the error paths must free what they allocated."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SAN = os.path.join(ROOT, "tests", "sanitize")
OUT = os.path.join(SAN, "_build")
FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def _run(cmd, **kw):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw)
    return p.returncode, p.stdout, p.stderr


def _clean_report(err):
    for bad in ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:"):
        assert bad not in err, err[-4000:]


def _fnv(h, a):
    for byte in np.ascontiguousarray(a).tobytes():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_oracle_under_asan_ubsan(tmp_path):
    import oracle
    from real_time_audio_sync_amd import synth
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "oracle_driver")
    rc, out, err = _run(["gcc"] + FLAGS + ["-ffp-contract=off", os.path.join(ROOT, "oracle", "rtsync_oracle.c"),
                                           os.path.join(SAN, "oracle_driver.c"), "-o", exe, "-lm"])
    assert rc == 0, err[-3000:]
    N, F = 90, 12
    ref = synth.synth_ref(N, seed=5)
    live = synth.synth_live(ref, seed=6, lo=0.5, hi=0.8)             # longer than the reference: overflow / stop paths
    live = np.concatenate([live, live[:, -1:].repeat(2 * N, axis=1) + 0.01 * np.random.RandomState(7).rand(12, 2 * N)], axis=1)
    T = live.shape[1]
    # kind, variant, c | W, max_run_count | hop, cost, frames used
    cases = [(0, 0, 20, 3, 0, T), (0, 1, 7, 2, 0, T), (0, 2, 33, 3, 1, T), (1, 0, 20, 3, 0, 100), (1, 2, 15, 4, 0, T),
             (2, 1, 10, 3, 0, 60), (2, 0, 200, 3, 0, T), (3, 0, 0, 0, 0, 70), (4, 0, 20, 10, 0, T), (4, 0, 64, 7, 0, 150),
             (0, 0, 1, 1, 0, 50)]
    f = tmp_path / "cases.bin"
    with open(f, "wb") as g:
        g.write(np.array([N, T, F, len(cases)], dtype=np.int32).tobytes())
        g.write(np.ascontiguousarray(ref.T).tobytes())
        g.write(np.ascontiguousarray(live.T).tobytes())
        for c in cases:
            g.write(np.array(c, dtype=np.int32).tobytes())
    rc, out, err = _run([exe, str(f)], env=ENV, timeout=600)
    _clean_report(err)
    assert rc == 0, (out, err[-2000:])
    got = [l.split() for l in out.splitlines() if l.startswith("case")]
    assert len(got) == len(cases)
    for k, (kind, variant, c, mrc, cost, tu) in enumerate(cases):
        h = 1469598103934665603
        lv = live[:, :tu]
        if kind <= 2:
            o = oracle.OtwOracle(ref, c, mrc, variant=variant, cost=cost)
            if kind == 0:
                o.run(lv)
            elif kind == 1:
                o.set_live(lv)
            else:
                for i in range(lv.shape[1]):
                    if o.insert(lv[:, i]) != 0:
                        break
            s = o.state
            st = np.array([s["t"], s["j"], s["direction"], s["previous"], s["run_count"], s["status"], s["first_insert"]], dtype=np.int32)
            rb, cb = o.bands()
            path = o.path
            for a in (path, st, rb, cb):
                h = _fnv(h, a)
        elif kind == 3:
            _, acc, path, back = oracle.dtw(lv, ref)
            for a in (path, acc, back):
                h = _fnv(h, a)
        else:
            w = oracle.WtwOracle(ref, c, mrc)
            w.insert_precheck()
            for i in range(lv.shape[1]):
                if w.push_col(lv[:, i]) != 0:
                    break
            s = w.state
            path = w.path
            h = _fnv(_fnv(h, path), np.array([s["chroma_ptr"], s["live_ptr"], s["ref_ptr"], s["status"]], dtype=np.int32))
        assert int(got[k][5]) == len(path) and got[k][7] == "%016x" % h, (k, cases[k], got[k])


def test_host_shim_validation_paths_under_asan_ubsan():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not (os.path.exists(hipcc) and os.path.exists(clang)):
        pytest.skip("no ROCm toolchain")
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, "real_time_audio_sync_amd", "csrc")
    srcs = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp")))
    lib = os.path.join(OUT, "librtsync_hostasan.so")
    cmd = [hipcc, "--cuda-host-only", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-function"] + FLAGS
    for s in srcs:
        cmd += ["-x", "hip", s]
    rc, out, err = _run(cmd + ["-o", lib], timeout=900)
    assert rc == 0, err[-3000:]
    # host-only objects refer to their (absent) device code objects: satisfy those symbols with empty stubs
    rc, out, err = _run(["nm", "-D", "--undefined-only", lib])
    syms = [l.split()[-1] for l in out.splitlines() if "__hip_fatbin_" in l]
    stub = os.path.join(OUT, "fatbin_stub.c")
    with open(stub, "w") as g:
        g.write("".join("__attribute__((aligned(4096))) const char %s[4096] = {0};\n" % s for s in syms))
    exe = os.path.join(OUT, "shim_driver")
    rc, out, err = _run([clang] + FLAGS + [os.path.join(SAN, "shim_driver.c"), stub, "-rdynamic", "-o", exe, "-L" + OUT,
                                           "-lrtsync_hostasan", "-Wl,-rpath," + OUT])
    assert rc == 0, err[-3000:]
    # hide any GPU: this test is about the host paths (and must behave the same on the GPU box)
    env = dict(ENV, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    rc, out, err = _run([exe], env=env, timeout=300)
    _clean_report(err)
    assert rc == 0 and "ok (0 mismatches)" in out, (out, err[-3000:])
