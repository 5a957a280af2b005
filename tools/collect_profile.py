#!/usr/bin/env python3
"""Run on the GPU box: the round's measurement set for the headline workload.

    python3 tools/collect_profile.py TAG        -> gpurun_out/profile_TAG/

  1. bench.py (default flags)                                   -> TAG_bench_final.json
  2. rocprofv3 --kernel-trace --stats -- python3 bench.py ...   -> TAG_otw_kernel_stats.csv, TAG_bench_under_rocprof.json
  3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes, --no-cpu, 3 steps)
                                                                -> TAG_pmc.json, otw_traffic.json (with the sha of csrc/otw.hip)
  4. per secondary entry of the bench line: rocprofv3 --kernel-trace --stats -- python3 bench.py --secondary-only KEY ...
                                                                -> TAG_secondary_kernel_stats.json
Every profiler pass is its own child process with the program right after `--` (no shell, no env wrapper).
Everything is written under gpurun_out/profile_TAG/ only; copy what should be judged into profiles/ by hand
(profiles/otw_traffic.json included: bench.py reports it as roofline.traffic only while its otw_hip_sha16 matches).
"""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "otw_advance_kernel"


def run(cmd, out_path=None, timeout=600):
    print("+", " ".join(cmd), flush=True)
    env = dict(os.environ, TMPDIR="/tmp")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stdout[-2000:] + "\n" + p.stderr[-4000:] + "\n")
        raise SystemExit("command failed: %s" % " ".join(cmd))
    if out_path:
        line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
        with open(out_path, "w") as f:
            f.write(line + "\n")
        return json.loads(line)
    return None


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits[0]


def pmc_pass(out, name, counters):
    d = os.path.join(out, "pmc_" + name)
    run(["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--",
         "python3", "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu"])
    sums, calls = {}, {}
    with open(find(d, "counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if KERNEL not in row["Kernel_Name"]:
                continue
            c = row["Counter_Name"]
            sums[c] = sums.get(c, 0.0) + float(row["Counter_Value"])
            calls.setdefault(c, set()).add(row["Dispatch_Id"])
    return {c: sums[c] / len(calls[c]) for c in sums}, {c: len(calls[c]) for c in sums}


def main():
    tag = sys.argv[1]
    out = os.path.join(ROOT, "gpurun_out", "profile_" + tag)
    os.makedirs(out, exist_ok=True)
    d = os.path.join(out, "trace")
    # --no-cpu: the CPU legs fork worker processes, which must not happen under the profiler's preloaded runtime
    under = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--",
                 "python3", "bench.py", "--steps", "20", "--warmup", "3", "--no-cpu"],
                os.path.join(out, tag + "_bench_under_rocprof.json"))
    stats = find(d, "kernel_stats.csv")
    with open(stats) as f, open(os.path.join(out, tag + "_otw_kernel_stats.csv"), "w") as g:
        for i, line in enumerate(f):
            if i < 8:
                g.write(line[:600] + ("\n" if len(line) > 600 else ""))
    with open(stats) as f:
        krow = [r for r in csv.DictReader(f) if KERNEL in r["Name"]][0]
    print("rocprofv3: %s calls=%s avg=%.3f ms (bench under rocprof, HIP events: %.3f ms)"
          % (krow["Name"][:60], krow["Calls"], float(krow["AverageNs"]) / 1e6, under["roofline"]["launch_ms"]), flush=True)

    pmc, n = {}, {}
    for name, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                           ("sq1", ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"]),
                           ("sq2", ["SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT"])):
        v, c = pmc_pass(out, name, counters)
        pmc.update(v)
        n.update(c)
    traffic = {
        "otw_hip_sha16": hashlib.sha256(open(os.path.join(ROOT, "real_time_audio_sync_amd", "csrc", "otw.hip"), "rb").read()).hexdigest()[:16],
        "hbm_bytes_per_launch": int((2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024),
        "FETCH_SIZE_KB": pmc["FETCH_SIZE"], "WRITE_SIZE_KB": pmc["WRITE_SIZE"],
        "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024  -- FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per "
                   "128-B request); this kernel's reads are 4-8 B per lane, an uncalibrated width, so the true figure lies "
                   "between the undoubled and the doubled sum",
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 1 --no-cpu, "
                  "kernel %s, %s (tools/collect_profile.py)" % (krow["Name"], tag)}
    # the un-profiled bench line last, reading the traffic figure this very pass measured (bench.py takes
    # roofline.traffic from profiles/otw_traffic.json), so that the two files agree
    with open(os.path.join(out, "otw_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    final = run(["python3", "bench.py", "--traffic-json", os.path.join(out, "otw_traffic.json")],
                os.path.join(out, tag + "_bench_final.json"))
    print("bench: %.3f ms/step, %.3e frames/s" % (final["ms_per_step"], final["value"]), flush=True)
    summary = {"kernel": krow["Name"], "kernel_trace": {k: krow[k] for k in ("Calls", "AverageNs", "MinNs", "MaxNs", "StdDev")},
               "pmc_mean_per_dispatch": pmc, "pmc_dispatches": n,
               "bench_final": {k: final[k] for k in ("value", "ms_per_step")},
               "bench_under_rocprof_launch_ms": under["roofline"]["launch_ms"]}
    with open(os.path.join(out, tag + "_pmc.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary["pmc_mean_per_dispatch"], indent=1))
    bench_secondary(tag, out, final)
    secondary(tag, out)


def bench_secondary(tag, out, final):
    """Kernel statistics of every secondary entry of the bench line, one rocprofv3 run per entry, next to the entry's own
    HIP-event figure from that run and from the un-profiled bench line."""
    rows = []
    plain = {e["key"]: e for e in final.get("secondary", [])}
    for key in ("dtw322", "dtw1289", "otw_b1", "otw_b64_f64", "chroma", "wtw20", "wtw100", "wtw10k"):
        d = os.path.join(out, "trace_sec_" + key)
        line = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py",
                    "--no-fork", "--no-numpy", "--steps", "2", "--warmup", "1", "--secondary-only", key],
                   os.path.join(out, "sec_%s.json" % key))
        entry = [e for e in line["secondary"] if e["key"] == key][0]
        want = [k.split(" ")[0] for k in entry["kernels"]]
        ks = []
        with open(find(d, "kernel_stats.csv")) as f:
            for r in csv.DictReader(f):
                if any(w in r["Name"] for w in want) and not (key != "otw_b64_f64" and key != "otw_b1" and "otw_" in r["Name"]):
                    ks.append({"name": r["Name"][:120], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                               "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3})
        rows.append({"key": key, "workload": entry["workload"], "ms_hip_events_under_rocprof": entry["ms"],
                     "ms_hip_events_plain_bench": plain.get(key, {}).get("ms"), "kernels": ks})
        print("secondary %-12s %.4f ms (plain %.4f)  %s" % (key, entry["ms"], plain.get(key, {}).get("ms") or float("nan"),
                                                           ", ".join("%s x%d %.1f us" % (k["name"].split("(")[0][-28:], k["calls"], k["avg_us"]) for k in ks)), flush=True)
    with open(os.path.join(out, tag + "_secondary_kernel_stats.json"), "w") as f:
        json.dump(rows, f, indent=1)


def kernel_rows(stats_csv, names):
    with open(stats_csv) as f:
        return [r for r in csv.DictReader(f) if any(n in r["Name"] for n in names)]


def secondary(tag, out):
    """Strip-DP (DTW / WTW) and chroma kernels: rocprofv3 kernel statistics of tools/bench_sdp.py and of a chroma run,
    and the HBM counters of the W = 10 000 window kernel."""
    d = os.path.join(out, "trace_sdp")
    p = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--",
                        "python3", "tools/bench_sdp.py", "dtw", "wtw", "big", "chroma"], cwd=ROOT,
                       env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-3000:])
        raise SystemExit("bench_sdp under rocprofv3 failed")
    with open(os.path.join(out, tag + "_sdp_under_rocprof.jsonl"), "w") as f:
        f.write("".join(l + "\n" for l in p.stdout.splitlines() if l.startswith("{")))
    stats = find(d, "kernel_stats.csv")
    keep = ("sdp_kernel", "big_dp", "big_ctl", "big_hops", "big_segment", "dtw_hops", "dtw_segment", "dtw_cost", "dtw_prep",
            "chroma_frames", "wtw_advance", "wtw_win", "_tail_kernel", "tail_ctl", "big_fill")
    with open(stats) as f, open(os.path.join(out, tag + "_sdp_kernel_stats.csv"), "w") as g:
        for i, line in enumerate(f):
            if i == 0 or any(k in line for k in keep):
                g.write(line[:400] + ("\n" if len(line) > 400 else ""))
    # the W = 10 000 window: HBM counters of wtw_big_dp_kernel, separate passes
    pmc = {}
    for name, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"])):
        dd = os.path.join(out, "pmc_sdp_" + name)
        run(["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", dd, "--",
             "python3", "tools/bench_sdp.py", "wtw10k"])
        vals = []
        with open(find(dd, "counter_collection.csv")) as f:
            for row in csv.DictReader(f):
                if "wtw_big_dp_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counters[0]:
                    vals.append(float(row["Counter_Value"]))
        vals = [v for v in vals if v > 0.5 * max(vals)] if vals else []  # rounds with a pending window
        pmc[counters[0]] = sum(vals) / max(len(vals), 1)
        pmc[counters[0] + "_dispatches"] = len(vals)
    with open(os.path.join(out, tag + "_sdp_pmc.json"), "w") as f:
        json.dump({"kernel": "wtw_big_dp_kernel (W = 10 000, one window per dispatch)", "KB_per_dispatch": pmc,
                   "hbm_bytes_per_window": int((2 * pmc.get("FETCH_SIZE", 0) + pmc.get("WRITE_SIZE", 0)) * 1024),
                   "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH_SIZE doubled per MI355X_MICROARCH.md"}, f, indent=1)


if __name__ == "__main__":
    main()
