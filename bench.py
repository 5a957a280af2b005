#!/usr/bin/env python3
"""Headline benchmark: aligned chroma frames/sec, batch=64 concurrent live streams against one
reference, OTW (otw_eran semantics) c=500, max_run_count=3 -- BASELINE.json configs[2] on one GPU,
configs[3] (64 streams per GPU, independent shards, no collective on the data path) on N GPUs.

A "step" = one whole-sequence pass of the hot path over the batch: rts_otw_run on B streams of
~2150 synthetic chroma frames each (inputs already resident in HBM).  value = live frames consumed
by all ranks per second of wall time (barrier + synchronize on both sides, max over ranks).

Extra JSON objects (see the task contract):
  roofline      dominant kernel (otw_advance_kernel): algorithmic bytes per launch / mean launch
                duration from HIP events recorded on the launch stream, against HBM 8 TB/s.
  cpu_baseline  timed on this box's host cores on the same streams, before the GPU is touched (the worker
                processes are forked while the process is still GPU-free): the C port of otw_eran.py's insert
                loop (oracle/) on 1 core and on all usable cores (one stream per task), and the numpy
                restatement (oracle/otw_numpy.py -- the reference's own per-cell numpy calls) on 1 core for one
                stream.  The C port's paths are also the parity gate: every stream's path must equal them.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def usable_cpus():
    """Cores this process may actually use: scheduler affinity, further limited by a cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_worker(task):
    """One stream through the C oracle in a worker process: (frames, seconds in the insert loop, path)."""
    ref, live, c, mrc = task
    import oracle
    o = oracle.OtwOracle(ref, c, mrc)
    lv = np.ascontiguousarray(live.T)
    t0 = time.perf_counter()
    n = o._L.orc_otw_run(o._h, lv.ctypes.data, lv.shape[0])  # insert loop only, like SURVEY 6
    dt = time.perf_counter() - t0
    return n, dt, o.path, os.getpid()


def cpu_legs(ref, lives, c, mrc, n_cpu, numpy_leg=True):
    """The CPU baseline legs.  Must run before anything touches the GPU (it forks)."""
    import multiprocessing as mp
    import oracle
    oracle.build()
    tasks = [(ref, lives[b], c, mrc) for b in range(n_cpu)]
    # 1 core, streams one after another
    seq = [_cpu_worker(t) for t in tasks]
    frames1 = sum(r[0] for r in seq)
    secs1 = sum(r[1] for r in seq)
    paths = [r[2] for r in seq]
    legs = {"c_port_1core": {"value": frames1 / secs1, "unit": "frames/s", "cores": 1,
                             "sample": "%d streams, %d frames, one after another" % (n_cpu, frames1)}}
    # all usable cores, one stream per task; the rate is frames / the busiest worker's time in the insert loop
    procs = min(n_cpu, usable_cpus())
    if procs > 1:
        with mp.get_context("fork").Pool(procs) as pool:
            pool.map(_cpu_worker, tasks[:procs], chunksize=1)  # warm every worker (library load, first touch)
            w0 = time.perf_counter()
            par = pool.map(_cpu_worker, tasks, chunksize=1)
            wall = time.perf_counter() - w0
        per_proc = {}
        for n, dt, _, pid in par:
            per_proc[pid] = per_proc.get(pid, 0.0) + dt
        framesp = sum(r[0] for r in par)
        legs["c_port_allcores"] = {"value": framesp / max(per_proc.values()), "unit": "frames/s", "cores": procs,
                                   "wall_value": framesp / wall,
                                   "sample": "%d streams over %d forked processes; value = frames / busiest process's "
                                             "insert-loop time, wall_value includes the 2N x N matrix allocation per stream"
                                             % (n_cpu, procs)}
    if numpy_leg:
        from oracle import otw_numpy
        o = otw_numpy.NumpyOTW(ref, c, mrc)
        t0 = time.perf_counter()
        n = o.run(lives[0])
        dt = time.perf_counter() - t0
        ok = np.array_equal(np.array(o.path, dtype=np.int32).reshape(-1, 2), paths[0])
        legs["numpy_1core"] = {"value": n / dt, "unit": "frames/s", "cores": 1, "path_equals_c_port": bool(ok),
                               "sample": "stream 0 only (%d frames): the reference's per-cell numpy calls, pure Python "
                                         "loop; the reference itself: 262 frames/s (BASELINE.md 3a, survey container)" % n}
    return legs, paths


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="streams per GPU")
    ap.add_argument("--n-ref", type=int, default=2200)
    ap.add_argument("--c", type=int, default=500)
    ap.add_argument("--max-run-count", type=int, default=3)
    ap.add_argument("--waves", type=int, default=0, help="waves per stream workgroup (0 = library default)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="feature dtype in HBM")
    ap.add_argument("--cpu-streams", type=int, default=-1, help="streams timed on the CPU oracle (-1 = all on rank 0)")
    ap.add_argument("--no-cpu", action="store_true", help="skip cpu_baseline / parity gate (profiling runs)")
    ap.add_argument("--no-numpy", action="store_true", help="skip the (slow) numpy leg of cpu_baseline")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "otw_traffic.json"),
                    help="PMC-derived HBM bytes per launch (from the committed rocprofv3 passes) to report as roofline.traffic")
    args = ap.parse_args()
    verbose = bool(os.environ.get("BENCH_VERBOSE"))

    def note(msg):
        if verbose:
            print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)

    if verbose:
        import faulthandler
        faulthandler.dump_traceback_later(90, repeat=True, file=sys.stderr)
    note("start rank=%s world=%s" % (os.environ.get("RANK"), os.environ.get("WORLD_SIZE")))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)

    from real_time_audio_sync_amd import shard, synth

    # ---- workload: one reference, `batch` different time-warped noisy renditions per rank
    B = args.batch
    ref = synth.synth_ref(args.n_ref, seed=1000)
    lo, hi = shard.partition(B * world, world, rank)  # contiguous slice of the global stream list
    lives = [synth.synth_live(ref, seed=shard.stream_seed(1000, g)) for g in range(lo, hi)]

    # ---- CPU baseline legs (rank 0, N = 1 only), before the GPU is initialised: the workers are forked
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        n_cpu = B if args.cpu_streams < 0 else min(B, args.cpu_streams)
        cpu = cpu_legs(ref, lives, args.c, args.max_run_count, n_cpu, numpy_leg=not args.no_numpy)
        note("cpu legs done")

    import torch
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # Rehearsal switches for a one-GPU box (never set by the driver): all ranks on device 0 and the two
    # tiny report reductions over gloo, since RCCL refuses two ranks on one device.
    single_device = bool(os.environ.get("BENCH_SINGLE_DEVICE"))
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = 0 if single_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from real_time_audio_sync_amd import otw_batch

    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    eng = otw_batch.BatchedOTW(ref, args.c, args.max_run_count, batch=B, variant="otw", dtype=tdt, device=dev,
                               waves=(args.waves or None))
    live_dev, len_dev = eng.pack(lives)
    note("inputs resident")

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        eng.run(live_dev, len_dev)
    barrier()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record()
        eng.run(live_dev, len_dev)
        ev1[i].record()
    barrier()
    t1 = time.perf_counter()
    note("timed region done")
    elapsed = t1 - t0
    launch_ms = [a.elapsed_time(b) for a, b in zip(ev0, ev1)]

    states = eng.states()
    from real_time_audio_sync_amd import _native as nat
    frames = int(states[:, nat.ST_CONSUMED].sum())
    cells = int(sum((int(np.uint32(s[nat.ST_CELLS_HI])) << 32) | int(np.uint32(s[nat.ST_CELLS_LO])) for s in states))
    n_col = int(states[:, nat.ST_COL_STRIPS].sum())
    n_path = int(states[:, nat.ST_N_PATH].sum())
    # A = 4*cells + 48 + 48*n_col + 8*n_path bytes per live frame, summed exactly
    alg_bytes = 4 * cells + 48 * frames + 48 * n_col + 8 * n_path

    # the only collectives in the run: max clock and frame count for the report (never on the data path)
    elapsed, total_frames = shard.reduce_clock_and_count(elapsed, frames, device=dev if backend == "nccl" else None)

    result = None
    if rank == 0:
        value = total_frames * args.steps / elapsed
        mean_launch_s = float(np.mean(launch_ms)) * 1e-3
        achieved = alg_bytes / mean_launch_s / 1e9
        traffic = None
        default_workload = (B == 64 and args.c == 500 and args.n_ref == 2200 and args.dtype == "f32")
        if default_workload and args.traffic_json and os.path.exists(args.traffic_json):
            traffic = json.load(open(args.traffic_json)).get("hbm_bytes_per_launch")
        # latency model: every stream is one workgroup walking its own dependent chain of steps
        # (one decide() = one path point per step), so a launch lasts as long as its longest stream:
        #   frames/s <= B * f_clk * frames_per_step / cycles_per_step
        steps_max = int(states[:, nat.ST_N_PATH].max())
        f_clk = 2.4e9
        cycles_per_step = mean_launch_s * f_clk / max(steps_max, 1)
        result = {
            "metric": "aligned chroma frames/sec, batch=64 OTW c=500; path-index match vs CPU ref",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.dtype == "f64" else "f64 arithmetic, f32 features in HBM",
            "data": "synthetic",
            "config": {"workload": "configs[%d]: batch=%d concurrent live streams per GPU vs one reference, OTW c=%d, "
                                   "max_run_count=%d, ref %d frames, live ~%d frames/stream, %s chroma in HBM"
                                   % (2 if world == 1 else 3, B, args.c, args.max_run_count, args.n_ref,
                                      frames // B, args.dtype),
                       "streams_total": B * world, "frames_per_step": total_frames,
                       "cells_per_frame": cells / max(frames, 1), "waves_per_stream": args.waves or 8},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": eng.kernel_name, "launch_ms": mean_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "latency_model": {
                             "f_clk_hz": f_clk, "steps_longest_stream": steps_max,
                             "cycles_per_step": cycles_per_step,
                             "steps_per_frame": n_path / max(frames, 1),
                             "cells_per_step": cells / max(n_path, 1),
                             "bound_frames_per_s": B * f_clk * (frames / max(n_path, 1)) / cycles_per_step,
                             "formula": "frames/s <= B * f_clk / (steps_per_frame * cycles_per_step); a step is one "
                                        "decide(): ~c cells of a dependent float64 (min,+) chain plus the control "
                                        "decision, executed by one workgroup per stream"},
                         "note": "B=64 occupies 64 of 256 CUs and each strip is a dependent float64 "
                                 "(min,+) chain: latency/occupancy-bound by construction, not HBM-bound"},
        }

    # ---- parity gate (outside the timed region): every stream's path against the C oracle's.  At N=1 the oracle
    # paths come from the cpu_baseline legs computed before the GPU was touched; at N>1 a light gate (4 streams of
    # rank 0's shard) runs here -- the baseline is an N=1 figure by contract.
    if rank == 0 and not args.no_cpu:
        if cpu is not None:
            legs, opaths = cpu
        else:
            import oracle
            legs, opaths = None, []
            for b in range(min(B, 4)):
                o = oracle.OtwOracle(ref, args.c, args.max_run_count)
                o.run(lives[b])
                opaths.append(o.path)
                del o
        mismatches = sum(0 if np.array_equal(eng.path(b), opaths[b]) else 1 for b in range(len(opaths)))
        if legs is not None:
            main_leg = legs.get("c_port_allcores", legs["c_port_1core"])
            result["cpu_baseline"] = {
                "value": main_leg["value"], "unit": "frames/s", "cores": main_leg["cores"], "kind": "port",
                "sample": "all %d streams of the batch (%d frames each on average) through the C port of otw_eran.py's "
                          "insert loop (oracle/rtsync_oracle.c, dense 2N x N float64 matrices, constructor excluded) on "
                          "%d host cores; legs: the same on 1 core, and the numpy restatement on 1 core"
                          % (len(opaths), frames // B, main_leg["cores"]),
                "legs": legs, "host_cpus": os.cpu_count(), "usable_cpus": usable_cpus()}
        result["parity"] = {"streams_checked": len(opaths), "path_mismatches": mismatches}
        if mismatches:
            result["value"] = 0.0  # a fast kernel whose results differ is not done
            result["parity"]["note"] = "PATH MISMATCH vs CPU oracle: value voided"
    if rank == 0:
        print(json.dumps(result))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
