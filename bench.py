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
  cpu_baseline  the C oracle (a port of otw_eran.py, 1 core) timed on this box on the same streams;
                also the parity gate: every stream's path must equal the oracle's.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


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
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
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

    from real_time_audio_sync_amd import otw_batch, shard, synth

    # ---- workload: one reference, `batch` different time-warped noisy renditions per rank
    B = args.batch
    ref = synth.synth_ref(args.n_ref, seed=1000)
    lo, hi = shard.partition(B * world, world, rank)  # contiguous slice of the global stream list
    lives = [synth.synth_live(ref, seed=shard.stream_seed(1000, g)) for g in range(lo, hi)]
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
        result = {
            "metric": "aligned chroma frames/sec, batch=64 OTW c=500; path-index match vs CPU ref",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
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
                         "note": "B=64 occupies 64 of 256 CUs and each strip is a dependent float64 "
                                 "(min,+) chain: latency/occupancy-bound by construction, not HBM-bound"},
        }

    # ---- CPU baseline (rank 0, N=1 only) + parity gate, outside the timed region.  At N>1 only a light
    # parity gate runs (4 streams of rank 0's shard): the baseline is an N=1 figure by contract.
    if rank == 0 and not args.no_cpu:
        import oracle
        n_cpu = B if args.cpu_streams < 0 else min(B, args.cpu_streams)
        if world > 1:
            n_cpu = min(n_cpu, 4)
        cpu_frames, cpu_time, mismatches = 0, 0.0, 0
        for b in range(n_cpu):
            o = oracle.OtwOracle(ref, args.c, args.max_run_count)
            lv = np.ascontiguousarray(lives[b].T)
            c0 = time.perf_counter()
            n = o._L.orc_otw_run(o._h, lv.ctypes.data, lv.shape[0])  # insert loop only, like SURVEY 6
            cpu_time += time.perf_counter() - c0
            cpu_frames += n
            if not np.array_equal(eng.path(b), o.path):
                mismatches += 1
            del o
        if world == 1:
            result["cpu_baseline"] = {
                "value": cpu_frames / cpu_time, "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": "%d of the %d streams (%d frames), C port of otw_eran.py's insert loop (oracle/), "
                          "dense 2N x N float64 matrices, constructor excluded; reference Python itself: 262 frames/s "
                          "(BASELINE.md 3a, survey container)" % (n_cpu, B, cpu_frames),
                "host_cpus": os.cpu_count()}
        result["parity"] = {"streams_checked": n_cpu, "path_mismatches": mismatches}
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
