#!/bin/bash
# One pass over everything profiles/<tag>_* holds (run on the GPU box: gpurun -- bash tools/collect_all.sh r03d; build tools/_diag/librtsync_diag.so with RTS_DIAG_LEVEL=2 python tools/otw_phase_profile.py --build first).
set -e
TAG=${1:-r03d}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 tools/collect_profile.py $TAG > $OUT/collect.log 2>&1 || { tail -30 $OUT/collect.log; exit 1; }
tail -5 $OUT/collect.log
: > $OUT/${TAG}_batch_sweep.jsonl
for B in 128 256 512 1024 4096; do
  timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu --batch $B >> $OUT/${TAG}_batch_sweep.jsonl
done
timeout -k 10 300 python3 bench.py --dtype f64 --no-numpy > $OUT/${TAG}_bench_f64.json
timeout -k 10 300 python3 bench.py --dtype f64 --no-cpu --batch 4096 --steps 5 --warmup 2 >> $OUT/${TAG}_batch_sweep.jsonl
timeout -k 10 600 python3 tools/bench_secondary.py > $OUT/${TAG}_secondary_kernels.jsonl 2> $OUT/secondary.err || { tail -20 $OUT/secondary.err; exit 1; }
timeout -k 10 300 python3 tools/bench_live.py 120 > $OUT/${TAG}_live.jsonl 2> $OUT/live.err || { tail -20 $OUT/live.err; exit 1; }
timeout -k 10 300 python3 tools/bench_wtw.py 20 64 100 128 > $OUT/${TAG}_wtw_paths.jsonl 2> $OUT/wtw.err || { tail -20 $OUT/wtw.err; exit 1; }
if [ -f tools/_diag/librtsync_diag.so ]; then
  RTS_DIAG_LEVEL=2 timeout -k 10 200 python3 tools/otw_phase_profile.py > $OUT/${TAG}_phase_light.txt 2>/dev/null || true
fi
echo collected
