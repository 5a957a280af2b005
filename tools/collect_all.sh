#!/bin/bash
# One pass over everything profiles/<tag>_* holds (run on the GPU box: gpurun -- bash tools/collect_all.sh r02b).
set -e
TAG=${1:-r02b}
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
echo collected
