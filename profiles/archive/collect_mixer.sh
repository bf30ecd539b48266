#!/usr/bin/env bash
# Evidence for the W-inside-the-step mixer (psf_mixer_fwd_f32), run on the GPU box from the repo root:
#     bash profiles/collect_mixer.sh <tag>
# For each route of the Adding forward (N = 16384, B = 64): kernel durations (--kernel-trace --stats) and, in separate
# passes, the memory-side counters FETCH_SIZE and WRITE_SIZE per kernel. The W bytes the producer writes (14 x 63 MB per
# forward) and the chain reads must be absent from the second route. Program directly after `--`; counters and traces never combined.
set -u
TAG=${1:-r04e}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_mixer
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for route in never always recipe; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$route" -- python3 $ROOT/profiles/mixer_route_run.py $route > "$OUT/stats_$route.log" 2>&1
  echo "stats $route rc=$?"
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$route" -- python3 $ROOT/profiles/mixer_route_run.py $route > "$OUT/fetch_$route.log" 2>&1
  echo "fetch $route rc=$?"
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_$route" -- python3 $ROOT/profiles/mixer_route_run.py $route > "$OUT/write_$route.log" 2>&1
  echo "write $route rc=$?"
done
cd "$ROOT"
python3 profiles/summarize_mixer.py "$TAG" > gpurun_out/${TAG}_mixer_summary.md 2>&1
cat gpurun_out/${TAG}_mixer_summary.md
