#!/bin/bash
# HISTORIC: ran against the library of its round; the SE3D_* environment switches named here were replaced by se3d_options in round 5 (tools/profile_r5.sh is the current form).
# Runs on the GPU box (gpurun): the three rocprofv3 passes behind profiles/<round>/ for one bench.py configuration.
#   bash tools/profile_bench.sh <tag> [bench.py args...]
# 1. --kernel-trace --stats            -> per-kernel durations (must agree with bench.py's in-library HIP-event times)
# 2. --pmc FETCH_SIZE (own pass)       -> HBM read bytes  (TCC: FETCH_SIZE needs 3 of 4 slots)
# 3. --pmc WRITE_SIZE (own pass)       -> HBM write bytes
# Outputs land in gpurun_out/prof_<tag>/ ; tools/pmc_traffic.py turns them into the committed summaries.
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench_stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench_write.log" 2>&1
timeout 600 python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench_plain.log" 2>&1
python3 tools/pmc_traffic.py "$OUT" "$TAG" "$@"
ls -la "$OUT"
