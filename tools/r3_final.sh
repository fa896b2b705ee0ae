#!/bin/bash
# Evidence run of round 3 on the GPU box (gpurun -- bash tools/r3_final.sh): the GPU suite, the default bench line, the parity sweep as a tool
# (one JSON report per seed) and the rocprofv3 passes behind profiles/r3/.  Outputs under gpurun_out/r3_final/ and gpurun_out/prof_r3/.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_final
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gputest.log 2>&1; echo "pytest rc $?" >> $OUT/gputest.log; tail -3 $OUT/gputest.log
timeout 900 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench rc $?"
: > $OUT/fuzz_parity.log
for s in 4 15 37 39 0 1 2 3 104 203 101 102 103 105 106 107 108 109 110 111 112 201 202 204 205 206; do timeout 200 python tools/fuzz_parity.py $s 40 2>> $OUT/fuzz_parity.err | tail -1 >> $OUT/fuzz_parity.log; done
FUZZ_POSE=h36m timeout 200 python tools/fuzz_parity.py 5 40 2>> $OUT/fuzz_parity.err | tail -1 >> $OUT/fuzz_parity.log
FUZZ_NORM_HEIGHT=1 timeout 200 python tools/fuzz_parity.py 6 40 2>> $OUT/fuzz_parity.err | tail -1 >> $OUT/fuzz_parity.log
FUZZ_POSE=h36m FUZZ_NORM_HEIGHT=1 timeout 200 python tools/fuzz_parity.py 7 40 2>> $OUT/fuzz_parity.err | tail -1 >> $OUT/fuzz_parity.log
wc -l $OUT/fuzz_parity.log
bash tools/profile_r3.sh serial c256 > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log
