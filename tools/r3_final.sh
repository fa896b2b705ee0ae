#!/bin/bash
# HISTORIC: ran against the library of its round; the SE3D_* environment switches named here were replaced by se3d_options in round 5 (tools/profile_r5.sh is the current form).
# Evidence run of round 3 on the GPU box (gpurun -- bash tools/r3_final.sh): the GPU suite, the default bench line, the parity sweep as a tool
# (one JSON report per seed) and the rocprofv3 passes behind profiles/r3/.  Outputs under gpurun_out/r3_final/ and gpurun_out/prof_r3/.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_final
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gputest.log 2>&1; echo "pytest rc $?" >> $OUT/gputest.log; tail -3 $OUT/gputest.log
timeout 900 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench rc $?"
bash tools/fuzz_sweep.sh $OUT/fuzz_parity.log | tail -1 | tee $OUT/fuzz_parity_summary.txt
wc -l $OUT/fuzz_parity.log
bash tools/profile_r3.sh serial c256 > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log
