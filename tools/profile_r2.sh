#!/bin/bash
# HISTORIC: ran against the library of its round; the SE3D_* environment switches named here were replaced by se3d_options in round 5 (tools/profile_r5.sh is the current form).
# Runs on the GPU box (gpurun -- bash tools/profile_r2.sh <what> ...): the rocprofv3 passes behind profiles/r2/.
#   serial   : bench.py --serial (one context, one stream) kernel stats + SQ PMC + HBM traffic PMC
#   c256     : 256cam128p (1 context and the default 3) kernel stats + SQ PMC of one context
#   lm       : the four SE3D_LM modes of the per-step LM kernel, kernel stats
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r2
mkdir -p "$OUT"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for what in "$@"; do
case $what in
serial)
  rm -rf $OUT/serial_stats $OUT/serial_sq $OUT/serial_fetch $OUT/serial_write
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_stats -- python3 bench.py --serial --steps 5 --warmup 1 --no-cpu-baseline > $OUT/serial_bench_stats.log 2>&1
  python3 tools/kernel_stats.py $OUT/serial_stats $OUT/serial_16cam6p_kernel_stats.csv
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/serial_sq -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > $OUT/serial_bench_sq.log 2>&1
  python3 tools/pmc_sq.py $OUT/serial_sq $OUT/pmc_sq_16cam6p_serial.json "bench.py --serial --steps 2 --warmup 1 (16cam6p, one context, 6144 sequences x 16 frames)"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/serial_fetch -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > $OUT/serial_bench_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/serial_write -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline > $OUT/serial_bench_write.log 2>&1
  python3 tools/pmc_traffic2.py $OUT/serial_fetch $OUT/serial_write $OUT/pmc_traffic_16cam6p_serial.json 6144 "bench.py --serial --steps 2 --warmup 1"
  ;;
c256)
  rm -rf $OUT/c256_stats1 $OUT/c256_stats3 $OUT/c256_sq $OUT/c256_fetch $OUT/c256_write
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c256_stats1 -- python3 bench.py --workload 256cam128p --contexts 1 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c256_bench_1ctx.log 2>&1
  python3 tools/kernel_stats.py $OUT/c256_stats1 $OUT/c256_1ctx_kernel_stats.csv
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c256_stats3 -- python3 bench.py --workload 256cam128p --steps 3 --warmup 2 --no-cpu-baseline > $OUT/c256_bench_4ctx.log 2>&1
  python3 tools/kernel_stats.py $OUT/c256_stats3 $OUT/c256_4ctx_kernel_stats.csv
  timeout 900 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/c256_sq -- python3 bench.py --workload 256cam128p --contexts 1 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/c256_bench_sq.log 2>&1
  python3 tools/pmc_sq.py $OUT/c256_sq $OUT/pmc_sq_256cam128p_1ctx.json "bench.py --workload 256cam128p --contexts 1 --steps 1 --warmup 1"
  ;;
lm)
  for mode in tree tree_mfma schur dense; do
    rm -rf $OUT/lm_$mode
    SE3D_LM=$mode timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lm_$mode -- python3 bench.py --workload 16cam6p --walk steps --contexts 1 --seqs 128 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/lm_bench_$mode.log 2>&1
    python3 tools/kernel_stats.py $OUT/lm_$mode $OUT/lm_${mode}_kernel_stats.csv
  done
  ;;
esac
done
ls -la $OUT
