#!/bin/bash
# Runs on the GPU box (gpurun -- bash tools/profile_r6.sh <what> ...): the rocprofv3 passes behind profiles/r6/.
#   serial      : bench.py --serial (one context, one stream) kernel stats
#   serial_pmc  : SQ + HBM traffic PMC passes of the serial pass
#   valumix     : VALU instruction-mix counters of the serial pass
#   c256        : 256cam128p, one context: kernel stats (+ PMC with c256_pmc)
#   ablate      : a variant library tools/tmp/libse3d_$ABLATE.so through the serial pass
#   c256_ab     : variant libraries tools/tmp/libse3d_<name>.so (tools/build_variants.sh) through the one-context 256cam128p pass: AB="name ..."
#   c256_ab_pmc : SQ + LDS counters of the pair kernel for such variants
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r6
mkdir -p "$OUT"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for what in "$@"; do
case $what in
serial)
  rm -rf $OUT/serial_stats
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_stats -- python3 bench.py --serial --steps 5 --warmup 1 --no-cpu-baseline --no-parity > $OUT/serial_bench_stats.log 2>&1
  python3 tools/kernel_stats.py $OUT/serial_stats $OUT/serial_16cam6p_kernel_stats.csv
  ;;
serial_pmc)
  rm -rf $OUT/serial_sq $OUT/serial_fetch $OUT/serial_write
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/serial_sq -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $OUT/serial_bench_sq.log 2>&1
  python3 tools/pmc_sq.py $OUT/serial_sq $OUT/pmc_sq_16cam6p_serial.json "bench.py --serial --steps 2 --warmup 1 (16cam6p, one context, 6144 sequences x 16 frames)"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/serial_fetch -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $OUT/serial_bench_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/serial_write -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $OUT/serial_bench_write.log 2>&1
  python3 tools/pmc_traffic2.py $OUT/serial_fetch $OUT/serial_write $OUT/pmc_traffic_16cam6p_serial.json 6144 "bench.py --serial --steps 2 --warmup 1"
  ;;
ablate)
  # a variant library (tools/tmp/libse3d_<name>.so, built with an ablation macro) through the serial pass: ablate=<name> via ABLATE env
  rm -rf $OUT/serial_stats_$ABLATE
  export SE3D_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/tmp/libse3d_$ABLATE.so
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_stats_$ABLATE -- python3 bench.py --serial --steps 5 --warmup 1 --no-cpu-baseline --no-parity > $OUT/serial_bench_stats_$ABLATE.log 2>&1
  unset SE3D_LIB_OVERRIDE
  python3 tools/kernel_stats.py $OUT/serial_stats_$ABLATE $OUT/serial_16cam6p_${ABLATE}_kernel_stats.csv
  ;;
valumix)
  # instruction mix of the vector ALU per kernel (two passes: the counters do not fit one)
  rm -rf $OUT/serial_mix64 $OUT/serial_mix32
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --output-format csv -d $OUT/serial_mix64 -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $OUT/serial_bench_mix64.log 2>&1
  python3 tools/pmc_sq.py $OUT/serial_mix64 $OUT/pmc_valumix64_16cam6p_serial.json "bench.py --serial --steps 2 --warmup 1: fp64 VALU instruction counters"
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/serial_mix32 -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $OUT/serial_bench_mix32.log 2>&1
  python3 tools/pmc_sq.py $OUT/serial_mix32 $OUT/pmc_valumix32_16cam6p_serial.json "bench.py --serial --steps 2 --warmup 1: fp32 / integer VALU instruction counters"
  ;;
c256)
  rm -rf $OUT/c256_stats1
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c256_stats1 -- python3 bench.py --workload 256cam128p --contexts 1 --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $OUT/c256_bench_1ctx.log 2>&1
  python3 tools/kernel_stats.py $OUT/c256_stats1 $OUT/c256_1ctx_kernel_stats.csv
  ;;
c256_ab)
  # variant libraries tools/tmp/libse3d_<name>.so through the one-context 256cam128p pass: AB="name name ..."
  for V in $AB; do
    rm -rf $OUT/c256_stats_$V
    export SE3D_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/tmp/libse3d_$V.so
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c256_stats_$V -- python3 bench.py --workload 256cam128p --contexts 1 --steps 2 --warmup 1 --no-cpu-baseline --no-parity ${AB_FLAGS:-} > $OUT/c256_bench_1ctx_$V.log 2>&1
    unset SE3D_LIB_OVERRIDE
    python3 tools/kernel_stats.py $OUT/c256_stats_$V $OUT/c256_1ctx_${V}_kernel_stats.csv | grep pairs_rows | sed "s/^/$V: /"
  done
  ;;
c256_ab_pmc)
  # SQ + LDS counters of the pair kernel for variant libraries: AB="name ..."
  for V in $AB; do
    export SE3D_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/tmp/libse3d_$V.so
    rm -rf $OUT/c256_sq_$V $OUT/c256_lds_$V
    timeout 900 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/c256_sq_$V -- python3 bench.py --workload 256cam128p --contexts 1 --steps 1 --warmup 1 --no-cpu-baseline --no-parity > $OUT/c256_bench_sq_$V.log 2>&1
    python3 tools/pmc_sq.py $OUT/c256_sq_$V $OUT/pmc_sq_256cam128p_1ctx_$V.json "variant $V"
    timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/c256_lds_$V -- python3 bench.py --workload 256cam128p --contexts 1 --steps 1 --warmup 1 --no-cpu-baseline --no-parity > $OUT/c256_bench_lds_$V.log 2>&1
    python3 tools/pmc_sq.py $OUT/c256_lds_$V $OUT/pmc_lds_256cam128p_1ctx_$V.json "variant $V: LDS counters"
    unset SE3D_LIB_OVERRIDE
    python3 - <<PY
import json
for f in ("$OUT/pmc_sq_256cam128p_1ctx_$V.json", "$OUT/pmc_lds_256cam128p_1ctx_$V.json"):
    d = json.load(open(f))["kernels"].get("epipolar_pairs")
    print("$V", json.dumps(d))
PY
  done
  ;;
c256_pmc)
  rm -rf $OUT/c256_sq $OUT/c256_fetch $OUT/c256_write
  timeout 900 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/c256_sq -- python3 bench.py --workload 256cam128p --contexts 1 --steps 1 --warmup 1 --no-cpu-baseline --no-parity > $OUT/c256_bench_sq.log 2>&1
  python3 tools/pmc_sq.py $OUT/c256_sq $OUT/pmc_sq_256cam128p_1ctx.json "bench.py --workload 256cam128p --contexts 1 --steps 1 --warmup 1"
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c256_fetch -- python3 bench.py --workload 256cam128p --contexts 1 --steps 1 --warmup 1 --no-cpu-baseline --no-parity > $OUT/c256_bench_fetch.log 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c256_write -- python3 bench.py --workload 256cam128p --contexts 1 --steps 1 --warmup 1 --no-cpu-baseline --no-parity > $OUT/c256_bench_write.log 2>&1
  python3 tools/pmc_traffic2.py $OUT/c256_fetch $OUT/c256_write $OUT/pmc_traffic_256cam128p_1ctx.json 4 "bench.py --workload 256cam128p --contexts 1 --steps 1 --warmup 1 (one context, 4 sequences x 16 frames)"
  ;;
esac
done
