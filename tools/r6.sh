#!/bin/bash
# Round 6's measurements on the GPU box (gpurun --timeout N -- bash tools/r6.sh <parts>): written to gpurun_out/r6/, the summaries that are
# quoted are copied into profiles/r6/ by hand.
#   robust [tag]   : bench.py headline + serial + `robust` block (rough scenes) without the other blocks -> robust_<tag>.log
#   robust_prof [tag]: rocprofv3 kernel stats of the rough-scene serial pass -> serial_16cam6p_rough_<tag>_kernel_stats.csv
#   sdma           : tools/microbench/sdma_pair (does an upload stream + a download stream overlap, and what decides it), two processes
#   c256 [tag]     : 256cam128p one context, kernel stats
#   suite          : pytest -m gpu
#   bench N        : N default bench.py runs
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6
mkdir -p $O
while [ $# -gt 0 ]; do
part=$1; shift
case $part in
robust)
  tag=${1:-x}; shift || true
  timeout 900 python bench.py --no-secondary --no-streaming --no-parity --no-cpu-baseline > $O/robust_$tag.log 2> $O/robust_$tag.err
  tail -1 $O/robust_$tag.log | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
r = d['robust']
print('clean', d['value'], d['ms_per_step'], '| robust', r['value'], r['ms_per_step'], 'frac_of_clean', r['fraction_of_clean_value'], 'second pass', r['second_pass_fraction'], 'serial', json.dumps(r['serial']))
print('outlier-heavy serial', json.dumps(r.get('outlier_heavy_serial')))"
  tail -3 $O/robust_$tag.err
  ;;
robust_prof)
  tag=${1:-x}; shift || true
  rm -rf $O/rough_stats_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rough_stats_$tag -- python3 bench.py --workload ${ROUGH_WORKLOAD:-16cam6p_rough} --serial --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/rough_serial_$tag.log 2>&1
  python3 tools/kernel_stats.py $O/rough_stats_$tag $O/serial_16cam6p_rough_${tag}_kernel_stats.csv | head -12
  rm -rf $O/rough_stats_$tag
  ;;
rough_ab)
  # variant libraries tools/tmp/libse3d_<name>.so (tools/build_variants.sh) through the rough-scene serial pass: AB="name ..."
  for V in ${AB:-}; do
    rm -rf $O/rough_stats_$V
    export SE3D_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/tmp/libse3d_$V.so
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rough_stats_$V -- python3 bench.py --workload 16cam6p_rough --serial --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/rough_serial_$V.log 2>&1
    unset SE3D_LIB_OVERRIDE
    python3 tools/kernel_stats.py $O/rough_stats_$V $O/serial_16cam6p_rough_${V}_kernel_stats.csv | grep "triangulate" | sed "s/^/$V: /"
    rm -rf $O/rough_stats_$V
  done
  ;;
rough_pmc)
  tag=${1:-x}; shift || true
  SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
  rm -rf $O/rough_sq_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/rough_sq_$tag -- python3 bench.py --workload 16cam6p_rough --serial --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $O/rough_sq_$tag.log 2>&1
  python3 tools/pmc_sq.py $O/rough_sq_$tag $O/pmc_sq_16cam6p_rough_$tag.json "bench.py --workload 16cam6p_rough --serial --steps 2 --warmup 1"
  python3 -c "
import json
d = json.load(open('$O/pmc_sq_16cam6p_rough_$tag.json'))['kernels']
for k in ('triangulate_second_pass', 'triangulate'):
    print(k, json.dumps(d.get(k)))"
  rm -rf $O/rough_sq_$tag
  ;;
rough_scan)
  # second-pass fraction and rates of the rough-scene serial pass under parameter overrides: ROUGHS="a=b,c=d e=f ..."
  for R in ${ROUGHS:-}; do
    timeout 600 python3 bench.py --workload 16cam6p_rough --serial --steps 3 --warmup 1 --no-cpu-baseline --no-parity --rough "$R" 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$R', d['value'], d['ms_per_step'], d['second_pass'], {k: v for k, v in d['roofline'].items() if k in ('kernel', 'avg_launch_ms')})"
  done
  ;;
stream_diag)
  # the default run's passes (secondary, headline, serial) followed by REPEATED streaming passes in the same process: n processes x r passes
  n=${1:-3}; shift || true
  r=${1:-6}; shift || true
  grep -H "MemFree\|MemTotal" /sys/devices/system/node/node*/meminfo > $O/stream_diag_numa.txt 2>&1
  for i in $(seq 1 $n); do
    timeout 900 python bench.py --no-parity --no-cpu-baseline --no-robust --stream-repeat $r ${STREAM_FLAGS:-} > $O/stream_diag_$i.log 2> $O/stream_diag_$i.err
    grep "streaming pass" $O/stream_diag_$i.err | sed "s/^/proc $i: /"
    grep -H "MemFree" /sys/devices/system/node/node*/meminfo >> $O/stream_diag_numa.txt 2>&1
  done
  cat $O/stream_diag_numa.txt | head -8
  ;;
probe_bisect)
  # which part of se3d_copy_stream_probe (diagnostic builds tools/tmp/libse3d_probediag<mode>.so, -DSE3D_PROBE_DIAG=<bit mask>: 1 warm both, 2 up alone, 4 down alone, 8 both)
  # flips the streaming path into its slow mode?  three streaming passes per process, the probe runs after each
  # (build first: for M in 15 9 2 4 0; do tools/build_variants.sh probediag$M "-DSE3D_PROBE_DIAG=$M" "" api; done — the mode is a compile-time constant: no getenv in csrc/)
  for M in ${MODES:-15 9 2 4 0}; do
    SE3D_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/tmp/libse3d_probediag$M.so timeout 600 python bench.py --no-secondary --no-serial --no-parity --no-cpu-baseline --no-robust --stream-repeat ${PASSES:-3} > $O/probe_bisect_$M.log 2> $O/probe_bisect_$M.err
    grep "streaming pass\|probe-repair" $O/probe_bisect_$M.err | cut -c1-${CUT:-90} | sed "s/^/mode $M: /"
  done
  ;;
sdma)
  for i in 1 2; do timeout 300 tools/microbench/sdma_pair 3 > $O/sdma_pair_$i.txt 2>&1; done
  cat $O/sdma_pair_1.txt
  ;;
c256)
  tag=${1:-x}; shift || true
  rm -rf $O/c256_stats_$tag
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c256_stats_$tag -- python3 bench.py --workload 256cam128p --contexts 1 --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/c256_bench_1ctx_$tag.log 2>&1
  python3 tools/kernel_stats.py $O/c256_stats_$tag $O/c256_1ctx_${tag}_kernel_stats.csv | head -12
  rm -rf $O/c256_stats_$tag
  ;;
c256_ab)
  # variant libraries through the one-context 256cam128p pass: AB="name ..."
  for V in ${AB:-}; do
    rm -rf $O/c256_stats_$V
    export SE3D_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/tmp/libse3d_$V.so
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c256_stats_$V -- python3 bench.py --workload 256cam128p --contexts 1 --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/c256_bench_1ctx_$V.log 2>&1
    unset SE3D_LIB_OVERRIDE
    python3 tools/kernel_stats.py $O/c256_stats_$V $O/c256_1ctx_${V}_kernel_stats.csv | grep "pairs_rows" | sed "s/^/$V: /"
    rm -rf $O/c256_stats_$V
  done
  ;;
tests)
  # selected test files: TESTS="tests/a.py tests/b.py"
  timeout 1500 python -m pytest ${TESTS:-tests} -m gpu -q -x > $O/pytest_sel.log 2>&1
  tail -5 $O/pytest_sel.log
  ;;
suite)
  timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest_gpu.log 2>&1
  tail -5 $O/pytest_gpu.log
  ;;
bench)
  n=${1:-1}; shift || true
  for i in $(seq 1 $n); do
    timeout 900 python bench.py ${BENCH_FLAGS:-} > $O/bench_default_$i.log 2> $O/bench_default_$i.err
    tail -1 $O/bench_default_$i.log | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('run $i:', d['value'], d['ms_per_step'], 'secondary', d['secondary']['value'], 'streaming', d['value_streaming'], d['streaming'].get('h2d_GB_per_s'), 'max step', max(d['streaming'].get('step_ms_context0')), 'gc', d['streaming'].get('python_gc'), 'probe', d['streaming'].get('copy_stream_probe'), 'robust', d.get('robust', {}).get('value'), 'frac', d['roofline']['frac'])"
  done
  ;;
esac
done
