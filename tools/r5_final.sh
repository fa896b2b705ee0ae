#!/bin/bash
# Round 5's evidence run on the GPU box (gpurun --timeout 3000 -- bash tools/r5_final.sh [parts]): everything profiles/r5/ quotes, written to
# gpurun_out/r5_final/ (+ gpurun_out/prof_r5/ for the rocprofv3 passes) and copied into profiles/r5/ by hand afterwards.
#   parts (default: all): suite bench profiles stream micro stamps      (stamps: build the diagnostic library first, make -C smartedgesensor3dhumanpose_amd/csrc diag)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_final
mkdir -p $O
PARTS=${*:-suite bench profiles stream micro stamps}
for part in $PARTS; do
case $part in
suite)
  timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.log 2>&1
  tail -3 $O/pytest_gpu.log
  ;;
bench)
  # the default line three times (spread of value, secondary and value_streaming on one box)
  for i in 1 2 3; do
    timeout 900 python bench.py > $O/bench_default_$i.log 2> $O/bench_default_$i.err
    tail -1 $O/bench_default_$i.log | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('run $i:', d['value'], d['ms_per_step'], 'secondary', d['secondary']['value'], 'streaming', d['value_streaming'], d['streaming'].get('h2d_GB_per_s'), d['streaming'].get('d2h_GB_per_s'), 'frac', d['roofline']['frac'], 'latency', d['latency_B1']['host_call_ms'])"
  done
  ;;
profiles)
  bash tools/profile_r5.sh serial serial_pmc valumix c256 c256_pmc > $O/profile_r5.log 2>&1
  tail -5 $O/profile_r5.log
  ;;
stream)
  for i in 1 2 3; do STREAM_CONTEXTS="8" bash tools/stream_scan.sh; done > $O/stream_scan_8ctx_x3.txt 2>&1
  STREAM_CONTEXTS="4 12" bash tools/stream_scan.sh > $O/stream_scan_4_12ctx.txt 2>&1
  STREAM_CONTEXTS="8" STREAM_OUT=1 bash tools/stream_scan.sh > $O/stream_scan_8ctx_direct_writes.txt 2>&1
  cat $O/stream_scan_8ctx_x3.txt | cut -c1-200
  ;;
micro)
  /opt/rocm/bin/hipcc -O2 -o /tmp/pcie_bw_multi tools/microbench/pcie_bw_multi.hip && timeout 300 /tmp/pcie_bw_multi > $O/microbench_pcie_bw_multi.txt 2>&1
  cat $O/microbench_pcie_bw_multi.txt
  ;;
stamps)
  timeout 600 python3 tools/rb_stamps.py > $O/rb_stamps.txt 2> $O/rb_stamps.err
  cat $O/rb_stamps.txt
  ;;
esac
done
