#!/bin/bash
# PMC passes on the 256cam128p pair kernel (one context, one step): bash tools/pmc_c256.sh [SE3D_PAIRS mode]
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_c256
mkdir -p $OUT
MODE=${1:-rows}
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $OUT/counters_list.txt | sort -u > $OUT/sq_counters.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rm -rf $OUT/${MODE}_$i
  SE3D_PAIRS=$MODE timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/${MODE}_$i -- python3 bench.py --workload 256cam128p --contexts 1 --stage 1 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${MODE}_bench_$i.log 2>&1
  python3 tools/pmc_sq.py $OUT/${MODE}_$i $OUT/pmc_${MODE}_$i.json "$SET" > /dev/null 2>&1
  python3 - <<PY
import json
d=json.load(open("$OUT/pmc_${MODE}_$i.json"))
for k,v in d["kernels"].items():
    if "pairs" in k: print(k, json.dumps(v))
PY
done
