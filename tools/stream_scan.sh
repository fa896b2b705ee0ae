#!/bin/bash
# GPU box: the PCIe-inclusive rate (value_streaming) against the number of streaming contexts (two batches in flight each).
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for nc in ${STREAM_CONTEXTS:-4 8 12}; do
  timeout 600 python3 bench.py --no-secondary --no-serial --no-cpu-baseline --no-parity --stream-contexts $nc --stream-out ${STREAM_OUT:-0} --stream-affinity ${STREAM_AFFINITY:-local} --steps ${STEPS:-12} 2>gpurun_out/stream_scan_$nc.err | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('contexts $nc:', round(d['value_streaming']), json.dumps(d['streaming']))"
done
