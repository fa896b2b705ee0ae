#!/bin/bash
# Runs on the GPU box: every rank's synthetic scenes of an 8-GPU run, one after the other on this GPU, to make sure that no rank's
# seeds overflow the hypothesis capacity (bench.py asserts) or publish more skeletons per frame than the exchange carries
# (n_persons + 2 slots).  bash tools/check_rank_scenes.sh [bench.py args]
for r in 0 1 2 3 4 5 6 7; do
  SE3D_BENCH_SEED_RANK=$r timeout 300 python3 bench.py --steps ${CHECK_STEPS:-20} --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('seed rank $r: %.2f M skel/s, max hypotheses per frame %d (capacity 12), max published per frame %d (exchange slots %d)' % (d['value'] / 1e6, c['max_hypotheses_per_frame'], c['max_published_per_frame'], c['n_persons'] + 2))"
done
