#!/bin/bash
# Runs on the GPU box: every rank's synthetic scenes of an 8-GPU run, one after the other on this GPU and under the launcher (one rank, so
# that the exchange path with its assertions runs), to make sure that no rank's seeds overflow the hypothesis capacity, publish more
# skeletons than the exchange blocks carry (dense blocks: n_persons + 0.25 persons per frame; slot blocks at 256cam128p: max_hyp) or trip
# any other assertion of bench.py.  bash tools/check_rank_scenes.sh [16cam6p|256cam128p ...]
WL="${@:-16cam6p 256cam128p}"
for wl in $WL; do
for r in 0 1 2 3 4 5 6 7; do
  st=4; [ $wl = 256cam128p ] && st=2
  timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + r)) \
      bench.py --gpus 1 --workload $wl --steps $st --warmup 1 --no-cpu-baseline --seed-rank $r 2>/tmp/check_rank.err | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); c = d['config']
    print('$wl seed rank $r: %.3g skel/s, max hypotheses per frame %d, max published per frame %d, exchange %s, verified: %s' % (d['value'], c['max_hypotheses_per_frame'], c['max_published_per_frame'], (c.get('exchange') or {}).get('form'), c['sharding'][-6:]))
except Exception as e:
    print('$wl seed rank $r: FAILED', e); print(open('/tmp/check_rank.err').read()[-1500:])"
done
done
