#!/bin/bash
# ablation of se3d_k_pairs_tiled at 256cam128p (diagnostic builds; numbers are kernel ms per 16-frame step)
for v in "" nostore nocompute; do
  if [ -n "$v" ]; then export SE3D_LIB_OVERRIDE=$PWD/smartedgesensor3dhumanpose_amd/libse3d_exp_$v.so; else unset SE3D_LIB_OVERRIDE; fi
  echo -n "variant=${v:-product} "; timeout 600 python bench.py --workload 256cam128p --steps 1 --warmup 1 --stage 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_per_step'])"
done
