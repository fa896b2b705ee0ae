#!/bin/bash
# GPU box: bench.py --gpus 2 with BOTH ranks on the one GPU of the box through the test-only exchange backend (shard.HostStagedDist: device
# buffers staged through host memory, gloo between the processes; RCCL refuses two ranks on one device).  What tests/test_gpu_bench.py asserts,
# with the JSON lines kept (profiles/r5/bench_two_ranks_one_gpu_*.json).
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29651 bench.py --gpus 2 --exchange-backend gloo-staged \
  --no-cpu-baseline --workload 256cam128p --contexts 1 --seqs 1 --tsteps 4 --steps 2 --warmup 1 --hyp-shard 2>gpurun_out/r5_two_ranks_256.err | grep "^{" > gpurun_out/r5_two_ranks_256.json
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29652 bench.py --gpus 2 --exchange-backend gloo-staged \
  --no-cpu-baseline --workload 16cam6p --seqs 32 --contexts 2 --steps 3 --warmup 1 2>gpurun_out/r5_two_ranks_16.err | grep "^{" > gpurun_out/r5_two_ranks_16.json
python3 - <<'PY'
import json
for f in ("gpurun_out/r5_two_ranks_256.json", "gpurun_out/r5_two_ranks_16.json"):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(d["n_gpus"], round(d["value"]), d["config"]["sharding"], d["collective"]["world_size"], d["collective"]["backend"][:40],
          d.get("hypothesis_shard", {}).get("verified_equal_to_single_rank"), d["exchange_ms_per_step"])
PY
