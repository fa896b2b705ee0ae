#!/usr/bin/env python3
"""Randomised parity sweep (GPU box) as a tool: the same comparison tests/test_gpu_fuzz_parity.py asserts on, over any seed / trial count,
printing one JSON report per seed (stage-1 index results abort the run when they differ; numeric results and stage-2 results are tallied
inside / outside the input-derived exclusion sets of tests/parity.py).   python tools/fuzz_parity.py [seed] [trials]
FUZZ_MODE=wave|wg|steps selects the stage-2 walk form, FUZZ_POSE=h36m / FUZZ_NORM_HEIGHT=1 the nodes' other configurations."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_py as O      # noqa: E402
from fuzz_core import run_trials       # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t0 = time.time()
rep = run_trials(O, seed, trials, os.environ.get("FUZZ_MODE", "wave"), pose_method=os.environ.get("FUZZ_POSE", "simple"), norm_height=bool(os.environ.get("FUZZ_NORM_HEIGHT")))
rep["seconds"] = round(time.time() - t0, 1)
for v in rep.pop("violations"):
    print("violation:", json.dumps(v), file=sys.stderr)
print(json.dumps(rep))
