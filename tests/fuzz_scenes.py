"""Scene draw of the randomised parity sweep (tests/test_gpu_fuzz_parity.py, tools/fuzz_parity.py): rigs of 16 / 24 cameras, 2-7 persons
0.5-1.2 m apart, up to 15 % missing views, up to 30 % low-confidence joints, up to 1.2 % gross 2-D outliers, 0.5-4 px noise.  One generator
consumed in a fixed order, so (seed, trial) names a scene."""
import numpy as np

from smartedgesensor3dhumanpose_amd import rig as R, scene as S
from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors

import os

# FUZZ_LARGE=1 (tool runs only): rigs of 32 / 48 cameras with 12 - 24 persons — the large-rig kernels (rows pair kernel with camera-column
# chunks, per-camera sweep launches, per-joint triangulation, per-step stage 2) under the same comparison; one sequence of 6 frames per scene
LARGE = bool(os.environ.get("FUZZ_LARGE"))
T_STEPS, N_SEQ, MAX_HYP, MAX_DET = (6, 1, 128, 32) if LARGE else (12, 3, 16, 12)


def draw_trial(rng):
    """Returns (rig, n_persons, seqs [N_SEQ][T_STEPS] frames, scene parameters)."""
    ncam = int(rng.choice([32, 48])) if LARGE else int(rng.choice([16, 16, 24]))
    rig = R.extr_rig_16cam() if ncam == 16 else R.elliptical_rig(ncam, seed=int(rng.integers(100)))
    rig.set_tensors(*rig_tensors(rig))
    persons = int(rng.integers(12, 25)) if LARGE else int(rng.integers(2, 8))
    extra = S.draw_rough_params(rng)   # the parameter ranges live next to the generator (scene.py; bench.py's `robust` block uses their middle)
    seqs = [S.generate_sequence(rig, S.SceneConfig(n_persons=persons, n_frames=T_STEPS, seed=int(rng.integers(1 << 30)), **extra))[0] for _ in range(N_SEQ)]
    return rig, persons, seqs, extra
