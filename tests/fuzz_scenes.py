"""Scene draw of the randomised parity sweep (tests/test_gpu_fuzz_parity.py, tools/fuzz_parity.py): rigs of 16 / 24 cameras, 2-7 persons
0.5-1.2 m apart, up to 15 % missing views, up to 30 % low-confidence joints, up to 1.2 % gross 2-D outliers, 0.5-4 px noise.  One generator
consumed in a fixed order, so (seed, trial) names a scene."""
import numpy as np

from smartedgesensor3dhumanpose_amd import rig as R, scene as S
from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors

T_STEPS, N_SEQ, MAX_HYP = 12, 3, 16


def draw_trial(rng):
    """Returns (rig, n_persons, seqs [N_SEQ][T_STEPS] frames, scene parameters)."""
    ncam = int(rng.choice([16, 16, 24]))
    rig = R.extr_rig_16cam() if ncam == 16 else R.elliptical_rig(24, seed=int(rng.integers(100)))
    rig.set_tensors(*rig_tensors(rig))
    persons = int(rng.integers(2, 8))
    extra = dict(p_miss=float(rng.uniform(0, 0.15)), low_score_frac=float(rng.uniform(0, 0.3)), outlier_frac=float(rng.uniform(0, 0.012)),
                 pixel_noise=float(rng.uniform(0.5, 4.0)), spacing=float(rng.uniform(0.5, 1.2)))
    seqs = [S.generate_sequence(rig, S.SceneConfig(n_persons=persons, n_frames=T_STEPS, seed=int(rng.integers(1 << 30)), **extra))[0] for _ in range(N_SEQ)]
    return rig, persons, seqs, extra
