#!/usr/bin/env python3
"""Golden vectors for row M (Munkres): produced by the REFERENCE's own Hungarian.cpp, compiled in place from
/root/reference/skeleton_3d/src/Hungarian.cpp into oracle/_ref/libhungarian_ref.so (oracle/Makefile target `ref`).

Small cases carry the matrix explicitly; large cases carry the numpy recipe (seed/kind/shape) plus a sha256 of the
matrix bytes so RNG drift is detected.  Run in the build container only (the reference does not travel).
"""
import hashlib, json, os, sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle_py as O

OUT = os.path.join(os.path.dirname(__file__), "hungarian_golden.json")


def make_matrix(kind, nr, nc, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((nr, nc))
    if kind == "integer":      # many exact ties
        return rng.integers(0, 6, (nr, nc)).astype(np.float64)
    if kind == "maxcost":      # MAX_COSTS = 1e6 entries as produced by calcCost vetoes (S3D:339-342, 386-389)
        m = rng.random((nr, nc)) * 0.2
        m[rng.random((nr, nc)) < 0.4] = 1e6
        return m
    if kind == "epipolar":     # realistic: one small entry per row/col + large rest, some duplicated columns
        m = 0.05 + rng.random((nr, nc)) * 2.0
        for i in range(min(nr, nc)):
            m[i, (i * 7 + 3) % nc] = rng.random() * 0.04
        if nc > 2:
            m[:, 1] = m[:, 0]
        return m
    if kind == "constant":
        return np.full((nr, nc), 0.5)
    raise ValueError(kind)


def main():
    O.build()
    assert O.ref_hungarian_available(), "reference Hungarian not built"
    cases = []
    shapes_small = [(1, 1), (1, 3), (3, 1), (2, 2), (3, 3), (2, 3), (3, 2), (4, 7), (7, 4), (6, 6), (8, 8), (5, 12), (12, 5), (16, 16)]
    seed = 0
    for kind in ["uniform", "integer", "maxcost", "epipolar", "constant"]:
        for (nr, nc) in shapes_small:
            for rep in range(2):
                m = make_matrix(kind, nr, nc, seed)
                a, c = O.ref_hungarian(m)
                cases.append({"kind": kind, "nr": nr, "nc": nc, "seed": seed, "matrix": m.tolist(), "assignment": a.tolist(), "cost": c})
                seed += 1
    for kind in ["uniform", "integer", "maxcost", "epipolar"]:
        for (nr, nc) in [(32, 32), (24, 40), (40, 24), (64, 64), (128, 128), (128, 96), (96, 128), (150, 128)]:
            m = make_matrix(kind, nr, nc, seed)
            a, c = O.ref_hungarian(m)
            cases.append({"kind": kind, "nr": nr, "nc": nc, "seed": seed, "sha256": hashlib.sha256(np.ascontiguousarray(m).tobytes()).hexdigest(),
                          "assignment": a.tolist(), "cost": c})
            seed += 1
    json.dump({"generator": "tests/golden/make_hungarian_golden.py", "source": "skeleton_3d/src/Hungarian.cpp:60 (compiled reference)",
               "cases": cases}, open(OUT, "w"))
    print("wrote", len(cases), "cases,", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
