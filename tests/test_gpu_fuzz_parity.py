"""Randomised GPU-vs-oracle parity sweep inside the suite: fixed seeds — among them 4, 15, 37, whose scenes made the default stage-2 path
diverge from the oracle in round 2, 1, 3, 39 with joints more than 1 mm off in round 2 (up to 26.6 m: ill-conditioned DLT systems, which
round 3 put into an exclusion set), and 104, 203 (a joint with a 90 km covariance, a covariance 1.2 % over the flat tolerance at kappa = 744).

Round 4: NO exclusion set for stage-1 positions and scores — the library sends the joints its fp64 fast path cannot guarantee through a second
pass with the oracle's restatement of the reference's float32 arithmetic (csrc/se3d_exact32.h) — and the library's flag is itself checked against the inputs
(`ill_not_listed`).  tests/fuzz_core.py states what is compared; stage 2 keeps its garbage-person exclusion (tests/parity.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEDS = [4, 15, 37, 39, 0, 1, 2, 3, 104, 203]
TRIALS = 40   # 40 scenes x 3 sequences x 12 frames per seed: ~1 440 frames, ~110 k joints, ~1 430 sequence-steps


VARIANTS = [(s, "simple", False) for s in SEEDS] + [(5, "h36m", False), (6, "simple", True), (7, "h36m", True)]   # the nodes' other configurations
LARGE = (1004, "large", False)


def _launch(seed, pose_method, norm_height, trials):
    env = dict(os.environ)
    if pose_method == "large":
        env["FUZZ_LARGE"] = "1"
    elif pose_method != "simple":
        env["FUZZ_POSE"] = pose_method
    if norm_height:
        env["FUZZ_NORM_HEIGHT"] = "1"
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), str(seed), str(trials)], env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True)


@pytest.fixture(scope="module")
def reports(oracle):   # (the fixture: the oracle library is built before the children import it)
    """Every run of this module as its own process (tools/fuzz_parity.py: tests/fuzz_core.run_trials, the report on stdout, the violations on stderr),
    all of them at once: a run is a single host thread of oracle and numpy work around small launches (13 s each, 180 s one after the other in round 4's
    suite).  A stage-1 index mismatch ends its run with the assertion on stderr."""
    procs = {v: _launch(*v, TRIALS) for v in VARIANTS}
    procs[LARGE] = _launch(*LARGE, 8)
    out = {}
    for v, p in procs.items():
        try:
            so, se = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
        out[v] = (p.returncode, so, se)
    return out


def _report(reports, key):
    rc, so, se = reports[key]
    assert rc == 0, se[-3000:]
    rep = json.loads(so.strip().split("\n")[-1])
    print(json.dumps({k: v for k, v in rep.items() if k != "tolerances"}))
    return rep, [ln for ln in se.split("\n") if ln.startswith("violation:")]


@pytest.mark.parametrize("seed,pose_method,norm_height", VARIANTS)
def test_random_scenes_match_the_oracle_on_every_joint(reports, seed, pose_method, norm_height):
    rep, viol = _report(reports, (seed, pose_method, norm_height))
    assert rep["frames"] == TRIALS * 36 and rep["joints"] > 50_000
    # stage 1: every valid joint (positions < 1 mm, scores bit-equal, covariances inside parity.cov_tolerance); nothing excluded
    assert rep["joints_over_1mm"] == 0 and rep["scores_over_rtol"] == 0 and rep["cov_over_tol"] == 0, viol[:5]
    assert rep["joints_excluded"] == 0
    # the library's conditioning flag covers what the inputs say is ill-conditioned
    assert rep["ill_not_listed"] == 0, rep
    # stage 2, every sequence-step outside the exclusion set
    assert rep["stage2_index_divergences"] == 0 and rep["stage2_numeric_failures"] == 0, viol[:5]
    assert rep["seq_steps_excluded"] <= 0.15 * rep["seq_steps"], rep
    # ... and on EVERY sequence-step, exempt or not: where the published joints differ, the library's final LM objective is not the worse one
    assert rep["objective_worse"] == 0, viol[:5]
    # inside the exclusion set the two sides actually differ on a small part of the steps only (round 6: 0.5 % — measured 0.08 - 0.15 % over the blind
    # batches of round 5 — with one mismatch per run allowed: a run has ~100 - 170 exempt steps; the aggregate bounds are test_exempt_set_budget_*)
    assert rep["excluded_stage2_mismatches"] <= max(1, 0.005 * rep["seq_steps_excluded"]), rep


def test_exempt_set_budget_over_the_suite_runs(reports):
    """Round 6: the budgets of stage 2's exempt set over ALL small-rig runs of the module together (per run the numbers are too small to bound
    below one event): mismatches of any kind <= 0.5 % of the exempt steps, person- / track-count or id divergences <= 0.1 % — with TWO events
    allowed at this sample size (~1 150 exempt steps, where 0.1 % is 1.1 events: the suite's own seeds hold 2, a rate of 0.06 % over all batches
    makes two events a one-in-seven draw); the 0.1 % itself is checked where the sample carries it, over the 100 seeds of tools/fuzz_blind.sh
    (round 6's batch: 6 of 8 365).  Predicted-only differences are counted."""
    reps = [_report(reports, v)[0] for v in VARIANTS]
    exempt = sum(r["seq_steps_excluded"] for r in reps)
    mism = sum(r["excluded_stage2_mismatches"] for r in reps)
    count = sum(r["excluded_count_mismatches"] for r in reps)
    kinds = {}
    for r in reps:
        for k, v in r["excluded_mismatch_kinds"].items():
            kinds[k] = kinds.get(k, 0) + v
    print("exempt steps %d of %d; mismatches inside %d (%s); count / id divergences %d; predicted-only person differences on comparable steps %d"
          % (exempt, sum(r["seq_steps"] for r in reps), mism, kinds, count, sum(r["objective_pred_only"] for r in reps)))
    assert exempt > 500
    assert mism <= max(2, 0.005 * exempt), (mism, exempt, kinds)
    assert count <= max(2, 0.001 * exempt), (count, exempt, kinds)


def test_large_rig_scenes_match_the_oracle_on_every_joint(reports):
    """The same comparison on rigs of 32 / 48 cameras with 12 - 24 persons (FUZZ_LARGE=1: the rows pair kernel with camera-column chunks, the
    per-camera sweep launches, the per-joint triangulation kernel, the per-step stage 2; the scene sizes are module constants of
    tests/fuzz_scenes.py read from the environment); 40 such seeds run in tools/fuzz_sweep.sh."""
    rep, _ = _report(reports, LARGE)
    assert rep["frames"] == 48 and rep["joints"] > 10_000
    assert rep["joints_over_1mm"] == 0 and rep["scores_over_rtol"] == 0 and rep["cov_over_tol"] == 0
    assert rep["joints_excluded"] == 0 and rep["ill_not_listed"] == 0
    assert rep["stage2_index_divergences"] == 0 and rep["stage2_numeric_failures"] == 0
    # the exempt share of this run (one garbage joint among ~400 per frame taints its sequence: 40 of 48 steps in round 4) and what differs inside it
    print("large rigs: %d of %d sequence-steps exempt, %d mismatches inside the exempt set, %d persons with different joints, of them with the worse objective: %d"
          % (rep["seq_steps_excluded"], rep["seq_steps"], rep["excluded_stage2_mismatches"], rep["objective_disagree"], rep["objective_worse"]))
    assert rep["seq_steps_excluded"] <= 44 and rep["objective_worse"] == 0
    assert rep["excluded_stage2_mismatches"] <= max(1, 0.02 * rep["seq_steps_excluded"]), rep   # (5 % until round 5; 40 exempt steps: one event)
