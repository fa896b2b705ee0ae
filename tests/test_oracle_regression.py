"""CPU: the oracle against its own committed regression vectors (tests/golden/oracle_regression.json): an intended change of the restatement's
arithmetic has to be accompanied by re-generated vectors (tests/golden/make_oracle_regression_golden.py); an unintended one fails here."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def test_oracle_reproduces_its_regression_vectors(oracle):
    import make_oracle_regression_golden as G
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_regression.json")))
    assert [s["name"] for s in want["scenes"]] == [s["name"] for s in G.SCENES]
    n_pub = 0
    for scene, ref in zip(G.SCENES, want["scenes"]):
        got = G.run(scene)
        for k, (a, b) in enumerate(zip(got["frames"], ref["frames"])):
            assert a == b, (scene["name"], k, {key: (a[key], b[key]) for key in a if a[key] != b[key]})
            n_pub += a["fused_n"]
    assert n_pub > 0   # the sequences are long enough for tracks to be published (num_obs > 10)
