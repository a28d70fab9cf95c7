"""Inputs of the driver-flow fixture (tests/golden/driver_flow.npz; made by
tests/golden/make_golden_flow.py from the reference's own classes): the same frames and
hyper-parameters, built without the reference."""
import numpy as np

from relevance_factorizationmachine_amd import synth

SHAPE = "kuairec_small"
TOP_K = [1, 3, 5, 7, 9]
LR = {"FM": {"IPS": 1e-4, "Naive": 3e-4}, "MF": {"IPS": 0.01, "Naive": 0.03}}
CASES = [(m, e) for m in ("FM", "MF") for e in ("IPS", "Naive")]


def frames(seed: int):
    """(interaction frame, {"FM": csr, "MF": pairs}) of an evaluation split: first occurrences
    of every (user, item) pair (repeats have identical features, hence tied scores)."""
    _, fm = synth.make_log(SHAPE, "FM", "IPS", seed=seed)
    _, mf = synth.make_log(SHAPE, "MF", "IPS", seed=seed)
    keep = synth.first_occurrences(mf["features"])
    frame = synth.interaction_frame({k: v[keep] for k, v in mf.items()}, mf["features"][keep])
    return frame, {"FM": fm["features"][keep], "MF": mf["features"][keep]}


def check_metric_columns(g, base, results, rtol=1e-10):
    """``results`` = {metric: [value per K]} of one model against the fixture's metric.csv columns."""
    for metric_name, values in results.items():
        want = g[f"metric_{base}_{metric_name}@K"]
        np.testing.assert_allclose(np.asarray(values, dtype=np.float64), want, rtol=rtol, atol=0, equal_nan=True,
                                   err_msg=f"{base} {metric_name}@K")
    assert {f"{base}_{m}@K" for m in results} == {c for c in map(str, g["metric_columns"]) if c.startswith(base + "_")}
