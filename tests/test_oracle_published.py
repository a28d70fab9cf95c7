"""The reference's PUBLISHED operating point (k=400 / B=2000 on the KuaiRec-shaped log, k=300 /
B=500 on the Coat-shaped one, 60-494 mini-batch steps; conf/setting/kuairec.yaml:50-59,
conf/setting/coat.yaml:27-36, data/best_params/*/*.json): the CPU oracle against the vectors
the reference produced there (tests/golden/make_golden_published.py).  No GPU needed.

Hundreds of non-linear steps are what BASELINE.json's "1e-5 relative on the learned V" is
about; both sides are float64 and differ in summation order only, so the oracle is held to
1e-10 norm-wise and to the contract tolerance element by element."""
import numpy as np
import pytest

from conftest import assert_elementwise, check_matrix_summary, load_golden, rel_err
from oracle import cpu_ref
from relevance_factorizationmachine_amd import synth
from test_oracle_golden import _log_digest

TIGHT = 1e-10

FM_CASES = [("fm_kuairec_k400_IPS", "kuairec_small", "IPS"), ("fm_kuairec_k400_Naive", "kuairec_small", "Naive"),
            ("fm_coat_k300_IPS", "coat", "IPS"), ("fm_coat_k300_Naive", "coat", "Naive")]
MF_CASES = [("mf_kuairec_k400_IPS", "kuairec_small", "IPS"), ("mf_coat_k300_IPS", "coat", "IPS")]


@pytest.mark.parametrize("case,shape,est", FM_CASES)
def test_fm_published_fit_matches_reference(case, shape, est):
    g = load_golden("published_" + case)
    train, val = synth.make_log(shape, "FM", est, seed=0)
    assert _log_digest(train, val) == str(g["input_digest"]), "synthetic inputs drifted"
    out = cpu_ref.fm_fit(train, val, n_epochs=int(g["n_epochs"]), n_factors=int(g["n_factors"]),
                         lr=float(g["lr"]), batch_size=int(g["batch_size"]), seed=int(g["seed"]), form="closed")
    check_matrix_summary(g, "V", out["V"], TIGHT, case)
    for name in ("w", "w0", "train_loss", "val_loss"):
        assert rel_err(out[name], g[name]) < TIGHT, (case, name)
        assert_elementwise(out[name], g[name], what=f"{case} {name}")
    pred = cpu_ref.fm_predict(val["features"], out["w0"], out["w"], out["V"])
    assert rel_err(pred, g["pred_val"]) < TIGHT
    assert_elementwise(pred, g["pred_val"], what=f"{case} predict(val)")


@pytest.mark.parametrize("case,shape,est", MF_CASES)
def test_mf_published_fit_matches_reference(case, shape, est):
    g = load_golden("published_" + case)
    sh = synth.SHAPES[shape]
    train, val = synth.make_log(sh, "MF", est, seed=0)
    assert _log_digest(train, val) == str(g["input_digest"]), "synthetic inputs drifted"
    out = cpu_ref.mf_fit(train, val, n_epochs=int(g["n_epochs"]), n_factors=int(g["n_factors"]),
                         lr=float(g["lr"]), batch_size=int(g["batch_size"]), seed=int(g["seed"]),
                         n_users=sh.n_users, n_items=sh.n_items, reg=float(g["reg"]))
    check_matrix_summary(g, "P", out["P"], TIGHT, case)
    check_matrix_summary(g, "Q", out["Q"], TIGHT, case)
    assert out["b"] == float(g["b"])
    for name in ("b_u", "b_i", "train_loss", "val_loss"):
        assert rel_err(out[name], g[name]) < TIGHT, (case, name)
        assert_elementwise(out[name], g[name], what=f"{case} {name}")
    pred = cpu_ref.mf_predict(val["features"], out["P"], out["Q"], out["b_u"], out["b_i"], out["b"])
    assert rel_err(pred, g["pred_val"]) < TIGHT
    assert_elementwise(pred, g["pred_val"], what=f"{case} predict(val)")
