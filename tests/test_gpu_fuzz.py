"""Randomised parity (hypothesis): small FM and MF fits of arbitrary shape against the
oracle, through the C ABI.  Needs an MI355X: ``pytest -m gpu``.  Tolerance as in
test_gpu_parity.py (float64 both sides: 1e-9 of the 1e-5 contract)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st
from scipy.sparse import random as sprandom

from conftest import rel_err
from oracle import cpu_ref

pytestmark = pytest.mark.gpu

TIGHT = 1e-9
SETTINGS = dict(max_examples=200, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)


@settings(**SETTINGS)
@given(n_rows=st.integers(1, 900), n_cols=st.integers(1, 80), density=st.floats(0.0, 0.5),
       k=st.integers(1, 70), frac=st.floats(0.01, 1.0), hot=st.sampled_from([0, -1, 1, 8]),
       dense_cols=st.integers(0, 3), seed=st.integers(0, 10 ** 6), epochs=st.integers(1, 6))
def test_fm_fit_random_shapes(n_rows, n_cols, density, k, frac, hot, dense_cols, seed, epochs):
    import relevance_factorizationmachine_amd as pkg
    rng = np.random.default_rng(seed)

    def log(m):
        X = sprandom(m, n_cols, density=density, format="csr", random_state=rng,
                     data_rvs=lambda s: rng.standard_normal(s)).tolil()
        for c in range(min(dense_cols, n_cols)):
            X[:, c] = rng.standard_normal(m)[:, None]
        X = X.tocsr()
        X.sort_indices()
        return {"features": X, "labels": (rng.random(m) < 0.5).astype(np.int64),
                "pscores": rng.uniform(0.1, 1.0, size=m) ** 0.5}

    train, val = log(n_rows), log(max(1, n_rows // 3))
    batch = max(1, int(round(frac * n_rows)))
    kw = dict(n_epochs=epochs, n_factors=k, lr=1e-4, batch_size=batch, seed=seed % 1000)
    model = pkg.FactorizationMachines(estimator="IPS", n_features=n_cols, **kw)
    model.hot_min_count = hot
    tr, va = model.fit(train, val)
    ref = cpu_ref.fm_fit(train, val, **kw)
    assert rel_err(model.V(), ref["V"]) < TIGHT
    assert rel_err(model.w(), ref["w"]) < TIGHT
    assert abs(model.w0(0) - ref["w0"][0]) < TIGHT * max(1.0, abs(ref["w0"][0]))
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT
    assert rel_err(model.predict(val["features"]),
                   cpu_ref.fm_predict(val["features"], ref["w0"], ref["w"], ref["V"])) < TIGHT


@settings(**SETTINGS)
@given(n_rows=st.integers(1, 700), nu=st.integers(1, 60), ni=st.integers(1, 60), k=st.integers(1, 140),
       frac=st.floats(0.01, 1.0), zipf=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_mf_fit_random_shapes(n_rows, nu, ni, k, frac, zipf, seed):
    import relevance_factorizationmachine_amd as pkg
    rng = np.random.default_rng(seed)

    def log(m):
        items = (rng.zipf(1.3, size=m) - 1) % ni if zipf else rng.integers(0, ni, size=m)
        pairs = np.stack([rng.integers(0, nu, size=m), items], axis=1).astype(np.int64)
        return {"features": pairs, "labels": (rng.random(m) < 0.5).astype(np.int64),
                "pscores": rng.uniform(0.1, 1.0, size=m) ** 0.5}

    train, val = log(n_rows), log(max(1, n_rows // 3))
    batch = max(1, int(round(frac * n_rows)))
    kw = dict(n_epochs=2, n_factors=k, lr=0.01, batch_size=batch, seed=seed % 1000, n_users=nu, n_items=ni,
              reg=0.5)
    model = pkg.LogisticMatrixFactorization(estimator="IPS", **kw)
    tr, va = model.fit(train, val)
    ref = cpu_ref.mf_fit(train, val, **kw)
    for nm in ("P", "Q", "b_u", "b_i"):
        assert rel_err(getattr(model, nm)(), ref[nm]) < TIGHT, nm
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT


@settings(**SETTINGS)
@given(n=st.integers(1, 3000), n_users=st.integers(1, 60), k=st.integers(1, 12), levels=st.integers(1, 40),
       p_pos=st.floats(0.0, 1.0), binary_p=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_device_evaluator_with_ties_equals_numpy_ranking(n, n_users, k, levels, p_pos, binary_p, seed):
    """Scores quantised to a few levels (ties everywhere, as saturated sigmoids give): the device
    value with its flagged users redone on the host must equal the reference's computation --
    per-user ``argsort()[::-1]`` with whatever order NumPy leaves ties in -- so a tie that can
    change the value and is NOT flagged shows up here."""
    from relevance_factorizationmachine_amd import evaluate, runtime
    rt = runtime.Runtime.get()
    rng = np.random.default_rng(seed)
    users = rng.integers(0, n_users, size=n)
    labels = (rng.random(n) < p_pos).astype(np.int64)
    pscore = rng.choice([0.25, 1.0], size=n) if binary_p else rng.uniform(0.1, 1.0, size=n)
    scores = rng.integers(0, levels, size=n) / float(levels)
    frame = {"user": users, "label": labels, "pscore": pscore, "ones_pscore": np.ones(n)}
    fr = evaluate.DeviceValFrame(rt, users, labels, pscore, k)
    d = rt.upload(scores)
    out = rt.empty((2,), d.dtype)
    fr.dcg_into(d, out.data_ptr())
    rt.sync()
    got = fr.resolve(scores, fr.scratch.cpu().numpy())
    if labels.sum() == 0:
        assert np.isnan(got)
        return
    want = cpu_ref.val_dcg(frame, scores, "IPS", k=k)
    assert got == pytest.approx(want, rel=1e-12)


@settings(max_examples=200, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(n_rows=st.integers(1, 700), n_val=st.integers(1, 500), n_cols=st.integers(1, 260),
       density=st.floats(0.0, 0.6), half_k=st.integers(65, 300), frac=st.floats(0.05, 1.0),
       dense_cols=st.integers(0, 4), seed=st.integers(0, 10 ** 6), hot=st.sampled_from([0, -1]))
def test_sliced_loss_forward_random_shapes(n_rows, n_val, n_cols, density, half_k, frac, dense_cols, seed, hot):
    """Even factor counts 130 .. 600 (one, two and four slices; any width of the last slice), logs of any
    density (rows of zero to hundreds of entries: translated rows of 16 / 32 / 64 records and rows read
    from the CSR arrays), any batch: the fit() loop with the sliced loss forward forced on
    (RFM_SLICED_MIN_ROWS=1; the train-loss rows in it, the logarithms once per call) against the oracle --
    parameters and both loss curves -- and the plan's scoring forward against predict()."""
    import os

    import relevance_factorizationmachine_amd as pkg
    rng = np.random.default_rng(seed)
    k = 2 * half_k

    def log(m):
        X = sprandom(m, n_cols, density=density, format="csr", random_state=rng,
                     data_rvs=lambda s: rng.standard_normal(s)).tolil()
        for c in range(min(dense_cols, n_cols)):
            X[:, c] = rng.standard_normal(m)[:, None]
        X = X.tocsr()
        X.sort_indices()
        return {"features": X, "labels": (rng.random(m) < 0.5).astype(np.int64),
                "pscores": rng.uniform(0.1, 1.0, size=m) ** 0.5}

    train, val = log(n_rows), log(n_val)
    batch = max(1, int(round(frac * n_rows)))
    kw = dict(n_epochs=2, n_factors=k, lr=1e-5, batch_size=batch, seed=seed % 1000)
    old = os.environ.get("RFM_SLICED_MIN_ROWS")
    os.environ["RFM_SLICED_MIN_ROWS"] = "1"
    try:
        model = pkg.FactorizationMachines(estimator="IPS", n_features=n_cols, **kw)
        model.hot_min_count = hot
        tr, va = model.fit(train, val)
    finally:
        if old is None:
            del os.environ["RFM_SLICED_MIN_ROWS"]
        else:
            os.environ["RFM_SLICED_MIN_ROWS"] = old
    assert model.plan_info["slices"] in ((1, 2, 4) if train["features"].nnz else (0,))  # (no entries: nothing to slice)
    ref = cpu_ref.fm_fit(train, val, n_features=n_cols, **kw)
    assert rel_err(model.V(), ref["V"]) < TIGHT and rel_err(model.w(), ref["w"]) < TIGHT
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT
