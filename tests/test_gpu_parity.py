"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the
golden vectors the reference produced.  Needs an MI355X: ``pytest -m gpu``.

Tolerance: BASELINE.json asks for <= 1e-5 relative on learned parameters and
scores.  Everything is float64 on both sides and differs only in summation
order, so the tests hold the path to CONTRACT/1e4 = 1e-9 to catch regressions
early; ``rel_err`` is max|a-b| / max|b|.
"""
import ctypes as C

import numpy as np
import pytest
from scipy.sparse import csr_matrix, random as sprandom, vstack

from conftest import assert_elementwise, load_golden, rel_err
from oracle import cpu_ref
from relevance_factorizationmachine_amd import synth

pytestmark = pytest.mark.gpu

CONTRACT = 1e-5
TIGHT = 1e-9


@pytest.fixture(scope="module")
def rfm():
    import relevance_factorizationmachine_amd as pkg
    from relevance_factorizationmachine_amd import _lib, runtime
    rt = runtime.Runtime.get()
    return pkg, _lib, runtime, rt


def _fm(pkg, **kw):
    base = dict(estimator="IPS", n_epochs=1, n_factors=8, lr=1e-3, batch_size=1, seed=12345)
    base.update(kw)
    return pkg.FactorizationMachines(**base)


# --------------------------------------------------------------------------
# golden end-to-end fits (configs 1 and 2 of BASELINE.json)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("est", ["IPS", "Naive"])
@pytest.mark.parametrize("fixture,shape", [("fm_coat_k8", "coat"), ("fm_kuairec_small_k16", "kuairec_small")])
def test_fm_fit_matches_reference_golden(rfm, fixture, shape, est):
    pkg = rfm[0]
    g = load_golden(fixture)
    sh = synth.SHAPES[shape]
    train, val = synth.make_log(sh, "FM", est, seed=0)
    model = pkg.FactorizationMachines(
        estimator=est, n_epochs=int(g["n_epochs"]), n_factors=sh.n_factors,
        n_features=train["features"].shape[1], lr=float(g[f"{est}_lr"]),
        batch_size=sh.batch_size, seed=int(g["seed"]))
    tr, va = model.fit(train, val)
    assert isinstance(tr, list) and isinstance(va, list) and len(tr) == len(va) == int(g["n_epochs"])
    assert rel_err(model.V(), g[f"{est}_V"]) < TIGHT
    assert rel_err(model.w(), g[f"{est}_w"]) < TIGHT
    assert rel_err(model.w0(), g[f"{est}_w0"]) < TIGHT
    assert abs(model.w0(0) - g[f"{est}_w0"][0]) <= TIGHT * max(1.0, abs(g[f"{est}_w0"][0]))
    assert rel_err(tr, g[f"{est}_train_loss"]) < TIGHT
    assert rel_err(va, g[f"{est}_val_loss"]) < TIGHT
    pred = model.predict(X=val["features"])
    assert pred.dtype == np.float64 and pred.ndim == 1
    assert rel_err(pred, g[f"{est}_pred_val"]) < TIGHT
    assert rel_err(pred, g[f"{est}_pred_val"]) < CONTRACT
    # ... and the contract tolerance element by element (|a-b| <= 1e-5 |b| + 1e-12)
    for got, name in ((model.V(), "V"), (model.w(), "w"), (model.w0(), "w0"), (tr, "train_loss"),
                      (va, "val_loss"), (pred, "pred_val")):
        assert_elementwise(got, g[f"{est}_{name}"], what=f"{fixture} {est} {name}")
    np.testing.assert_array_equal(model.predict(val["features"]), pred)  # positional call too


@pytest.mark.parametrize("est", ["IPS", "Naive"])
def test_mf_fit_matches_reference_golden(rfm, est):
    pkg = rfm[0]
    g = load_golden("mf_small")
    sh = synth.SHAPES["kuairec_small"]
    train, val = synth.make_log(sh, "MF", est, seed=0)
    model = pkg.LogisticMatrixFactorization(
        estimator=est, n_epochs=3, n_factors=16, n_users=sh.n_users, n_items=sh.n_items,
        lr=0.01, reg=0.5, batch_size=2000, seed=12345)
    with pytest.raises(AttributeError):
        model.predict(val["features"])  # the global bias exists only after fit (src/mf.py:84)
    tr, va = model.fit(train, val)
    for nm in ("P", "Q", "b_u", "b_i"):
        assert rel_err(getattr(model, nm)(), g[f"{est}_{nm}"]) < TIGHT, nm
    assert model.b == float(g[f"{est}_b"])
    assert rel_err(tr, g[f"{est}_train_loss"]) < TIGHT
    assert rel_err(va, g[f"{est}_val_loss"]) < TIGHT
    assert rel_err(model.predict(val["features"]), g[f"{est}_pred_val"]) < TIGHT
    for got, name in ((model.P(), "P"), (model.Q(), "Q"), (model.b_u(), "b_u"), (model.b_i(), "b_i"),
                      (tr, "train_loss"), (va, "val_loss"), (model.predict(val["features"]), "pred_val")):
        assert_elementwise(got, g[f"{est}_{name}"], what=f"mf_small {est} {name}")


def test_fit_with_ids_sampled_in_chunks(rfm):
    """The row-id lists arrive in chunks of iterations (sampled while the GPU works on the
    chunk before): three iterations per chunk here, same result as the reference's fit."""
    pkg, _lib, runtime, rt = rfm
    old = runtime.BatchIdStream.CHUNK_IDS
    rt.clear_caches()  # (ids resident from an earlier fit would bypass the chunks)
    try:
        for fixture, shape, cls in (("fm_coat_k8", "coat", "FM"), ("mf_small", "kuairec_small", "MF")):
            g = load_golden(fixture)
            sh = synth.SHAPES[shape]
            train, val = synth.make_log(sh, cls, "IPS", seed=0)
            runtime.BatchIdStream.CHUNK_IDS = 3 * sh.batch_size
            if cls == "FM":
                model = pkg.FactorizationMachines(
                    estimator="IPS", n_epochs=int(g["n_epochs"]), n_factors=sh.n_factors,
                    n_features=train["features"].shape[1], lr=float(g["IPS_lr"]), batch_size=sh.batch_size,
                    seed=int(g["seed"]))
                names = ("V", "w")
            else:
                runtime.BatchIdStream.CHUNK_IDS = 2 * sh.batch_size  # three iterations: chunks of 2 + 1
                model = pkg.LogisticMatrixFactorization(
                    estimator="IPS", n_epochs=3, n_factors=16, n_users=sh.n_users, n_items=sh.n_items,
                    lr=0.01, reg=0.5, batch_size=2000, seed=12345)
                names = ("P", "Q", "b_u", "b_i")
            tr, va = model.fit(train, val)
            for nm in names:
                assert rel_err(getattr(model, nm)(), g[f"IPS_{nm}"]) < TIGHT, (fixture, nm)
            assert rel_err(tr, g["IPS_train_loss"]) < TIGHT and rel_err(va, g["IPS_val_loss"]) < TIGHT
    finally:
        runtime.BatchIdStream.CHUNK_IDS = old


def test_fit_remembers_the_split_and_notices_edits(rfm):
    """A second fit on the same split reuses the device copies of the log, the labels and the
    row ids (nothing sampled or uploaded) and gives the same result; labels edited in place
    are uploaded again; `remember_splits = False` uploads every time.  All against the oracle."""
    pkg, _lib, runtime, rt = rfm
    rng = np.random.default_rng(99)
    train = _random_log(rng, 3000, 70, 0.1, 1)
    val = _random_log(rng, 200, 70, 0.1, 1)
    kw = dict(n_factors=6, n_features=70, lr=1e-3, batch_size=512, n_epochs=40, seed=3)
    rt.clear_caches()

    def check(model, tr, va, data):
        ref = cpu_ref.fm_fit(data, val, n_epochs=40, n_factors=6, lr=1e-3, batch_size=512, seed=3)
        assert rel_err(model.V(), ref["V"]) < TIGHT and rel_err(model.w(), ref["w"]) < TIGHT
        assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT

    m1 = _fm(pkg, **kw)
    check(m1, *m1.fit(train, val), train)
    assert runtime.ID_CACHE.get_device(rt.device, 3000, 512, 40) is not None
    dev_log = rt.log_cache().get(train["features"])
    m2 = _fm(pkg, **kw)
    check(m2, *m2.fit(train, val), train)
    assert rt.log_cache().get(train["features"]) is dev_log  # the same device copy served both
    assert rel_err(m1.V(), m2.V()) < 1e-12  # (hot-column sums are LDS atomics: last bits may differ)
    # labels flipped in place (same array object): the stale device copy must not be used
    train["labels"][:] = 1 - train["labels"]
    m3 = _fm(pkg, **kw)
    check(m3, *m3.fit(train, val), train)
    # features scaled in place
    train["features"].data *= 0.5
    m4 = _fm(pkg, **kw)
    check(m4, *m4.fit(train, val), train)
    # ONE element edited in place, in each of the arrays a fit reads (the caches hash every
    # byte: no edit is too small to be seen)
    for edit in ("label", "pscore", "value", "val_label"):
        before = _fm(pkg, **kw)
        before.fit(train, val)
        if edit == "label":
            train["labels"][1501] = 1 - train["labels"][1501]
        elif edit == "pscore":
            train["pscores"][777] *= 0.5
        elif edit == "value":
            train["features"].data[train["features"].nnz // 2 + 1] += 3.0
        else:
            val["labels"][101] = 1 - val["labels"][101]
        after = _fm(pkg, **kw)
        tr_a, va_a = after.fit(train, val)
        check(after, tr_a, va_a, train)
        if edit != "val_label":
            assert not np.array_equal(after.V(), before.V()), edit  # the edit changed the result
    # and with the caches switched off
    old = runtime.Runtime.remember_splits
    try:
        runtime.Runtime.remember_splits = False
        m5 = _fm(pkg, **kw)
        check(m5, *m5.fit(train, val), train)
    finally:
        runtime.Runtime.remember_splits = old


def test_dcg_parity_on_gpu_scores(rfm):
    """DCG@5 of the GPU's validation scores equals the reference's (fixture G7)."""
    pkg = rfm[0]
    g, gd = load_golden("fm_kuairec_small_k16"), load_golden("val_dcg")
    sh = synth.SHAPES["kuairec_small"]
    for est in ("IPS", "Naive"):
        train, val = synth.make_log(sh, "FM", est, seed=0)
        _, val_mf = synth.make_log(sh, "MF", est, seed=0)
        frame = synth.interaction_frame(val_mf, val_mf["features"])

        class Hook:  # the ValEvaluator contract (utils/evaluate.py:160-207)
            features = {"FM": val["features"]}

            def evaluate(self, y_scores, estimator):
                return cpu_ref.val_dcg(frame, y_scores, estimator, k=5)

        model = pkg.FactorizationMachines(
            estimator=est, n_epochs=int(g["n_epochs"]), n_factors=16, n_features=train["features"].shape[1],
            lr=float(g[f"{est}_lr"]), batch_size=2000, seed=12345, evaluator=Hook())
        assert model.model_name == "FM" and model.val_metrics == []
        model.fit(train, val)
        assert len(model.val_metrics) == int(g["n_epochs"])
        assert model.val_metrics[-1] == pytest.approx(float(gd[f"g2_val_dcg_{est}"]), rel=1e-9)


# --------------------------------------------------------------------------
# known-answer step (fixture G3: k=3, odd factor count, non-unit values)
# --------------------------------------------------------------------------
def test_fm_one_step_known_answer(rfm):
    pkg, _lib, runtime, rt = rfm
    g = load_golden("fm_one_step_tiny")
    X = csr_matrix(g["dense"])
    train = {"features": X, "labels": g["y"], "pscores": g["p"]}
    model = _fm(pkg, n_factors=3, n_features=5, lr=float(g["lr"]), batch_size=6, seed=int(g["seed"]))
    np.testing.assert_array_equal(model.V(), g["V_init"])
    np.testing.assert_array_equal(model.w(), g["w_init"])
    # gradients through rfm_fm_grad
    from relevance_factorizationmachine_amd.fm import FmPlan
    dev = runtime.DeviceCSR(rt, X)
    plan = FmPlan(rt, dev, g["y"], g["p"], 3, 6)
    ids = rt.upload(cpu_ref.batch_ids(6, 6, 0).astype(np.int32))
    y, p = rt.upload(g["y"], dtype=np.float64), rt.upload(g["p"], dtype=np.float64)
    grad = rt.empty((5 * 3 + 5 + 1,), y.dtype)
    _lib.check(rt.lib.rfm_fm_grad(rt.ctx, plan.handle, dev.indptr.data_ptr(), dev.indices.data_ptr(),
                                  dev.values.data_ptr(), y.data_ptr(), p.data_ptr(), ids.data_ptr(), 6,
                                  model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr(),
                                  grad.data_ptr()))
    rt.sync()
    gh = grad.cpu().numpy()
    assert rel_err(gh[:15].reshape(5, 3), g["G_V"]) < 1e-8  # fixture gradients are (before-after)/lr
    assert rel_err(gh[15:20], g["g_w"]) < 1e-8
    assert rel_err(gh[20], g["g_w0"]) < 1e-8
    plan.close()
    tr, va = model.fit(train, train)
    assert rel_err(model.V(), g["V_after"]) < TIGHT
    assert rel_err(model.w(), g["w_after"]) < TIGHT
    assert rel_err(model.w0(), g["w0_after"]) < TIGHT
    assert rel_err(tr, g["train_loss"]) < TIGHT and rel_err(va, g["val_loss"]) < TIGHT


# --------------------------------------------------------------------------
# edge cases: ragged / empty rows, odd and large k, full batch, untouched columns
# --------------------------------------------------------------------------
def _random_log(rng, n_rows, n_cols, density, dense_cols=0):
    X = sprandom(n_rows, n_cols, density=density, format="csr", random_state=rng,
                 data_rvs=lambda s: rng.standard_normal(s)).tolil()
    for c in range(dense_cols):  # a few columns present in every row (long column lists)
        X[:, c] = rng.standard_normal(n_rows)[:, None]
    X = X.tocsr()
    X.sort_indices()
    y = (rng.random(n_rows) < 0.5).astype(np.int64)
    p = rng.uniform(0.1, 1.0, size=n_rows) ** 0.5
    return {"features": X, "labels": y, "pscores": p}


@pytest.mark.parametrize("k", [1, 2, 3, 7, 8, 16, 30, 32, 33, 64, 100, 128, 129, 300, 400])
def test_fm_forward_all_factor_counts(rfm, k):
    pkg = rfm[0]
    rng = np.random.default_rng(k)
    log = _random_log(rng, 257, 90, 0.08)
    log["features"][5, :] = 0  # an explicitly emptied row
    log["features"].eliminate_zeros()
    model = _fm(pkg, n_factors=k, n_features=90)
    w0, w, V = cpu_ref.fm_init(12345, 90, k)
    assert rel_err(model.predict(log["features"]), cpu_ref.fm_predict(log["features"], w0, w, V)) < TIGHT


@pytest.mark.parametrize("hot", [0, -1, -2, 4])
@pytest.mark.parametrize("k,batch,dense_cols", [(3, 64, 0), (8, 500, 2), (32, 6000, 3), (64, 999, 1), (300, 128, 1)])
def test_fm_fit_ragged_logs(rfm, k, batch, dense_cols, hot):
    """Variable nnz per row, empty rows, columns nobody touches, column lists
    longer than one chunk (dense columns x 6000 rows), batch == n_rows; with the
    on-chip hot-column sums at their default threshold, off, and very eager."""
    pkg = rfm[0]
    rng = np.random.default_rng(100 + k)
    n_rows = max(batch, 6000 if batch == 6000 else 1500)
    train = _random_log(rng, n_rows, 140, 0.05, dense_cols)
    val = _random_log(rng, 333, 140, 0.05, dense_cols)
    lr = 1e-4
    model = _fm(pkg, n_factors=k, n_features=140, lr=lr, batch_size=batch, n_epochs=4, seed=3)
    model.hot_min_count = hot
    tr, va = model.fit(train, val)
    ref = cpu_ref.fm_fit(train, val, n_epochs=4, n_factors=k, lr=lr, batch_size=batch, seed=3)
    assert rel_err(model.V(), ref["V"]) < TIGHT
    assert rel_err(model.w(), ref["w"]) < TIGHT
    assert rel_err(model.w0(), ref["w0"]) < TIGHT
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT


@pytest.mark.parametrize("hot", [0, -1, -2])
@pytest.mark.parametrize("k,density,dense_cols", [(3, 0.04, 1), (32, 0.25, 4), (33, 0.05, 0), (96, 0.05, 2),
                                                  (200, 0.03, 1)])
def test_fm_fit_full_chip_batches(rfm, k, density, dense_cols, hot):
    """Batches large enough for the forward's many-rows-in-flight shape (one 1024-thread
    workgroup per CU): odd factor counts, rows longer than one lane group's round of
    entries (density 0.25 x 120 columns), rows of several chunks per lane (k = 200), with
    and without the on-chip hot sums.  Two iterations against the oracle."""
    pkg = rfm[0]
    rng = np.random.default_rng(7 * k + hot)
    n_rows, batch, n_cols = 45_000, 40_000, 120
    train = _random_log(rng, n_rows, n_cols, density, dense_cols)
    val = _random_log(rng, 400, n_cols, density, dense_cols)
    lr = 2e-6
    model = _fm(pkg, n_factors=k, n_features=n_cols, lr=lr, batch_size=batch, n_epochs=2, seed=5)
    model.hot_min_count = hot
    tr, va = model.fit(train, val)
    ref = cpu_ref.fm_fit(train, val, n_epochs=2, n_factors=k, lr=lr, batch_size=batch, seed=5)
    assert rel_err(model.V(), ref["V"]) < TIGHT
    assert rel_err(model.w(), ref["w"]) < TIGHT
    assert rel_err(model.w0(), ref["w0"]) < TIGHT
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT


def _bounded_log(rng, n_rows, n_cols, max_len, dense_cols):
    """Rows of 0..max_len entries (a few empty), `dense_cols` columns in (almost) every row."""
    lens = rng.integers(0, max_len + 1, size=n_rows)
    lens[rng.integers(0, n_rows, size=5)] = 0
    lens[rng.integers(0, n_rows, size=5)] = max_len
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cols = np.empty(indptr[-1], dtype=np.int32)
    for r in range(n_rows):
        m = lens[r]
        if m:
            d = min(dense_cols, m)
            rest = rng.choice(np.arange(dense_cols, n_cols), size=m - d, replace=False) if m > d else []
            cols[indptr[r]: indptr[r + 1]] = np.sort(np.concatenate([np.arange(d), rest]).astype(np.int32))
    X = csr_matrix((rng.standard_normal(indptr[-1]), cols, indptr), shape=(n_rows, n_cols))
    y = (rng.random(n_rows) < 0.5).astype(np.int64)
    p = rng.uniform(0.1, 1.0, size=n_rows) ** 0.5
    return {"features": X, "labels": y, "pscores": p}


@pytest.mark.parametrize("hot", [0, -1, -2])
@pytest.mark.parametrize("k,max_len,n_rows", [(4, 4, 140_000), (8, 4, 72_000), (16, 8, 72_000), (30, 16, 40_000),
                                              (64, 20, 24_000), (97, 33, 12_000), (128, 64, 12_000),
                                              (200, 40, 12_000), (7, 5, 72_000), (33, 30, 12_000)])
def test_fm_fit_padded_row_blocks(rfm, k, max_len, n_rows, hot):
    """Logs whose longest row fits one round of a lane group, at batches that take the
    many-rows shape: the plan keeps padded row blocks and the forward reads them (every
    lanes-per-row count, one and several factor chunks per lane, empty and full rows, a
    last partial trip).  Two iterations against the oracle."""
    pkg = rfm[0]
    rng = np.random.default_rng(31 * k + hot)
    n_cols = 150
    batch = n_rows - 1234
    train = _bounded_log(rng, n_rows, n_cols, max_len, 2)
    val = _bounded_log(rng, 300, n_cols, max_len, 2)
    lr = 2e-6
    model = _fm(pkg, n_factors=k, n_features=n_cols, lr=lr, batch_size=batch, n_epochs=2, seed=5)
    model.hot_min_count = hot
    tr, va = model.fit(train, val)
    assert model.plan_info["row_blocks"] == 1 and model.plan_info["longest_row"] == max_len
    assert model.plan_info["row_block_bytes"] == 16 * model.plan_info["lanes_per_row"]
    ref = cpu_ref.fm_fit(train, val, n_epochs=2, n_factors=k, lr=lr, batch_size=batch, seed=5)
    assert rel_err(model.V(), ref["V"]) < TIGHT
    assert rel_err(model.w(), ref["w"]) < TIGHT
    assert rel_err(model.w0(), ref["w0"]) < TIGHT
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT


def test_non_canonical_csr_inputs(rfm):
    """CSR as SciPy allows it: duplicate column entries inside a row (SciPy sums them
    before squaring -- X.power(2) de-duplicates -- and so must we), unsorted indices,
    explicitly stored zeros, int64 index arrays.  The caller's matrix is not modified."""
    pkg = rfm[0]
    rng = np.random.default_rng(11)
    n_rows, n_cols, z = 700, 60, 9
    indptr = np.arange(0, n_rows * z + 1, z, dtype=np.int64)
    indices = rng.integers(0, n_cols, size=n_rows * z).astype(np.int64)  # duplicates, unsorted
    data = rng.standard_normal(n_rows * z)
    data[rng.integers(0, data.size, size=300)] = 0.0  # stored zeros
    def log(m):
        X = csr_matrix((data[: m * z].copy(), indices[: m * z].copy(), indptr[: m + 1].copy()), shape=(m, n_cols))
        assert not X.has_canonical_format
        return {"features": X, "labels": (rng.random(m) < 0.5).astype(np.int64),
                "pscores": rng.uniform(0.1, 1.0, size=m) ** 0.5}
    train, val = log(700), log(200)
    nnz_before = train["features"].nnz
    kw = dict(n_epochs=5, n_factors=6, lr=1e-3, batch_size=256, seed=5)
    for hot in (0, -1, 2):
        model = _fm(pkg, n_features=n_cols, **kw)
        model.hot_min_count = hot
        tr, va = model.fit(train, val)
        ref = cpu_ref.fm_fit(train, val, **kw)
        assert rel_err(model.V(), ref["V"]) < TIGHT and rel_err(model.w(), ref["w"]) < TIGHT
        assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT
        assert rel_err(model.predict(val["features"]),
                       cpu_ref.fm_predict(val["features"].copy(), ref["w0"], ref["w"], ref["V"])) < TIGHT
    # the oracle, like the reference, de-duplicates the matrices it is handed in place;
    # this package must leave the caller's matrices as given
    t2, v2 = log(700), log(200)
    model = _fm(pkg, n_features=n_cols, **kw)
    model.fit(t2, v2)
    model.predict(v2["features"])
    for X in (t2["features"], v2["features"]):
        assert X.nnz in (nnz_before, 200 * z) and not X.has_canonical_format


def test_saturated_logits_and_empty_inputs(rfm):
    """_sigmoid clips at +-700 (fixture G9); rows without entries score sigmoid(w0)."""
    pkg = rfm[0]
    s = load_golden("sigmoid_edges")
    model = _fm(pkg, n_factors=2, n_features=3)
    empty_rows = csr_matrix((4, 3))
    for x, want in zip(s["x"], s["y"]):
        model.w0.set(np.array([x]))
        got = model.predict(empty_rows)
        assert got.shape == (4,)
        assert np.all(np.abs(got - want) <= 1e-14 * want + 1e-320), (x, got[0], want)
    assert model.predict(csr_matrix((0, 3))).shape == (0,)


def test_logloss_cases(rfm):
    pkg = rfm[0]
    g = load_golden("logloss_cases")
    model = _fm(pkg, n_factors=2, n_features=3)
    assert model._cross_entropy_loss(g["y"], g["scores"], g["pscores"]) == pytest.approx(float(g["loss"]), rel=1e-13)
    assert model._cross_entropy_loss(g["y"], g["scores"], np.ones(8)) == pytest.approx(float(g["loss_naive"]), rel=1e-13)
    assert model._cross_entropy_loss(g["y"][:3], g["scores"][:3], g["pscores"][:3]) == pytest.approx(
        float(g["loss_first3"]), rel=1e-13)


def test_batch_larger_than_log_raises(rfm):
    pkg = rfm[0]
    train, val = synth.make_log("coat", "FM", "IPS", seed=0)
    model = _fm(pkg, n_factors=8, n_features=train["features"].shape[1], batch_size=10 ** 6)
    with pytest.raises(ValueError, match="Cannot sample"):
        model.fit(train, val)


@pytest.mark.parametrize("k,batch", [(1, 50), (5, 700), (16, 2000), (128, 512), (300, 64)])
def test_mf_fit_vs_oracle(rfm, k, batch):
    pkg = rfm[0]
    rng = np.random.default_rng(k)
    nu, ni, n = 300, 40, 3000
    def log(m):
        pairs = np.stack([rng.integers(0, nu, size=m), (rng.zipf(1.3, size=m) - 1) % ni], axis=1).astype(np.int64)
        return {"features": pairs, "labels": (rng.random(m) < 0.5).astype(np.int64),
                "pscores": rng.uniform(0.1, 1.0, size=m) ** 0.5}
    train, val = log(n), log(500)
    kw = dict(n_epochs=2, n_factors=k, lr=0.02, batch_size=batch, seed=9, n_users=nu, n_items=ni, reg=0.5)
    model = pkg.LogisticMatrixFactorization(estimator="IPS", **kw)
    tr, va = model.fit(train, val)
    ref = cpu_ref.mf_fit(train, val, **kw)
    for nm in ("P", "Q", "b_u", "b_i"):
        assert rel_err(getattr(model, nm)(), ref[nm]) < TIGHT, nm
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT


@pytest.mark.parametrize("k,nu,ni,batch", [(4, 5, 3, 600), (16, 3, 1, 300), (128, 7, 2, 400), (33, 2000, 1, 900),
                                           (200, 50, 2, 300), (16, 1, 500, 700)])
def test_mf_long_chains_vs_oracle(rfm, k, nu, ni, batch):
    """Few users and/or items: every level holds a handful of examples, user rows are
    rewritten within a few levels (so they cannot be read ahead) and the chain of levels is
    as long as the batch."""
    pkg = rfm[0]
    rng = np.random.default_rng(k + nu)
    def log(m):
        pairs = np.stack([rng.integers(0, nu, size=m), rng.integers(0, ni, size=m)], axis=1).astype(np.int64)
        return {"features": pairs, "labels": (rng.random(m) < 0.5).astype(np.int64),
                "pscores": rng.uniform(0.1, 1.0, size=m) ** 0.5}
    train, val = log(1000), log(200)
    kw = dict(n_epochs=2, n_factors=k, lr=0.005, batch_size=batch, seed=3, n_users=nu, n_items=ni, reg=0.5)
    model = pkg.LogisticMatrixFactorization(estimator="IPS", **kw)
    tr, va = model.fit(train, val)
    ref = cpu_ref.mf_fit(train, val, **kw)
    for nm in ("P", "Q", "b_u", "b_i"):
        assert rel_err(getattr(model, nm)(), ref[nm]) < TIGHT, nm
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT


def test_mf_hogwild_is_a_labelled_non_parity_mode(rfm):
    """HOGWILD updates race on shared users/items (updates of a popular item are
    lost), so it learns more slowly than the sequential result and its parameters
    differ: it is a throughput mode, never the graded path."""
    pkg = rfm[0]
    sh = synth.SHAPES["kuairec_small"]
    train, val = synth.make_log(sh, "MF", "IPS", seed=0)
    kw = dict(estimator="IPS", n_epochs=20, n_factors=16, lr=0.01, batch_size=2000, seed=12345,
              n_users=sh.n_users, n_items=sh.n_items, reg=0.5)
    exact = pkg.LogisticMatrixFactorization(**kw)
    tr_e, va_e = exact.fit(train, val)
    hog = pkg.LogisticMatrixFactorization(**kw)
    hog.hogwild = True
    tr_h, va_h = hog.fit(train, val)
    assert np.all(np.isfinite(va_h)) and va_h[-1] < va_h[0]
    assert va_e[-1] < va_e[0]
    assert rel_err(hog.P(), exact.P()) > 1e-9  # not the reference's result: documented


# --------------------------------------------------------------------------
# full-size properties (config 3: 7176 x 10728 + side features, k=32, 1M rows)
# --------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big_log():
    sh = synth.SHAPES["kuairec_big"]
    train, val = synth.make_log(sh, "FM", "IPS", seed=0, n_train=1_000_000, n_val=20_000)
    return sh, train, val


@pytest.mark.parametrize("hot", [0, -1, -2])
@pytest.mark.parametrize("batch", [2000, 65536])
def test_full_size_step_properties(rfm, big_log, batch, hot):
    """At BASELINE's full size the oracle is too slow for whole fits, so check
    size-independent properties: (1) step == grad + apply, (2) a step is
    reproducible -- bitwise with the hot-column class off (every sum has a fixed
    order), to the last bits with it on (LDS atomics inside a workgroup),
    (3) g_w0 is minus the sum of residuals and the gradient of a column nobody
    touched is exactly zero, (4) one oracle step on the same batch."""
    pkg, _lib, runtime, rt = rfm
    from relevance_factorizationmachine_amd.fm import FmPlan
    sh, train, val = big_log
    n, k = train["features"].shape[1], sh.n_factors
    lr = 9e-6
    dev = runtime.DeviceCSR(rt, train["features"])
    y = rt.upload(train["labels"], dtype=np.float64)
    p = rt.upload(train["pscores"], dtype=np.float64)
    plan = FmPlan(rt, dev, train["labels"], train["pscores"], k, batch, hot)
    assert (plan.info()["hot_columns"] > 0) == (hot in (0, -2))  # (-2: on chip, in a fixed order)
    ids_h = runtime.sample_batches(dev.shape[0], batch, 0, 1)[0]
    ids = rt.upload(ids_h)
    csr = (dev.indptr.data_ptr(), dev.indices.data_ptr(), dev.values.data_ptr(), y.data_ptr(), p.data_ptr())

    def fresh():
        return _fm(pkg, n_factors=k, n_features=n, lr=lr, batch_size=batch)

    def params(m):
        return m.w0.dev.data_ptr(), m.w.dev.data_ptr(), m.V.dev.data_ptr()

    a, b, c = fresh(), fresh(), fresh()
    _lib.check(rt.lib.rfm_fm_step(rt.ctx, plan.handle, *csr, ids.data_ptr(), batch, *params(a), lr))
    _lib.check(rt.lib.rfm_fm_step(rt.ctx, plan.handle, *csr, ids.data_ptr(), batch, *params(b), lr))
    grad = rt.empty((n * k + n + 1,), y.dtype)
    _lib.check(rt.lib.rfm_fm_grad(rt.ctx, plan.handle, *csr, ids.data_ptr(), batch, *params(c), grad.data_ptr()))
    gh = grad.cpu().numpy().copy()
    _lib.check(rt.lib.rfm_fm_apply(rt.ctx, *params(c), grad.data_ptr(), n, k, lr))
    rt.sync()
    Va, Vb, Vc = a.V(), b.V(), c.V()
    if hot < 0:
        np.testing.assert_array_equal(Va, Vb)  # (2) bitwise reproducible
        np.testing.assert_array_equal(a.w(), b.w())
    else:
        assert rel_err(Va, Vb) < 1e-13 and rel_err(a.w(), b.w()) < 1e-13
    assert rel_err(Vc, Va) < 1e-13 and rel_err(c.w(), a.w()) < 1e-13  # (1)
    # (4) one oracle step on the same rows
    w0, w, V = cpu_ref.fm_init(12345, n, k)
    Xb = train["features"][ids_h]
    err, g_w0, g_w, G_V = cpu_ref.fm_gradients(Xb, train["labels"][ids_h], train["pscores"][ids_h], w0, w, V)
    assert rel_err(gh[: n * k].reshape(n, k), G_V) < TIGHT
    assert rel_err(gh[n * k: n * k + n], g_w) < TIGHT
    assert abs(gh[-1] - g_w0) <= TIGHT * max(1.0, abs(g_w0))
    assert abs(gh[-1] + np.sum(err)) <= 1e-9 * max(1.0, abs(np.sum(err)))  # (3)
    untouched = np.setdiff1d(np.arange(n), np.unique(Xb.indices))
    if untouched.size:
        assert np.all(gh[: n * k].reshape(n, k)[untouched] == 0.0)
        np.testing.assert_array_equal(Va[untouched], V[untouched])
    assert rel_err(Va, V - lr * G_V) < TIGHT
    plan.close()


@pytest.mark.parametrize("hot", [0, -2])
@pytest.mark.parametrize("small", [1, 300, 2000, 8000])
def test_small_step_on_a_plan_made_for_many_rows(rfm, small, hot):
    """A plan whose max_batch takes the many-rows forward keeps only the padded row blocks
    (layout()['row_blocks'] == 1); the header allows any batch 1..max_batch on it, and a shard
    of a global batch is exactly that: the small-batch forward shape must read the row blocks
    too.  rfm_fm_step, rfm_fm_grad and rfm_fm_grad_rows against the oracle."""
    pkg, _lib, runtime, rt = rfm
    from relevance_factorizationmachine_amd.fm import FmPlan
    sh = synth.SHAPES["kuairec_big"]
    train, _ = synth.make_log(sh, "FM", "IPS", seed=0, n_train=60_000, n_val=16)
    n, k, lr = train["features"].shape[1], 32, 1e-4
    dev = runtime.DeviceCSR(rt, train["features"])
    y = rt.upload(train["labels"], dtype=np.float64)
    p = rt.upload(train["pscores"], dtype=np.float64)
    plan = FmPlan(rt, dev, train["labels"], train["pscores"], k, 40_000, hot)
    assert plan.layout()["row_blocks"] == 1
    ids_h = runtime.sample_batches(dev.shape[0], small, 3, 1)[0]
    ids = rt.upload(ids_h)
    csr = (dev.indptr.data_ptr(), dev.indices.data_ptr(), dev.values.data_ptr(), y.data_ptr(), p.data_ptr())
    w0, w, V = cpu_ref.fm_init(12345, n, k)
    Xb = train["features"][ids_h]
    err, g_w0, g_w, G_V = cpu_ref.fm_gradients(Xb, train["labels"][ids_h], train["pscores"][ids_h], w0, w, V)

    m = _fm(pkg, n_factors=k, n_features=n, lr=lr, batch_size=small)
    params = (m.w0.dev.data_ptr(), m.w.dev.data_ptr(), m.V.dev.data_ptr())
    grad = rt.empty((n * k + n + 1,), y.dtype)
    _lib.check(rt.lib.rfm_fm_grad(rt.ctx, plan.handle, *csr, ids.data_ptr(), small, *params, grad.data_ptr()))
    gh = grad.cpu().numpy()
    assert rel_err(gh[: n * k].reshape(n, k), G_V) < TIGHT
    assert rel_err(gh[n * k: n * k + n], g_w) < TIGHT
    assert abs(gh[-1] - g_w0) <= TIGHT * max(1.0, abs(g_w0))
    # touched-row form
    cap = n
    rows = rt.empty((cap, k + 2), y.dtype)
    n_rows = rt.empty((1,), ids.dtype)
    gw0 = rt.empty((1,), y.dtype)
    _lib.check(rt.lib.rfm_fm_grad_rows(rt.ctx, plan.handle, ids.data_ptr(), small, *params, rows.data_ptr(), cap,
                                       n_rows.data_ptr(), gw0.data_ptr(), None, 0, None))
    cnt = int(n_rows.cpu().numpy()[0])
    rec = rows.cpu().numpy()[:cnt]
    cols = rec[:, 0].astype(np.int64)
    # ascending, every touched column present; the on-chip (hot) columns are always listed,
    # with a zero gradient when no row of the batch holds them
    assert np.all(np.diff(cols) > 0)
    touched = np.unique(Xb.indices)
    assert np.isin(touched, cols).all()
    extra = np.setdiff1d(cols, touched)
    assert np.isin(extra, plan.hot_columns()).all()
    assert rel_err(rec[:, 1: k + 1], G_V[cols]) < TIGHT and rel_err(rec[:, k + 1], g_w[cols]) < TIGHT
    assert np.all(rec[np.isin(cols, extra), 1:] == 0.0)
    # the step itself
    _lib.check(rt.lib.rfm_fm_step(rt.ctx, plan.handle, *csr, ids.data_ptr(), small, *params, lr))
    rt.sync()
    assert rel_err(m.V(), V - lr * G_V) < TIGHT and rel_err(m.w(), w - lr * g_w) < TIGHT
    assert abs(m.w0(0) - (w0[0] - lr * g_w0)) <= TIGHT
    plan.close()


def test_full_size_forward_permutation_invariance(rfm, big_log):
    pkg, _lib, runtime, rt = rfm
    sh, train, val = big_log
    model = _fm(pkg, n_factors=sh.n_factors, n_features=val["features"].shape[1])
    X = val["features"]
    base = model.predict(X)
    perm = np.random.default_rng(0).permutation(X.shape[0])
    np.testing.assert_array_equal(model.predict(X[perm]), base[perm])
    w0, w, V = cpu_ref.fm_init(12345, X.shape[1], sh.n_factors)
    assert rel_err(base, cpu_ref.fm_predict(X, w0, w, V)) < TIGHT


# --------------------------------------------------------------------------
# configs 4 and 5 of BASELINE.json at their parameter sizes (V 563 MB, P 1 GB)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("batch", [2000, 65536])
def test_config4_fit_vs_oracle(rfm, batch):
    """1M users x 100k items, n = 1 100 110 columns, k = 64: two fit() iterations against
    the oracle's closed-form step (the log is cut to 300k rows so the CPU side takes seconds;
    the parameter tables keep their full size, so V streams from HBM here)."""
    pkg = rfm[0]
    sh = synth.SHAPES["synthetic_1m"]
    train, val = synth.make_log(sh, "FM", "IPS", seed=0, n_train=300_000, n_val=5_000)
    n = train["features"].shape[1]
    assert n == 1_100_110
    kw = dict(n_epochs=2, n_factors=64, lr=9e-6, batch_size=batch, seed=12345)
    model = pkg.FactorizationMachines(estimator="IPS", n_features=n, **kw)
    tr, va = model.fit(train, val)
    ref = cpu_ref.fm_fit(train, val, form="closed", **kw)
    assert rel_err(model.V(), ref["V"]) < TIGHT
    assert rel_err(model.w(), ref["w"]) < TIGHT
    assert rel_err(model.w0(), ref["w0"]) < TIGHT
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT
    assert rel_err(model.predict(val["features"]), cpu_ref.fm_predict(val["features"], ref["w0"], ref["w"], ref["V"])) < TIGHT


def test_config5_mf_fit_vs_oracle(rfm):
    """MF with P 1M x 128 and Q 100k x 128: two sequential-SGD batches against the oracle."""
    pkg = rfm[0]
    sh = synth.SHAPES["synthetic_1m"]
    train, val = synth.make_log(sh, "MF", "IPS", seed=0, n_train=300_000, n_val=5_000)
    kw = dict(n_epochs=2, n_factors=128, lr=0.01, batch_size=2000, seed=12345, n_users=sh.n_users,
              n_items=sh.n_items, reg=0.5)
    model = pkg.LogisticMatrixFactorization(estimator="IPS", **kw)
    tr, va = model.fit(train, val)
    ref = cpu_ref.mf_fit(train, val, **kw)
    for nm in ("P", "Q", "b_u", "b_i"):
        assert rel_err(getattr(model, nm)(), ref[nm]) < TIGHT, nm
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT
    assert rel_err(model.predict(val["features"]),
                   cpu_ref.mf_predict(val["features"], ref["P"], ref["Q"], ref["b_u"], ref["b_i"], ref["b"])) < TIGHT


def test_import_shims_resolve_to_this_package(rfm):
    pkg = rfm[0]
    from src.fm import FactorizationMachines as FM
    from src.mf import LogisticMatrixFactorization as MF
    from src.base import PointwiseBaseRecommender as Base
    from utils.optimizer import SGD
    assert FM is pkg.FactorizationMachines and MF is pkg.LogisticMatrixFactorization
    assert issubclass(FM, Base) and SGD is pkg.DeviceSGD


# --------------------------------------------------------------------------
# ABI hardening (round 2): entry points called directly, large factor counts,
# non-finite parameters, id validation, empty validation sets, cache staleness
# --------------------------------------------------------------------------
def test_forward_loss_entry_point(rfm):
    """rfm_fm_forward_loss (fused scores + IPS log-loss, src/fm.py:90-102) directly
    through the C ABI, with and without a row-id list."""
    pkg, _lib, runtime, rt = rfm
    rng = np.random.default_rng(21)
    log = _random_log(rng, 900, 70, 0.1, 1)
    X, k = log["features"], 12
    model = _fm(pkg, n_factors=k, n_features=70)
    w0, w, V = cpu_ref.fm_init(12345, 70, k)
    dev = runtime.DeviceCSR(rt, X)
    y, p = rt.upload(log["labels"], dtype=np.float64), rt.upload(log["pscores"], dtype=np.float64)
    for rows in (None, rng.permutation(900)[:333].astype(np.int32)):
        m = 900 if rows is None else len(rows)
        d_rows = None if rows is None else rt.upload(rows)
        pred, loss = rt.empty((m,), y.dtype), rt.empty((1,), y.dtype)
        _lib.check(rt.lib.rfm_fm_forward_loss(
            rt.ctx, dev.indptr.data_ptr(), dev.indices.data_ptr(), dev.values.data_ptr(), y.data_ptr(),
            p.data_ptr(), None if rows is None else d_rows.data_ptr(), m, model.w0.dev.data_ptr(),
            model.w.dev.data_ptr(), model.V.dev.data_ptr(), 70, k, 1e-8, pred.data_ptr(), loss.data_ptr()))
        rt.sync()
        sel = slice(None) if rows is None else rows
        want = cpu_ref.fm_predict(X[sel], w0, w, V)
        assert rel_err(pred.cpu().numpy(), want) < TIGHT
        assert float(loss.cpu()[0]) == pytest.approx(
            cpu_ref.ips_logloss(log["labels"][sel], want, log["pscores"][sel]), rel=1e-12)
    with pytest.raises(ValueError):  # a loss over no rows is a caller error at this level
        _lib.check(rt.lib.rfm_fm_forward_loss(
            rt.ctx, dev.indptr.data_ptr(), dev.indices.data_ptr(), dev.values.data_ptr(), y.data_ptr(),
            p.data_ptr(), None, 0, model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr(),
            70, k, 1e-8, None, loss.data_ptr()))


@pytest.mark.parametrize("k,batch", [(4, 300), (32, 2000), (130, 257)])
def test_mf_plain_schedule_entry_points(rfm, k, batch):
    """rfm_mf_schedule + rfm_mf_sgd_levels (the order / level_ptr form of the exact MF
    batch, src/mf.py:97-108) directly through the C ABI against the sequential oracle."""
    pkg, _lib, runtime, rt = rfm
    rng = np.random.default_rng(k)
    nu, ni, n = 200, 30, 2500
    pairs = np.stack([rng.integers(0, nu, size=n), (rng.zipf(1.3, size=n) - 1) % ni], axis=1).astype(np.int64)
    y = (rng.random(n) < 0.5).astype(np.float64)
    p = rng.uniform(0.1, 1.0, size=n) ** 0.5
    P, Q, bu, bi = cpu_ref.mf_init(7, nu, ni, k)
    b, lr, reg = 0.4, 0.02, 0.5
    rows = cpu_ref.batch_ids(n, batch, 0).astype(np.int32)
    order, level_ptr = runtime.mf_schedule(pairs[rows, 0], pairs[rows, 1], nu, ni)
    dP, dQ, dbu, dbi = rt.upload(P), rt.upload(Q), rt.upload(bu), rt.upload(bi)
    du, di = rt.upload(pairs[:, 0].astype(np.int32)), rt.upload(pairs[:, 1].astype(np.int32))
    dy, dp = rt.upload(y), rt.upload(p)
    d_rows, d_order, d_lptr = rt.upload(rows), rt.upload(order), rt.upload(level_ptr)
    _lib.check(rt.lib.rfm_mf_sgd_levels(
        rt.ctx, du.data_ptr(), di.data_ptr(), dy.data_ptr(), dp.data_ptr(), d_rows.data_ptr(),
        d_order.data_ptr(), level_ptr.ctypes.data, d_lptr.data_ptr(), len(level_ptr) - 1, dP.data_ptr(),
        dQ.data_ptr(), dbu.data_ptr(), dbi.data_ptr(), b, k, lr, reg))
    rt.sync()
    cpu_ref.mf_sgd_batch(pairs[rows], y[rows], p[rows], P, Q, bu, bi, b, lr, reg)
    assert rel_err(dP.cpu().numpy(), P) < TIGHT and rel_err(dQ.cpu().numpy(), Q) < TIGHT
    assert rel_err(dbu.cpu().numpy(), bu) < TIGHT and rel_err(dbi.cpu().numpy(), bi) < TIGHT


@pytest.mark.parametrize("k", [512, 513, 1024])
def test_fm_largest_factor_counts(rfm, k):
    """Factor counts up to RFM_MAX_FACTORS (forward variants with 8 and 16 chunks per
    lane): scores, and two training iterations, against the oracle."""
    pkg = rfm[0]
    rng = np.random.default_rng(k)
    train = _random_log(rng, 700, 60, 0.1, 1)
    val = _random_log(rng, 150, 60, 0.1, 1)
    model = _fm(pkg, n_factors=k, n_features=60, lr=1e-6, batch_size=256, n_epochs=2, seed=9)
    w0, w, V = cpu_ref.fm_init(9, 60, k)
    assert rel_err(model.predict(val["features"]), cpu_ref.fm_predict(val["features"], w0, w, V)) < TIGHT
    tr, va = model.fit(train, val)
    ref = cpu_ref.fm_fit(train, val, n_epochs=2, n_factors=k, lr=1e-6, batch_size=256, seed=9)
    assert rel_err(model.V(), ref["V"]) < TIGHT and rel_err(model.w(), ref["w"]) < TIGHT
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT
    with pytest.raises(ValueError):
        _fm(pkg, n_factors=1025, n_features=60).predict(val["features"])


def test_non_finite_row_reaches_only_the_rows_that_hold_its_column(rfm):
    """Rows are padded to the lane group's round with entries that gather zeros, not
    V[0,:]: a NaN in row 0 of V (or w[0]) must leave every row that does not hold column 0
    exactly as the reference scores it (src/fm.py:114-133 never touches that row for them)."""
    pkg, _lib, runtime, rt = rfm
    rng = np.random.default_rng(3)
    log = _random_log(rng, 3000, 50, 0.12)
    X = log["features"]
    holds0 = np.zeros(3000, dtype=bool)
    holds0[X.tocsc()[:, 0].indices] = True
    for k in (6, 32):
        model = _fm(pkg, n_factors=k, n_features=50, lr=1e-4, batch_size=2048, n_epochs=1, seed=2)
        w0, w, V = cpu_ref.fm_init(2, 50, k)
        V[0, :] = np.nan
        w[0] = np.inf
        model.V.set(V)
        model.w.set(w)
        got = model.predict(X)
        want = cpu_ref.fm_predict(X, w0, w, V)
        assert np.isfinite(got[~holds0]).all() and (~holds0).sum() > 1000
        assert rel_err(got[~holds0], want[~holds0]) < TIGHT
        assert np.isnan(got[holds0]).all()
        # the training forward (plan records, several rows per lane group): one step leaves the
        # columns that only finite rows touch finite
        rows = np.flatnonzero(~holds0)[:2048]
        sub = {"features": X[rows], "labels": log["labels"][rows], "pscores": log["pscores"][rows]}
        model.fit(sub, sub)
        Vg = model.V()
        assert np.isfinite(Vg[1:]).all() and np.isnan(Vg[0]).all()


def test_check_ids_env_rejects_bad_row_ids(rfm, monkeypatch):
    """RFM_CHECK_IDS=1: a repeated or out-of-range row id is a RFM_ERR_BAD_ARG (ValueError);
    without it the precondition is the caller's (include/rfm_hip.h)."""
    pkg, _lib, runtime, rt = rfm
    from relevance_factorizationmachine_amd.fm import FmPlan
    train, _ = synth.make_log("coat", "FM", "IPS", seed=0)
    X = train["features"]
    model = _fm(pkg, n_factors=8, n_features=X.shape[1], batch_size=500)
    dev = runtime.DeviceCSR(rt, X)
    y, p = rt.upload(train["labels"], dtype=np.float64), rt.upload(train["pscores"], dtype=np.float64)
    plan = FmPlan(rt, dev, train["labels"], train["pscores"], 8, 500)
    good = cpu_ref.batch_ids(X.shape[0], 500, 0).astype(np.int32)

    def step(ids):
        d = rt.upload(ids)
        _lib.check(rt.lib.rfm_fm_step(rt.ctx, plan.handle, dev.indptr.data_ptr(), dev.indices.data_ptr(),
                                      dev.values.data_ptr(), y.data_ptr(), p.data_ptr(), d.data_ptr(), len(ids),
                                      model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr(), 1e-4))
        rt.sync()

    monkeypatch.setenv("RFM_CHECK_IDS", "1")
    step(good)
    step(good)  # the same ids in the next step are fine: distinct WITHIN a step
    dup = good.copy()
    dup[17] = dup[3]
    with pytest.raises(ValueError, match="twice"):
        step(dup)
    far = good.copy()
    far[5] = X.shape[0]
    with pytest.raises(ValueError, match="outside"):
        step(far)
    neg = good.copy()
    neg[0] = -1
    with pytest.raises(ValueError, match="outside"):
        step(neg)
    step(good)
    plan.close()


def test_fit_with_empty_validation_set_gives_nan_losses(rfm):
    """The reference's loss over an empty validation set is nan (mean of nothing,
    src/base.py:61); fit() must not refuse it."""
    pkg = rfm[0]
    train, val = synth.make_log("coat", "FM", "IPS", seed=0)
    empty = {"features": val["features"][:0], "labels": val["labels"][:0], "pscores": val["pscores"][:0]}
    model = _fm(pkg, n_factors=8, n_features=train["features"].shape[1], batch_size=500, n_epochs=3, lr=1e-4)
    tr, va = model.fit(train, empty)
    ref = cpu_ref.fm_fit(train, val, n_epochs=3, n_factors=8, lr=1e-4, batch_size=500, seed=12345)
    assert rel_err(tr, ref["train_loss"]) < TIGHT and len(va) == 3 and np.isnan(va).all()
    tmf, vmf = synth.make_log("coat", "MF", "IPS", seed=0)
    mf = pkg.LogisticMatrixFactorization(estimator="IPS", n_epochs=2, n_factors=4, lr=0.02, batch_size=500,
                                         seed=1, n_users=290, n_items=300, reg=0.5)
    tr, va = mf.fit(tmf, {"features": vmf["features"][:0], "labels": vmf["labels"][:0],
                          "pscores": vmf["pscores"][:0]})
    assert np.isfinite(tr).all() and np.isnan(va).all()


def test_predict_sees_in_place_edits_of_a_cached_matrix(rfm):
    """predict() keeps the device copy of the last matrices it was handed; editing one in
    place must not be scored from the stale copy."""
    pkg = rfm[0]
    rng = np.random.default_rng(8)
    log = _random_log(rng, 400, 30, 0.2)
    X = log["features"]
    model = _fm(pkg, n_factors=5, n_features=30)
    w0, w, V = cpu_ref.fm_init(12345, 30, 5)
    assert rel_err(model.predict(X), cpu_ref.fm_predict(X, w0, w, V)) < TIGHT
    X.data[:] = X.data * 2.0 + 0.25
    assert rel_err(model.predict(X), cpu_ref.fm_predict(X, w0, w, V)) < TIGHT
    X.data[X.nnz // 3] -= 7.0  # a single element
    assert rel_err(model.predict(X), cpu_ref.fm_predict(X, w0, w, V)) < TIGHT


def test_plan_builders_agree_and_reject_malformed_logs(rfm):
    """The training plan is built on the device, from the device copy of the log or from
    the caller's host arrays (rfm_fm_plan_create uploads a transient copy): both give the
    same step, and both reject what SciPy would not produce."""
    pkg, _lib, runtime, rt = rfm
    from relevance_factorizationmachine_amd.fm import FmPlan
    rng = np.random.default_rng(13)
    log = _random_log(rng, 3000, 200, 0.04, 2)
    X, k, B = log["features"], 10, 1500
    dev = runtime.DeviceCSR(rt, X)
    y, p = rt.upload(log["labels"], dtype=np.float64), rt.upload(log["pscores"], dtype=np.float64)
    hy, hp = log["labels"].astype(np.float64), log["pscores"].astype(np.float64)
    ids = rt.upload(cpu_ref.batch_ids(3000, B, 0).astype(np.int32))

    def host_plan(indptr, indices, values, n_rows=3000, n_cols=200):
        h = C.c_void_p()
        _lib.check(rt.lib.rfm_fm_plan_create(rt.ctx, indptr.ctypes.data, indices.ctypes.data, values.ctypes.data,
                                             hy.ctypes.data, hp.ctypes.data, n_rows, n_cols, k, B, 0, C.byref(h)))
        return h

    results = []
    for which in ("device", "host"):
        model = _fm(pkg, n_factors=k, n_features=200, lr=1e-4, batch_size=B)
        if which == "device":
            plan = FmPlan(rt, dev, y, p, k, B)
            handle = plan.handle
        else:
            handle = host_plan(dev.h_indptr, dev.h_indices, dev.h_values)
        _lib.check(rt.lib.rfm_fm_step(rt.ctx, handle, dev.indptr.data_ptr(), dev.indices.data_ptr(),
                                      dev.values.data_ptr(), y.data_ptr(), p.data_ptr(), ids.data_ptr(), B,
                                      model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr(), 1e-4))
        rt.sync()
        results.append((model.V(), model.w(), model.w0()))
        if which == "device":
            info = plan.info()
            assert info["nnz"] == X.nnz and info["hot_columns"] >= 2 and info["slots"] % 64 == 0
            plan.close()
        else:
            rt.lib.rfm_fm_plan_destroy(handle)
    w0, w, V = cpu_ref.fm_init(12345, 200, k)
    rows = cpu_ref.batch_ids(3000, B, 0)
    cpu_ref.fm_step_closed(X[rows], log["labels"][rows], log["pscores"][rows], w0, w, V, 1e-4)
    for Vg, wg, w0g in results:
        assert rel_err(Vg, V) < TIGHT and rel_err(wg, w) < TIGHT and rel_err(w0g, w0) < TIGHT
    assert rel_err(results[0][0], results[1][0]) < 1e-13  # same plan either way (hot sums: LDS atomics)

    indptr, indices, values = dev.h_indptr.copy(), dev.h_indices.copy(), dev.h_values.copy()
    bad = indices.copy()
    r = int(np.flatnonzero(np.diff(indptr) >= 2)[0])
    bad[indptr[r] + 1] = bad[indptr[r]]  # a row names a column twice
    with pytest.raises(ValueError, match="twice"):
        host_plan(indptr, bad, values)
    bad = indices.copy()
    bad[7] = 200
    with pytest.raises(ValueError, match="column index"):
        host_plan(indptr, bad, values)
    bad = indices.copy()
    bad[7] = -3
    with pytest.raises(ValueError, match="column index"):
        host_plan(indptr, bad, values)
    badp = indptr.copy()
    badp[5], badp[6] = badp[6], badp[5] - 1
    with pytest.raises(ValueError, match="monotone"):
        host_plan(badp, indices, values)
    # a log without entries still gives a plan (rows score sigmoid(w0))
    empty = host_plan(np.zeros(4, np.int64), np.zeros(1, np.int32), np.zeros(1), n_rows=3, n_cols=5)
    rt.lib.rfm_fm_plan_destroy(empty)


@pytest.mark.parametrize("shape,k,batch,n_train", [("kuairec_small", 16, 2000, None), ("kuairec_big", 32, 40_000, 120_000),
                                                   ("kuairec_big", 64, 30_000, 100_000), ("kuairec_big", 7, 5000, 60_000)])
def test_deterministic_switch_gives_bitwise_reproducible_fits(rfm, shape, k, batch, n_train):
    """model.deterministic = True (hot_min_count = -2): the frequent columns are still summed on
    chip, but in a fixed order (or, where the factor count has no such kernel, through their
    column lists): two fits are equal bit for bit, at the one-row and the many-rows forward
    shape; the default (LDS atomics) agrees to ~1e-12, and so does the oracle."""
    pkg = rfm[0]
    train, val = synth.make_log(shape, "FM", "IPS", seed=0, **({"n_train": n_train, "n_val": 500} if n_train else {}))
    kw = dict(estimator="IPS", n_epochs=3, n_factors=k, lr=9e-6, batch_size=batch, seed=12345,
              n_features=train["features"].shape[1])
    fits = []
    for det in (True, True, False):
        m = pkg.FactorizationMachines(**kw)
        m.deterministic = det
        tr, va = m.fit(train, val)
        fits.append((m.V(), m.w(), tr, va, m.plan_info))
    assert fits[0][4]["hot_columns"] > 0  # (the class is on in this mode)
    np.testing.assert_array_equal(fits[0][0], fits[1][0])
    np.testing.assert_array_equal(fits[0][1], fits[1][1])
    assert fits[0][2] == fits[1][2] and fits[0][3] == fits[1][3]
    assert rel_err(fits[2][0], fits[0][0]) < 1e-12 and rel_err(fits[2][2], fits[0][2]) < 1e-12
    ref = cpu_ref.fm_fit(train, val, n_epochs=3, n_factors=k, lr=9e-6, batch_size=batch, seed=12345)
    assert rel_err(fits[0][0], ref["V"]) < TIGHT and rel_err(fits[0][1], ref["w"]) < TIGHT
    assert rel_err(fits[0][2], ref["train_loss"]) < TIGHT and rel_err(fits[0][3], ref["val_loss"]) < TIGHT


@pytest.mark.parametrize("shape,k,batch,its", [("kuairec_small", 16, 2000, 150), ("kuairec_small", 400, 2000, 24),
                                              ("coat", 8, 500, 150)])
def test_prepared_steps_experiment_gives_the_same_fit(rfm, monkeypatch, shape, k, batch, its):
    """RFM_PREP=1 (an opt-in experiment, off by default: profiles/r3i): the batches' row blocks and
    the tasks' records of a chunk of iterations are laid out ahead of the loop and the gradient
    launch takes its PREP form.  The sums and their order are those of the default form: in the
    bitwise-reproducible mode the two fits are equal bit for bit -- more than one chunk here."""
    pkg, _lib, runtime, rt = rfm
    sh = synth.SHAPES[shape]
    train, val = synth.make_log(sh, "FM", "IPS", seed=0)

    def fit():
        rt.clear_caches()  # (a remembered plan would keep the form it was built with)
        m = pkg.FactorizationMachines(estimator="IPS", n_epochs=its, n_factors=k, lr=9e-6, batch_size=batch,
                                      seed=12345, n_features=train["features"].shape[1])
        m.deterministic = True
        return m, m.fit(train, val)

    # (48 MiB of chunk buffers: 10 iterations per chunk at k = 400, 64 at the small factor counts --
    # several chunks in every case, and the hand-over between the two buffers)
    monkeypatch.setenv("RFM_PREP_MB", "48")
    base, (tr0, va0) = fit()
    for mode in ("1", "2"):  # rows + records laid out ahead; records only
        monkeypatch.setenv("RFM_PREP", mode)
        prep, (tr1, va1) = fit()
        monkeypatch.delenv("RFM_PREP")
        rt.clear_caches()
        np.testing.assert_array_equal(prep.V(), base.V())
        np.testing.assert_array_equal(prep.w(), base.w())
        assert prep.w0(0) == base.w0(0)
        # (the loss curves: the same forwards, but the default form may score the batch and the
        # validation rows in one launch -- another partition of the same sum)
        assert rel_err(tr1, tr0) < 1e-13 and rel_err(va1, va0) < 1e-13


@pytest.mark.parametrize("k", [129, 130, 191, 257, 258, 300, 383, 384, 385, 400, 511])
def test_fm_fit_every_chunk_count(rfm, k):
    """Factor counts of several chunks per lane (2, 3, 4, 8 chunks of 64 lanes; one and two
    factors per lane; the last chunk full, nearly empty, or one lane wide): the forward's
    in-register chunks and the gradient launch's and the finalize's chunk-per-workgroup forms
    against the oracle -- a log with four columns in every row (longer than a workgroup's
    tasks: pieces + finalize, short and long), ragged rows and untouched columns; the step, the
    dense gradient and the touched-row records."""
    pkg, _lib, runtime, rt = rfm
    from relevance_factorizationmachine_amd.fm import FmPlan
    rng = np.random.default_rng(k)
    train = _random_log(rng, 9000, 120, 0.04, 4)
    val = _random_log(rng, 300, 120, 0.04, 4)
    lr, batch, its = 2e-6, 3000, 3
    model = _fm(pkg, n_factors=k, n_features=120, lr=lr, batch_size=batch, n_epochs=its, seed=5)
    tr, va = model.fit(train, val)
    assert model.plan_info["hot_columns"] == 0 and model.plan_info["split_columns"] > 0
    ref = cpu_ref.fm_fit(train, val, n_epochs=its, n_factors=k, lr=lr, batch_size=batch, seed=5)
    assert rel_err(model.V(), ref["V"]) < TIGHT and rel_err(model.w(), ref["w"]) < TIGHT
    assert rel_err(model.w0(), ref["w0"]) < TIGHT
    assert rel_err(tr, ref["train_loss"]) < TIGHT and rel_err(va, ref["val_loss"]) < TIGHT
    # gradient forms on a fresh model: dense, and as touched-row records
    dev = runtime.DeviceCSR(rt, train["features"])
    y = rt.upload(train["labels"], dtype=np.float64)
    p = rt.upload(train["pscores"], dtype=np.float64)
    plan = FmPlan(rt, dev, y, p, k, batch)
    ids_h = runtime.sample_batches(dev.shape[0], batch, 1, 1)[0]
    ids = rt.upload(ids_h)
    m = _fm(pkg, n_factors=k, n_features=120, lr=lr, batch_size=batch, seed=5)
    params = (m.w0.dev.data_ptr(), m.w.dev.data_ptr(), m.V.dev.data_ptr())
    csr = (dev.indptr.data_ptr(), dev.indices.data_ptr(), dev.values.data_ptr(), y.data_ptr(), p.data_ptr())
    w0, w, V = cpu_ref.fm_init(5, 120, k)
    Xb = train["features"][ids_h]
    _, g_w0, g_w, G_V = cpu_ref.fm_gradients(Xb, train["labels"][ids_h], train["pscores"][ids_h], w0, w, V)
    grad = rt.empty((120 * k + 120 + 1,), y.dtype)
    for _ in range(2):  # twice: the slot bitmap alternates between two buffers by step parity
        _lib.check(rt.lib.rfm_fm_grad(rt.ctx, plan.handle, *csr, ids.data_ptr(), batch, *params, grad.data_ptr()))
        gh = grad.cpu().numpy()
        assert rel_err(gh[: 120 * k].reshape(120, k), G_V) < TIGHT and rel_err(gh[120 * k: -1], g_w) < TIGHT
        assert abs(gh[-1] - g_w0) <= TIGHT * max(1.0, abs(g_w0))
    rows = rt.empty((120, k + 2), y.dtype)
    n_rows = rt.empty((1,), ids.dtype)
    gw0 = rt.empty((1,), y.dtype)
    _lib.check(rt.lib.rfm_fm_grad_rows(rt.ctx, plan.handle, ids.data_ptr(), batch, *params, rows.data_ptr(), 120,
                                       n_rows.data_ptr(), gw0.data_ptr(), None, 0, None))
    rec = rows.cpu().numpy()[: int(n_rows.cpu().numpy()[0])]
    cols = rec[:, 0].astype(np.int64)
    np.testing.assert_array_equal(cols, np.unique(Xb.indices))
    assert rel_err(rec[:, 1: k + 1], G_V[cols]) < TIGHT and rel_err(rec[:, k + 1], g_w[cols]) < TIGHT
    plan.close()


@pytest.mark.parametrize("shape,k,batch", [("kuairec_small", 400, 2000), ("kuairec_small", 16, 2000), ("coat", 300, 500)])
def test_merged_loss_forward_gives_the_same_losses(rfm, monkeypatch, shape, k, batch):
    """The train-loss and validation-loss forwards of an iteration as ONE launch (the SEG form of the
    forward: rows of two logs, two sets of loss partials; default for factor counts of several
    chunks per lane, RFM_MERGE_LOSS=2 forces it, 0 forbids it): same parameters bit for bit, same
    loss curves up to the order of the sums, both against the oracle."""
    pkg, _lib, runtime, rt = rfm
    sh = synth.SHAPES[shape]
    train, val = synth.make_log(sh, "FM", "IPS", seed=0)
    fits = {}
    monkeypatch.setenv("RFM_SLICED_LOSS", "0")  # (the sliced loss forward would take k = 400 over)
    for mode in ("0", "2"):
        monkeypatch.setenv("RFM_MERGE_LOSS", mode)
        m = pkg.FactorizationMachines(estimator="IPS", n_epochs=6, n_factors=k, lr=9e-6, batch_size=batch,
                                      seed=12345, n_features=train["features"].shape[1])
        m.deterministic = True
        fits[mode] = (m, *m.fit(train, val))
    monkeypatch.delenv("RFM_MERGE_LOSS")
    (a, tra, vaa), (b, trb, vab) = fits["0"], fits["2"]
    np.testing.assert_array_equal(a.V(), b.V())
    assert rel_err(trb, tra) < 1e-13 and rel_err(vab, vaa) < 1e-13
    ref = cpu_ref.fm_fit(train, val, n_epochs=6, n_factors=k, lr=9e-6, batch_size=batch, seed=12345)
    assert rel_err(trb, ref["train_loss"]) < TIGHT and rel_err(vab, ref["val_loss"]) < TIGHT


@pytest.mark.parametrize("k,n_cols,density,val_kind", [
    (130, 300, 0.03, "same"), (192, 300, 0.03, "long"), (258, 120, 0.15, "same"), (300, 300, 0.03, "mixed"),
    (400, 300, 0.03, "same"), (400, 40, 0.9, "long"), (514, 300, 0.03, "empty_rows"), (1022, 200, 0.05, "mixed"),
    (1024, 200, 0.05, "same")])
def test_sliced_loss_forward_gives_the_same_losses(rfm, monkeypatch, k, n_cols, density, val_kind):
    """The loss forwards of fit() sliced by factors (rfm_fm_sliced.hpp: every even k > 128, taken when
    the batch and the validation log together have enough rows; RFM_SLICED_MIN_ROWS=1 forces it,
    RFM_SLICED_LOSS=0 forbids it): same parameters bit for bit, loss curves equal to the plain
    forwards' up to the order of the sums, both against the oracle.  Slices of 2, 4 and 8 pieces with a
    narrower last slice; staged rows of 16 / 32 / 64 entries; validation rows LONGER than the staging
    stride derived from the training log (read straight from the log), empty rows, rows of cached
    columns only and of uncached columns only; twice in a row (the second fit finds a warm plan)."""
    pkg = rfm[0]
    rng = np.random.default_rng(7 * k + n_cols)
    train = _random_log(rng, 3000, n_cols, density, 4)
    if val_kind == "same":
        val = _random_log(rng, 700, n_cols, density, 4)
    elif val_kind == "long":  # every validation row longer than any training row
        val = _random_log(rng, 300, n_cols, min(1.0, density * 6 + 0.3), 4)
    else:
        a = _random_log(rng, 400, n_cols, density, 4)
        b = _random_log(rng, 200, n_cols, min(1.0, density * 8 + 0.3), 0)
        X = vstack([a["features"], b["features"]]).tolil()
        if val_kind == "empty_rows":
            X[0, :] = 0
            X[17, :] = 0
            X[599, :] = 0
        X[3, 4:] = 0   # only columns that are in every training row
        X[5, :4] = 0   # none of them
        X = X.tocsr()
        X.eliminate_zeros()
        X.sort_indices()
        val = {"features": X, "labels": np.concatenate([a["labels"], b["labels"]]),
               "pscores": np.concatenate([a["pscores"], b["pscores"]])}
    lr, batch, its = 2e-6, 1000, 4
    fits = {}
    for mode in ("plain", "sliced"):
        monkeypatch.setenv("RFM_SLICED_LOSS", "0" if mode == "plain" else "1")
        monkeypatch.setenv("RFM_SLICED_MIN_ROWS", "1")
        m = _fm(pkg, n_factors=k, n_features=n_cols, lr=lr, batch_size=batch, n_epochs=its, seed=5)
        first = m.fit(train, val)
        m2 = _fm(pkg, n_factors=k, n_features=n_cols, lr=lr, batch_size=batch, n_epochs=its, seed=5)
        again = m2.fit(train, val)
        np.testing.assert_array_equal(np.asarray(first), np.asarray(again))
        fits[mode] = (m, *first)
    monkeypatch.delenv("RFM_SLICED_LOSS")
    monkeypatch.delenv("RFM_SLICED_MIN_ROWS")
    (a, tra, vaa), (b, trb, vab) = fits["plain"], fits["sliced"]
    np.testing.assert_array_equal(a.V(), b.V())
    assert rel_err(trb, tra) < 1e-12 and rel_err(vab, vaa) < 1e-12
    info = b.plan_info
    assert info["slices"] == (1 if k <= 256 else 2 if k <= 512 else 4) and info["cached_columns"] >= 4
    assert info["slices"] * info["factors_per_slice"] >= k > (info["slices"] - 1) * info["factors_per_slice"]
    ref = cpu_ref.fm_fit(train, val, n_epochs=its, n_factors=k, lr=lr, batch_size=batch, seed=5)
    assert rel_err(trb, ref["train_loss"]) < TIGHT and rel_err(vab, ref["val_loss"]) < TIGHT
    assert_elementwise(trb, ref["train_loss"], what="train loss")
    assert_elementwise(vab, ref["val_loss"], what="validation loss")


@pytest.mark.parametrize("bad_col,holders", [(10, 100), (150, 5)])
def test_sliced_loss_forward_keeps_a_non_finite_row_to_itself(rfm, monkeypatch, bad_col, holders):
    """A non-finite row of V reaches exactly the rows that hold its column: a validation log without
    the column keeps a finite loss, equal to the plain forward's, and one with it turns NaN -- with
    the column among the cached ones (100 training rows hold it) and not (5 do).  The holders are
    training rows that no batch of the fit samples, so the step never spreads the NaN."""
    pkg, _lib, runtime, rt = rfm
    rng = np.random.default_rng(3)
    train = _random_log(rng, 2000, 200, 0.04, 3)
    val = _random_log(rng, 500, 200, 0.04, 3)
    batch, its = 500, 3
    sampled = np.unique(runtime.sample_batches(2000, batch, 0, its))
    free = np.setdiff1d(np.arange(2000), sampled)
    assert free.size >= holders
    Xt = train["features"].tolil()
    Xt[:, bad_col] = 0
    Xt[free[:holders], bad_col] = 1.5
    train["features"] = Xt.tocsr()
    train["features"].eliminate_zeros()
    Xv = val["features"].tolil()
    Xv[:, bad_col] = 0
    without = dict(val, features=Xv.tocsr())
    without["features"].eliminate_zeros()
    Xv[::7, bad_col] = 0.5
    with_col = dict(val, features=Xv.tocsr())
    monkeypatch.setenv("RFM_SLICED_MIN_ROWS", "1")
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RFM_SLICED_LOSS", mode)
        for name, v in (("without", without), ("with", with_col)):
            m = _fm(pkg, n_factors=300, n_features=200, lr=1e-6, batch_size=batch, n_epochs=its, seed=5)
            V = m.V()
            V[bad_col, 7] = np.inf
            V[bad_col, 290] = np.nan
            m.V.set(V)
            out[mode, name] = m.fit(train, v)
    monkeypatch.delenv("RFM_SLICED_LOSS")
    monkeypatch.delenv("RFM_SLICED_MIN_ROWS")
    for name in ("without", "with"):
        (tr0, va0), (tr1, va1) = out["0", name], out["1", name]
        assert np.isfinite(tr0).all() and np.isfinite(tr1).all() and rel_err(tr1, tr0) < 1e-12
        if name == "without":
            assert np.isfinite(va1).all() and rel_err(va1, va0) < 1e-12
        else:
            assert np.isnan(va0).all() and np.isnan(va1).all()


@pytest.mark.parametrize("shape,k,batch", [("kuairec_small", 16, 2000), ("coat", 300, 500), ("kuairec_small", 64, 20000)])
def test_deferred_loss_logarithms_give_the_same_losses(rfm, monkeypatch, shape, k, batch):
    """The plain loss forwards of rfm_fm_train leave their rows' scores and one launch per run of
    iterations takes the logarithms (default; RFM_DEFER_LOSS=0: inside the forward, staged through LDS
    in the many-rows shape): same parameters bit for bit, same loss curves up to the order of the sums,
    both against the oracle.  150 iterations: more than one run of 128 (a run's last train-loss rows are
    scored in the NEXT run's first step when they ride in the forward launch, RFM_RIDE_LOSS)."""
    pkg = rfm[0]
    sh = synth.SHAPES[shape]
    train, val = synth.make_log(sh, "FM", "IPS", seed=0)
    its = 150 if batch <= 2000 else 5
    fits = {}
    monkeypatch.setenv("RFM_SLICED_LOSS", "0")
    monkeypatch.setenv("RFM_MERGE_LOSS", "0")
    for mode, defer, ride, ride_val in (("0", "0", "1", "1"), ("1", "1", "1", "1"), ("apart", "1", "0", "1"),
                                        ("train_only", "1", "1", "0")):
        monkeypatch.setenv("RFM_DEFER_LOSS", defer)
        monkeypatch.setenv("RFM_RIDE_LOSS", ride)
        monkeypatch.setenv("RFM_RIDE_VAL", ride_val)
        m = pkg.FactorizationMachines(estimator="IPS", n_epochs=its, n_factors=k, lr=9e-6, batch_size=batch,
                                      seed=12345, n_features=train["features"].shape[1])
        m.hot_min_count = -1  # (no on-chip class: every sum in a fixed order, and the rows may ride)
        fits[mode] = (m, *m.fit(train, val))
    for name in ("RFM_DEFER_LOSS", "RFM_RIDE_LOSS", "RFM_RIDE_VAL", "RFM_SLICED_LOSS", "RFM_MERGE_LOSS"):
        monkeypatch.delenv(name)
    (a, tra, vaa), (b, trb, vab), (c, trc, vac) = fits["0"], fits["1"], fits["apart"]
    # (the train-loss rows -- and the registered validation log's -- riding in the next step's forward
    # launch at small batches, or scored by launches of their own: the same scores, bit for bit)
    np.testing.assert_array_equal(np.asarray(trb), np.asarray(trc))
    np.testing.assert_array_equal(np.asarray(vab), np.asarray(vac))
    np.testing.assert_array_equal(np.asarray(vab), np.asarray(fits["train_only"][2]))
    np.testing.assert_array_equal(np.asarray(trb), np.asarray(fits["train_only"][1]))
    np.testing.assert_array_equal(a.V(), b.V())
    np.testing.assert_array_equal(c.V(), b.V())
    assert len(trb) == its and rel_err(trb, tra) < 1e-13 and rel_err(vab, vaa) < 1e-13
    n_ref = min(its, 8)
    ref = cpu_ref.fm_fit(train, val, n_epochs=n_ref, n_factors=k, lr=9e-6, batch_size=batch, seed=12345)
    assert rel_err(trb[:n_ref], ref["train_loss"]) < TIGHT and rel_err(vab[:n_ref], ref["val_loss"]) < TIGHT


def test_sliced_loss_forward_one_iteration_per_call_and_registration(rfm, monkeypatch):
    """A fit() with a host evaluator trains one iteration per rfm_fm_train call: the sliced loss
    forward then reads the validation log that fit() REGISTERED with the plan
    (rfm_fm_plan_register_log: translated once, not per call) and must give the losses of the
    plain many-iterations-per-call fit bit for bit; a call that names other validation arrays than
    the registered ones is translated per call and is right as well."""
    pkg, _lib, runtime, rt = rfm
    from relevance_factorizationmachine_amd.fm import FmPlan
    rng = np.random.default_rng(11)
    train = _random_log(rng, 3000, 200, 0.05, 4)
    val = _random_log(rng, 700, 200, 0.05, 4)
    other = _random_log(rng, 500, 200, 0.05, 4)
    monkeypatch.setenv("RFM_SLICED_MIN_ROWS", "1")

    class Mean:
        features = {"FM": val["features"]}

        def evaluate(self, y_scores, estimator):
            return float(np.mean(y_scores))

    kw = dict(n_factors=300, n_features=200, lr=2e-6, batch_size=1000, n_epochs=5, seed=5)
    base = _fm(pkg, **kw)
    base.hot_min_count = -1
    tr0, va0 = base.fit(train, val)
    m = _fm(pkg, evaluator=Mean(), **kw)
    m.hot_min_count = -1
    m.device_evaluator = False
    tr1, va1 = m.fit(train, val)
    np.testing.assert_array_equal(np.asarray(tr1), np.asarray(tr0))
    np.testing.assert_array_equal(np.asarray(va1), np.asarray(va0))
    np.testing.assert_array_equal(m.V(), base.V())
    assert len(m.val_metrics) == 5
    # through the C ABI: the plan holds `val` registered, the call names `other`
    dev = runtime.DeviceCSR(rt, train["features"])
    y = rt.upload(train["labels"], dtype=np.float64)
    p = rt.upload(train["pscores"], dtype=np.float64)
    dv = runtime.DeviceCSR(rt, val["features"])
    do = runtime.DeviceCSR(rt, other["features"])
    oy = rt.upload(other["labels"], dtype=np.float64)
    op = rt.upload(other["pscores"], dtype=np.float64)
    ids = rt.upload(runtime.sample_batches(3000, 1000, 0, 5))
    out = {}
    for registered in (False, True):
        plan = FmPlan(rt, dev, y, p, 300, 1000, -1)
        if registered:
            _lib.check(rt.lib.rfm_fm_plan_register_log(
                rt.ctx, plan.handle, 0, dv.indptr.data_ptr(), dv.indices.data_ptr(), dv.values.data_ptr(), 700))
        mm = _fm(pkg, **kw)
        tl = rt.empty((5,), y.dtype)
        vl = rt.empty((5,), y.dtype)
        _lib.check(rt.lib.rfm_fm_train(
            rt.ctx, plan.handle, dev.indptr.data_ptr(), dev.indices.data_ptr(), dev.values.data_ptr(),
            y.data_ptr(), p.data_ptr(), ids.data_ptr(), 1000, 5, mm.w0.dev.data_ptr(), mm.w.dev.data_ptr(),
            mm.V.dev.data_ptr(), 2e-6, do.indptr.data_ptr(), do.indices.data_ptr(), do.values.data_ptr(),
            oy.data_ptr(), op.data_ptr(), 500, 1e-8, tl.data_ptr(), vl.data_ptr()))
        rt.sync()
        out[registered] = vl.cpu().numpy()
        plan.close()
    monkeypatch.delenv("RFM_SLICED_MIN_ROWS")
    np.testing.assert_array_equal(out[True], out[False])
    ref = cpu_ref.fm_fit(train, other, n_epochs=5, n_factors=300, lr=2e-6, batch_size=1000, seed=5)
    assert rel_err(out[True], ref["val_loss"]) < TIGHT


@pytest.mark.parametrize("k", [300, 400, 16])
def test_plan_forward_scores(rfm, monkeypatch, k):
    """rfm_fm_plan_forward: the scores of an evaluation log through the plan -- by the sliced forward
    where the plan has one (even k > 128; the log registered in slot 1 or translated per call), by the
    plain forward otherwise -- against rfm_fm_forward and the oracle; rows longer than the records of
    a translated row, empty rows."""
    pkg, _lib, runtime, rt = rfm
    from relevance_factorizationmachine_amd.fm import FmPlan
    rng = np.random.default_rng(k)
    train = _random_log(rng, 3000, 200, 0.05, 4)
    a = _random_log(rng, 900, 200, 0.05, 4)
    b = _random_log(rng, 100, 200, 0.6, 0)
    X = vstack([a["features"], b["features"]]).tolil()
    X[7, :] = 0
    X = X.tocsr()
    X.eliminate_zeros()
    X.sort_indices()
    monkeypatch.setenv("RFM_SLICED_MIN_ROWS", "1")
    dev = runtime.DeviceCSR(rt, train["features"])
    y = rt.upload(train["labels"], dtype=np.float64)
    p = rt.upload(train["pscores"], dtype=np.float64)
    ev = runtime.DeviceCSR(rt, X)
    m = _fm(pkg, n_factors=k, n_features=200, seed=9)
    params = (m.w0.dev.data_ptr(), m.w.dev.data_ptr(), m.V.dev.data_ptr())
    plan = FmPlan(rt, dev, y, p, k, 1000)
    assert (plan.sliced()["slices"] > 0) == (k > 128)
    want = m.predict(X)
    for registered in (False, True, False):
        _lib.check(rt.lib.rfm_fm_plan_register_log(
            rt.ctx, plan.handle, 1, *( (ev.indptr.data_ptr(), ev.indices.data_ptr(), ev.values.data_ptr(), 1000)
                                       if registered else (None, None, None, 0))))
        out = rt.empty((1000,), y.dtype)
        _lib.check(rt.lib.rfm_fm_plan_forward(rt.ctx, plan.handle, ev.indptr.data_ptr(), ev.indices.data_ptr(),
                                              ev.values.data_ptr(), 1000, *params, out.data_ptr()))
        rt.sync()
        got = out.cpu().numpy()
        assert rel_err(got, want) < 1e-12
        assert_elementwise(got, want, what="scores")
    monkeypatch.delenv("RFM_SLICED_MIN_ROWS")
    plan.close()
    w0, w, V = cpu_ref.fm_init(9, 200, k)
    assert rel_err(got, cpu_ref.fm_predict(X, w0, w, V)) < TIGHT
