"""The per-iteration validation metric on the device (``rfm_val_dcg``, SURVEY.md
8f N1) against the oracle's restatement of ``ValEvaluator.evaluate`` and the
reference's own outputs (fixture G7).  Needs an MI355X: ``pytest -m gpu``.

Tolerance: float64 on both sides; per user the terms are added in the reference's
order, the mean over users is a different (fixed) summation order: 1e-12 relative.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import cpu_ref
from relevance_factorizationmachine_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-12


@pytest.fixture(scope="module")
def ev():
    from relevance_factorizationmachine_amd import evaluate, runtime
    return evaluate, runtime.Runtime.get()


def _stable_rule_dcg(frame, scores, pscores, k):
    """IPS-DCG@k with the documented tie rule (later row first), per user."""
    vals, ok = [], []
    for rows in cpu_ref._per_user_slices(frame["user"]):
        rank = np.argsort(scores[rows], kind="stable")[::-1]
        ys = frame["label"][rows][rank]
        if np.sum(ys) == 0:
            vals.append(0.0)
            ok.append(False)
            continue
        vals.append(cpu_ref.ips_dcg_at_k(ys, k, pscores[rows][rank]))
        ok.append(True)
    return np.array(vals), np.array(ok)


def test_matches_reference_fixture(ev):
    """Values the reference's ValEvaluator / TestEvaluator produced (make_golden_eval.py)."""
    evaluate, rt = ev
    g = load_golden("val_dcg_distinct")
    for est, col in (("IPS", "pscore"), ("Naive", "ones_pscore")):
        for k in (1, 3, 5, 10):
            fr = evaluate.DeviceValFrame(rt, g["user"], g["label"], g[col], k)
            val, order_dependent = fr.dcg_checked(g["scores"])
            assert val == pytest.approx(float(g[f"val_dcg_{est}_k{k}"]), rel=TOL), (est, k)
            assert order_dependent == 0
        # ties between rows of equal label and propensity: any tie order gives the same value
        fr = evaluate.DeviceValFrame(rt, g["user"], g["tied_label"],
                                     g["tied_pscore"] if est == "IPS" else g[col], 5)
        val, order_dependent = fr.dcg_checked(g["tied_scores"])
        assert val == pytest.approx(float(g[f"tied_val_dcg_{est}"]), rel=TOL)
        assert order_dependent == 0
    # TestEvaluator's DCG@K (utils/metrics.py:83-107) is the same sum without propensities
    got = [evaluate.DeviceValFrame(rt, g["user"], g["label"], None, k).dcg(g["scores"]) for k in (1, 3, 5, 7, 9)]
    np.testing.assert_allclose(got, g["test_dcg"], rtol=TOL)


def test_saturated_ties_follow_the_documented_rule(ev):
    """Fixture G7 has 10 % of its scores tied at 1.0 with different labels: the reference's
    value there is whatever order NumPy's unstable, CPU-dependent default sort leaves (parity
    unpinned for that case); the device follows argsort(kind="stable")[::-1] and reports the
    users concerned, which is what sends such an iteration back to the host evaluator."""
    evaluate, rt = ev
    g = load_golden("val_dcg")
    frame = {k: g[k] for k in ("user", "label", "pscore")}
    fr = evaluate.DeviceValFrame(rt, g["user"], g["label"], g["pscore"], 5)
    got, order_dependent = fr.dcg_checked(g["scores"])
    vals, ok = _stable_rule_dcg(frame, g["scores"], g["pscore"], 5)
    assert got == pytest.approx(float(np.mean(vals[ok])), rel=TOL)
    assert order_dependent == _order_dependent_users(frame, g["scores"], g["pscore"], 5).sum() > 0
    # and stays within the spread of tie orders around the reference's value
    assert abs(got - float(g["val_dcg_IPS"])) < 0.05 * float(g["val_dcg_IPS"])


def _order_dependent_users(frame, scores, pscores, k):
    """Users (with a positive label) for whom a score tie between rows of different label or
    propensity reaches into the first k ranks."""
    flags = []
    for rows in cpu_ref._per_user_slices(frame["user"]):
        if np.sum(frame["label"][rows]) == 0:
            flags.append(False)
            continue
        rank = np.argsort(scores[rows], kind="stable")[::-1]
        sc, ys, ps = scores[rows][rank], frame["label"][rows][rank], pscores[rows][rank]
        top = sc[: min(k, len(sc))]
        flag = bool(np.isnan(sc).any())
        for v in np.unique(top[~np.isnan(top)]):
            m = sc == v
            flag = flag or len(set(zip(ys[m].tolist(), ps[m].tolist()))) > 1
        flags.append(flag)
    return np.array(flags)


@pytest.mark.parametrize("k", [1, 5, 9])
def test_ragged_frames_against_oracle(ev, k):
    evaluate, rt = ev
    rng = np.random.default_rng(k)
    # users with one row, fewer than k rows, thousands of rows, and no positive label at all
    sizes = np.concatenate([[1, 1, 2, 3, 4, 4000, 777], rng.integers(1, 200, size=300)])
    users = np.repeat(rng.permutation(len(sizes)) * 3 + 1, sizes)
    perm = rng.permutation(users.shape[0])  # frame order is not grouped
    users = users[perm]
    labels = (rng.random(users.shape[0]) < 0.08).astype(np.int64)
    labels[users == users[0]] = 0
    pscore = rng.uniform(0.1, 1.0, users.shape[0]) ** 0.5
    scores = rng.random(users.shape[0])
    frame = {"user": users, "label": labels, "pscore": pscore, "ones_pscore": np.ones_like(pscore)}
    fr = evaluate.DeviceValFrame(rt, users, labels, pscore, k)
    got = fr.dcg(scores)
    assert got == pytest.approx(cpu_ref.val_dcg(frame, scores, "IPS", k=k), rel=TOL)
    vals, ok, dep = fr.per_user()
    ref_vals, ref_ok = _stable_rule_dcg(frame, scores, pscore, k)
    np.testing.assert_array_equal(ok, ref_ok)
    np.testing.assert_allclose(vals[ok], ref_vals[ref_ok], rtol=TOL)
    assert (~ok).sum() >= 1 and not dep.any()
    # the evaluator's frame is not touched
    assert fr.n_rows == users.shape[0]


def test_tie_rule_and_extremes(ev):
    evaluate, rt = ev
    rng = np.random.default_rng(3)
    users = rng.integers(0, 40, size=4000)
    labels = (rng.random(4000) < 0.3).astype(np.int64)
    pscore = rng.uniform(0.1, 1.0, 4000)
    scores = np.round(rng.random(4000), 1)  # eleven distinct values: ties everywhere
    scores[rng.integers(0, 4000, 50)] = 1.0
    scores[rng.integers(0, 4000, 50)] = 0.0
    frame = {"user": users, "label": labels, "pscore": pscore}
    fr = evaluate.DeviceValFrame(rt, users, labels, pscore, 5)
    fr.dcg(scores)
    vals, ok, dep = fr.per_user()
    ref_vals, ref_ok = _stable_rule_dcg(frame, scores, pscore, 5)
    np.testing.assert_array_equal(ok, ref_ok)
    np.testing.assert_allclose(vals[ok], ref_vals[ok], rtol=TOL)
    np.testing.assert_array_equal(dep, _order_dependent_users(frame, scores, pscore, 5))
    # infinities rank like numbers
    s2 = rng.random(4000)
    s2[::7] = np.inf
    s2[3::11] = -np.inf
    fr.dcg(s2)
    vals, ok, dep = fr.per_user()
    ref_vals, ref_ok = _stable_rule_dcg(frame, s2, pscore, 5)
    np.testing.assert_allclose(vals[ok], ref_vals[ok], rtol=TOL)
    np.testing.assert_array_equal(dep, _order_dependent_users(frame, s2, pscore, 5))
    # a NaN score is never ranked here (NumPy ranks it first): the user is reported
    s3 = rng.random(4000)
    s3[17] = np.nan
    _, n_dep = fr.dcg_checked(s3)
    assert n_dep == int(labels[users == users[17]].sum() > 0)


def test_degenerate_frames(ev):
    evaluate, rt = ev
    # nobody has a positive label: np.mean([]) is nan in the reference
    fr = evaluate.DeviceValFrame(rt, np.array([0, 0, 1]), np.zeros(3), np.ones(3), 5)
    assert np.isnan(fr.dcg(np.array([0.3, 0.2, 0.1])))
    # empty frame
    fr = evaluate.DeviceValFrame(rt, np.zeros(0, np.int64), np.zeros(0), np.zeros(0), 5)
    assert fr.n_segments == 0 and np.isnan(fr.dcg(np.zeros(0)))
    with pytest.raises(ValueError):
        evaluate.DeviceValFrame(rt, np.zeros(3, np.int64), np.zeros(3), np.zeros(3), 0)
    with pytest.raises(ValueError):
        evaluate.DeviceValFrame(rt, np.zeros(3, np.int64), np.zeros(2), np.zeros(3), 5)
    fr = evaluate.DeviceValFrame(rt, np.zeros(3, np.int64), np.ones(3), np.ones(3), 5)
    with pytest.raises(ValueError):
        fr.dcg(np.zeros(4))


class _ValEvaluatorLike:
    """The attributes of the reference's ValEvaluator (utils/evaluate.py:22-33,160-207)
    with the oracle's restatement as its evaluate()."""

    metric_name = "DCG"
    rfm_device_evaluator = True  # opts in: its evaluate() IS the reference's metric

    def __init__(self, frame, features, k=5, as_pandas=False):
        self.k = k
        self.features = features
        self._frame = frame
        self.calls = 0
        if as_pandas:
            import pandas as pd
            self.interaction_df = pd.DataFrame(frame)
        else:
            self.interaction_df = frame

    def evaluate(self, y_scores, estimator):
        self.calls += 1
        return cpu_ref.val_dcg(self._frame, y_scores, estimator, k=self.k)


def _unique_pair_frame(val_fm, val_mf):
    """The validation rows whose (user, item) pair occurs for the first time (repeats have
    identical features, hence exactly tied scores, and the synthetic labels differ)."""
    keep = synth.first_occurrences(val_mf["features"])
    frame = synth.interaction_frame({k: v[keep] for k, v in val_mf.items()}, val_mf["features"][keep])
    return keep, frame, (val_fm["features"][keep] if val_fm is not None else None)


@pytest.mark.parametrize("unique_pairs", [False, True])
@pytest.mark.parametrize("est", ["IPS", "Naive"])
def test_fm_fit_device_evaluator(ev, est, unique_pairs):
    """fit(evaluator=ValEvaluator-like) equals the host callback iteration by iteration --
    users whose value hangs on the order of tied scores (saturated scores, repeated pairs) are
    redone on the host with NumPy's own argsort -- and ends at the value the reference's
    ValEvaluator gives for the reference's final predictions (fixtures G7 /
    make_golden_eval.py)."""
    import relevance_factorizationmachine_amd as pkg
    g, g7, gd = load_golden("fm_kuairec_small_k16"), load_golden("val_dcg"), load_golden("val_dcg_distinct")
    sh = synth.SHAPES["kuairec_small"]
    train, val = synth.make_log(sh, "FM", est, seed=0)
    _, val_mf = synth.make_log(sh, "MF", est, seed=0)
    if unique_pairs:
        keep, frame, ev_X = _unique_pair_frame(val, val_mf)
        assert keep.shape[0] == int(gd[f"g2_unique_rows_{est}"])
        want = float(gd[f"g2_unique_val_dcg_{est}"])
    else:
        frame, ev_X = synth.interaction_frame(val_mf, val_mf["features"]), val["features"]
        want = float(g7[f"g2_val_dcg_{est}"])
    E = int(g["n_epochs"])
    kw = dict(estimator=est, n_epochs=E, n_factors=16, n_features=train["features"].shape[1],
              lr=float(g[f"{est}_lr"]), batch_size=2000, seed=12345)

    hook = _ValEvaluatorLike(frame, {"FM": ev_X}, as_pandas=(est == "IPS"))
    dev = pkg.FactorizationMachines(evaluator=hook, **kw)
    dev.fit(train, val)
    assert len(dev.val_metrics) == E
    assert hook.calls == 0 and dev.evaluator_host_calls <= E  # the evaluator object is never called
    assert dev.val_metrics[-1] == pytest.approx(want, rel=1e-9)

    hook2 = _ValEvaluatorLike(frame, {"FM": ev_X})
    host = pkg.FactorizationMachines(evaluator=hook2, **kw)
    host.device_evaluator = False
    host.fit(train, val)
    assert hook2.calls == E
    np.testing.assert_allclose(dev.val_metrics, host.val_metrics, rtol=TOL)
    np.testing.assert_allclose(dev.V(), host.V(), rtol=1e-11, atol=1e-14)  # the evaluator never touches training


def test_fit_device_evaluator_without_ties_stays_on_device(ev):
    """Small initial parameters: no saturated scores, unique pairs -> no host call at all."""
    import relevance_factorizationmachine_amd as pkg
    sh = synth.SHAPES["kuairec_small"]
    train, val = synth.make_log(sh, "FM", "IPS", seed=0)
    _, val_mf = synth.make_log(sh, "MF", "IPS", seed=0)
    _, frame, ev_X = _unique_pair_frame(val, val_mf)
    kw = dict(estimator="IPS", n_epochs=4, n_factors=16, n_features=train["features"].shape[1], lr=1e-4,
              batch_size=2000, seed=12345, alpha=0.05)
    hook = _ValEvaluatorLike(frame, {"FM": ev_X})
    dev = pkg.FactorizationMachines(evaluator=hook, **kw)
    dev.fit(train, val)
    assert hook.calls == 0 and dev.evaluator_host_calls == 0 and len(dev.val_metrics) == 4
    hook2 = _ValEvaluatorLike(frame, {"FM": ev_X})
    host = pkg.FactorizationMachines(evaluator=hook2, **kw)
    host.device_evaluator = False
    host.fit(train, val)
    np.testing.assert_allclose(dev.val_metrics, host.val_metrics, rtol=TOL)


def test_eval_loop_chunks(ev):
    """More iterations than one chunk of score slots holds."""
    import relevance_factorizationmachine_amd as pkg
    evaluate, rt = ev
    sh = synth.SHAPES["coat"]
    train, val = synth.make_log(sh, "FM", "IPS", seed=0)
    _, val_mf = synth.make_log(sh, "MF", "IPS", seed=0)
    frame = synth.interaction_frame(val_mf, val_mf["features"])
    kw = dict(estimator="IPS", n_epochs=7, n_factors=8, n_features=train["features"].shape[1], lr=1e-4,
              batch_size=500, seed=12345)
    old = evaluate.EvalLoop.CHUNK_BYTES
    evaluate.EvalLoop.CHUNK_BYTES = 3 * 8 * val["features"].shape[0]  # three iterations per chunk
    try:
        hook = _ValEvaluatorLike(frame, {"FM": val["features"]})
        dev = pkg.FactorizationMachines(evaluator=hook, **kw)
        dev.fit(train, val)
    finally:
        evaluate.EvalLoop.CHUNK_BYTES = old
    hook2 = _ValEvaluatorLike(frame, {"FM": val["features"]})
    host = pkg.FactorizationMachines(evaluator=hook2, **kw)
    host.device_evaluator = False
    host.fit(train, val)
    assert len(dev.val_metrics) == 7
    np.testing.assert_allclose(dev.val_metrics, host.val_metrics, rtol=TOL)


def test_mf_fit_device_evaluator(ev):
    import relevance_factorizationmachine_amd as pkg
    sh = synth.SHAPES["kuairec_small"]
    train, val = synth.make_log(sh, "MF", "IPS", seed=0)
    frame = synth.interaction_frame(val, val["features"])
    kw = dict(estimator="IPS", n_epochs=3, n_factors=16, n_users=sh.n_users, n_items=sh.n_items, lr=0.01,
              reg=0.5, batch_size=2000, seed=12345)
    hook = _ValEvaluatorLike(frame, {"MF": val["features"]})
    dev = pkg.LogisticMatrixFactorization(evaluator=hook, **kw)
    dev.fit(train, val)
    assert len(dev.val_metrics) == 3 and hook.calls == 0 and dev.evaluator_host_calls <= 3
    hook2 = _ValEvaluatorLike(frame, {"MF": val["features"]})
    host = pkg.LogisticMatrixFactorization(evaluator=hook2, **kw)
    host.device_evaluator = False
    host.fit(train, val)
    assert hook2.calls == 3
    np.testing.assert_allclose(dev.val_metrics, host.val_metrics, rtol=TOL)
    # unique pairs: nothing is tie-order dependent, nothing goes to the host
    keep, frame_u, _ = _unique_pair_frame(None, val)
    hook3 = _ValEvaluatorLike(frame_u, {"MF": val["features"][keep]})
    dev3 = pkg.LogisticMatrixFactorization(evaluator=hook3, **kw)
    dev3.fit(train, val)
    hook4 = _ValEvaluatorLike(frame_u, {"MF": val["features"][keep]})
    host4 = pkg.LogisticMatrixFactorization(evaluator=hook4, **kw)
    host4.device_evaluator = False
    host4.fit(train, val)
    assert hook3.calls == 0 and dev3.evaluator_host_calls == 0
    np.testing.assert_allclose(dev3.val_metrics, host4.val_metrics, rtol=TOL)


def test_unrecognised_evaluators_stay_host_callbacks(ev):
    evaluate, rt = ev
    frame = {"user": np.arange(4), "label": np.ones(4), "pscore": np.ones(4), "ones_pscore": np.ones(4)}

    class Other:
        k, metric_name, interaction_df, rfm_device_evaluator = 5, "Recall", frame, True

    class NoFrame:
        k, metric_name, rfm_device_evaluator = 5, "DCG", True

    class Short:
        k, metric_name, interaction_df, rfm_device_evaluator = 5, "DCG", {"user": np.arange(4)}, True

    class Unknown:  # the right attributes, but an evaluate() nobody vouches for
        k, metric_name, interaction_df = 5, "DCG", frame

        def evaluate(self, y_scores, estimator):
            return 0.0

    for obj in (Other(), NoFrame(), Short(), Unknown(), object()):
        assert evaluate.device_frame(rt, obj, "IPS", 4) is None

    class Good:
        k, metric_name, interaction_df, rfm_device_evaluator = 5, "DCG", frame, True

    assert evaluate.device_frame(rt, Good(), "IPS", 4) is not None
    assert evaluate.device_frame(rt, Good(), "IPS", 5) is None  # scores would not match the frame


# --------------------------------------------------------------------------
# test-set metrics: rfm_topk_users + DeviceTestEvaluator
# --------------------------------------------------------------------------
def test_device_test_evaluator_matches_reference_fixture(ev):
    """Every metric of the reference's TestEvaluator (outputs stored by make_golden_eval.py),
    with the per-user ranking done by rfm_topk_users."""
    import pandas as pd

    evaluate, rt = ev
    g = load_golden("val_dcg_distinct")
    K = [1, 3, 5, 7, 9]
    for prefix, users, used in (("test_", g["user"], {"DCG", "CatalogCoverage", "Recall", "MAP", "Gini"}),
                                ("short_test_", g["short_user"], {"CatalogCoverage", "DCG"})):
        df = pd.DataFrame({"user": users, "item": g["item"], "label": g["label"], "pscore": g["pscore"],
                           "ones_pscore": np.ones(len(users))})
        te = evaluate.DeviceTestEvaluator(interaction_df=df, features={}, K=K, used_metrics=used, n_items=400)
        res = te.evaluate(g["scores"])
        assert set(res) == {"ME"} | used and te.host_users == 0  # distinct scores: nothing redone
        for m in res:
            np.testing.assert_allclose(res[m], g[prefix + m], rtol=1e-13, atol=0, err_msg=prefix + m)
        np.testing.assert_array_equal(df["y_score"].to_numpy(), g["scores"])  # the reference's side effect
        # device-resident scores too
        res2 = te.evaluate(rt.upload(g["scores"]))
        assert {m: list(v) for m, v in res2.items()} == {m: list(v) for m, v in res.items()}
    with pytest.raises(ValueError):
        te.evaluate(g["scores"][:-1])


@pytest.mark.parametrize("seed", range(6))
def test_device_test_evaluator_with_ties_equals_numpy_ranking(ev, seed):
    """Tie-heavy scores (few distinct levels, saturated ones, NaNs), ragged users, users
    shorter than K, users without positives: the device ranks, flags the users whose first
    max(K) rows depend on the tie order, and those are ranked again with NumPy's own sort --
    the result is the oracle's restatement of TestEvaluator.evaluate, metric by metric."""
    evaluate, rt = ev
    rng = np.random.default_rng(100 + seed)
    n, n_users, n_items = 4000, [30, 400, 1500, 3, 800, 4000][seed], 90
    frame = {"user": rng.integers(0, n_users, size=n) * 3 + 1, "item": rng.integers(0, n_items + 5, size=n),
             "label": (rng.random(n) < [0.3, 0.05, 0.5, 0.3, 1.0, 0.2][seed]).astype(np.int64),
             "pscore": np.round(rng.uniform(0.1, 1.0, size=n), 1 if seed % 2 else 3)}
    levels = [5, 50, 3, 1000, 2, 7][seed]
    scores = rng.integers(0, levels, size=n) / levels
    scores[rng.random(n) < 0.1] = 1.0
    if seed == 3:
        scores[rng.integers(0, n, size=5)] = np.nan
    K, used = (1, 2, 5, 10), ("DCG", "CatalogCoverage", "Recall", "MAP", "Gini")
    te = evaluate.DeviceTestEvaluator(interaction_df=dict(frame), features={}, K=K, used_metrics=used,
                                      n_items=n_items)
    got = te.evaluate(scores)
    want = cpu_ref.test_metrics(frame, scores, K, used, n_items)
    for m in want:
        np.testing.assert_allclose(got[m], want[m], rtol=1e-12, atol=0, equal_nan=True, err_msg=m)
    # positions under the device's own rule (later row first among equal scores)
    pos, flags = te.topk(scores)
    fr = te.frame
    for u in rng.integers(0, fr.n_segments, size=40):
        lo, hi = fr.h_seg_ptr[u], fr.h_seg_ptr[u + 1]
        sc = scores[fr.h_order[lo:hi]]
        assert (flags[u] & 1) == int(fr.h_labels[lo:hi].sum() > 0)
        if np.isnan(sc).any():
            assert flags[u] & 2
            continue
        rank = np.argsort(sc, kind="stable")[::-1][: max(K)]
        np.testing.assert_array_equal(pos[u][: len(rank)], lo + rank)
        assert (pos[u][len(rank):] == -1).all()
