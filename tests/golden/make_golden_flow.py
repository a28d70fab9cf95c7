"""Golden vectors of the DRIVER FLOW, by RUNNING THE REFERENCE's classes (build container only)::

    python tests/golden/make_golden_flow.py

The reference's drivers cannot run here (hydra / omegaconf / the datasets are absent), so this
replays their loop bodies with the reference's own ``src.fm`` / ``src.mf`` /
``utils.evaluate.{ValEvaluator, TestEvaluator}`` on the C2-shaped synthetic split:

  1. ``utils/search_params.py:79-123``: for FM, MF x IPS, Naive -- fit ``max_epoch`` iterations
     with the ``ValEvaluator`` on (every iteration scores the full validation frame), take
     ``best_epoch = argmax(val_metrics)``;
  2. ``main_kuairec.py:85-125``: a fresh model with ``n_epochs = best_epoch`` (the drivers'
     off-by-one included), ``fit``, ``predict(X=evaluator.features[model])`` on the test frame,
     ``TestEvaluator.evaluate`` -> the columns of ``metric.csv``; then the Random baseline
     (``main_kuairec.py:127-134``).

Deviations from the drivers' configuration, on purpose: ``max_epoch`` 30 instead of 500 and
n_factors 8 instead of 400 (fixture size / time); FM is built with ``alpha=0.25`` (a constructor
field, ``src/fm.py:28``; the drivers leave the default 2.0).  With the default, a fifth of the
FM scores saturate at exactly 1.0 and the reference's DCG then depends on the order in which
NumPy's unstable default sort leaves tied rows (``utils/evaluate.py:93,197``) -- CPU-dependent,
not pinnable by a fixture.  The script ASSERTS that no evaluation it stores has two equal scores
inside one user.  Evaluation frames hold the first occurrence of every (user, item) pair
(repeats have identical features, hence tied scores).  Only the reference's numeric OUTPUTS
and the seeds / hyper-parameters are stored.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import _import_reference  # noqa: E402
from relevance_factorizationmachine_amd import synth  # noqa: E402

SHAPE, K_FACTORS, BATCH, MAX_EPOCH, SEED, FM_ALPHA = "kuairec_small", 8, 2000, 30, 12345, 0.25
LR = {"FM": {"IPS": 1e-4, "Naive": 3e-4}, "MF": {"IPS": 0.01, "Naive": 0.03}}  # (MF: kuairec.yaml:53-59)
REG = 0.5
TOP_K = [1, 3, 5, 7, 9]  # main_kuairec.py: K


def frames(seed: int):
    """(interaction frame, {"FM": csr, "MF": pairs}) of an evaluation split: first occurrences."""
    _, fm = synth.make_log(SHAPE, "FM", "IPS", seed=seed)
    _, mf = synth.make_log(SHAPE, "MF", "IPS", seed=seed)
    keep = synth.first_occurrences(mf["features"])
    frame = synth.interaction_frame({k: v[keep] for k, v in mf.items()}, mf["features"][keep])
    return frame, {"FM": fm["features"][keep], "MF": mf["features"][keep]}


def assert_no_ties(users, scores, what):
    order = np.lexsort((scores, users))
    u, s = users[order], scores[order]
    assert not np.any((u[1:] == u[:-1]) & (s[1:] == s[:-1])), f"{what}: tied scores inside a user"


def main() -> None:
    FM, MF, ValEvaluator, TestEvaluator = _import_reference()
    shape = synth.SHAPES[SHAPE]
    val_frame, val_feats = frames(1)
    test_frame, test_feats = frames(2)
    out = {"n_factors": np.int64(K_FACTORS), "batch_size": np.int64(BATCH), "max_epoch": np.int64(MAX_EPOCH),
           "seed": np.int64(SEED), "fm_alpha": np.float64(FM_ALPHA), "reg": np.float64(REG),
           "val_rows": np.int64(len(val_frame["user"])), "test_rows": np.int64(len(test_frame["user"]))}

    def build(model_name, est, n_epochs, evaluator=None):
        train, val = synth.make_log(shape, model_name, est, seed=0)
        if model_name == "FM":
            m = FM(estimator=est, n_epochs=n_epochs, n_factors=K_FACTORS, n_features=train["features"].shape[1],
                   lr=LR["FM"][est], batch_size=BATCH, seed=SEED, alpha=FM_ALPHA, evaluator=evaluator)
        else:
            m = MF(estimator=est, n_epochs=n_epochs, n_factors=K_FACTORS, n_users=shape.n_users,
                   n_items=shape.n_items, lr=LR["MF"][est], reg=REG, batch_size=BATCH, seed=SEED,
                   evaluator=evaluator)
        return m, train, val

    te = TestEvaluator(interaction_df=pd.DataFrame(test_frame), features=test_feats, n_items=shape.n_items,
                       used_metrics={"DCG", "CatalogCoverage"}, K=TOP_K)
    metric_columns = {}
    for model_name in ("FM", "MF"):
        for est in ("IPS", "Naive"):
            base = f"{model_name}_{est}"
            out[f"{base}_lr"] = np.float64(LR[model_name][est])
            # ---- search_params.py:79-123 ------------------------------------------------
            ve = ValEvaluator(interaction_df=pd.DataFrame(val_frame), features=val_feats, k=5, metric_name="DCG")
            inner = ve.evaluate

            def checked(y_scores, estimator, _inner=inner, _base=base):
                assert_no_ties(val_frame["user"], np.asarray(y_scores), _base + " validation")
                return _inner(y_scores=y_scores, estimator=estimator)

            ve.evaluate = checked
            m, train, val = build(model_name, est, MAX_EPOCH, ve)
            tl, vl = m.fit(train, val)
            best = int(np.argmax(m.val_metrics))
            out[f"{base}_val_metrics"] = np.asarray(m.val_metrics, dtype=np.float64)
            out[f"{base}_search_train_loss"] = np.asarray(tl)
            out[f"{base}_search_val_loss"] = np.asarray(vl)
            out[f"{base}_best_epoch"] = np.int64(best)
            # ---- main_kuairec.py:85-125 ---------------------------------------------------
            m2, train, val = build(model_name, est, best)
            m2.fit(train, val)
            pred = m2.predict(X=te.features[model_name])
            assert_no_ties(test_frame["user"], pred, base + " test")
            results = te.evaluate(pred)
            out[f"{base}_test_pred"] = np.asarray(pred)
            for metric_name, values in results.items():
                metric_columns[f"{base}_{metric_name}@K"] = np.asarray(values, dtype=np.float64)
            print(base, "best epoch", best, "val DCG@5", m.val_metrics[best], flush=True)
    np.random.seed(SEED)  # main_kuairec.py:127-134
    rnd = np.random.uniform(0, 1, size=len(test_frame["user"]))
    for metric_name, values in te.evaluate(y_scores=rnd).items():
        metric_columns[f"Random_{metric_name}@K"] = np.asarray(values, dtype=np.float64)
    out["metric_columns"] = np.array(sorted(metric_columns))
    for name, vals in metric_columns.items():
        out["metric_" + name] = vals
    path = os.path.join(HERE, "driver_flow.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); columns: {sorted(metric_columns)}")


if __name__ == "__main__":
    main()
