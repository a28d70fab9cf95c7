"""Golden vectors for the device evaluator, produced by RUNNING THE
REFERENCE's ValEvaluator / TestEvaluator in the build container::

    python tests/golden/make_golden_eval.py

G7 (val_dcg.npz) carries saturated ties, whose ranking NumPy's default (unstable,
CPU-dependent) sort decides; no fixed rule reproduces that.  This fixture holds the
same frame with DISTINCT scores, and a tied variant in which every tie is between
rows of equal (label, pscore), so the reference's value does not depend on the order
it leaves ties in.  Only numeric inputs and the reference's OUTPUTS are stored.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from make_golden import _import_reference  # noqa: E402


def main() -> None:
    _, _, ValEvaluator, TestEvaluator = _import_reference()
    rng = np.random.default_rng(11)
    n = 6000
    sizes_user = rng.integers(0, 150, size=n)
    sizes_user[:1500] = 7  # one long group (ranking past one wavefront's first trip)
    frame = {
        "user": sizes_user.astype(np.int64) * 5 + 2,
        "item": rng.integers(0, 400, size=n).astype(np.int64),
        "label": (rng.random(n) < 0.25).astype(np.int64),
        "pscore": rng.uniform(0.1, 1.0, size=n) ** 0.5,
    }
    frame["label"][frame["user"] == frame["user"][-1]] = 0  # a user the metric leaves out
    frame["ones_pscore"] = np.ones(n)
    scores = rng.permutation(n) / n + rng.random(n) / (4 * n)  # distinct
    out = dict(frame)
    out["scores"] = scores
    df = pd.DataFrame(frame)
    for k in (1, 3, 5, 10):
        ve = ValEvaluator(interaction_df=df.copy(), features={}, k=k, metric_name="DCG")
        for est in ("IPS", "Naive"):
            out[f"val_dcg_{est}_k{k}"] = np.float64(ve.evaluate(y_scores=scores, estimator=est))
    te = TestEvaluator(interaction_df=df.copy(), features={}, K=(1, 3, 5, 7, 9), used_metrics={"DCG"},
                       n_items=400)
    out["test_dcg"] = np.asarray(te.evaluate(scores)["DCG"])
    # every metric TestEvaluator knows (utils/metrics.py:169-178) on the same frame and scores
    te_all = TestEvaluator(interaction_df=df.copy(), features={}, K=(1, 3, 5, 7, 9),
                           used_metrics={"DCG", "CatalogCoverage", "Recall", "MAP", "Gini"}, n_items=400)
    res = te_all.evaluate(scores)
    assert set(res) == {"ME", "DCG", "CatalogCoverage", "Recall", "MAP", "Gini"}
    for name, vals in res.items():
        out[f"test_{name}"] = np.asarray(vals, dtype=np.float64)
    # the drivers' own configuration (main_kuairec.py:76-82) on a frame of short users, where
    # ME@K of a user with fewer than K rows is nan
    short = rng.integers(0, 1200, size=n).astype(np.int64)
    fr3 = dict(frame, user=short)
    te3 = TestEvaluator(interaction_df=pd.DataFrame(fr3), features={}, K=[1, 3, 5, 7, 9],
                        used_metrics={"CatalogCoverage", "DCG"}, n_items=400)
    res3 = te3.evaluate(scores)
    out["short_user"] = short
    for name, vals in res3.items():
        out[f"short_test_{name}"] = np.asarray(vals, dtype=np.float64)

    # ties that cannot matter: label and pscore are functions of the (rounded) score
    tied = np.round(scores, 2)
    lab2 = (np.floor(tied * 100).astype(np.int64) % 3 == 0).astype(np.int64)
    ps2 = 0.2 + 0.7 * tied
    fr2 = dict(frame, label=lab2, pscore=ps2)
    ve = ValEvaluator(interaction_df=pd.DataFrame(fr2), features={}, k=5, metric_name="DCG")
    out["tied_scores"], out["tied_label"], out["tied_pscore"] = tied, lab2, ps2
    out["tied_val_dcg_IPS"] = np.float64(ve.evaluate(y_scores=tied, estimator="IPS"))
    out["tied_val_dcg_Naive"] = np.float64(ve.evaluate(y_scores=tied, estimator="Naive"))
    # G2's final predictions over the rows of the (synthetic) validation log whose
    # (user, item) pair occurs for the first time: a repeated pair has identical features,
    # hence identical scores, and the log's labels differ between the repeats
    from relevance_factorizationmachine_amd import synth
    g2 = np.load(os.path.join(HERE, "fm_kuairec_small_k16.npz"))
    for est in ("IPS", "Naive"):
        _, val = synth.make_log("kuairec_small", "MF", est, seed=0)
        keep = synth.first_occurrences(val["features"])
        fr = synth.interaction_frame({k2: v[keep] for k2, v in val.items()}, val["features"][keep])
        ve2 = ValEvaluator(interaction_df=pd.DataFrame(fr), features={}, k=5, metric_name="DCG")
        out[f"g2_unique_val_dcg_{est}"] = np.float64(
            ve2.evaluate(y_scores=g2[f"{est}_pred_val"][keep], estimator=est))
        out[f"g2_unique_rows_{est}"] = np.int64(keep.shape[0])
    np.savez_compressed(os.path.join(HERE, "val_dcg_distinct.npz"), **out)
    print("wrote val_dcg_distinct.npz:", {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


if __name__ == "__main__":
    main()
