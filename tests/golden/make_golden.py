"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (``/root/reference`` does not exist on the GPU
box)::

    python tests/golden/make_golden.py

It imports ``src.fm`` / ``src.mf`` / ``utils.evaluate`` from ``/root/reference``
(nothing of the reference is copied: only inputs' digests and the reference's
numeric OUTPUTS are stored), feeds them the seeded synthetic logs of
``relevance_factorizationmachine_amd.synth`` and writes ``*.npz`` fixtures:

G1 fm_coat_k8            G2 fm_kuairec_small_k16   G3 fm_one_step_tiny
G4 mf_small              G5 batch_ids              G6 init_rng
G7 val_dcg               G8 logloss_cases          G9 sigmoid_edges
(SURVEY.md section 8c).
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import pandas as pd
from scipy.sparse import csr_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from relevance_factorizationmachine_amd import synth  # noqa: E402


def _import_reference():
    # the reference's packages are called ``src`` and ``utils``; this repo has
    # shims of the same name, so resolve the reference ones explicitly.
    for name in [m for m in sys.modules if m == "src" or m.startswith("src.") or m == "utils" or m.startswith("utils.")]:
        del sys.modules[name]
    sys.path.insert(0, REF)
    try:
        from src.fm import FactorizationMachines
        from src.mf import LogisticMatrixFactorization
        from utils.evaluate import TestEvaluator, ValEvaluator
        import src.fm as _fm
        assert _fm.__file__.startswith(REF), _fm.__file__
    finally:
        sys.path.remove(REF)
    return FactorizationMachines, LogisticMatrixFactorization, ValEvaluator, TestEvaluator


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def csr_digest(X: csr_matrix) -> str:
    return digest(X.indptr.astype(np.int64), X.indices.astype(np.int64), X.data)


def log_digest(train, val) -> str:
    parts = []
    for d in (train, val):
        f = d["features"]
        parts.append(csr_digest(f) if hasattr(f, "indptr") else digest(f))
        parts.append(digest(d["labels"], d["pscores"]))
    return "|".join(parts)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def fm_case(FM, shape_name, n_epochs, lrs, fixture):
    shape = synth.SHAPES[shape_name]
    out = {}
    for est in ("IPS", "Naive"):
        train, val = synth.make_log(shape, "FM", est, seed=0)
        model = FM(estimator=est, n_epochs=n_epochs, n_factors=shape.n_factors,
                   n_features=train["features"].shape[1], lr=lrs[est],
                   batch_size=shape.batch_size, seed=12345)
        tr, va = model.fit(train, val)
        out[f"{est}_train_loss"] = np.asarray(tr)
        out[f"{est}_val_loss"] = np.asarray(va)
        out[f"{est}_w0"] = model.w0().copy()
        out[f"{est}_w"] = model.w().copy()
        out[f"{est}_V"] = model.V().copy()
        out[f"{est}_pred_val"] = model.predict(X=val["features"])
        out[f"{est}_lr"] = np.float64(lrs[est])
        out[f"{est}_input_digest"] = np.array(log_digest(train, val))
    out["n_epochs"] = np.int64(n_epochs)
    out["seed"] = np.int64(12345)
    save(fixture, **out)
    return out


def main():
    FM, MF, ValEvaluator, TestEvaluator = _import_reference()

    # ---- G1: Coat-shaped, k=8, B=500, lr 1e-4, E=20 ------------------------
    fm_case(FM, "coat", 20, {"IPS": 1e-4, "Naive": 1e-4}, "fm_coat_k8")

    # ---- G2: KuaiRec-small-shaped, k=16, B=2000, E=10 ----------------------
    g2 = fm_case(FM, "kuairec_small", 10, {"IPS": 9e-6, "Naive": 3e-4}, "fm_kuairec_small_k16")

    # ---- G3: one hand-checkable step ---------------------------------------
    dense = np.array([
        [1.0, 0.0, 0.5, 0.0, -2.0],
        [0.0, 1.0, 0.5, 0.0, 0.0],
        [1.0, 0.0, 0.0, 3.0, 0.0],
        [0.0, 1.0, -1.5, 0.0, 0.25],
        [1.0, 0.0, 0.0, 0.0, 0.0],
        [0.0, 0.0, 2.0, 0.0, 1.0],
    ])
    X = csr_matrix(dense)
    X.indices = X.indices.astype(np.int32)
    X.indptr = X.indptr.astype(np.int32)
    y = np.array([1, 0, 1, 1, 0, 1], dtype=np.int64)
    p = np.array([0.5, 1.0, 0.25, 0.8, 1.0, 0.4])
    train = {"features": X, "labels": y, "pscores": p}
    model = FM(estimator="IPS", n_epochs=1, n_factors=3, n_features=5, lr=0.05, batch_size=6, seed=7)
    w0_0, w_0, V_0 = model.w0().copy(), model.w().copy(), model.V().copy()
    tr, va = model.fit(train, train)
    from sklearn.utils import resample
    Xb, yb, pb = resample(X, y, p, replace=False, n_samples=6, random_state=0)
    # residual with the initial parameters, through the reference's predict
    probe = FM(estimator="IPS", n_epochs=1, n_factors=3, n_features=5, lr=0.05, batch_size=6, seed=7)
    err = yb / pb - probe.predict(Xb)
    save("fm_one_step_tiny", dense=dense, y=y, p=p, lr=np.float64(0.05), seed=np.int64(7),
         w0_init=w0_0, w_init=w_0, V_init=V_0, error=err,
         g_w0=(w0_0 - model.w0()) / 0.05, g_w=(w_0 - model.w()) / 0.05, G_V=(V_0 - model.V()) / 0.05,
         w0_after=model.w0().copy(), w_after=model.w().copy(), V_after=model.V().copy(),
         train_loss=np.asarray(tr), val_loss=np.asarray(va))

    # ---- G4: MF small -------------------------------------------------------
    shape = synth.SHAPES["kuairec_small"]
    out = {}
    for est in ("IPS", "Naive"):
        train, val = synth.make_log(shape, "MF", est, seed=0)
        model = MF(estimator=est, n_epochs=3, n_factors=16, n_users=shape.n_users, n_items=shape.n_items,
                   lr=0.01, reg=0.5, batch_size=2000, seed=12345)
        tr, va = model.fit(train, val)
        out[f"{est}_train_loss"] = np.asarray(tr)
        out[f"{est}_val_loss"] = np.asarray(va)
        for nm in ("P", "Q", "b_u", "b_i"):
            out[f"{est}_{nm}"] = getattr(model, nm)().copy()
        out[f"{est}_b"] = np.float64(model.b)
        out[f"{est}_pred_val"] = model.predict(val["features"])
        out[f"{est}_input_digest"] = np.array(log_digest(train, val))
    save("mf_small", **out)

    # ---- G5: batch ids ------------------------------------------------------
    out = {}
    for n, epochs in ((3660, range(4)), (43036, range(4)), (10 ** 6, range(1))):
        for e in epochs:
            idx = resample(np.arange(n), replace=False, n_samples=32, random_state=e)
            out[f"n{n}_e{e}"] = np.asarray(idx, dtype=np.int64)
    save("batch_ids", **out)

    # ---- G6: init draw order -----------------------------------------------
    out = {}
    for n, k in ((637, 8), (4848, 16)):
        m = FM(estimator="IPS", n_epochs=0, n_factors=k, n_features=n, lr=0.1, batch_size=1, seed=12345)
        out[f"fm_n{n}_k{k}_w"] = m.w()[:8].copy()
        out[f"fm_n{n}_k{k}_V"] = m.V()[:2, :8].copy()
    for nu, ni, k in ((290, 300, 8), (1411, 3327, 16)):
        m = MF(estimator="IPS", n_epochs=0, n_factors=k, n_users=nu, n_items=ni, lr=0.1, reg=0.5,
               batch_size=1, seed=12345)
        out[f"mf_{nu}_{ni}_k{k}_P"] = m.P()[:2, :8].copy()
        out[f"mf_{nu}_{ni}_k{k}_Q"] = m.Q()[-2:, :8].copy()
        out[f"mf_{nu}_{ni}_k{k}_b_u"] = m.b_u()[:4].copy()
        out[f"mf_{nu}_{ni}_k{k}_b_i"] = m.b_i()[-4:].copy()
    save("init_rng", **out)

    # ---- G7: DCG checkers ---------------------------------------------------
    rng = np.random.default_rng(5)
    n = 2000
    frame = {
        "user": rng.integers(0, 120, size=n).astype(np.int64),
        "item": rng.integers(0, 400, size=n).astype(np.int64),
        "label": (rng.random(n) < 0.3).astype(np.int64),
        "pscore": rng.uniform(0.1, 1.0, size=n) ** 0.5,
    }
    frame["ones_pscore"] = np.ones(n)
    scores = rng.random(n)
    scores[rng.integers(0, n, size=200)] = 1.0  # saturated ties
    out = {k: v for k, v in frame.items()}
    out["scores"] = scores
    ve = ValEvaluator(interaction_df=pd.DataFrame(frame), features={}, k=5, metric_name="DCG")
    out["val_dcg_IPS"] = np.float64(ve.evaluate(y_scores=scores, estimator="IPS"))
    out["val_dcg_Naive"] = np.float64(ve.evaluate(y_scores=scores, estimator="Naive"))
    te = TestEvaluator(interaction_df=pd.DataFrame(frame), features={}, K=(1, 3, 5, 7, 9),
                       used_metrics={"DCG"}, n_items=400)
    out["test_dcg"] = np.asarray(te.evaluate(scores)["DCG"])
    # the same checker on G2's predictions over the (synthetic) val log
    for est in ("IPS", "Naive"):
        _, val = synth.make_log("kuairec_small", "MF", est, seed=0)
        fr = synth.interaction_frame(val, val["features"])
        ve2 = ValEvaluator(interaction_df=pd.DataFrame(fr), features={}, k=5, metric_name="DCG")
        out[f"g2_val_dcg_{est}"] = np.float64(ve2.evaluate(y_scores=g2[f"{est}_pred_val"], estimator=est))
    save("val_dcg", **out)

    # ---- G8 / G9: loss and sigmoid edge cases ------------------------------
    m = FM(estimator="IPS", n_epochs=0, n_factors=2, n_features=3, lr=0.1, batch_size=1, seed=0)
    y = np.array([1, 0, 1, 1, 0, 0, 1, 0], dtype=np.int64)
    s = np.array([0.9, 0.1, 0.0, 1.0, 1.0, 0.0, 0.5, 0.5])
    pp = np.array([0.2, 1.0, 0.5, 0.1, 1.0, 0.3, 1.0, 0.7])
    save("logloss_cases", y=y, scores=s, pscores=pp,
         loss=np.float64(m._cross_entropy_loss(y, s, pp)),
         loss_naive=np.float64(m._cross_entropy_loss(y, s, np.ones(8))),
         loss_first3=np.float64(m._cross_entropy_loss(y[:3], s[:3], pp[:3])))
    xs = np.array([0.0, 1.0, -1.0, 36.0, -36.0, 37.0, -37.0, 700.0, -700.0, 701.0, -701.0, 1e4, -1e4])
    save("sigmoid_edges", x=xs, y=m._sigmoid(xs))


if __name__ == "__main__":
    main()
