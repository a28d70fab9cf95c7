"""Golden vectors at the reference's PUBLISHED operating point, by RUNNING THE REFERENCE.

Build container only (``/root/reference`` does not exist on the GPU box)::

    python tests/golden/make_golden_published.py            # all cases, one process each
    python tests/golden/make_golden_published.py --case fm_kuairec_k400_IPS

Every published run of the reference is k=400 / B=2000 (KuaiRec) or k=300 / B=500 (Coat) for
85-494 mini-batch steps (``conf/setting/kuairec.yaml:50-59``, ``conf/setting/coat.yaml:27-36``,
``data/best_params/{kuairec,coat}/*.json``).  The datasets are absent, so the inputs are the
seeded synthetic logs of ``relevance_factorizationmachine_amd.synth`` with the published
hyper-parameters and iteration counts:

  fm_kuairec_k400_IPS    C2-shaped log, FM k=400, B=2000, lr 9e-6, 221 steps (FM_IPS.json)
  fm_kuairec_k400_Naive  same log, Naive, lr 3e-4, 494 steps (FM_Naive.json)
  fm_coat_k300_IPS       Coat-shaped log, FM k=300, B=500, lr 1e-4, 401 steps (coat/FM_IPS.json)
  fm_coat_k300_Naive     same, Naive, lr 2e-4, 386 steps (coat/FM_Naive.json)
  mf_kuairec_k400_IPS    C2-shaped pairs, MF k=400, B=2000, lr 0.01, reg 0.5, 60 steps
  mf_coat_k300_IPS       Coat-shaped pairs, MF k=300, B=500, lr 0.02, reg 0.5, 85 steps (coat/MF_IPS.json)

Only the reference's numeric OUTPUTS are stored (``published_<case>.npz``): both loss curves,
``predict(val)``, scalars, and -- the parameter matrices being 1.5-15 MB -- every ``stride``-th
row of each matrix plus the row sums, column sums and the sum of squares of the WHOLE matrix, so
that every element is covered by at least one stored number.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import _import_reference, log_digest  # noqa: E402
from relevance_factorizationmachine_amd import synth  # noqa: E402

CASES = {
    # name: (model, shape, estimator, k, batch, lr, reg, steps, row stride of the stored matrices)
    "fm_kuairec_k400_IPS": ("FM", "kuairec_small", "IPS", 400, 2000, 9e-6, None, 221, 16),
    "fm_kuairec_k400_Naive": ("FM", "kuairec_small", "Naive", 400, 2000, 3e-4, None, 494, 16),
    "fm_coat_k300_IPS": ("FM", "coat", "IPS", 300, 500, 1e-4, None, 401, 4),
    "fm_coat_k300_Naive": ("FM", "coat", "Naive", 300, 500, 2e-4, None, 386, 4),
    "mf_kuairec_k400_IPS": ("MF", "kuairec_small", "IPS", 400, 2000, 0.01, 0.5, 60, 16),
    "mf_coat_k300_IPS": ("MF", "coat", "IPS", 300, 500, 0.02, 0.5, 85, 4),
}


def matrix_summary(prefix: str, M: np.ndarray, stride: int) -> dict:
    """Every stride-th row, and sums that cover all elements."""
    M = np.asarray(M, dtype=np.float64)
    return {f"{prefix}_rows": M[::stride].copy(), f"{prefix}_stride": np.int64(stride),
            f"{prefix}_rowsum": M.sum(axis=1), f"{prefix}_colsum": M.sum(axis=0),
            f"{prefix}_sqsum": np.float64((M * M).sum()), f"{prefix}_shape": np.array(M.shape, dtype=np.int64)}


def run_case(name: str) -> None:
    model_kind, shape_name, est, k, batch, lr, reg, steps, stride = CASES[name]
    FM, MF, _, _ = _import_reference()
    shape = synth.SHAPES[shape_name]
    train, val = synth.make_log(shape, model_kind, est, seed=0)
    t0 = time.perf_counter()
    out = {"n_epochs": np.int64(steps), "seed": np.int64(12345), "lr": np.float64(lr),
           "n_factors": np.int64(k), "batch_size": np.int64(batch),
           "input_digest": np.array(log_digest(train, val))}
    if model_kind == "FM":
        m = FM(estimator=est, n_epochs=steps, n_factors=k, n_features=train["features"].shape[1], lr=lr,
               batch_size=batch, seed=12345)
        tr, va = m.fit(train, val)
        out.update(matrix_summary("V", m.V(), stride))
        out["w"] = m.w().copy()
        out["w0"] = m.w0().copy()
        out["pred_val"] = m.predict(X=val["features"])
    else:
        m = MF(estimator=est, n_epochs=steps, n_factors=k, n_users=shape.n_users, n_items=shape.n_items, lr=lr,
               reg=reg, batch_size=batch, seed=12345)
        tr, va = m.fit(train, val)
        out.update(matrix_summary("P", m.P(), stride))
        out.update(matrix_summary("Q", m.Q(), stride))
        out["b_u"] = m.b_u().copy()
        out["b_i"] = m.b_i().copy()
        out["b"] = np.float64(m.b)
        out["reg"] = np.float64(reg)
        out["pred_val"] = m.predict(val["features"])
    out["train_loss"] = np.asarray(tr, dtype=np.float64)
    out["val_loss"] = np.asarray(va, dtype=np.float64)
    out["reference_fit_seconds"] = np.float64(time.perf_counter() - t0)
    path = os.path.join(HERE, f"published_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB) in {time.perf_counter() - t0:.0f} s", flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", choices=sorted(CASES))
    args = ap.parse_args()
    if args.case:
        run_case(args.case)
        return
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--case", name]) for name in CASES]
    rc = [p.wait() for p in procs]
    if any(rc):
        raise SystemExit(f"cases failed: {dict(zip(CASES, rc))}")


if __name__ == "__main__":
    main()
