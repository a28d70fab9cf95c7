"""Golden vectors for the feature assembly, produced by RUNNING THE
REFERENCE's dataset preparers in the build container::

    python tests/golden/make_golden_features.py

``utils/dataloader/coat/_preparer.py`` (``_get_fm_features``, ``_nagative_sampling``) and
``utils/dataloader/kuairec/_preparer.py`` (``_negative_sample``, ``_prepare_fm_datasets``) are
importable here; ``kuairec/_feature.py`` is not (it imports omegaconf, which is not
installed), so the one-hot / standardise / multi-hot table builders are checked against the
pandas / scikit-learn calls that file makes (oracle ``feature_table``), not against its own
output: parity of those three helpers is unpinned by the reference.  Only numeric inputs and
the reference's OUTPUTS are stored.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import pandas as pd
from scipy import sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_preparers():
    for name in [m for m in sys.modules if m in ("utils", "conf") or m.startswith(("utils.", "conf."))]:
        del sys.modules[name]
    sys.path.insert(0, REF)
    try:
        from utils.dataloader.coat._preparer import DatasetPreparer as CoatPreparer
        from utils.dataloader.kuairec._preparer import DatasetPreparer as KuaiPreparer
        import utils.dataloader.coat._preparer as mod
        assert mod.__file__.startswith(REF), mod.__file__
    finally:
        sys.path.remove(REF)
    return CoatPreparer, KuaiPreparer


def main() -> None:
    CoatPreparer, KuaiPreparer = _import_preparers()
    rng = np.random.default_rng(2024)
    out = {}
    # ---- Coat: [one-hot user | user features | one-hot item | item features] ----------
    nu, ni, n = 37, 23, 400
    user_feats = (rng.random((nu, 9)) < 0.3).astype(np.float64)
    item_feats = (rng.random((ni, 6)) < 0.4).astype(np.float64) * rng.integers(1, 3, size=(ni, 6))
    df = pd.DataFrame({"user": rng.integers(0, nu, size=n), "item": rng.integers(0, ni, size=n),
                       "label": (rng.random(n) < 0.3).astype(np.int64)})
    prep = CoatPreparer(_seed=12345, _params=None, _pow_used=0.5)
    fm = prep._get_fm_features(dfs={"train": df}, onehot_user_ids=sp.identity(nu, format="csr"),
                               onehot_item_ids=sp.identity(ni, format="csr"),
                               user_features=sp.csr_matrix(user_feats), item_features=sp.csr_matrix(item_feats))
    X = fm["train"].tocsr()
    X.sort_indices()
    out.update(coat_user=df["user"].values, coat_item=df["item"].values, coat_label=df["label"].values,
               coat_user_feats=user_feats, coat_item_feats=item_feats, coat_shape=np.array(X.shape),
               coat_indptr=X.indptr.astype(np.int64), coat_indices=X.indices.astype(np.int64), coat_data=X.data,
               coat_sampled=np.asarray(prep._nagative_sampling(df), dtype=np.int64))
    # ---- KuaiRec: 1:1 negative sampling and features[indices] --------------------------
    m = 5000
    kdf = pd.DataFrame({"label": (rng.random(m) < 0.2).astype(np.int64)})
    kp = KuaiPreparer(_seed=12345)
    out["kuai_label"] = kdf["label"].values
    out["kuai_sampled"] = np.asarray(kp._negative_sample(df=kdf), dtype=np.int64)
    out["kuai_sampled_x2"] = np.asarray(kp._negative_sample(df=kdf, negative_multiple=2), dtype=np.int64)
    feats = sp.random(m, 50, density=0.1, format="csr", random_state=np.random.default_rng(7))
    picked = kp._prepare_fm_datasets(features=feats, feature_indices={"val": out["kuai_sampled"]})["val"].tocsr()
    out.update(kuai_feat_indptr=feats.indptr.astype(np.int64), kuai_feat_indices=feats.indices.astype(np.int64),
               kuai_feat_data=feats.data, kuai_pick_indptr=picked.indptr.astype(np.int64),
               kuai_pick_indices=picked.indices.astype(np.int64), kuai_pick_data=picked.data)
    path = os.path.join(HERE, "feature_assembly.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB):", {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
