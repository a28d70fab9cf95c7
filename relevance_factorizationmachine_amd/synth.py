"""Seeded synthetic training logs shaped like the reference's loaders' output.

The KuaiRec / Coat datasets are git-ignored in the reference and absent here,
so every parity case, fixture and bench line runs on synthetic logs that keep
the *layout* the hot path sees (SURVEY.md section 8a row IN, section 8d):

* FM features: ``scipy.sparse.csr_matrix`` float64 data, int32 sorted indices,
  built as one-hot user (+) one-hot item (+) standardised reals (+) one-hot
  groups, mirroring ``utils/dataloader/kuairec/_feature.py:54-84`` and the
  column list of ``conf/setting/kuairec.yaml:16-44`` (Coat:
  ``utils/dataloader/coat/_preparer.py:154-168``).
* MF features: int64 ``(N, 2)`` ``[user, item]`` pairs
  (``utils/dataloader/kuairec/_preparer.py:151-154``).
* labels int64 in {0, 1} (1:1 negative sampling, ``_preparer.py:90-115``),
  pscores float64 = U(0.1, 1) ** pow_used for IPS (``_click.py:202``) or ones.

Nothing here is on the measured path; it only makes inputs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
from scipy.sparse import csr_matrix

# one-hot side blocks of the KuaiRec-shaped log (sizes from SURVEY.md 8d)
KUAIREC_USER_GROUPS = (4, 2, 7, 50, 7, 2, 2)
KUAIREC_N_TAGS = 31
# Coat-shaped side blocks: 14 user-feature columns, 33 item-feature columns
COAT_USER_GROUPS = (2, 6, 3, 3)
COAT_ITEM_GROUPS = (2, 16, 13, 2)


@dataclass(frozen=True)
class LogShape:
    """One workload of SURVEY.md section 8d."""

    name: str
    n_users: int
    n_items: int
    n_train: int
    n_val: int
    n_factors: int
    batch_size: int
    kind: str = "kuairec"  # or "coat"


SHAPES: Dict[str, LogShape] = {
    # C1: Coat 290x300, FM k=8, B=500 (conf/setting/coat.yaml:29)
    "coat": LogShape("coat", 290, 300, 3660, 842, 8, 500, "coat"),
    # C2: KuaiRec small_matrix 1411x3327, k=16, B=2000 (kuairec.yaml:52)
    "kuairec_small": LogShape("kuairec_small", 1411, 3327, 43036, 14308, 16, 2000),
    # C3: KuaiRec big_matrix-shaped 7176x10728 + side features, k=32
    "kuairec_big": LogShape("kuairec_big", 7176, 10728, 1_000_000, 100_000, 32, 2000),
    # C4: 1M x 100k, k=64
    "synthetic_1m": LogShape("synthetic_1m", 1_000_000, 100_000, 2_000_000, 100_000, 64, 2000),
}


def _zipf_items(rng: np.random.Generator, n: int, n_items: int) -> np.ndarray:
    return ((rng.zipf(1.3, size=n) - 1) % n_items).astype(np.int64)


def _labels_pscores(rng, n: int, estimator: str, pow_used: float):
    labels = (rng.random(n) < 0.5).astype(np.int64)
    if estimator == "IPS":
        pscores = rng.uniform(0.1, 1.0, size=n) ** pow_used
    elif estimator == "Naive":
        pscores = np.ones(n, dtype=np.float64)
    else:
        raise ValueError(f"estimator must be IPS or Naive, got {estimator!r}")
    return labels, pscores


def _kuairec_features(rng, users, items, n_users, n_items, tables) -> csr_matrix:
    n = users.shape[0]
    z = 2 + 1 + len(KUAIREC_USER_GROUPS) + 4 + 1
    cols = np.empty((n, z), dtype=np.int32)
    vals = np.ones((n, z), dtype=np.float64)
    c = 0
    off = 0
    cols[:, c] = users
    c += 1
    off += n_users
    cols[:, c] = off + items
    c += 1
    off += n_items
    # one standardised real per interaction (the "timestamp" column)
    cols[:, c] = off
    vals[:, c] = rng.standard_normal(n)
    c += 1
    off += 1
    # one-hot user groups, value keyed by the user
    for g, size in enumerate(KUAIREC_USER_GROUPS):
        cols[:, c] = off + tables["user_groups"][g][users]
        c += 1
        off += size
    # four standardised reals keyed by the item (item daily features)
    for r in range(4):
        cols[:, c] = off
        vals[:, c] = tables["item_reals"][r][items]
        c += 1
        off += 1
    # one of 31 tags keyed by the item
    cols[:, c] = off + tables["item_tag"][items]
    c += 1
    off += KUAIREC_N_TAGS
    assert c == z
    indptr = np.arange(0, n * z + 1, z, dtype=np.int32)
    X = csr_matrix((vals.ravel(), cols.ravel(), indptr), shape=(n, off))
    return X


def _coat_features(users, items, n_users, n_items, tables) -> csr_matrix:
    n = users.shape[0]
    z = 2 + len(COAT_USER_GROUPS) + len(COAT_ITEM_GROUPS)
    cols = np.empty((n, z), dtype=np.int32)
    c = 0
    off = 0
    cols[:, c] = users
    c += 1
    off += n_users
    cols[:, c] = off + items
    c += 1
    off += n_items
    for g, size in enumerate(COAT_USER_GROUPS):
        cols[:, c] = off + tables["user_groups"][g][users]
        c += 1
        off += size
    for g, size in enumerate(COAT_ITEM_GROUPS):
        cols[:, c] = off + tables["item_groups"][g][items]
        c += 1
        off += size
    vals = np.ones(n * z, dtype=np.float64)
    indptr = np.arange(0, n * z + 1, z, dtype=np.int32)
    return csr_matrix((vals, cols.ravel(), indptr), shape=(n, off))


def n_features_of(shape: LogShape) -> int:
    if shape.kind == "coat":
        return shape.n_users + shape.n_items + sum(COAT_USER_GROUPS) + sum(COAT_ITEM_GROUPS)
    return shape.n_users + shape.n_items + 1 + sum(KUAIREC_USER_GROUPS) + 4 + KUAIREC_N_TAGS


def make_log(
    shape: LogShape | str,
    model: str = "FM",
    estimator: str = "IPS",
    seed: int = 0,
    pow_used: float = 0.5,
    n_train: int | None = None,
    n_val: int | None = None,
) -> Tuple[dict, dict]:
    """Return ``(train, val)`` dicts with keys ``features/labels/pscores``.

    Same contract as ``DataLoader.load(model_name, estimator)``
    (``utils/dataloader/kuairec/loader.py:78-117``).  The (user, item) pairs,
    labels and propensities are identical for ``model="FM"`` and ``"MF"``.
    """
    if isinstance(shape, str):
        shape = SHAPES[shape]
    if model not in ("FM", "MF"):
        raise ValueError(f"model must be FM or MF, got {model!r}")
    rng = np.random.default_rng(seed)
    nu, ni = shape.n_users, shape.n_items
    if shape.kind == "coat":
        tables = {
            "user_groups": [rng.integers(0, s, size=nu) for s in COAT_USER_GROUPS],
            "item_groups": [rng.integers(0, s, size=ni) for s in COAT_ITEM_GROUPS],
        }
    else:
        tables = {
            "user_groups": [rng.integers(0, s, size=nu) for s in KUAIREC_USER_GROUPS],
            "item_reals": [rng.standard_normal(ni) for _ in range(4)],
            "item_tag": rng.integers(0, KUAIREC_N_TAGS, size=ni),
        }
    out = []
    for n in (shape.n_train if n_train is None else n_train, shape.n_val if n_val is None else n_val):
        users = rng.integers(0, nu, size=n).astype(np.int64)
        items = _zipf_items(rng, n, ni)
        if shape.kind == "coat":
            X = _coat_features(users, items, nu, ni, tables)
        else:
            X = _kuairec_features(rng, users, items, nu, ni, tables)
        labels, pscores = _labels_pscores(rng, n, estimator, pow_used)
        feats = X if model == "FM" else np.stack([users, items], axis=1)
        out.append({"features": feats, "labels": labels, "pscores": pscores})
    return out[0], out[1]


def interaction_frame(val: dict, pairs: np.ndarray) -> dict:
    """Columns a ``ValEvaluator`` frame holds (``utils/evaluate.py:222-239``):
    user, item, label, pscore, ones_pscore -- as plain arrays."""
    return {
        "user": pairs[:, 0].astype(np.int64),
        "item": pairs[:, 1].astype(np.int64),
        "label": val["labels"].astype(np.int64),
        "pscore": val["pscores"].astype(np.float64),
        "ones_pscore": np.ones(len(val["labels"]), dtype=np.float64),
    }


def first_occurrences(pairs: np.ndarray) -> np.ndarray:
    """Ascending row numbers of the first occurrence of every (user, item) pair."""
    pairs = np.asarray(pairs)
    key = pairs[:, 0].astype(np.int64) * (int(pairs[:, 1].max()) + 1) + pairs[:, 1]
    return np.sort(np.unique(key, return_index=True)[1])
