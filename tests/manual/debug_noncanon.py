import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scipy.sparse import csr_matrix
import relevance_factorizationmachine_amd as pkg
from oracle import cpu_ref
rng = np.random.default_rng(11)
n_rows, n_cols, z = 700, 60, 9
indptr = np.arange(0, n_rows * z + 1, z, dtype=np.int64)
indices = rng.integers(0, n_cols, size=n_rows * z).astype(np.int64)
data = rng.standard_normal(n_rows * z)
data[rng.integers(0, data.size, size=300)] = 0.0
def log(m):
    X = csr_matrix((data[: m * z].copy(), indices[: m * z].copy(), indptr[: m + 1].copy()), shape=(m, n_cols))
    return {"features": X, "labels": (rng.random(m) < 0.5).astype(np.int64), "pscores": rng.uniform(0.1, 1.0, size=m) ** 0.5}
train, val = log(700), log(200)
for variant in ("dups", "nodups"):
    if variant == "nodups":
        X = train["features"].copy(); X.sum_duplicates(); train = dict(train, features=X)
    for E in (1, 5):
        kw = dict(n_epochs=E, n_factors=6, lr=1e-3, batch_size=256, seed=5)
        for hot in (0, -1, 2):
            m = pkg.FactorizationMachines(estimator="IPS", n_features=n_cols, **kw); m.hot_min_count = hot
            tr, va = m.fit(train, val)
            ref = cpu_ref.fm_fit(train, val, **kw)
            dV = np.abs(m.V() - ref["V"]).max(axis=1); dw = np.abs(m.w() - ref["w"])
            bad = np.flatnonzero(dV > 1e-9)
            print(variant, "E", E, "hot", hot, "maxdV %.2e maxdw %.2e dw0 %.2e" % (dV.max(), dw.max(), abs(m.w0(0) - ref["w0"][0])),
                  "bad cols", bad[:12], "of", len(bad), "loss", abs(tr[0] - ref["train_loss"][0]))
