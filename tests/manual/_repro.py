import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_parity import _random_log, _fm
from oracle import cpu_ref
import relevance_factorizationmachine_amd as pkg
k, density, dense_cols, hot = 32, 0.25, 4, -1
rng = np.random.default_rng(7 * k + hot)
train = _random_log(rng, 45000, 120, density, dense_cols); val = _random_log(rng, 400, 120, density, dense_cols)
for batch in (40000, 1000):
    model = _fm(pkg, n_factors=k, n_features=120, lr=2e-6, batch_size=batch, n_epochs=1, seed=5)
    model.hot_min_count = hot
    model.fit(train, val)
    ref = cpu_ref.fm_fit(train, val, n_epochs=1, n_factors=k, lr=2e-6, batch_size=batch, seed=5)
    d = np.abs(model.V() - ref["V"]).max(axis=1)
    print("batch", batch, "max err", d.max(), "bad cols", np.flatnonzero(d > 1e-9)[:20], "w err", np.abs(model.w()-ref["w"]).max())
    lens = np.diff(train["features"].tocsc().indptr)
    print(" lens of bad", lens[np.flatnonzero(d > 1e-9)[:10]], "all lens min/max", lens.min(), lens.max())
