"""GPU box: where the sliced loss forward starts to pay (RFM_SLICED_MIN_ROWS, default 4 096 rows of
batch + validation log): warm fit() at the published point (k = 400, B = 2 000) with validation logs
of 500 .. 14 308 rows, the sliced forward forced on and off.   usage: python profiles/sliced_threshold.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relevance_factorizationmachine_amd import synth
from relevance_factorizationmachine_amd.fm import FactorizationMachines

train, val = synth.make_log("kuairec_small", "FM", "IPS", seed=0)
kw = dict(estimator="IPS", n_factors=400, lr=9e-6, seed=12345, n_features=train["features"].shape[1], batch_size=2000)
os.environ["RFM_SLICED_MIN_ROWS"] = "1"
for n_val in (500, 1000, 2000, 4000, 8000, 14308):
    v = {k: x[:n_val] for k, x in val.items()}
    row = []
    for mode in ("0", "1"):
        os.environ["RFM_SLICED_LOSS"] = mode
        for rep in range(2):
            m = FactorizationMachines(n_epochs=300, **kw)
            t0 = time.perf_counter(); m.fit(train, v); w = time.perf_counter() - t0
        row.append(1e3 * w / 300)
    print(f"batch 2000 + validation {n_val:6d} rows: plain {row[0]:.4f} ms/it, sliced {row[1]:.4f} ms/it")
