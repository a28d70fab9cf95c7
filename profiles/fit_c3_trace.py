"""GPU box, under rocprofv3 --kernel-trace --stats: FactorizationMachines.fit() on the config-3 log
(N = 1 M, k = 32, validation 100 000 rows) at B = 2 000, 400 iterations, twice (the second fit is
the warm one) -- which kernels an iteration is made of.   usage: rocprofv3 ... -- python3 profiles/fit_c3_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relevance_factorizationmachine_amd import synth
from relevance_factorizationmachine_amd.fm import FactorizationMachines
B = int(os.environ.get("FIT_BATCH", "2000"))
train, val = synth.make_log("kuairec_big", "FM", "IPS", seed=0)
kw = dict(estimator="IPS", n_factors=32, lr=9e-6, seed=12345, n_features=train["features"].shape[1], batch_size=B)
for rep in range(2):
    m = FactorizationMachines(n_epochs=400, **kw)
    t0 = time.perf_counter(); m.fit(train, val); print("fit wall ms/it", 1e3 * (time.perf_counter() - t0) / 400)
