"""GPU box: host-side profile (cProfile) of a WARM fit() on the config-3 log (N = 1 M, k = 32, B = 2 000,
200 iterations): 12.3 ms = 2 ms of content hashes of the split (before any launch) + the GPU's 9.2 ms
(46 us per iteration: forward with the riding train-loss rows 11.6, gradient launch 8.1, validation
forward 26.2; profiles/fit_c3_trace.py) + read-back.   usage: python profiles/fit_host_profile_c3.py"""
import cProfile, pstats, time, sys, os
sys.path.insert(0, os.getcwd())
from relevance_factorizationmachine_amd import synth
from relevance_factorizationmachine_amd.fm import FactorizationMachines
train, val = synth.make_log("kuairec_big", "FM", "IPS", seed=0)
kw = dict(estimator="IPS", n_factors=32, lr=9e-6, seed=12345, n_features=train["features"].shape[1], batch_size=2000)
for rep in range(2):
    m = FactorizationMachines(n_epochs=200, **kw); t0 = time.perf_counter(); m.fit(train, val); print("wall ms", 1e3*(time.perf_counter()-t0))
m = FactorizationMachines(n_epochs=200, **kw)
pr = cProfile.Profile(); pr.enable(); m.fit(train, val); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
