#!/usr/bin/env python3
"""Where a fit() at config-3 size spends its wall time (one-time set-up vs per-iteration)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from relevance_factorizationmachine_amd import synth
from relevance_factorizationmachine_amd.fm import FactorizationMachines, FmPlan
from relevance_factorizationmachine_amd.runtime import BatchIdStream, DeviceCSR, Runtime, sample_batches

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
shape = synth.SHAPES["kuairec_big"]
train, val = synth.make_log(shape, "FM", "IPS", seed=0)
rt = Runtime.get(0)
X = train["features"]
def T(label, fn):
    rt.sync(); t0 = time.perf_counter(); r = fn(); rt.sync(); print(f"{label}: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True); return r
m = T("model init (host RNG + upload)", lambda: FactorizationMachines(estimator="IPS", n_epochs=200, n_factors=32, lr=9e-6, batch_size=B, seed=12345, n_features=X.shape[1]))
tr = T("DeviceCSR(train)", lambda: DeviceCSR(rt, X))
va = T("DeviceCSR(val)", lambda: DeviceCSR(rt, val["features"]))
plan = T("FmPlan build", lambda: FmPlan(rt, tr, train["labels"], train["pscores"], 32, B))
ids = T("sampler, 200 iterations", lambda: sample_batches(X.shape[0], B, 0, 200))
T("upload ids", lambda: rt.upload(ids))
m2 = FactorizationMachines(estimator="IPS", n_epochs=200, n_factors=32, lr=9e-6, batch_size=B, seed=12345, n_features=X.shape[1])
T("fit(200 it) total", lambda: m2.fit(train, val))
