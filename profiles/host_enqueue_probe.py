#!/usr/bin/env python3
"""Is fit()'s loop bound by the host's enqueue rate?  Times the return of one rfm_fm_train call
(all launches enqueued) against the drain of the stream, with and without the loss forwards
(GPU box; config 3, B = 2 000)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from relevance_factorizationmachine_amd import _lib, synth
from relevance_factorizationmachine_amd.fm import FactorizationMachines, FmPlan
from relevance_factorizationmachine_amd.runtime import DeviceCSR, Runtime, sample_batches

B, K = int(os.environ.get("PROBE_BATCH", "2000")), 400
sh = synth.SHAPES["kuairec_big"]
train, val = synth.make_log(sh, "FM", "IPS", seed=0)
rt = Runtime.get(0)
X = train["features"]; n = X.shape[1]
m = FactorizationMachines(estimator="IPS", n_epochs=1, n_factors=32, lr=9e-6, batch_size=B, seed=12345, n_features=n)
csr, vcsr = DeviceCSR(rt, X), DeviceCSR(rt, val["features"])
y, p = rt.upload(train["labels"], dtype=np.float64), rt.upload(train["pscores"], dtype=np.float64)
vy, vp = rt.upload(val["labels"], dtype=np.float64), rt.upload(val["pscores"], dtype=np.float64)
plan = FmPlan(rt, csr, y, p, 32, B)
ids = rt.upload(sample_batches(X.shape[0], B, 0, K))
tl, vl = rt.empty((K,), torch.float64), rt.empty((K,), torch.float64)
par = (m.w0.dev.data_ptr(), m.w.dev.data_ptr(), m.V.dev.data_ptr())
tr = (csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr(), y.data_ptr(), p.data_ptr())
va = (vcsr.indptr.data_ptr(), vcsr.indices.data_ptr(), vcsr.values.data_ptr(), vy.data_ptr(), vp.data_ptr())
for name, losses in (("steps only", False), ("steps + both loss forwards", True), ("steps + both loss forwards", True)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(rt.lib.rfm_fm_train(rt.ctx, plan.handle, *tr, ids.data_ptr(), B, K, *par, 9e-6,
                                   *(va if losses else (None,) * 5), vcsr.shape[0] if losses else 0, 1e-8,
                                   tl.data_ptr() if losses else None, vl.data_ptr() if losses else None))
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: call returned after {1e6*(t1-t0)/K:.1f} us/it, stream drained after {1e6*(t2-t0)/K:.1f} us/it")
