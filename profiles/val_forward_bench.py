#!/usr/bin/env python3
"""The plain forward (predict / validation loss: rfm_fm_forward on the caller's CSR arrays) alone:
time per launch, back to back, for the validation sets of the configs (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from relevance_factorizationmachine_amd import _lib, synth
from relevance_factorizationmachine_amd.fm import FactorizationMachines
from relevance_factorizationmachine_amd.runtime import DeviceCSR, Runtime

rt = Runtime.get(0)
out = []
for shape, k, n_val in (("kuairec_big", 32, 100_000), ("kuairec_big", 32, 20_000), ("kuairec_small", 16, None),
                        ("kuairec_small", 400, None), ("synthetic_1m", 64, 100_000)):
    sh = synth.SHAPES[shape]
    _, val = synth.make_log(sh, "FM", "IPS", seed=0, n_train=1000, n_val=n_val)
    X = val["features"]
    m = FactorizationMachines(estimator="IPS", n_epochs=1, n_factors=k, lr=1e-5, batch_size=1, seed=12345,
                              n_features=X.shape[1])
    d = DeviceCSR(rt, X)
    scores = rt.empty((X.shape[0],), torch.float64)
    def fwd():
        _lib.check(rt.lib.rfm_fm_forward(rt.ctx, d.indptr.data_ptr(), d.indices.data_ptr(), d.values.data_ptr(), None,
                                         X.shape[0], m.w0.dev.data_ptr(), m.w.dev.data_ptr(), m.V.dev.data_ptr(),
                                         X.shape[1], k, scores.data_ptr()))
    y = rt.upload(val["labels"], dtype=np.float64); p = rt.upload(val["pscores"], dtype=np.float64)
    loss = rt.empty((1,), torch.float64)
    def fwd_loss():
        _lib.check(rt.lib.rfm_fm_forward_loss(rt.ctx, d.indptr.data_ptr(), d.indices.data_ptr(), d.values.data_ptr(),
                                              y.data_ptr(), p.data_ptr(), None, X.shape[0], m.w0.dev.data_ptr(),
                                              m.w.dev.data_ptr(), m.V.dev.data_ptr(), X.shape[1], k, 1e-8, None,
                                              loss.data_ptr()))
    res = []
    for fn in (fwd, fwd_loss):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 200)
    out.append(f"{shape} k={k} rows={X.shape[0]}: {1e6*res[0]:.1f} us, with loss (+ finish launch) {1e6*res[1]:.1f} us")
print(" | ".join(out))
