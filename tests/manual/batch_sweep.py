#!/usr/bin/env python3
"""Step time against the batch size: in-place step (rfm_fm_train) and the data-parallel
pieces gradient + apply (rfm_fm_grad / rfm_fm_apply, no exchange) on one GPU.
usage: python tests/manual/batch_sweep.py <n_train> <B> [<B> ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from relevance_factorizationmachine_amd import _lib, synth
from relevance_factorizationmachine_amd.dist import hip_fm_worker
from relevance_factorizationmachine_amd.fm import FactorizationMachines, FmPlan
from relevance_factorizationmachine_amd.runtime import DeviceCSR, Runtime, sample_batches

n_train = int(sys.argv[1])
shape = synth.SHAPES["kuairec_big"]
t0 = time.perf_counter()
train, _ = synth.make_log(shape, "FM", "IPS", seed=0, n_train=n_train, n_val=16)
print(f"make_log {time.perf_counter()-t0:.1f} s", flush=True)
X = train["features"]; n = X.shape[1]; k = shape.n_factors; lr = 9e-6
rt = Runtime.get(0)
csr = DeviceCSR(rt, X)
y = rt.upload(train["labels"], dtype=np.float64); p = rt.upload(train["pscores"], dtype=np.float64)
for B in map(int, sys.argv[2:]):
    model = FactorizationMachines(estimator="IPS", n_epochs=1, n_factors=k, lr=lr, batch_size=B, seed=12345, n_features=n)
    t0 = time.perf_counter(); ids = sample_batches(n_train, B, 0, 30); ts = time.perf_counter() - t0
    d_ids = rt.upload(ids)
    t0 = time.perf_counter(); plan = FmPlan(rt, csr, train["labels"], train["pscores"], k, B); tp = time.perf_counter() - t0
    ptrs = (csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr(), y.data_ptr(), p.data_ptr())
    params = (model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr())
    def run(first, count):
        _lib.check(rt.lib.rfm_fm_train(rt.ctx, plan.handle, *ptrs, d_ids.data_ptr() + first * B * 4, B, count,
                                       *params, lr, None, None, None, None, None, 0, 1e-8, None, None))
    run(0, 10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(10, 20); torch.cuda.synchronize(); t_step = (time.perf_counter() - t0) / 20
    grad = rt.empty((n * (k + 1) + 1,), torch.float64)
    w = hip_fm_worker(rt, plan, csr, y, p, d_ids, B, model, grad, 1, 0, lr)
    for it in range(10): w.step(it, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(10, 30): w.step(it, B)
    torch.cuda.synchronize(); t_dp = (time.perf_counter() - t0) / 20
    info = plan.info() if hasattr(plan, "info") else None
    print(f"B={B}: step {1e6*t_step:.1f} us = {B/t_step/1e6:.0f} M ex/s; grad+apply {1e6*t_dp:.1f} us; "
          f"sampler {1e3*ts/30:.1f} ms/batch; plan build {tp:.2f} s; info {info}", flush=True)
    plan.close()
