#!/usr/bin/env python3
"""Config 4 (1M users x 100k items, n = 1.1M features, k = 64, V = 563 MB) on one GPU:
one step against the oracle's closed-form gradients, then step timings."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import cpu_ref
from relevance_factorizationmachine_amd import _lib, synth
from relevance_factorizationmachine_amd.fm import FactorizationMachines, FmPlan
from relevance_factorizationmachine_amd.runtime import DeviceCSR, Runtime, sample_batches

shape = synth.SHAPES["synthetic_1m"]; k = 64; lr = 9e-6
t0 = time.perf_counter()
train, _ = synth.make_log(shape, "FM", "IPS", seed=0, n_train=2_000_000, n_val=16)
X = train["features"]; n = X.shape[1]
print(f"log: {X.shape}, nnz {X.nnz}, gen {time.perf_counter()-t0:.1f}s", flush=True)
rt = Runtime.get(0)
for B in (2000, 65536):
    m = FactorizationMachines(estimator="IPS", n_epochs=1, n_factors=k, lr=lr, batch_size=B, seed=12345, n_features=n)
    csr = DeviceCSR(rt, X)
    y = rt.upload(train["labels"], dtype=np.float64); p = rt.upload(train["pscores"], dtype=np.float64)
    t0 = time.perf_counter(); plan = FmPlan(rt, csr, train["labels"], train["pscores"], k, B); tp = time.perf_counter() - t0
    ids_h = sample_batches(X.shape[0], B, 0, 25); ids = rt.upload(ids_h)
    args = (csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr(), y.data_ptr(), p.data_ptr())
    par = (m.w0.dev.data_ptr(), m.w.dev.data_ptr(), m.V.dev.data_ptr())
    # one step vs oracle
    w0, w, V = cpu_ref.fm_init(12345, n, k)
    rows = ids_h[0]
    err, g0, gw, GV = cpu_ref.fm_gradients(X[rows], train["labels"][rows], train["pscores"][rows], w0, w, V)
    _lib.check(rt.lib.rfm_fm_step(rt.ctx, plan.handle, *args, ids.data_ptr(), B, *par, lr)); rt.sync()
    Vg = m.V(); Vr = V - lr * GV
    relV = np.max(np.abs(Vg - Vr)) / np.max(np.abs(Vr)); relW = np.max(np.abs(m.w() - (w - lr * gw))) / np.max(np.abs(w))
    def run(first, count):
        _lib.check(rt.lib.rfm_fm_train(rt.ctx, plan.handle, *args, ids.data_ptr() + first * B * 4, B, count, *par, lr,
                                       None, None, None, None, None, 0, 1e-8, None, None))
    run(1, 4); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(5, 20); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20
    ms = (C.c_double * 4)(); cnt = (C.c_int64 * 4)()
    _lib.check(rt.lib.rfm_profile_begin(rt.ctx)); run(5, 20); _lib.check(rt.lib.rfm_profile_end(rt.ctx, ms, cnt))
    print(f"B={B}: plan {plan.info()} built in {tp:.2f}s; step-vs-oracle rel err V {relV:.1e} w {relW:.1e}; "
          f"step {1e6*wall:.1f} us = {B/wall/1e6:.1f} M ex/s; fwd/cons/fin {1e3*ms[0]/cnt[0]:.1f}/{1e3*ms[1]/cnt[1]:.1f}/{1e3*ms[2]/cnt[2]:.1f} us", flush=True)
    plan.close(); del m, csr, y, p, ids
