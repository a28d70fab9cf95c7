#!/usr/bin/env python3
"""Config 4 (n = 1.1M features, k = 64, V = 563 MB) on one GPU: what one rank of the
data-parallel step costs with the dense gradient (rfm_fm_grad = memset + kernels, then
rfm_fm_apply over n*(k+1)+1 doubles) against the touched-row form (rfm_fm_grad_rows, then
rfm_fm_apply_rows).  The exchange itself needs more than one GPU; this prices the part of
the step that used to move O(n*k) bytes per iteration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from relevance_factorizationmachine_amd import _lib, synth
from relevance_factorizationmachine_amd.fm import FactorizationMachines, FmPlan
from relevance_factorizationmachine_amd.runtime import DeviceCSR, Runtime, sample_batches

shape = synth.SHAPES["synthetic_1m"]; k = 64; lr = 9e-6; IT = 10
train, _ = synth.make_log(shape, "FM", "IPS", seed=0, n_train=2_000_000, n_val=16)
X = train["features"]; n = X.shape[1]
rt = Runtime.get(0)
csr = DeviceCSR(rt, X)
y = rt.upload(train["labels"], dtype=np.float64); p = rt.upload(train["pscores"], dtype=np.float64)
args = (csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr(), y.data_ptr(), p.data_ptr())
dense = rt.empty((n * (k + 1) + 1,), torch.float64)
for B in (2000, 16384, 65536):
    m = FactorizationMachines(estimator="IPS", n_epochs=1, n_factors=k, lr=lr, batch_size=B, seed=12345, n_features=n)
    par = (m.w0.dev.data_ptr(), m.w.dev.data_ptr(), m.V.dev.data_ptr())
    plan = FmPlan(rt, csr, train["labels"], train["pscores"], k, B)
    ids = rt.upload(sample_batches(X.shape[0], B, 0, 2 * IT + 2))
    cap = min(n, 16 * B + 256)
    rows = rt.empty((cap, k + 2), torch.float64); n_rows = rt.empty((1,), torch.int32); gw0 = rt.empty((1,), torch.float64)

    def dense_step(it):
        _lib.check(rt.lib.rfm_fm_grad(rt.ctx, plan.handle, *args, ids.data_ptr() + it * B * 4, B, *par, dense.data_ptr()))
        _lib.check(rt.lib.rfm_fm_apply(rt.ctx, *par, dense.data_ptr(), n, k, lr))

    def rows_step(it):
        _lib.check(rt.lib.rfm_fm_grad_rows(rt.ctx, plan.handle, ids.data_ptr() + it * B * 4, B, *par, rows.data_ptr(),
                                           cap, n_rows.data_ptr(), gw0.data_ptr(), None, 0, None))
        _lib.check(rt.lib.rfm_fm_apply_rows(rt.ctx, rows.data_ptr(), n_rows.data_ptr(), cap, gw0.data_ptr(), *par, n, k, lr))

    def fused_step(it):
        _lib.check(rt.lib.rfm_fm_step(rt.ctx, plan.handle, *args, ids.data_ptr() + it * B * 4, B, *par, lr))

    out = {}
    for name, fn in (("fused step", fused_step), ("dense grad+apply", dense_step), ("rows grad+apply", rows_step)):
        fn(0); fn(1); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(2, 2 + IT):
            fn(it)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / IT
    cnt = int(n_rows.cpu()[0])
    print(f"B={B}: " + "; ".join(f"{k_} {1e6*v:.0f} us" for k_, v in out.items()) +
          f"; touched rows {cnt} of {n} = {cnt*(k+2)*8/1e6:.1f} MB of records vs {8*(n*(k+1)+1)/1e6:.0f} MB dense", flush=True)
    plan.close(); del m
