#!/usr/bin/env python3
"""fit() wall time exactly as the reference runs it (step + train-loss forward +
val-loss forward per iteration, src/fm.py:71-102), GPU vs the CPU oracle's
reference-structured step.  Timing experiment for DESIGN.md, not the headline bench.
usage: python tests/manual/fit_bench.py <shape> <k> <batch> <epochs> [cpu_epochs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import relevance_factorizationmachine_amd as pkg
from oracle import cpu_ref
from relevance_factorizationmachine_amd import synth

shape = synth.SHAPES[sys.argv[1]]
k, B, E = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
Ecpu = int(sys.argv[5]) if len(sys.argv) > 5 else 3
train, val = synth.make_log(shape, "FM", "IPS", seed=0)
n = train["features"].shape[1]
kw = dict(estimator="IPS", n_factors=k, lr=9e-6, batch_size=B, seed=12345, n_features=n)
pkg.FactorizationMachines(n_epochs=3, **kw).fit(train, val)  # warm-up (library load, caches)
from relevance_factorizationmachine_amd.runtime import Runtime
Runtime.get().clear_caches()  # cold: the split is uploaded, ids sampled, plan built inside the timed fit
m = pkg.FactorizationMachines(n_epochs=E, **kw)
t0 = time.perf_counter(); tr, va = m.fit(train, val); gpu = time.perf_counter() - t0
m2 = pkg.FactorizationMachines(n_epochs=E, **kw)  # the next model on the same split
t0 = time.perf_counter(); m2.fit(train, val); again = time.perf_counter() - t0
t0 = time.perf_counter()
ref = cpu_ref.fm_fit(train, val, n_epochs=Ecpu, n_factors=k, lr=9e-6, batch_size=B, seed=12345, form="refstruct")
cpu = (time.perf_counter() - t0) / Ecpu
err = max(abs(a - b) / abs(b) for a, b in zip(va[:Ecpu], ref["val_loss"]))
print(f"{shape.name} n={n} k={k} B={B} N={train['features'].shape[0]}/{val['features'].shape[0]}: "
      f"GPU fit({E} it) {gpu:.3f} s = {1e3*gpu/E:.3f} ms/it = {E*B/gpu:,.0f} ex/s (incl. sampler, uploads, plan; "
      f"next fit on the same split {1e3*again/E:.3f} ms/it); "
      f"CPU oracle(refstruct) {1e3*cpu:.1f} ms/it = {B/cpu:,.0f} ex/s; speed-up {cpu/(gpu/E):,.0f}x; "
      f"val-loss rel diff first {Ecpu} it {err:.1e}")
