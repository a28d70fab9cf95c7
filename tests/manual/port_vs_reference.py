#!/usr/bin/env python3
"""Build container only: time the oracle's reference-structured FM fit (``bench.py``'s
``cpu_baseline`` of kind "port") beside the REAL reference (``/root/reference/src/fm.py``,
imported) on the same synthetic log, same seeds, same iterations -- SURVEY.md 8d asks that the
port's timing stay within a few % of the reference's.  Results are recorded in BASELINE.md
section 2.   usage: python tests/manual/port_vs_reference.py [k] [batch] [iterations]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from make_golden import _import_reference  # noqa: E402
from oracle import cpu_ref  # noqa: E402
from relevance_factorizationmachine_amd import synth  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
E = int(sys.argv[3]) if len(sys.argv) > 3 else 10
FM = _import_reference()[0]
train, val = synth.make_log("kuairec_big", "FM", "IPS", seed=0, n_train=1_000_000, n_val=14308)
n = train["features"].shape[1]
res = {}
for name in ("reference", "port", "reference", "port"):  # twice: the second pair is reported
    t0 = time.perf_counter()
    if name == "reference":
        m = FM(estimator="IPS", n_epochs=E, n_factors=k, n_features=n, lr=9e-6, batch_size=B, seed=12345)
        m.fit(train, val)
        V = m.V()
    else:
        V = cpu_ref.fm_fit(train, val, n_epochs=E, n_factors=k, lr=9e-6, batch_size=B, seed=12345, form="refstruct")["V"]
    res[name] = ((time.perf_counter() - t0) / E, V.copy())
ref_ms, port_ms = 1e3 * res["reference"][0], 1e3 * res["port"][0]
print(f"C3-shaped log (N=1M, n={n}), k={k}, B={B}, {E} iterations of fit(): reference {ref_ms:.1f} ms/it, "
      f"oracle refstruct port {port_ms:.1f} ms/it, port/reference = {port_ms / ref_ms:.3f}; "
      f"max |V_port - V_ref| = {np.max(np.abs(res['port'][1] - res['reference'][1])):.2e}")
