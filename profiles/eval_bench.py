#!/usr/bin/env python3
"""fit() with the per-iteration validation DCG@5 hook: the device
evaluator (rfm_val_dcg; users whose value depends on the order of tied scores redone on the host) against the
host callback (predict -> download -> the oracle's restatement of ValEvaluator.evaluate).
Timing experiment for DESIGN.md; usage: python profiles/eval_bench.py [n_val] [epochs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import relevance_factorizationmachine_amd as pkg
from oracle import cpu_ref
from relevance_factorizationmachine_amd import synth

n_val = int(sys.argv[1]) if len(sys.argv) > 1 else 65471   # full validation frame of config 2
E = int(sys.argv[2]) if len(sys.argv) > 2 else 50
sh = synth.SHAPES["kuairec_small"]
train, val = synth.make_log(sh, "FM", "IPS", seed=0, n_val=n_val)
_, val_mf = synth.make_log(sh, "MF", "IPS", seed=0, n_val=n_val)
keep = synth.first_occurrences(val_mf["features"])


class Hook:
    metric_name, k = "DCG", 5
    rfm_device_evaluator = True  # its evaluate() is the reference's metric (oracle restatement)

    def __init__(self, rows):
        self.interaction_df = synth.interaction_frame({k: v[rows] for k, v in val_mf.items()}, val_mf["features"][rows])
        self.features = {"FM": val["features"][rows]}
        self.calls = 0

    def evaluate(self, y_scores, estimator):
        self.calls += 1
        return cpu_ref.val_dcg(self.interaction_df, y_scores, estimator, k=5)


for name, rows, alpha in (("all rows, alpha=2 (saturated scores -> ties)", np.arange(n_val), 2.0),
                          ("unique pairs, alpha=0.05 (no ties)", keep, 0.05)):
    kw = dict(estimator="IPS", n_factors=16, n_features=train["features"].shape[1], lr=9e-6, batch_size=2000,
              seed=12345, alpha=alpha)
    pkg.FactorizationMachines(n_epochs=2, evaluator=Hook(rows), **kw).fit(train, val)  # warm-up
    res = {}
    for mode in (True, False):
        h = Hook(rows)
        m = pkg.FactorizationMachines(n_epochs=E, evaluator=h, **kw)
        m.device_evaluator = mode
        t0 = time.perf_counter(); m.fit(train, val); dt = time.perf_counter() - t0
        res[mode] = (dt, h.calls, m.val_metrics, getattr(m, "evaluator_host_users", 0))
    same = np.allclose(res[True][2], res[False][2], rtol=1e-12)
    print(f"{name}: frame {len(rows)} rows; fit({E} it) device evaluator {1e3*res[True][0]/E:.2f} ms/it "
          f"({res[True][3]} users redone on the host over all iterations, evaluator called {res[True][1]}x), host callback {1e3*res[False][0]/E:.2f} ms/it; "
          f"speed-up {res[False][0]/res[True][0]:.1f}x; metrics equal: {same}", flush=True)
