"""Where the wall time of FactorizationMachines.fit() goes on config 3 (manual, GPU box):
cProfile of a fit on a log that was already fitted once (ids cached, CSR resident)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from relevance_factorizationmachine_amd import synth  # noqa: E402
from relevance_factorizationmachine_amd.fm import FactorizationMachines  # noqa: E402

B = int(os.environ.get("FIT_BATCH", "65536"))
ITS = int(os.environ.get("FIT_ITS", "200"))
PROFILED = os.environ.get("FIT_PROFILED", "third")
shape = synth.SHAPES["kuairec_big"]
train, val = synth.make_log(shape, "FM", "IPS", seed=0)
kw = dict(estimator="IPS", n_factors=shape.n_factors, lr=9e-6, seed=12345, n_features=train["features"].shape[1])
FactorizationMachines(n_epochs=ITS, batch_size=B, **kw).fit(train, val)
for label in ("second", "third", "fourth", "fifth"):
    m = FactorizationMachines(n_epochs=ITS, batch_size=B, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if label == PROFILED:
        pr = cProfile.Profile()
        pr.enable()
    m.fit(train, val)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if label == PROFILED:
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
    print(f"{label} fit: {1e3 * dt:.1f} ms = {1e3 * dt / ITS:.3f} ms/it at B={B}")
