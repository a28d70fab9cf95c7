import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import relevance_factorizationmachine_amd as pkg
from relevance_factorizationmachine_amd import synth
shape = synth.SHAPES[sys.argv[1]]; k=int(sys.argv[2]); B=int(sys.argv[3]); E=int(sys.argv[4])
train, val = synth.make_log(shape, "FM", "IPS", seed=0)
kw = dict(estimator="IPS", n_factors=k, lr=9e-6, batch_size=B, seed=12345, n_features=train["features"].shape[1])
pkg.FactorizationMachines(n_epochs=3, **kw).fit(train, val)
pkg.FactorizationMachines(n_epochs=E, **kw).fit(train, val)
