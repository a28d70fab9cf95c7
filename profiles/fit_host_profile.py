"""GPU box: where a WARM fit() at the reference's published operating points spends its wall time on
the host side (cProfile): it is the final rt.sync() -- the loop is GPU-bound (KuaiRec shape: 16.5 ms
for 221 iterations = 75 us each; Coat shape: 11.6 ms for 401).   usage: python profiles/fit_host_profile.py"""
import cProfile, pstats, time, sys
sys.path.insert(0, '.')
import numpy as np
from relevance_factorizationmachine_amd import synth
from relevance_factorizationmachine_amd.fm import FactorizationMachines
for shape,k,B,its,lr in (("kuairec_small",400,2000,221,9e-6),("coat",300,500,401,1e-4)):
    train,val=synth.make_log(shape,"FM","IPS",seed=0)
    kw=dict(estimator="IPS",n_factors=k,lr=lr,seed=12345,n_features=train["features"].shape[1],batch_size=B)
    FactorizationMachines(n_epochs=3,**kw).fit(train,val)
    for rep in range(2):
        m=FactorizationMachines(n_epochs=its,**kw)
        t0=time.perf_counter(); m.fit(train,val); print(shape,"fit wall ms",1e3*(time.perf_counter()-t0))
    m=FactorizationMachines(n_epochs=its,**kw)
    pr=cProfile.Profile(); pr.enable(); m.fit(train,val); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
