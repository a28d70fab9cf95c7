#!/usr/bin/env python3
"""Timing experiments on the GPU box (not part of the product or the tests).

Runs the FM training step of config 3 under different launch geometries /
ablation masks, each in a fresh process, and prints per-kernel average times
(HIP events, rfm_profile_*).  Usage (through gpurun):
    python profiles/ablate.py [--batch 65536] NAME=ENV1=v,ENV2=v ...
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
import torch
from relevance_factorizationmachine_amd import _lib, synth
from relevance_factorizationmachine_amd.fm import FactorizationMachines, FmPlan
from relevance_factorizationmachine_amd.runtime import DeviceCSR, Runtime, sample_batches
B = int(os.environ.get("ABL_BATCH", "65536")); K = 20; k = int(os.environ.get("ABL_K", "32"))
shape = synth.SHAPES[os.environ.get("ABL_SHAPE", "kuairec_big")]
cache = "/tmp/abl_%%s.npz" %% shape.name
train, _ = synth.make_log(shape, "FM", "IPS", seed=0, n_val=16, n_train=int(os.environ.get("ABL_NTRAIN", "0")) or None)
X = train["features"]; n = X.shape[1]
rt = Runtime.get(0)
m = FactorizationMachines(estimator="IPS", n_epochs=1, n_factors=k, lr=9e-6, batch_size=B, seed=12345, n_features=n)
csr = DeviceCSR(rt, X)
y = rt.upload(train["labels"], dtype=np.float64); p = rt.upload(train["pscores"], dtype=np.float64)
ids = rt.upload(sample_batches(X.shape[0], B, 0, K + 5))
plan = FmPlan(rt, csr, train["labels"], train["pscores"], k, B, int(os.environ.get("ABL_HOT", "0")))
args = (csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr(), y.data_ptr(), p.data_ptr())
par = (m.w0.dev.data_ptr(), m.w.dev.data_ptr(), m.V.dev.data_ptr())
def run(first, count):
    _lib.check(rt.lib.rfm_fm_train(rt.ctx, plan.handle, *args, ids.data_ptr() + first * B * 4, B, count, *par, 9e-6,
                                   None, None, None, None, None, 0, 1e-8, None, None))
run(0, 5); torch.cuda.synchronize()
t0 = time.perf_counter(); run(5, K); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / K
ms = (C.c_double * 4)(); cnt = (C.c_int64 * 4)()
_lib.check(rt.lib.rfm_profile_begin(rt.ctx)); run(5, K); _lib.check(rt.lib.rfm_profile_end(rt.ctx, ms, cnt))
print(json.dumps({"wall_us": 1e6 * wall, "fwd_us": 1e3 * ms[0] / cnt[0], "cons_us": 1e3 * ms[1] / cnt[1],
                  "fin_us": 1e3 * ms[2] / cnt[2], "step_us": 1e3 * ms[3] / cnt[3], "plan": plan.info()}))
"""


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":  # measure in this process (for rocprofv3 -- python3 ...)
        exec(compile(CHILD % {"root": ROOT}, "ablate_child", "exec"), {"__name__": "__main__"})
        return
    argv = sys.argv[1:]
    if "--batch" in argv:  # same as ABL_BATCH in the environment
        at = argv.index("--batch")
        os.environ["ABL_BATCH"] = argv[at + 1]
        del argv[at:at + 2]
    specs = [a for a in argv if "=" in a or a.isidentifier()]
    for spec in specs:
        name, _, rest = spec.partition("=")
        env = dict(os.environ)
        for kv in filter(None, rest.split(",")):
            key, _, val = kv.partition("=")
            env[key] = val
        proc = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True)
        line = proc.stdout.strip().splitlines()[-1] if proc.stdout.strip() else proc.stderr[-800:]
        print(f"{name:28s} {line}", flush=True)


if __name__ == "__main__":
    main()
