#!/usr/bin/env python3
"""Manual check at the parameter size of BASELINE config 4 (n = 1.1 M features, k = 64, V = 563 MB):
`fit_data_parallel` with 2 ranks sharing the one GPU of a box (touched-row exchange, host-staged
transport) against the single-process `model.fit` -- parameters and both loss curves.
usage (GPU box): python tests/manual/c4_dp_check.py [global_batch] [iterations] [exchange]"""
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from relevance_factorizationmachine_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ITS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
EXCHANGE = sys.argv[3] if len(sys.argv) > 3 else "rows"
OUT = "/tmp/c4_dp_check"


def make():
    import relevance_factorizationmachine_amd as pkg

    train, val = synth.make_log("synthetic_1m", "FM", "IPS", seed=0, n_train=400_000, n_val=20_000)
    model = pkg.FactorizationMachines(estimator="IPS", n_epochs=ITS, n_factors=64, lr=9e-6, batch_size=B,
                                      seed=12345, n_features=train["features"].shape[1])
    return model, train, val


def worker(rank, world, port):
    import torch.distributed as dist

    from relevance_factorizationmachine_amd.dist import HostStagedTransport, fit_data_parallel

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, train, val = make()
        t0 = time.perf_counter()
        tr, va = fit_data_parallel(model, train, val, exchange=EXCHANGE,
                                   transport=HostStagedTransport(world, rank, rt=model._rt))
        dt = time.perf_counter() - t0
        np.savez(os.path.join(OUT, f"rank{rank}.npz"), V=model.V(), w=model.w(), w0=model.w0(), tr=np.array(tr),
                 va=np.array(va), seconds=dt)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp

    os.makedirs(OUT, exist_ok=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
    model, train, val = make()
    tr, va = model.fit(train, val)
    V, w = model.V(), model.w()
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))  # noqa: E731
    outs = [np.load(os.path.join(OUT, f"rank{r}.npz")) for r in range(2)]
    print(f"config-4 size (n={V.shape[0]}, k=64, V={V.nbytes / 1e6:.0f} MB), global batch {B}, {ITS} iterations, "
          f"exchange {EXCHANGE}: V rel err {max(rel(o['V'], V) for o in outs):.1e}, w {max(rel(o['w'], w) for o in outs):.1e}, "
          f"train loss {max(rel(o['tr'], tr) for o in outs):.1e}, val loss {max(rel(o['va'], va) for o in outs):.1e}, "
          f"replicas identical: {bool(np.array_equal(outs[0]['V'], outs[1]['V']) and np.array_equal(outs[0]['w'], outs[1]['w']))}")
