"""world_size-2 check of the data-parallel step logic on CPU (gloo).

The sharding / all-reduce / apply driver (``dist.DataParallelStep``) is the
product code; the arithmetic plugged into it here is the CPU oracle, because no
GPU is present -- the HIP arithmetic is covered by the ``-m gpu`` tests.  The
2-rank result must equal the single-process oracle fit."""
import os
import socket

import numpy as np
import pytest

from oracle import cpu_ref
from relevance_factorizationmachine_amd import synth
from relevance_factorizationmachine_amd.dist import DataParallelStep, shard_bounds

N_EPOCHS, K, LR, BATCH = 4, 8, 1e-3, 501  # odd batch: uneven shards


def test_shard_bounds_cover_batch():
    for batch in (0, 1, 2, 7, 500, 501, 65536):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(batch, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        train, _ = synth.make_log("coat", "FM", "IPS", seed=0)
        X, y, p = train["features"], train["labels"], train["pscores"]
        n = X.shape[1]
        w0, w, V = cpu_ref.fm_init(12345, n, K)
        ids = np.stack([cpu_ref.batch_ids(X.shape[0], BATCH, e) for e in range(N_EPOCHS)])
        grad = torch.zeros(n * K + n + 1, dtype=torch.float64)

        def grad_fn(lo, hi, it):
            rows = ids[it, lo:hi]
            _, g_w0, g_w, G_V = cpu_ref.fm_gradients(X[rows], y[rows], p[rows], w0, w, V)
            grad[: n * K] = torch.from_numpy(G_V.ravel())
            grad[n * K: n * K + n] = torch.from_numpy(g_w)
            grad[-1] = float(g_w0)

        def apply_fn(g):
            gh = g.numpy()
            V[...] -= LR * gh[: n * K].reshape(n, K)
            w[...] -= LR * gh[n * K: n * K + n]
            w0[...] -= LR * gh[-1]

        step = DataParallelStep(grad, grad_fn, apply_fn,
                                lambda g: dist.all_reduce(g, op=dist.ReduceOp.SUM), world, rank)
        for it in range(N_EPOCHS):
            step.step(it, BATCH)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), w0=w0, w=w, V=V)
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_equals_single_process(tmp_path):
    import torch.multiprocessing as mp

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    train, val = synth.make_log("coat", "FM", "IPS", seed=0)
    ref = cpu_ref.fm_fit(train, val, n_epochs=N_EPOCHS, n_factors=K, lr=LR, batch_size=BATCH,
                         seed=12345, with_losses=False)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs:
        assert np.max(np.abs(o["V"] - ref["V"])) < 1e-12
        assert np.max(np.abs(o["w"] - ref["w"])) < 1e-12
        assert np.max(np.abs(o["w0"] - ref["w0"])) < 1e-12
    # replicas stay bitwise identical: same all-reduced gradient, same apply
    np.testing.assert_array_equal(outs[0]["V"], outs[1]["V"])
    np.testing.assert_array_equal(outs[0]["w"], outs[1]["w"])
