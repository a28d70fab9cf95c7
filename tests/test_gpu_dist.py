"""Two ranks sharing the one GPU of the box run the data-parallel FM step
(rfm_fm_grad -> all-reduce -> rfm_fm_apply) and must land on the single-process
rfm_fm_step result.  The ranks exchange through gloo with host staging, because
two RCCL ranks cannot sit on one device; the multi-GPU RCCL path itself is the
same code with torch.distributed's nccl backend (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest

from conftest import rel_err
from relevance_factorizationmachine_amd import synth

pytestmark = pytest.mark.gpu

N_STEPS, BATCH, K, LR = 3, 2000, 16, 9e-6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup():
    import relevance_factorizationmachine_amd as pkg
    from relevance_factorizationmachine_amd.fm import FmPlan
    from relevance_factorizationmachine_amd.runtime import DeviceCSR, Runtime, sample_batches

    train, _ = synth.make_log("kuairec_small", "FM", "IPS", seed=0)
    X = train["features"]
    rt = Runtime.get(0)
    model = pkg.FactorizationMachines(estimator="IPS", n_epochs=1, n_factors=K, lr=LR, batch_size=BATCH,
                                      seed=12345, n_features=X.shape[1])
    csr = DeviceCSR(rt, X)
    y = rt.upload(train["labels"], dtype=np.float64)
    p = rt.upload(train["pscores"], dtype=np.float64)
    plan = FmPlan(rt, csr, train["labels"], train["pscores"], K, BATCH)
    ids = rt.upload(sample_batches(X.shape[0], BATCH, 0, N_STEPS))
    return rt, model, csr, y, p, plan, ids


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from relevance_factorizationmachine_amd.dist import hip_fm_worker

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rt, model, csr, y, p, plan, ids = _setup()
        n = model.n_features
        grad = rt.empty((n * (K + 1) + 1,), torch.float64)

        def all_reduce(g):  # host-staged: both ranks share cuda:0
            h = g.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            g.copy_(h)

        worker = hip_fm_worker(rt, plan, csr, y, p, ids, BATCH, model, grad, world, rank, LR, all_reduce)
        for it in range(N_STEPS):
            worker.step(it, BATCH)
        rt.sync()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), V=model.V(), w=model.w(), w0=model.w0())
        plan.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_matches_single_gpu_step(tmp_path):
    import torch.multiprocessing as mp
    from relevance_factorizationmachine_amd import _lib

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rt, model, csr, y, p, plan, ids = _setup()
    for it in range(N_STEPS):
        _lib.check(rt.lib.rfm_fm_step(rt.ctx, plan.handle, csr.indptr.data_ptr(), csr.indices.data_ptr(),
                                      csr.values.data_ptr(), y.data_ptr(), p.data_ptr(),
                                      ids.data_ptr() + it * BATCH * 4, BATCH, model.w0.dev.data_ptr(),
                                      model.w.dev.data_ptr(), model.V.dev.data_ptr(), LR))
    rt.sync()
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs:
        assert rel_err(o["V"], model.V()) < 1e-12
        assert rel_err(o["w"], model.w()) < 1e-12
        assert rel_err(o["w0"], model.w0()) < 1e-12
    np.testing.assert_array_equal(outs[0]["V"], outs[1]["V"])  # replicas stay identical
    plan.close()


def test_rccl_binding_single_rank():
    """The C ABI's own RCCL all-reduce: with one rank it must leave the buffer
    unchanged (multi-rank runs need one GPU per rank)."""
    import ctypes as C
    import torch
    from relevance_factorizationmachine_amd import _lib
    from relevance_factorizationmachine_amd.runtime import Runtime

    rt = Runtime.get(0)
    uid = (C.c_uint8 * 128)()
    _lib.check(rt.lib.rfm_comm_unique_id(uid))
    _lib.check(rt.lib.rfm_comm_init(rt.ctx, 1, 0, uid))
    try:
        x = torch.arange(1000, dtype=torch.float64, device=rt.torch_device)
        _lib.check(rt.lib.rfm_allreduce_sum(rt.ctx, x.data_ptr(), 1000))
        rt.sync()
        np.testing.assert_array_equal(x.cpu().numpy(), np.arange(1000, dtype=np.float64))
        with pytest.raises(ValueError):
            _lib.check(rt.lib.rfm_comm_init(rt.ctx, 1, 0, uid))  # already initialised
    finally:
        _lib.check(rt.lib.rfm_comm_destroy(rt.ctx))


def test_train_dp_single_rank_equals_step():
    """rfm_fm_train_dp with the whole batch as this rank's shard (no exchange) is
    grad + apply of every iteration, i.e. the fused step."""
    import torch
    from relevance_factorizationmachine_amd import _lib
    from relevance_factorizationmachine_amd.dist import hip_fm_train_dp

    rt, model, csr, y, p, plan, ids = _setup()
    grad = rt.empty((model.n_features * (K + 1) + 1,), torch.float64)
    hip_fm_train_dp(rt, plan, ids, BATCH, 0, N_STEPS, model, grad, 1, 0, LR)
    rt.sync()
    rt2, ref, csr2, y2, p2, plan2, ids2 = _setup()
    for it in range(N_STEPS):
        _lib.check(rt2.lib.rfm_fm_step(rt2.ctx, plan2.handle, csr2.indptr.data_ptr(), csr2.indices.data_ptr(),
                                       csr2.values.data_ptr(), y2.data_ptr(), p2.data_ptr(),
                                       ids2.data_ptr() + it * BATCH * 4, BATCH, ref.w0.dev.data_ptr(),
                                       ref.w.dev.data_ptr(), ref.V.dev.data_ptr(), LR))
    rt2.sync()
    assert rel_err(model.V(), ref.V()) < 1e-13
    assert rel_err(model.w(), ref.w()) < 1e-13
    assert rel_err(model.w0(), ref.w0()) < 1e-13
    plan.close()
    plan2.close()
