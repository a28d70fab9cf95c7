"""``dist.fit_data_parallel`` on the HIP kernels: 1, 2 and 3 ranks run the whole fit() loop of
``rfm_fm_fit_dp`` (gradients of the shard, exchange, update, both losses -- everything enqueued by
one C call per run of iterations) and must land on the single-process ``model.fit``: parameters,
BOTH loss curves (src/fm.py:112) and per-iteration evaluator values, replicas bit-identical.

The ranks share the box's one GPU, so the collectives go through the C ABI's ``rfm_transport``
callbacks (host-staged gloo); with one GPU per rank the same loop calls RCCL on the compute
stream instead (transport=None) -- that path needs a multi-GPU node and is NOT covered here."""
import os
import socket

import numpy as np
import pytest

from conftest import rel_err
from relevance_factorizationmachine_amd import synth

pytestmark = pytest.mark.gpu

CASES = {
    # name: (shape, n_train, k, global batch, iterations, lr)
    "small_k16": ("kuairec_small", None, 16, 2001, 6, 9e-6),          # uneven shards
    "published_k400": ("kuairec_small", None, 400, 2000, 4, 9e-6),    # several chunks per lane
    "many_rows_shards": ("kuairec_big", 120_000, 32, 40_000, 3, 9e-6),  # shards take the many-rows forward
    "tiny_batch": ("coat", None, 8, 5, 4, 1e-4),                      # ranks with one or two rows
}


class _MeanEvaluator:
    def __init__(self, features):
        self.features = {"FM": features}

    def evaluate(self, y_scores, estimator):
        return float(np.mean(y_scores))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(case, with_evaluator):
    import relevance_factorizationmachine_amd as pkg

    shape, n_train, k, batch, its, lr = CASES[case]
    train, val = synth.make_log(shape, "FM", "IPS", seed=0, n_train=n_train, n_val=2000 if n_train else None)
    model = pkg.FactorizationMachines(
        estimator="IPS", n_epochs=its, n_factors=k, lr=lr, batch_size=batch, seed=12345,
        n_features=train["features"].shape[1],
        evaluator=_MeanEvaluator(val["features"]) if with_evaluator else None)
    return model, train, val


def _worker(rank, world, port, out_dir, case, exchange, with_evaluator):
    import torch.distributed as dist

    from relevance_factorizationmachine_amd.dist import HostStagedTransport, fit_data_parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, train, val = _make(case, with_evaluator)
        t = HostStagedTransport(world, rank, rt=model._rt)
        tr, va = fit_data_parallel(model, train, val, exchange=exchange, transport=t)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), V=model.V(), w=model.w(), w0=model.w0(),
                 tr=np.array(tr), va=np.array(va), pred=model.predict(val["features"]),
                 metrics=np.array(model.val_metrics if with_evaluator else []))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world,exchange,with_evaluator", [
    ("small_k16", 1, "rows", False), ("small_k16", 2, "dense", False), ("small_k16", 2, "rows", True),
    ("small_k16", 3, "rows", False), ("published_k400", 2, "rows", False), ("published_k400", 3, "dense", False),
    ("many_rows_shards", 2, "rows", False), ("many_rows_shards", 3, "dense", False), ("tiny_batch", 3, "rows", False),
])
def test_fit_data_parallel_equals_single_gpu_fit(tmp_path, case, world, exchange, with_evaluator):
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), case, exchange, with_evaluator), nprocs=world,
             join=True)
    model, train, val = _make(case, with_evaluator)
    tr, va = model.fit(train, val)
    pred = model.predict(val["features"])
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs:
        assert rel_err(o["V"], model.V()) < 1e-12 and rel_err(o["w"], model.w()) < 1e-12
        assert rel_err(o["w0"], model.w0()) < 1e-12
        assert rel_err(o["tr"], tr) < 1e-12 and rel_err(o["va"], va) < 1e-12
        assert rel_err(o["pred"], pred) < 1e-12
        if with_evaluator:
            assert rel_err(o["metrics"], model.val_metrics) < 1e-12
    for o in outs[1:]:  # every replica holds the same parameters and reports the same losses, bit for bit
        for name in ("V", "w", "w0", "tr", "va"):
            np.testing.assert_array_equal(outs[0][name], o[name])


@pytest.mark.parametrize("exchange", ["rows", "dense"])
def test_single_rank_through_the_rccl_exchange(exchange, monkeypatch):
    """One rank, transport=None, RFM_DP_FORCE_EXCHANGE=1: the whole multi-rank loop of rfm_fm_fit_dp
    with the library's OWN RCCL binding (communicator of one rank: ncclAllGather of the transfer
    plan, ncclAllReduce of the gradient / the loss sums; the all-to-alls reduce to the rank's own
    block) must give the plain fit.  This is as much of the RCCL path as one GPU can run."""
    import ctypes as C

    import relevance_factorizationmachine_amd as pkg
    from relevance_factorizationmachine_amd import _lib
    from relevance_factorizationmachine_amd.dist import HipDpEngine

    monkeypatch.setenv("RFM_DP_FORCE_EXCHANGE", "1")
    model, train, val = _make("small_k16", False)
    rt = model._rt
    uid = (C.c_uint8 * 128)()
    _lib.check(rt.lib.rfm_comm_unique_id(uid))
    _lib.check(rt.lib.rfm_comm_init(rt.ctx, 1, 0, uid))
    try:
        engine = HipDpEngine(model, train, val, 1, 0, exchange, None)
        try:
            for first, count, ids in engine.chunks():
                engine.run(first, count, first, ids)
            tr, va = engine.losses()
        finally:
            engine.close()
    finally:
        _lib.check(rt.lib.rfm_comm_destroy(rt.ctx))
    monkeypatch.delenv("RFM_DP_FORCE_EXCHANGE")
    ref, _, _ = _make("small_k16", False)
    tr0, va0 = ref.fit(train, val)
    assert rel_err(model.V(), ref.V()) < 1e-12 and rel_err(model.w(), ref.w()) < 1e-12
    assert rel_err(model.w0(), ref.w0()) < 1e-12
    assert rel_err(tr, tr0) < 1e-12 and rel_err(va, va0) < 1e-12
