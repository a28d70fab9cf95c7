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


# ---------------------------------------------------------------------------
# touched-row gradients and their exchange (rfm_fm_grad_rows / _apply_rows /
# _reduce_rows / _set_rows; SURVEY.md 8e option 1)
# ---------------------------------------------------------------------------
def _grad_rows(rt, plan, model, ids_ptr, batch, cap, ranges=None):
    import torch
    from relevance_factorizationmachine_amd import _lib

    k = model.n_factors
    rows = torch.full((max(cap, 1), k + 2), float("nan"), dtype=torch.float64, device=rt.torch_device)
    n_rows = rt.empty((1,), torch.int32)
    gw0 = rt.empty((1,), torch.float64)
    nr = 0 if ranges is None else len(ranges)
    d_lo = rt.upload(np.asarray(ranges, dtype=np.int32)) if nr else None
    d_bounds = rt.empty((nr + 1,), torch.int32) if nr else None
    _lib.check(rt.lib.rfm_fm_grad_rows(
        rt.ctx, plan.handle, ids_ptr, batch, model.w0.dev.data_ptr(), model.w.dev.data_ptr(),
        model.V.dev.data_ptr(), rows.data_ptr(), cap, n_rows.data_ptr(), gw0.data_ptr(),
        d_lo.data_ptr() if nr else None, nr, d_bounds.data_ptr() if nr else None))
    rt.sync()
    return rows, n_rows, gw0, (d_bounds.cpu().numpy() if nr else None)


def test_grad_rows_is_the_dense_gradient_on_the_touched_columns():
    import torch
    from relevance_factorizationmachine_amd import _lib

    rt, model, csr, y, p, plan, ids = _setup()
    n, k = model.n_features, K
    train, _ = synth.make_log("kuairec_small", "FM", "IPS", seed=0)
    host_ids = ids.cpu().numpy().reshape(N_STEPS, BATCH)
    ranges = [0, 1000, 1001, n // 2, n - 1]
    for it in (0, 1):  # twice: stamps of the first call must not leak into the second
        rows, n_rows, gw0, bounds = _grad_rows(rt, plan, model, ids.data_ptr() + it * BATCH * 4, BATCH, n, ranges)
        cnt = int(n_rows.cpu()[0])
        rec = rows.cpu().numpy()[:cnt]
        cols = rec[:, 0].astype(np.int64)
        touched = np.unique(train["features"][host_ids[it]].indices)
        hot = plan.hot_columns()
        assert np.all(np.diff(cols) > 0), "records ascend by column"
        np.testing.assert_array_equal(cols, np.union1d(touched, hot))
        dense = rt.empty((n * (k + 1) + 1,), torch.float64)
        _lib.check(rt.lib.rfm_fm_grad(rt.ctx, plan.handle, csr.indptr.data_ptr(), csr.indices.data_ptr(),
                                      csr.values.data_ptr(), y.data_ptr(), p.data_ptr(),
                                      ids.data_ptr() + it * BATCH * 4, BATCH, model.w0.dev.data_ptr(),
                                      model.w.dev.data_ptr(), model.V.dev.data_ptr(), dense.data_ptr()))
        rt.sync()
        g = dense.cpu().numpy()
        GV, gw = g[: n * k].reshape(n, k), g[n * k: n * k + n]
        assert rel_err(rec[:, 1:-1], GV[cols]) < 1e-13
        assert rel_err(rec[:, -1], gw[cols]) < 1e-13
        assert rel_err(gw0.cpu().numpy(), g[-1:]) < 1e-13
        untouched = np.setdiff1d(np.arange(n), cols)
        assert not GV[untouched].any() and not gw[untouched].any()
        np.testing.assert_array_equal(bounds, np.concatenate([np.searchsorted(cols, ranges), [cnt]]))
    # a list that does not fit reports the true count and fills only the room it was given
    rows, n_rows, _, _ = _grad_rows(rt, plan, model, ids.data_ptr(), BATCH, 100)
    cnt0 = len(np.union1d(np.unique(train["features"][host_ids[0]].indices), plan.hot_columns()))
    assert cnt0 > 100 and int(n_rows.cpu()[0]) == cnt0
    assert np.isfinite(rows.cpu().numpy()[:100]).all()
    # an empty shard: no records, zero g_w0
    rows, n_rows, gw0, bounds = _grad_rows(rt, plan, model, None, 0, n, ranges)
    assert int(n_rows.cpu()[0]) == 0 and float(gw0.cpu()[0]) == 0.0 and not bounds.any()
    plan.close()


def test_gradient_entry_points_at_a_many_rows_batch():
    """The gradient forms of the step at a batch that takes the forward's many-rows shape over
    padded row blocks (40 000 rows of the KuaiRec-big-shaped log, k = 32): the dense
    gradient against the oracle's closed form, the touched-row records against the dense
    gradient, and records applied = the fused step."""
    import torch
    import relevance_factorizationmachine_amd as pkg
    from oracle import cpu_ref
    from relevance_factorizationmachine_amd import _lib
    from relevance_factorizationmachine_amd.fm import FmPlan
    from relevance_factorizationmachine_amd.runtime import DeviceCSR, Runtime, sample_batches

    batch, k, lr = 40_000, 32, 9e-6
    train, _ = synth.make_log("kuairec_big", "FM", "IPS", seed=0, n_train=100_000, n_val=16)
    X = train["features"]
    n = X.shape[1]
    rt = Runtime.get(0)
    kw = dict(estimator="IPS", n_epochs=1, n_factors=k, lr=lr, batch_size=batch, seed=12345, n_features=n)
    model, ref = pkg.FactorizationMachines(**kw), pkg.FactorizationMachines(**kw)
    csr = DeviceCSR(rt, X)
    y = rt.upload(train["labels"], dtype=np.float64)
    p = rt.upload(train["pscores"], dtype=np.float64)
    plan = FmPlan(rt, csr, y, p, k, batch)
    assert plan.layout()["row_blocks"] == 1
    host_ids = sample_batches(X.shape[0], batch, 0, 2)
    ids = rt.upload(host_ids)
    csr_ptrs = (csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr(), y.data_ptr(), p.data_ptr())

    # dense gradient vs the oracle
    dense = rt.empty((n * (k + 1) + 1,), torch.float64)
    _lib.check(rt.lib.rfm_fm_grad(rt.ctx, plan.handle, *csr_ptrs, ids.data_ptr(), batch,
                                  model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr(),
                                  dense.data_ptr()))
    rt.sync()
    g = dense.cpu().numpy()
    GV, gw, gw0 = g[: n * k].reshape(n, k), g[n * k: n * k + n], g[-1]
    rows0 = host_ids[0]
    _, o_w0, o_w, o_V = cpu_ref.fm_gradients(X[rows0], train["labels"][rows0].astype(np.float64),
                                             train["pscores"][rows0], model.w0(), model.w(), model.V())
    assert rel_err(GV, o_V) < 1e-11 and rel_err(gw, o_w) < 1e-11 and abs(gw0 - o_w0) < 1e-9 * max(1.0, abs(o_w0))
    # touched-row records = the dense gradient on the touched columns
    rows, n_rows, r_w0, _ = _grad_rows(rt, plan, model, ids.data_ptr(), batch, n)
    cnt = int(n_rows.cpu()[0])
    rec = rows.cpu().numpy()[:cnt]
    cols = rec[:, 0].astype(np.int64)
    np.testing.assert_array_equal(cols, np.union1d(np.unique(X[rows0].indices), plan.hot_columns()))
    assert rel_err(rec[:, 1:-1], GV[cols]) < 1e-12 and rel_err(rec[:, -1], gw[cols]) < 1e-12
    assert rel_err(r_w0.cpu().numpy(), g[-1:]) < 1e-12
    # records applied = the fused step, two iterations
    plan2 = FmPlan(rt, csr, y, p, k, batch)
    for it in range(2):
        rows, n_rows, r_w0, _ = _grad_rows(rt, plan, model, ids.data_ptr() + it * batch * 4, batch, n)
        _lib.check(rt.lib.rfm_fm_apply_rows(rt.ctx, rows.data_ptr(), n_rows.data_ptr(), n, r_w0.data_ptr(),
                                            model.w0.dev.data_ptr(), model.w.dev.data_ptr(),
                                            model.V.dev.data_ptr(), n, k, lr))
        _lib.check(rt.lib.rfm_fm_step(rt.ctx, plan2.handle, *csr_ptrs, ids.data_ptr() + it * batch * 4, batch,
                                      ref.w0.dev.data_ptr(), ref.w.dev.data_ptr(), ref.V.dev.data_ptr(), lr))
    rt.sync()
    assert rel_err(model.V(), ref.V()) < 1e-12 and rel_err(model.w(), ref.w()) < 1e-12
    assert rel_err(model.w0(), ref.w0()) < 1e-12
    plan.close()
    plan2.close()


def test_grad_rows_then_apply_rows_equals_step():
    from relevance_factorizationmachine_amd import _lib

    rt, model, csr, y, p, plan, ids = _setup()
    n = model.n_features
    for it in range(N_STEPS):
        rows, n_rows, gw0, _ = _grad_rows(rt, plan, model, ids.data_ptr() + it * BATCH * 4, BATCH, n)
        _lib.check(rt.lib.rfm_fm_apply_rows(rt.ctx, rows.data_ptr(), n_rows.data_ptr(), n, gw0.data_ptr(),
                                            model.w0.dev.data_ptr(), model.w.dev.data_ptr(),
                                            model.V.dev.data_ptr(), n, K, LR))
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


def test_reduce_rows_and_set_rows_against_numpy():
    """Owner-side reduction: records of one column from several segments are added in
    segment order; positions of the output are those of the input."""
    import torch
    from relevance_factorizationmachine_amd import _lib
    from relevance_factorizationmachine_amd.runtime import Runtime

    rt = Runtime.get(0)
    rng = np.random.default_rng(5)
    n, k, lr = 500, 11, 0.37
    V, w, w0 = rng.standard_normal((n, k)), rng.standard_normal(n), rng.standard_normal(1)
    segs = []
    for s in range(5):
        cols = np.sort(rng.choice(n, size=[120, 0, 300, 1, 77][s], replace=False))
        segs.append(np.concatenate([cols[:, None].astype(float), rng.standard_normal((len(cols), k + 1))], axis=1))
    got = np.concatenate(segs)
    seg_ptr = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int32)
    # NumPy: in segment order
    want = np.zeros_like(got)
    want[:, 0] = -1
    acc, first = {}, {}
    for i, r in enumerate(got):
        c = int(r[0])
        if c in first:
            acc[c] = acc[c] + r[1:]
        else:
            first[c], acc[c] = i, 0.0 + r[1:]
    for c, i in first.items():
        want[i, 0] = c
        want[i, 1:-1] = V[c] - lr * acc[c][:-1]
        want[i, -1] = w[c] - lr * acc[c][-1]
    dV, dw, dw0 = rt.upload(V), rt.upload(w), rt.upload(w0)
    d_got, d_seg = rt.upload(got), rt.upload(seg_ptr)
    out = rt.empty(got.shape, torch.float64)
    _lib.check(rt.lib.rfm_fm_reduce_rows(rt.ctx, d_got.data_ptr(), d_seg.data_ptr(), 5, len(got),
                                         dw.data_ptr(), dV.data_ptr(), n, k, lr, out.data_ptr()))
    rt.sync()
    res = out.cpu().numpy()
    np.testing.assert_array_equal(res[:, 0], want[:, 0])
    live = want[:, 0] >= 0
    assert rel_err(res[live], want[live]) < 1e-14  # same operation order (the device may fuse a*b+c)
    parts = rt.upload(np.array([0.5, -1.25, 3.0]))
    _lib.check(rt.lib.rfm_fm_set_rows(rt.ctx, out.data_ptr(), len(got), parts.data_ptr(), 3, 1,
                                      dw0.data_ptr(), dw.data_ptr(), dV.data_ptr(), n, k, lr))
    rt.sync()
    V2, w2 = V.copy(), w.copy()
    cols = want[live, 0].astype(int)
    V2[cols], w2[cols] = want[live, 1:-1], want[live, -1]
    assert rel_err(dV.cpu().numpy(), V2) < 1e-14 and rel_err(dw.cpu().numpy(), w2) < 1e-14
    untouched = np.setdiff1d(np.arange(n), cols)
    np.testing.assert_array_equal(dV.cpu().numpy()[untouched], V[untouched])
    assert rel_err(dw0.cpu().numpy(), w0 - lr * ((0.5 + -1.25) + 3.0)) < 1e-15


def _rows_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from relevance_factorizationmachine_amd.dist import RowExchange, hip_fm_rows_worker

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rt, model, csr, y, p, plan, ids = _setup()
        comm = RowExchange.for_torch(dist, world, rank, model.n_features, K, backend="gloo")
        worker = hip_fm_rows_worker(rt, plan, ids, BATCH, model, world, rank, LR, comm)
        sent = []
        for it in range(N_STEPS):
            worker.step(it, BATCH)
            sent.append(int(worker.last_counts.sum()))
        rt.sync()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), V=model.V(), w=model.w(), w0=model.w0(),
                 sent=np.array(sent))
        plan.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_touched_row_exchange_matches_single_gpu_step(tmp_path, world):
    """Ranks sharing the one GPU exchange only touched rows (host-staged gloo) and land on
    the single-process rfm_fm_step result; replicas stay bitwise identical."""
    import torch.multiprocessing as mp
    from relevance_factorizationmachine_amd import _lib

    mp.spawn(_rows_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
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
        assert int(o["sent"].max()) <= model.n_features
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0]["V"], o["V"])
        np.testing.assert_array_equal(outs[0]["w"], o["w"])
        np.testing.assert_array_equal(outs[0]["w0"], o["w0"])
    plan.close()


# ---------------------------------------------------------------------------
# MF user-range partition on the HIP kernels
# ---------------------------------------------------------------------------
MF_KW = dict(n_factors=8, lr=0.02, reg=0.5, seed=12345, n_users=290, n_items=300)
MF_EPOCHS, MF_BATCH = 4, 500


def _mf_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    import relevance_factorizationmachine_amd as pkg
    from relevance_factorizationmachine_amd.dist import hip_mf_partition_worker
    from relevance_factorizationmachine_amd.runtime import Runtime

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        train, _ = synth.make_log("coat", "MF", "IPS", seed=0)
        rt = Runtime.get(0)
        model = pkg.LogisticMatrixFactorization(estimator="IPS", n_epochs=MF_EPOCHS, batch_size=MF_BATCH, **MF_KW)
        step, finish = hip_mf_partition_worker(rt, model, train, world, rank, stage_host=True)
        for it in range(MF_EPOCHS):
            step.step(it, step.users_of(it))
        finish()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), P=model.P(), Q=model.Q(), bu=model.b_u(), bi=model.b_i())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_mf_user_partition_on_the_device(tmp_path, world):
    """The partitioned MF mode with the HIP kernels (ranks share the one GPU, host-staged
    gloo) against a NumPy simulation of the same partition; one rank = the exact fit."""
    import torch.multiprocessing as mp
    from oracle import cpu_ref
    from relevance_factorizationmachine_amd.dist import user_ranges

    mp.spawn(_mf_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    train, _ = synth.make_log("coat", "MF", "IPS", seed=0)
    X, y, p = train["features"], train["labels"], train["pscores"]
    P, Q, bu, bi = cpu_ref.mf_init(MF_KW["seed"], MF_KW["n_users"], MF_KW["n_items"], MF_KW["n_factors"])
    b, lo = float(np.mean(y)), user_ranges(MF_KW["n_users"], world)
    for it in range(MF_EPOCHS):
        rows = cpu_ref.batch_ids(X.shape[0], MF_BATCH, it)
        users = X[rows, 0]
        dQ, dbi = np.zeros_like(Q), np.zeros_like(bi)
        for r in range(world):
            mine = rows[(users >= lo[r]) & (users < lo[r + 1])]
            Qr, bir = Q.copy(), bi.copy()
            cpu_ref.mf_sgd_batch(X[mine], y[mine], p[mine], P, Qr, bu, bir, b, MF_KW["lr"], MF_KW["reg"])
            dQ += Qr - Q
            dbi += bir - bi
        Q, bi = (Q + dQ, bi + dbi) if world > 1 else (Qr, bir)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs:
        for name, want in (("P", P), ("Q", Q), ("bu", bu), ("bi", bi)):
            assert rel_err(o[name], want) < 1e-9, name
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0]["Q"], o["Q"])
        np.testing.assert_array_equal(outs[0]["P"], o["P"])
