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


# ---------------------------------------------------------------------------
# touched-row exchange (SURVEY.md 8e option 1): RowExchangeStep + TorchRowComm are the
# product code; the arithmetic plugged in is NumPy from the oracle
# ---------------------------------------------------------------------------
def _np_grad_rows(X, y, p, w0, w, V, rows_idx, lo_cols, world):
    """Oracle gradient of the given rows as the record list rfm_fm_grad_rows emits."""
    n, k = V.shape
    if len(rows_idx) == 0:
        return np.zeros((0, k + 2)), np.zeros(world + 1), 0.0
    Xb = X[rows_idx]
    _, g_w0, g_w, G_V = cpu_ref.fm_gradients(Xb, y[rows_idx], p[rows_idx], w0, w, V)
    cols = np.unique(Xb.indices)
    rec = np.concatenate([cols[:, None].astype(np.float64), G_V[cols], g_w[cols, None]], axis=1)
    bounds = np.concatenate([np.searchsorted(cols, lo_cols), [len(cols)]]).astype(np.float64)
    return rec, bounds, float(g_w0)


def _np_reduce_rows(got, seg_ptr, w, V, lr):
    """Owner-side sum in segment (rank) order + update, in the layout rfm_fm_reduce_rows emits."""
    out = np.full_like(got, 0.0)
    out[:, 0] = -1.0
    first = {}
    acc = {}
    for s in range(len(seg_ptr) - 1):
        for i in range(seg_ptr[s], seg_ptr[s + 1]):
            c = int(got[i, 0])
            if c not in first:
                first[c] = i
                acc[c] = 0.0 + got[i, 1:].copy()
            else:
                acc[c] = acc[c] + got[i, 1:]
    for c, i in first.items():
        out[i, 0] = c
        out[i, 1:-1] = V[c] - lr * acc[c][:-1]
        out[i, -1] = w[c] - lr * acc[c][-1]
    return out


def _rows_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from relevance_factorizationmachine_amd.dist import RowExchange, RowExchangeStep, owner_ranges

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        train, _ = synth.make_log("coat", "FM", "IPS", seed=0)
        X, y, p = train["features"], train["labels"], train["pscores"]
        n = X.shape[1]
        w0, w, V = cpu_ref.fm_init(12345, n, K)
        ids = np.stack([cpu_ref.batch_ids(X.shape[0], BATCH, e) for e in range(N_EPOCHS)])
        lo_cols = owner_ranges(n, world)
        sent = []

        def grad_rows_fn(lo, hi, it):
            rec, bounds, g_w0 = _np_grad_rows(X, y, p, w0, w, V, ids[it, lo:hi], lo_cols, world)
            sent.append(len(rec))
            return torch.from_numpy(rec), torch.from_numpy(np.concatenate([bounds, [g_w0]]))

        def reduce_fn(got, seg_ptr):
            return torch.from_numpy(_np_reduce_rows(got.numpy(), seg_ptr, w, V, LR))

        def set_rows_fn(everything, gw0_parts):
            e = everything.numpy()
            live = e[e[:, 0] >= 0]
            cols = live[:, 0].astype(np.int64)
            assert len(np.unique(cols)) == len(cols)  # every column has exactly one owner row
            V[cols] = live[:, 1:-1]
            w[cols] = live[:, -1]
            s = 0.0
            for g in gw0_parts:
                s += g
            w0[...] -= LR * s

        comm = RowExchange.for_torch(dist, world, rank, n, K, backend="gloo")
        step = RowExchangeStep(world, rank, comm, grad_rows_fn, reduce_fn, set_rows_fn)
        for it in range(N_EPOCHS):
            step.step(it, BATCH)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), w0=w0, w=w, V=V, sent=np.array(sent))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_touched_row_exchange_equals_single_process(tmp_path, world):
    import torch.multiprocessing as mp

    from relevance_factorizationmachine_amd.dist import shard_bounds as sb

    mp.spawn(_rows_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    train, val = synth.make_log("coat", "FM", "IPS", seed=0)
    X, y, p = train["features"], train["labels"], train["pscores"]
    n = X.shape[1]
    ref = cpu_ref.fm_fit(train, val, n_epochs=N_EPOCHS, n_factors=K, lr=LR, batch_size=BATCH,
                         seed=12345, with_losses=False)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs:
        assert np.max(np.abs(o["V"] - ref["V"])) < 1e-12
        assert np.max(np.abs(o["w"] - ref["w"])) < 1e-12
        assert np.max(np.abs(o["w0"] - ref["w0"])) < 1e-12
    for o in outs[1:]:  # replicas bitwise identical: every row computed once, by its owner
        np.testing.assert_array_equal(outs[0]["V"], o["V"])
        np.testing.assert_array_equal(outs[0]["w"], o["w"])
        np.testing.assert_array_equal(outs[0]["w0"], o["w0"])
    # bit for bit the single-process oracle whose step adds the shards' gradients in rank order
    w0, w, V = cpu_ref.fm_init(12345, n, K)
    ids = np.stack([cpu_ref.batch_ids(X.shape[0], BATCH, e) for e in range(N_EPOCHS)])
    for it in range(N_EPOCHS):
        gV, gw, g0 = np.zeros_like(V), np.zeros_like(w), 0.0
        seen_V = np.zeros(n, dtype=bool)
        for r in range(world):
            lo, hi = sb(BATCH, world, r)
            rows = ids[it, lo:hi]
            _, a0, aw, aV = cpu_ref.fm_gradients(X[rows], y[rows], p[rows], w0, w, V)
            cols = np.unique(X[rows].indices)
            fresh = cols[~seen_V[cols]]
            again = cols[seen_V[cols]]
            gV[fresh], gw[fresh] = aV[fresh], aw[fresh]
            gV[again] += aV[again]
            gw[again] += aw[again]
            seen_V[cols] = True
            g0 += a0
        t = np.flatnonzero(seen_V)
        V[t] = V[t] - LR * gV[t]
        w[t] = w[t] - LR * gw[t]
        w0 -= LR * g0
    np.testing.assert_array_equal(outs[0]["V"], V)
    np.testing.assert_array_equal(outs[0]["w"], w)
    np.testing.assert_array_equal(outs[0]["w0"], w0)
    # only touched rows travel: far fewer records than columns x iterations would be
    assert all(int(o["sent"].max()) <= n for o in outs)


# ---------------------------------------------------------------------------
# MF user-range partition: MfUserPartitionStep is the product code; the
# arithmetic plugged in is the oracle's sequential SGD.  Not the reference's semantics for
# world > 1: checked against a single-process simulation of the same partition, and at the
# loss level against the exact fit.
# ---------------------------------------------------------------------------
MF_KW = dict(n_factors=6, lr=0.02, reg=0.5, seed=12345, n_users=290, n_items=300)
MF_EPOCHS, MF_BATCH = 5, 500


def _mf_log():
    return synth.make_log("coat", "MF", "IPS", seed=0)


def _simulate_partition(train, world):
    """What the partitioned mode computes, in one process: per batch every rank's examples
    run sequentially from the merged Q / b_i, then the deltas are added in rank order."""
    from relevance_factorizationmachine_amd.dist import user_ranges

    X, y, p = train["features"], train["labels"], train["pscores"]
    P, Q, bu, bi = cpu_ref.mf_init(MF_KW["seed"], MF_KW["n_users"], MF_KW["n_items"], MF_KW["n_factors"])
    b = float(np.mean(y))
    lo = user_ranges(MF_KW["n_users"], world)
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
        if world > 1:
            Q, bi = Q + dQ, bi + dbi
        else:
            Q, bi = Qr, bir
    return P, Q, bu, bi, b


def _mf_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from relevance_factorizationmachine_amd.dist import MfUserPartitionStep

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        train, _ = _mf_log()
        X, y, p = train["features"], train["labels"], train["pscores"]
        P, Q, bu, bi = cpu_ref.mf_init(MF_KW["seed"], MF_KW["n_users"], MF_KW["n_items"], MF_KW["n_factors"])
        b = float(np.mean(y))
        sync = {"Q": Q.copy(), "bi": bi.copy()}
        rows_of = {}

        def sgd_fn(it, positions):
            mine = rows_of[it][positions]
            cpu_ref.mf_sgd_batch(X[mine], y[mine], p[mine], P, Q, bu, bi, b, MF_KW["lr"], MF_KW["reg"])

        def delta_fn():
            return torch.from_numpy(np.concatenate([(Q - sync["Q"]).ravel(), bi - sync["bi"]]))

        def merge_fn(total):
            t = total.numpy()
            sync["Q"] = sync["Q"] + t[: Q.size].reshape(Q.shape)
            sync["bi"] = sync["bi"] + t[Q.size:]
            Q[...] = sync["Q"]
            bi[...] = sync["bi"]

        step = MfUserPartitionStep(world, rank, MF_KW["n_users"], sgd_fn, delta_fn, merge_fn,
                                   lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
        for it in range(MF_EPOCHS):
            rows_of[it] = cpu_ref.batch_ids(X.shape[0], MF_BATCH, it)
            step.step(it, X[rows_of[it], 0])
        # owners hand their rows of P / b_u to everybody
        for r in range(world):
            lo, hi = step.lo[r], step.lo[r + 1]
            for arr in (P, bu):
                t = torch.from_numpy(arr[lo:hi].copy())
                dist.broadcast(t, src=r)
                arr[lo:hi] = t.numpy()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), P=P, Q=Q, bu=bu, bi=bi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_mf_user_partition_step(tmp_path, world):
    import torch.multiprocessing as mp

    mp.spawn(_mf_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    train, val = _mf_log()
    P, Q, bu, bi, b = _simulate_partition(train, world)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs:
        for name, want in (("P", P), ("Q", Q), ("bu", bu), ("bi", bi)):
            assert np.max(np.abs(o[name] - want)) < 1e-12, name
    for o in outs[1:]:  # one all-reduced delta, one merge: replicas are bitwise identical
        for name in ("P", "Q", "bu", "bi"):
            np.testing.assert_array_equal(outs[0][name], o[name])
    exact = cpu_ref.mf_fit(train, val, n_epochs=MF_EPOCHS, batch_size=MF_BATCH, **MF_KW)
    if world == 1:  # one rank IS the reference's sequential batch
        for name in ("P", "Q"):
            np.testing.assert_array_equal(outs[0][name], exact[name])
    else:  # more ranks: same model at the loss level, not to 1e-5 on the parameters
        got = cpu_ref.ips_logloss(val["labels"], cpu_ref.mf_predict(val["features"], outs[0]["P"], outs[0]["Q"],
                                                                    outs[0]["bu"], outs[0]["bi"], b), val["pscores"])
        # (measured: 2 ranks 3 %, 3 ranks 8 % after five 500-row batches with Zipf items --
        # examples of different ranks that share an item do not see each other inside a batch)
        assert abs(got - exact["val_loss"][-1]) < 0.15 * abs(exact["val_loss"][-1])


# ---------------------------------------------------------------------------
# dist.fit_data_parallel (the multi-GPU fit()) and HostStagedTransport: product code; the
# arithmetic plugged in is the NumPy engine of tests/np_dp_engine.py
# ---------------------------------------------------------------------------
DP_KW = dict(estimator="IPS", n_epochs=5, n_factors=6, lr=1e-3, batch_size=501, seed=12345)


class _CountingEvaluator:
    """An evaluator of the host-callback kind (src/fm.py:104-110)."""

    def __init__(self, features):
        self.features = {"FM": features}

    def evaluate(self, y_scores, estimator):
        return float(np.mean(y_scores)) + (0.0 if estimator == "IPS" else 1.0)


def _dp_worker(rank, world, port, out_dir, exchange, with_evaluator):
    from types import SimpleNamespace

    import torch.distributed as dist

    from np_dp_engine import NumpyDpEngine
    from relevance_factorizationmachine_amd.dist import HostStagedTransport, fit_data_parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        train, val = synth.make_log("coat", "FM", "IPS", seed=0)
        model = SimpleNamespace(n_features=train["features"].shape[1], evaluator=None, **DP_KW)
        if with_evaluator:
            model.evaluator = _CountingEvaluator(val["features"])
            model.val_metrics, model.model_name = [], "FM"
        t = HostStagedTransport(world, rank)
        if rank == 0:  # the transport's host API on ragged payloads
            pass
        ragged = [np.arange(3 * (rank + 1) + p, dtype=np.float64).view(np.uint8) for p in range(world)]
        got = t.all_to_all_host(ragged, [8 * (3 * (s + 1) + rank) for s in range(world)])
        for s in range(world):
            np.testing.assert_array_equal(got[s].view(np.float64), np.arange(3 * (s + 1) + rank, dtype=np.float64))
        np.testing.assert_array_equal(t.all_gather_host(np.array([rank, 7], dtype=np.int32)),
                                      np.array([[r, 7] for r in range(world)], dtype=np.int32))
        np.testing.assert_array_equal(t.all_reduce_sum_host(np.array([1.0, rank])),
                                      np.array([world, world * (world - 1) / 2]))
        engines = []

        def factory(*a):
            engines.append(NumpyDpEngine(*a))
            return engines[-1]

        tr, va = fit_data_parallel(model, train, val, exchange=exchange, transport=t, engine_factory=factory)
        e = engines[0]
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), w0=e.w0, w=e.w, V=e.V, tr=np.array(tr), va=np.array(va),
                 metrics=np.array(model.val_metrics if with_evaluator else []))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange,with_evaluator", [(2, "dense", False), (2, "rows", True), (3, "rows", False),
                                                          (3, "dense", True)])
def test_fit_data_parallel_equals_single_process_fit(tmp_path, world, exchange, with_evaluator):
    """Both loss curves, the parameters and the per-iteration evaluator values of the multi-rank
    fit equal the single-process oracle fit; the replicas are bitwise identical."""
    import torch.multiprocessing as mp

    mp.spawn(_dp_worker, args=(world, _free_port(), str(tmp_path), exchange, with_evaluator), nprocs=world,
             join=True)
    train, val = synth.make_log("coat", "FM", "IPS", seed=0)
    kw = {k: v for k, v in DP_KW.items() if k != "estimator"}
    ev = _CountingEvaluator(val["features"])
    ref = cpu_ref.fm_fit(train, val, **kw, score_hook=(lambda s: ev.evaluate(s, "IPS")) if with_evaluator else None,
                         hook_features=val["features"] if with_evaluator else None)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs:
        for name in ("V", "w", "w0"):
            assert np.max(np.abs(o[name] - ref[name])) < 1e-12, name
        assert np.max(np.abs(o["tr"] - np.array(ref["train_loss"]))) < 1e-12
        assert np.max(np.abs(o["va"] - np.array(ref["val_loss"]))) < 1e-12
        if with_evaluator:
            assert np.max(np.abs(o["metrics"] - np.array(ref["val_metrics"]))) < 1e-12
    for o in outs[1:]:
        for name in ("V", "w", "w0", "tr", "va"):
            np.testing.assert_array_equal(outs[0][name], o[name])
