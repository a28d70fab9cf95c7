"""Data-parallel FM step over the GPUs of one node.

The step's gradient is a plain SUM over batch rows (no 1/|B|;
``src/fm.py:142,153,178-180``), so: parameters replicated on every rank, the
batch's row-id list cut into contiguous shards, each rank computes the dense
gradient ``[G_V | g_w | g_w0]`` of its shard (``rfm_fm_grad``), one all-reduce
(SUM) over RCCL/xGMI, and every rank applies the same update
(``rfm_fm_apply``).  Mathematically the single-GPU step; only the summation
order of the cross-rank add differs.

``torch.distributed`` is the transport (backend "nccl" is RCCL on ROCm; "gloo"
in the CPU tests).  The arithmetic is injected, so the sharding / exchange /
apply logic can be exercised without a GPU.
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_bounds(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ``batch`` rows for ``rank``; the first
    ``batch % world`` ranks take one extra row."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class DataParallelStep:
    """One exchange step per mini-batch.

    grad_fn(lo, hi, it)  writes the gradient of batch rows [lo, hi) of iteration
                         ``it`` into ``grad`` (all elements; zeros if lo == hi)
    all_reduce(grad)     in-place SUM over ranks
    apply_fn(grad)       theta -= lr * grad
    """

    def __init__(self, grad, grad_fn: Callable, apply_fn: Callable, all_reduce: Callable,
                 world: int, rank: int):
        self.grad, self.grad_fn, self.apply_fn, self.all_reduce = grad, grad_fn, apply_fn, all_reduce
        self.world, self.rank = world, rank

    def step(self, it: int, global_batch: int) -> None:
        lo, hi = shard_bounds(global_batch, self.world, self.rank)
        self.grad_fn(lo, hi, it)
        if self.world > 1:
            self.all_reduce(self.grad)
        self.apply_fn(self.grad)


def hip_fm_worker(rt, plan, csr, y, p, d_ids, global_batch: int, model, grad, world: int, rank: int,
                  lr: float, all_reduce=None) -> DataParallelStep:
    """Bind ``DataParallelStep`` to the HIP kernels.  ``d_ids`` holds the GLOBAL
    batches, ``(n_iters, global_batch)`` int32 on the device; ``grad`` is a
    float64 device tensor of ``n*(k+1)+1`` elements.  ``all_reduce`` defaults to
    ``torch.distributed.all_reduce`` (RCCL when the group's backend is nccl)."""
    import torch.distributed as dist

    from . import _lib

    n, k = model.n_features, model.n_factors
    params = (model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr())

    def grad_fn(lo: int, hi: int, it: int) -> None:
        if hi == lo:
            grad.zero_()
            return
        ids_ptr = d_ids.data_ptr() + (it * global_batch + lo) * 4
        _lib.check(rt.lib.rfm_fm_grad(
            rt.ctx, plan.handle, csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr(),
            y.data_ptr(), p.data_ptr(), ids_ptr, hi - lo, *params, grad.data_ptr()))

    def apply_fn(g) -> None:
        _lib.check(rt.lib.rfm_fm_apply(rt.ctx, *params, g.data_ptr(), n, k, float(lr)))

    def default_all_reduce(g) -> None:
        dist.all_reduce(g, op=dist.ReduceOp.SUM)

    return DataParallelStep(grad, grad_fn, apply_fn, all_reduce or default_all_reduce, world, rank)


def init_direct_rccl(rt, world: int, rank: int) -> bool:
    """Give the runtime's context its own RCCL communicator (``rfm_comm_init``): rank 0
    draws the id and the existing torch.distributed group carries it to the others.
    Returns False on every rank (and touches nothing) if rank 0 cannot bind RCCL."""
    import ctypes as C

    import torch.distributed as dist

    from . import _lib

    box = [None]
    if rank == 0:
        uid = (C.c_uint8 * 128)()
        try:
            _lib.check(rt.lib.rfm_comm_unique_id(uid))
            box = [bytes(uid)]
        except Exception:  # noqa: BLE001 -- reported by the caller through the False result
            box = [None]
    dist.broadcast_object_list(box, src=0)  # every rank takes part, success or not
    if box[0] is None:
        return False
    buf = (C.c_uint8 * 128).from_buffer_copy(box[0])
    _lib.check(rt.lib.rfm_comm_init(rt.ctx, world, rank, buf))
    # known-answer all-reduce before anything depends on the communicator: sum of the ranks'
    # numbers (a wrong enum value or ABI mismatch shows here, not as a diverged model)
    import torch

    probe = torch.full((3,), float(rank + 1), dtype=torch.float64, device=rt.device)
    _lib.check(rt.lib.rfm_allreduce_sum(rt.ctx, probe.data_ptr(), 3))
    rt.sync()
    want = world * (world + 1) / 2
    if not bool((probe == want).all().item()):
        raise RuntimeError(f"RCCL all-reduce self-test: got {probe.tolist()}, want {want}")
    return True


def hip_fm_train_dp(rt, plan, d_ids, global_batch: int, first: int, count: int, model, grad,
                    world: int, rank: int, lr: float) -> None:
    """``count`` data-parallel iterations starting at ``first`` in one C call
    (``rfm_fm_train_dp``): gradient of this rank's shard, RCCL all-reduce on the
    compute stream, identical apply -- no host round trip between the three."""
    from . import _lib

    lo, hi = shard_bounds(global_batch, world, rank)
    _lib.check(rt.lib.rfm_fm_train_dp(
        rt.ctx, plan.handle, d_ids.data_ptr() + first * global_batch * 4, global_batch, lo, hi, count,
        model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr(), float(lr),
        grad.data_ptr()))


# ---------------------------------------------------------------------------
# touched-row exchange (option 1: feature-range ownership)
# ---------------------------------------------------------------------------
def owner_ranges(n_features: int, world: int):
    """First column of every rank's range (ascending, ``world`` entries): rank r owns
    columns ``[lo[r], lo[r+1])`` (the last range ends at ``n_features``)."""
    import numpy as np

    return np.array([n_features * r // world for r in range(world)], dtype=np.int32)


class TorchRowComm:
    """The three collectives of one exchange over ``torch.distributed``.

    ``stage_host`` moves the payloads through host memory (gloo; also how two ranks that
    share one GPU in the tests talk); with backend nccl (= RCCL over xGMI) they stay on the
    device.  Records are rows of ``width`` float64."""

    def __init__(self, dist, world: int, rank: int, width: int, stage_host: bool):
        self.dist, self.world, self.rank, self.width, self.stage_host = dist, world, rank, width, stage_host

    def _wire(self, t):
        return t.cpu() if self.stage_host else t

    def gather_meta(self, meta):
        """all-gather of one small float64 vector per rank -> host ndarray [world][len]."""
        import torch

        m = self._wire(meta)
        out = [torch.empty_like(m) for _ in range(self.world)]
        self.dist.all_gather(out, m)
        return torch.stack(out).cpu().numpy()

    def all_to_all_rows(self, rows, send_counts, recv_counts):
        """Records [sum(send_counts)][width] cut by destination -> records [sum(recv_counts)][width]
        ordered by source rank."""
        import torch

        src = self._wire(rows).reshape(-1)
        out = torch.empty(int(sum(recv_counts)) * self.width, dtype=src.dtype, device=src.device)
        self.dist.all_to_all_single(out, src[: int(sum(send_counts)) * self.width],
                                    [int(c) * self.width for c in recv_counts],
                                    [int(c) * self.width for c in send_counts])
        out = out.reshape(-1, self.width)
        return out.to(rows.device) if self.stage_host else out

    def all_gather_rows(self, rows, counts):
        """Every rank's record list (``counts[r]`` records from rank r) -> all of them, by rank."""
        import torch

        mine = self._wire(rows).reshape(-1)[: int(counts[self.rank]) * self.width]
        parts = [torch.empty(int(c) * self.width, dtype=mine.dtype, device=mine.device) for c in counts]
        if len({int(c) for c in counts}) == 1:
            self.dist.all_gather(parts, mine)
        else:
            self._all_gather_v(parts, mine)
        out = torch.cat(parts).reshape(-1, self.width)
        return out.to(rows.device) if self.stage_host else out

    def _all_gather_v(self, parts, mine):
        # lists of different lengths: one broadcast per source rank
        for r in range(self.world):
            if r == self.rank:
                parts[r].copy_(mine)
            if parts[r].numel():
                self.dist.broadcast(parts[r], src=r)


class RowExchangeStep:
    """One data-parallel FM step that moves only the touched rows.

    Parameters are replicated; rank r owns the columns ``[lo[r], lo[r+1])``.  Per step:

    1. ``grad_rows_fn(lo, hi, it)`` -> this rank's gradient records of batch rows [lo, hi) of
       iteration ``it`` -- ``(rows [cnt][k+2] ascending by column, bounds int[world+1]``
       = position of each owner range in the list, ``g_w0 partial)``;
    2. all-gather of ``[bounds | g_w0]`` (one small vector per rank): every rank now knows
       every transfer size of the step and all partial g_w0;
    3. all-to-all: each record goes to the rank that owns its column;
    4. ``reduce_fn(recv, seg_ptr)``: the owner adds the records of a column in RANK ORDER and
       updates its row -> ``[column, V_new, w_new]`` (column -1 at the positions of a column's
       other records, so list sizes are known beforehand);
    5. all-gather of the updated rows; ``set_rows_fn(all_rows, gw0_parts)`` stores them into
       the local replica and applies ``w0 -= lr * sum(gw0_parts)`` (rank order).

    Every replica ends the step with identical parameters, bit for bit: each row is computed
    once, by its owner, and copied.  The arithmetic is injected (HIP kernels in
    ``hip_fm_rows_worker``; NumPy from the oracle in the CPU tests)."""

    def __init__(self, world: int, rank: int, comm, grad_rows_fn: Callable, reduce_fn: Callable,
                 set_rows_fn: Callable):
        self.world, self.rank, self.comm = world, rank, comm
        self.grad_rows_fn, self.reduce_fn, self.set_rows_fn = grad_rows_fn, reduce_fn, set_rows_fn
        self.last_counts = None  # records sent per destination in the last step (diagnostics)

    def step(self, it: int, global_batch: int) -> None:
        import numpy as np

        lo, hi = shard_bounds(global_batch, self.world, self.rank)
        rows, meta = self.grad_rows_fn(lo, hi, it)  # meta: float64 [world + 2] = bounds | g_w0
        allmeta = self.comm.gather_meta(meta)  # host [world][world + 2]; the step's one host sync
        bounds = np.rint(allmeta[:, : self.world + 1]).astype(np.int64)
        send = np.diff(bounds[self.rank])
        recv = np.array([bounds[s, self.rank + 1] - bounds[s, self.rank] for s in range(self.world)])
        self.last_counts = send
        got = self.comm.all_to_all_rows(rows, send, recv)
        seg_ptr = np.concatenate([[0], np.cumsum(recv)]).astype(np.int32)
        upd = self.reduce_fn(got, seg_ptr)
        # rank r emits as many (padded) records as it received
        out_counts = [int(sum(bounds[s, r + 1] - bounds[s, r] for s in range(self.world)))
                      for r in range(self.world)]
        everything = self.comm.all_gather_rows(upd, out_counts)
        self.set_rows_fn(everything, allmeta[:, self.world + 1])


def hip_fm_rows_worker(rt, plan, d_ids, global_batch: int, model, world: int, rank: int, lr: float,
                       comm, cap_rows: int = 0) -> RowExchangeStep:
    """Bind ``RowExchangeStep`` to the HIP kernels (``rfm_fm_grad_rows``, ``rfm_fm_reduce_rows``,
    ``rfm_fm_set_rows``).  ``d_ids`` holds the GLOBAL batches ``(n_iters, global_batch)`` int32
    on the device; ``cap_rows`` bounds a rank's record list (default: every column)."""
    import numpy as np
    import torch

    from . import _lib

    n, k = model.n_features, model.n_factors
    width = k + 2
    cap = int(cap_rows) if cap_rows else n
    params = (model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr())
    rows = rt.empty((cap, width), torch.float64)
    n_rows = rt.empty((1,), torch.int32)
    meta = rt.empty((world + 2,), torch.float64)
    bounds = rt.empty((world + 1,), torch.int32)
    gw0 = rt.empty((1,), torch.float64)
    range_lo = rt.upload(owner_ranges(n, world))

    def grad_rows_fn(lo: int, hi: int, it: int):
        ids_ptr = d_ids.data_ptr() + (it * global_batch + lo) * 4
        _lib.check(rt.lib.rfm_fm_grad_rows(
            rt.ctx, plan.handle, ids_ptr, hi - lo, *params, rows.data_ptr(), cap, n_rows.data_ptr(),
            gw0.data_ptr(), range_lo.data_ptr(), world, bounds.data_ptr()))
        meta[: world + 1] = bounds
        meta[world + 1] = gw0[0]
        return rows, meta

    def reduce_fn(got, seg_ptr):
        total = int(seg_ptr[-1])
        out = rt.empty((max(total, 1), width), torch.float64)
        d_seg = rt.upload(seg_ptr)
        _lib.check(rt.lib.rfm_fm_reduce_rows(
            rt.ctx, got.data_ptr() if total else None, d_seg.data_ptr(), world, total, params[1], params[2],
            n, k, float(lr), out.data_ptr() if total else None))
        return out

    def set_rows_fn(everything, gw0_parts):
        parts = rt.upload(np.ascontiguousarray(gw0_parts, dtype=np.float64))
        cnt = int(everything.shape[0])
        _lib.check(rt.lib.rfm_fm_set_rows(
            rt.ctx, everything.data_ptr() if cnt else None, cnt, parts.data_ptr(), world, 1, *params,
            n, k, float(lr)))

    return RowExchangeStep(world, rank, comm, grad_rows_fn, reduce_fn, set_rows_fn)


class RowExchange:
    """Factory for the transport of ``RowExchangeStep`` (kept apart so bench.py and the tests
    pick the same wiring)."""

    @staticmethod
    def for_torch(dist, world: int, rank: int, n_features: int, n_factors: int, backend: str = "nccl",
                  stage_host=None) -> TorchRowComm:
        del n_features
        if stage_host is None:
            stage_host = backend != "nccl"
        return TorchRowComm(dist, world, rank, n_factors + 2, stage_host)


# ---------------------------------------------------------------------------
# MF across GPUs: user-range partition; NOT the reference's semantics
# ---------------------------------------------------------------------------
def user_ranges(n_users: int, world: int):
    """First user of every rank's range (``world + 1`` entries, the last = ``n_users``)."""
    import numpy as np

    return np.array([n_users * r // world for r in range(world + 1)], dtype=np.int64)


class MfUserPartitionStep:
    """One mini-batch of logistic MF over ``world`` ranks.

    The reference's batch is strictly sequential (``src/mf.py:97-108``), so it does not
    shard with parameter parity; exact mode is one GPU (or independent replicas).  This is
    the throughput mode the north star names next to it: rank r owns the users
    ``[lo[r], lo[r+1])`` -- their rows of P / b_u are written by nobody else -- and holds a
    replica of Q / b_i.  Per batch it runs the EXACT sequential SGD over the batch's
    examples of its own users, in batch order, against its replica; then the replicas'
    changes to Q / b_i are summed (all-reduce of the deltas) and every replica continues
    from ``Q_before + sum of deltas``.  Examples of different ranks that share an item no
    longer see each other's update inside a batch: results match the reference at the loss /
    ranking level, not to 1e-5 on the parameters.  With one rank nothing is exchanged and
    the result IS the reference's.

    sgd_fn(it, positions)   exact sequential SGD over the given batch positions (ascending)
    delta_fn() -> tensor    replica's Q / b_i minus their values after the last merge
    merge_fn(total)         Q / b_i := values after the last merge + total
    all_reduce(tensor)      in-place SUM over ranks
    """

    def __init__(self, world: int, rank: int, n_users: int, sgd_fn: Callable, delta_fn: Callable,
                 merge_fn: Callable, all_reduce: Callable):
        self.world, self.rank = world, rank
        self.lo = user_ranges(n_users, world)
        self.sgd_fn, self.delta_fn, self.merge_fn, self.all_reduce = sgd_fn, delta_fn, merge_fn, all_reduce

    def mine(self, users):
        """Batch positions (ascending) of the examples whose user this rank owns."""
        import numpy as np

        users = np.asarray(users)
        return np.flatnonzero((users >= self.lo[self.rank]) & (users < self.lo[self.rank + 1]))

    def step(self, it: int, batch_users) -> None:
        self.sgd_fn(it, self.mine(batch_users))
        if self.world > 1:
            d = self.delta_fn()
            self.all_reduce(d)
            self.merge_fn(d)


def hip_mf_partition_worker(rt, model, train: dict, world: int, rank: int, all_reduce=None,
                            stage_host: bool = False):
    """Bind ``MfUserPartitionStep`` to the HIP kernels for ``model`` (a
    ``LogisticMatrixFactorization``) and the training log ``train``.  Returns
    ``(step, finish)``: ``step.step(it, users_of_batch)`` with the batch's row ids in
    ``step.batch_rows(it)``; ``finish()`` gathers the owners' P / b_u rows so that every rank
    holds the whole model."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from . import _lib
    from .mf import DevicePairs
    from .runtime import mf_cache_capacity, mf_schedule_ex, sample_batches

    tr = DevicePairs(rt, train["features"])
    h_y = np.ascontiguousarray(train["labels"], dtype=np.float64)
    h_p = np.ascontiguousarray(train["pscores"], dtype=np.float64)
    k, B = model.n_factors, model.batch_size
    model.b = float(np.mean(train["labels"]))
    params = (model.P.dev.data_ptr(), model.Q.dev.data_ptr(), model.b_u.dev.data_ptr(),
              model.b_i.dev.data_ptr())
    cache_cap = mf_cache_capacity(k)
    nq, ni = model.n_items * k, model.n_items
    sync = torch.cat([model.Q.dev.reshape(-1), model.b_i.dev.reshape(-1)]).clone()
    delta = torch.empty_like(sync)
    keep = []
    # The host side of a batch -- exact sampler, the rank's examples, their level schedule --
    # depends on the iteration number alone: a worker thread prepares iteration it + 1 while the
    # GPU runs iteration it (the calls release the GIL); a few iterations ahead, one thread each.
    from concurrent.futures import ThreadPoolExecutor

    import os

    depth = max(1, min(8, (os.cpu_count() or 2) // 2))  # iterations prepared ahead, one thread each
    pool = ThreadPoolExecutor(max_workers=depth, thread_name_prefix="rfm-mf-part")
    ahead, cur = {}, {}
    u_lo = user_ranges(model.n_users, world)

    def prepare(it: int):
        rows = sample_batches(tr.n, B, it, 1, n_threads=1)[0]
        users = tr.h_users[rows]
        pos = np.flatnonzero((users >= u_lo[rank]) & (users < u_lo[rank + 1]))
        mine = rows[pos]
        sched = (mf_schedule_ex(tr.h_users[mine], tr.h_items[mine], h_y[mine], h_p[mine], model.n_users,
                                model.n_items, cache_cap) if pos.size else None)
        return rows, pos, sched

    def batch(it: int):
        if cur.get("it") != it:
            fut = ahead.pop(it, None)
            cur.update(it=it, data=fut.result() if fut is not None else prepare(it))
            for stale in [j for j in ahead if not it < j <= it + depth]:
                ahead.pop(stale).cancel()
            for nxt in range(it + 1, it + 1 + depth):
                if nxt not in ahead:
                    ahead[nxt] = pool.submit(prepare, nxt)
        return cur["data"]

    def batch_rows(it: int) -> np.ndarray:
        return batch(it)[0]

    def sgd_fn(it: int, positions: np.ndarray) -> None:
        if positions.size == 0:
            return
        rows_all, pos, sched = batch(it)
        if sched is None or not np.array_equal(pos, positions):  # (positions other than the rank's own)
            rows = rows_all[positions]
            sched = mf_schedule_ex(tr.h_users[rows], tr.h_items[rows], h_y[rows], h_p[rows],
                                   model.n_users, model.n_items, cache_cap)
        ex, level_ptr, cache_items = sched
        d_ex, d_lptr = rt.upload(ex.view(np.uint8)), rt.upload(level_ptr)
        d_cache = rt.upload(cache_items if cache_items.size else np.zeros(1, np.int32))
        keep.append((d_ex, d_lptr, d_cache, level_ptr))
        if len(keep) > 32:
            rt.sync()
            del keep[:-1]
        _lib.check(rt.lib.rfm_mf_sgd_levels_ex(
            rt.ctx, d_ex.data_ptr(), level_ptr.ctypes.data, d_lptr.data_ptr(), len(level_ptr) - 1,
            d_cache.data_ptr(), int(cache_items.size), *params, float(model.b), k, float(model.lr),
            float(model.reg)))

    def delta_fn():
        _lib.check(rt.lib.rfm_mf_delta(rt.ctx, params[1], sync.data_ptr(), delta.data_ptr(), nq))
        _lib.check(rt.lib.rfm_mf_delta(rt.ctx, params[3], sync.data_ptr() + nq * 8, delta.data_ptr() + nq * 8, ni))
        return delta

    def merge_fn(total) -> None:
        _lib.check(rt.lib.rfm_mf_merge(rt.ctx, params[1], sync.data_ptr(), total.data_ptr(), nq))
        _lib.check(rt.lib.rfm_mf_merge(rt.ctx, params[3], sync.data_ptr() + nq * 8, total.data_ptr() + nq * 8, ni))

    def default_all_reduce(t) -> None:
        if stage_host:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    step = MfUserPartitionStep(world, rank, model.n_users, sgd_fn, delta_fn, merge_fn,
                               all_reduce or default_all_reduce)
    step.batch_rows = batch_rows
    step.users_of = lambda it: tr.h_users[batch_rows(it)]

    def finish() -> None:
        rt.sync()
        for fut in ahead.values():
            fut.cancel()
        pool.shutdown(wait=True)
        if world == 1:
            return
        lo = step.lo
        for r in range(world):
            for t, width in ((model.P.dev, k), (model.b_u.dev, 1)):
                part = t.reshape(-1)[lo[r] * width: lo[r + 1] * width]
                if part.numel() == 0:
                    continue
                wire = part.cpu() if stage_host else part
                dist.broadcast(wire, src=r)
                if stage_host:
                    part.copy_(wire)
        rt.sync()

    return step, finish


# ---------------------------------------------------------------------------
# data-parallel fit(): the reference's loop (src/fm.py:71-112) on several GPUs
# ---------------------------------------------------------------------------
class HostStagedTransport:
    """The three collectives of ``rfm_fm_fit_dp`` over ``torch.distributed`` with the payloads
    staged through host memory: the transport of the gloo rehearsals and of ranks that share
    one GPU in the tests.  (With backend nccl nothing of this is used: ``transport=None`` makes
    the library call RCCL itself, on the compute stream -- ``init_direct_rccl``.)

    Host API (NumPy in, NumPy out) for engines that keep their buffers on the host; ``c_struct``
    wraps it as the ``rfm_transport`` of include/rfm_hip.h for the HIP engine, which hands the
    callbacks raw device pointers (``rfm_copy_to_host`` / ``rfm_copy_to_device``)."""

    def __init__(self, world: int, rank: int, rt=None, group=None):
        import torch.distributed as dist

        self.dist, self.world, self.rank, self.rt, self.group = dist, world, rank, rt, group
        self.error = None
        self._keep = None

    # ---- host API ---------------------------------------------------------
    def all_gather_host(self, arr):
        import numpy as np
        import torch

        mine = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1))
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine, group=self.group)
        return np.stack([o.numpy() for o in out]).view(arr.dtype).reshape((self.world,) + tuple(arr.shape))

    def all_reduce_sum_host(self, arr):
        import numpy as np
        import torch

        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.numpy()

    def all_to_all_host(self, send, recv_bytes):
        """``send[p]``: uint8 array for peer p; returns the list of uint8 arrays received
        (``recv_bytes[p]`` bytes from peer p)."""
        import numpy as np
        import torch

        src = torch.from_numpy(np.concatenate([np.ascontiguousarray(s).view(np.uint8).reshape(-1) for s in send])
                               if sum(len(s) for s in send) else np.zeros(0, np.uint8))
        out = torch.empty(int(sum(recv_bytes)), dtype=torch.uint8)
        self.dist.all_to_all_single(out, src, [int(b) for b in recv_bytes], [int(len(s)) for s in send],
                                    group=self.group)
        got, at = [], 0
        for b in recv_bytes:
            got.append(out.numpy()[at:at + int(b)])
            at += int(b)
        return got

    # ---- the C struct -----------------------------------------------------
    def c_struct(self):
        import ctypes as C

        import numpy as np

        from . import _lib

        rt, W = self.rt, self.world
        assert rt is not None, "a transport for the HIP engine needs the runtime"

        def d2h(ptr, nbytes):
            buf = np.empty(int(nbytes), dtype=np.uint8)
            _lib.check(rt.lib.rfm_copy_to_host(rt.ctx, buf.ctypes.data, ptr, int(nbytes)))
            return buf

        def h2d(ptr, buf):
            buf = np.ascontiguousarray(buf)
            _lib.check(rt.lib.rfm_copy_to_device(rt.ctx, ptr, buf.ctypes.data, buf.nbytes))

        def guard(fn):
            def wrapped(*args):
                try:
                    fn(*args)
                    return 0
                except BaseException as exc:  # noqa: BLE001 -- reported through the C return code
                    self.error = exc
                    return 1
            return wrapped

        @guard
        def all_gather(_user, d_send, d_recv, nbytes):
            h2d(d_recv, self.all_gather_host(d2h(d_send, nbytes)).reshape(-1))

        @guard
        def all_reduce(_user, d_buf, count):
            h2d(d_buf, self.all_reduce_sum_host(d2h(d_buf, count * 8).view(np.float64)))

        @guard
        def all_to_all(_user, d_send, soff, sbytes, d_recv, roff, rbytes):
            send = [d2h(d_send + soff[p], sbytes[p]) for p in range(W)]
            got = self.all_to_all_host(send, [rbytes[p] for p in range(W)])
            for p in range(W):
                if rbytes[p]:
                    h2d(d_recv + roff[p], got[p])

        cbs = (_lib.TRANSPORT_ALL_GATHER(all_gather), _lib.TRANSPORT_ALL_REDUCE(all_reduce),
               _lib.TRANSPORT_ALL_TO_ALL(all_to_all))
        struct = _lib.Transport(None, W, self.rank, *cbs)
        self._keep = (cbs, struct)  # the callbacks must outlive the C calls
        return struct


def choose_exchange(n_features: int, n_factors: int, global_batch: int) -> str:
    """"rows" when the touched rows of a global batch are estimated at under a quarter of the
    dense gradient buffer (about two one-hot columns per row plus the side features are touched,
    and the records travel twice -- to the owner and back); else "dense"."""
    touched = min(n_features, 2 * global_batch + 256)
    return "rows" if 4 * touched * (n_factors + 2) * 2 < n_features * (n_factors + 1) else "dense"


class HipDpEngine:
    """One rank's side of ``fit_data_parallel`` on the HIP kernels: the log replicated in HBM,
    a training plan for this rank's shard size, and runs of iterations handed to
    ``rfm_fm_fit_dp`` (gradients, exchange, update and both losses enqueued on the stream)."""

    def __init__(self, model, train, val, world: int, rank: int, exchange: str, transport):
        import numpy as np

        from . import _lib
        from .fm import FmPlan
        from .runtime import BatchIdStream, DeviceCSR

        rt = model._rt
        self.rt, self.model, self.world, self.rank = rt, model, world, rank
        X = train["features"]
        if X.shape[1] != model.n_features:
            raise ValueError(f"train features have {X.shape[1]} columns, model has {model.n_features}")
        self.global_batch = model.batch_size
        self.ids = BatchIdStream(rt, X.shape[0], self.global_batch, model.n_epochs, need_host=False)
        try:
            self.tr = X if isinstance(X, DeviceCSR) else DeviceCSR(rt, X)
            self.y = rt.upload(np.asarray(train["labels"]), dtype=np.float64)
            self.p = rt.upload(np.asarray(train["pscores"]), dtype=np.float64)
            vX = val["features"]
            self.va = vX if isinstance(vX, DeviceCSR) else DeviceCSR(rt, vX)
            self.vy = rt.upload(np.asarray(val["labels"]), dtype=np.float64)
            self.vp = rt.upload(np.asarray(val["pscores"]), dtype=np.float64)
            lo, hi = shard_bounds(self.global_batch, world, 0)  # rank 0 holds a largest shard
            hot = -2 if model.deterministic else model.hot_min_count
            self.plan = FmPlan(rt, self.tr, self.y, self.p, model.n_factors, max(hi - lo, 1), hot)
        except BaseException:
            self.ids.close()
            raise
        self.exchange = {"dense": 0, "rows": 1}[exchange]
        self.transport = transport
        self._c_transport = transport.c_struct() if transport is not None else None
        self.tl = rt.empty((model.n_epochs,), self.y.dtype)
        self.vl = rt.empty((model.n_epochs,), self.y.dtype)
        self.has_val = self.va.shape[0] > 0
        if not self.has_val:
            self.vl.fill_(float("nan"))
        self._lib = _lib

    def chunks(self):
        for first, _host, dev_ids in self.ids.chunks():
            yield first, dev_ids.shape[0], dev_ids

    def run(self, first: int, count: int, chunk_first: int, dev_ids) -> None:
        import ctypes as C

        from .base import LOSS_EPS

        m, rt, B = self.model, self.rt, self.global_batch
        ids_ptr = dev_ids.data_ptr() + (first - chunk_first) * B * 4
        tp = C.byref(self._c_transport) if self._c_transport is not None else None
        rc = rt.lib.rfm_fm_fit_dp(
            rt.ctx, self.plan.handle, tp, self.exchange, ids_ptr, B, count,
            m.w0.dev.data_ptr(), m.w.dev.data_ptr(), m.V.dev.data_ptr(), float(m.lr),
            self.va.indptr.data_ptr(), self.va.indices.data_ptr(), self.va.values.data_ptr(),
            self.vy.data_ptr(), self.vp.data_ptr(), self.va.shape[0], LOSS_EPS,
            self.tl.data_ptr() + first * 8, self.vl.data_ptr() + first * 8 if self.has_val else None)
        if rc != 0 and self.transport is not None and self.transport.error is not None:
            raise self.transport.error
        self._lib.check(rc)

    def predict(self, X):
        return self.model.predict(X=X)

    def losses(self):
        self.rt.sync()
        return self.tl.cpu().numpy().tolist(), self.vl.cpu().numpy().tolist()

    def close(self) -> None:
        self.ids.close()
        self.rt.sync()
        self.plan.close()


def fit_data_parallel(model, train: dict, val: dict, exchange: str = "auto", transport=None,
                      engine_factory=None) -> tuple:
    """``model.fit(train, val)`` over the ranks of the initialised ``torch.distributed`` group:
    every rank calls this with the same model (same seed, hence the same initial parameters),
    the same split and the GLOBAL ``batch_size``; returns the same two loss lists on every rank
    (``src/fm.py:112``) and leaves every replica with the same parameters.

    Per iteration (``src/fm.py:71-102``) the global batch -- ``resample(..., random_state=epoch)``
    as ever -- is cut into contiguous shards (``shard_bounds``); each rank computes the batch-SUM
    gradient of its shard, the shards are combined (``exchange``: "dense" all-reduce of
    ``[G_V | g_w | g_w0]``, "rows" = touched rows through their owner, "auto" =
    ``choose_exchange``), every replica applies the same update; the train loss of the batch
    with the new parameters and the validation loss are computed as per-rank sums over row
    shards and combined with ONE all-reduce per run of iterations.  The sums
    differ from the single-GPU fit in summation order only.

    ``transport``: None = RCCL inside the library (backend nccl: one process per GPU);
    ``HostStagedTransport`` otherwise.  ``model.evaluator`` is called after every iteration on
    every rank (the replicas are identical, so are the metrics), as ``src/fm.py:104-110``.
    ``engine_factory`` replaces the arithmetic (tests)."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    if model.n_epochs <= 0:
        return [], []
    if exchange == "auto":
        exchange = choose_exchange(model.n_features, model.n_factors, model.batch_size)
    if exchange not in ("dense", "rows"):
        raise ValueError(f"exchange must be auto, dense or rows, got {exchange!r}")
    if engine_factory is None:
        if transport is None and world > 1:
            rt = model._rt
            if not getattr(rt, "_direct_rccl", False):
                if dist.get_backend() != "nccl" or not init_direct_rccl(rt, world, rank):
                    raise RuntimeError("no transport: backend nccl binds RCCL inside the library; other "
                                       "backends need HostStagedTransport(world, rank, rt)")
                rt._direct_rccl = True
        engine_factory = HipDpEngine
    engine = engine_factory(model, train, val, world, rank, exchange, transport)
    try:
        evaluator = getattr(model, "evaluator", None)
        for chunk_first, count, ids in engine.chunks():
            if evaluator is None:
                engine.run(chunk_first, count, chunk_first, ids)
                continue
            for epoch in range(chunk_first, chunk_first + count):
                engine.run(epoch, 1, chunk_first, ids)
                scores = engine.predict(evaluator.features[model.model_name])
                model.val_metrics.append(evaluator.evaluate(y_scores=scores, estimator=model.estimator))
        return engine.losses()
    finally:
        engine.close()
