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
