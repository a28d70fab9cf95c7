"""``LogisticMatrixFactorization`` with the reference's surface
(``src/mf.py:16-170``) on top of the HIP kernels of librfm_hip.so.

The reference updates a batch strictly example by example
(``src/mf.py:97-108``).  That order is kept exactly: the host derives, per
batch, the level schedule (``rfm_mf_schedule``) under which every example runs
after the latest earlier example sharing its user or item, and the device
executes the levels in order (``csrc/rfm_mf.hip``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from .base import LOSS_EPS, PointwiseBaseRecommender
from .evaluate import EvalLoop, device_frame
from .optimizer import DeviceSGD
from .runtime import BatchIdStream, Runtime, mf_cache_capacity, mf_schedule_ex


class _SchedulePipe:
    """Host side of the exact MF batches, off the critical path: a producer thread derives the
    level schedule of the coming iterations (``rfm_mf_schedule_ex``; ctypes releases the GIL)
    and packs it -- records, level pointers, cached items -- into a ring of PINNED host
    buffers; the consumer enqueues one asynchronous copy per iteration into the matching
    device slot and launches.  A host slot is handed back to the producer only after the copy
    out of it has completed (an event), a device slot is reused ``DEPTH`` iterations later on
    the same stream, i.e. after the kernels that read it."""

    DEPTH = 8

    def __init__(self, rt: Runtime, batch_size: int, cache_cap: int):
        import queue

        import torch

        self.rt, self.B, self.cache_cap = rt, batch_size, cache_cap
        self.ex_bytes = 24 * batch_size
        self.lptr_off = (self.ex_bytes + 15) // 16 * 16
        self.cache_off = (self.lptr_off + 4 * (batch_size + 1) + 15) // 16 * 16
        self.total = self.cache_off + 4 * max(cache_cap, 1)
        self.host = [torch.empty(self.total, dtype=torch.uint8, pin_memory=True) for _ in range(self.DEPTH)]
        self.dev = [rt.empty((self.total,), torch.uint8) for _ in range(self.DEPTH)]
        self.free: "queue.Queue" = queue.Queue()
        for s in range(self.DEPTH):
            self.free.put((s, None))
        self.ready: "queue.Queue" = queue.Queue(maxsize=self.DEPTH)
        self.torch = torch

    def produce(self, epochs, schedule_fn) -> None:
        """Runs in the producer thread: ``schedule_fn(rows) -> (ex, level_ptr, cache_items)``
        for every ``(epoch, rows, ids_ptr)`` of ``epochs``."""
        try:
            for epoch, rows, ids_ptr in epochs:
                slot, ev = self.free.get()
                if slot is None:
                    return
                if ev is not None:
                    ev.synchronize()  # the copy out of this host slot has completed
                ex, level_ptr, cache_items = schedule_fn(rows)
                h = self.host[slot].numpy()
                h[: self.ex_bytes] = ex.view(np.uint8)
                h[self.lptr_off: self.lptr_off + 4 * len(level_ptr)] = level_ptr.view(np.uint8)
                h[self.cache_off: self.cache_off + 4 * len(cache_items)] = cache_items.view(np.uint8)
                self.ready.put((epoch, ids_ptr, slot, len(level_ptr) - 1, len(cache_items)))
            self.ready.put(None)
        except BaseException as exc:  # noqa: BLE001 -- handed to the consumer
            self.ready.put(exc)
        finally:
            # the generator runs id_stream.chunks() in THIS thread: closing it here runs that
            # generator's cleanup (sampler thread stopped, pinned buffers released)
            close = getattr(epochs, "close", None)
            if close is not None:
                close()

    def take(self):
        """Next iteration: enqueues the copy and returns ``(epoch, ids_ptr, device ex ptr, HOST
        level_ptr ptr, device level_ptr ptr, n_levels, device cache ptr, n_cached, done)``;
        call ``done()`` after the launches that read the slot have been enqueued."""
        item = self.ready.get()
        if item is None:
            return None
        if isinstance(item, BaseException):
            raise item
        epoch, ids_ptr, slot, n_levels, n_cached = item
        self.dev[slot].copy_(self.host[slot], non_blocking=True)
        ev = self.torch.cuda.Event()
        ev.record()
        base, hbase = self.dev[slot].data_ptr(), self.host[slot].data_ptr()

        def done() -> None:
            self.free.put((slot, ev))

        return (epoch, ids_ptr, base, hbase + self.lptr_off, base + self.lptr_off, n_levels,
                base + self.cache_off, n_cached, done)

    def stop(self) -> None:
        self.free.put((None, None))


class DevicePairs:
    """(user, item) pairs of a log in HBM as two int32 arrays."""

    def __init__(self, rt: Runtime, pairs: np.ndarray):
        pairs = np.asarray(pairs)
        if pairs.ndim != 2 or pairs.shape[1] < 2:
            raise ValueError("MF features must be an (N, 2) array of [user, item]")
        self.n = int(pairs.shape[0])
        self.h_users = np.ascontiguousarray(pairs[:, 0], dtype=np.int32)
        self.h_items = np.ascontiguousarray(pairs[:, 1], dtype=np.int32)
        self.users = rt.upload(self.h_users if self.n else np.zeros(1, np.int32))
        self.items = rt.upload(self.h_items if self.n else np.zeros(1, np.int32))


@dataclass
class LogisticMatrixFactorization(PointwiseBaseRecommender):
    """Logistic matrix factorisation trained by per-example SGD on an MI355X.

    Args (``src/mf.py:20-32``): n_users, n_items, reg (L2 on touched rows),
    alpha (init scale), evaluator (optional ``ValEvaluator``-like object).
    """

    n_users: int
    n_items: int
    reg: float
    alpha: float = 4.0
    evaluator: Optional[object] = None

    # Not a constructor argument.  False (default): the reference's strictly
    # sequential per-example updates, reproduced exactly through the level
    # schedule.  True: HOGWILD-style unordered updates (rfm_mf_sgd_hogwild) -- a
    # throughput mode whose parameters do NOT match the reference to 1e-5.
    hogwild = False
    # Not a constructor argument: a ValEvaluator-like ``evaluator`` (see evaluate.py) is
    # computed on the device; False keeps the host callback for every evaluator.
    device_evaluator = True

    def __post_init__(self) -> None:
        # src/mf.py:34-66 -- the reference's NumPy calls, in its draw order
        np.random.seed(self.seed)
        limit = self.alpha * np.sqrt(6 / self.n_factors)
        P = np.random.uniform(low=-limit, high=limit, size=(self.n_users, self.n_factors))
        Q = np.random.uniform(low=-limit, high=limit, size=(self.n_items, self.n_factors))
        b_u = np.random.normal(scale=0.001, size=self.n_users)
        b_i = np.random.normal(scale=0.001, size=self.n_items)

        self._rt = Runtime.get()
        self.P = DeviceSGD(self._rt, P, self.lr)
        self.Q = DeviceSGD(self._rt, Q, self.lr)
        self.b_u = DeviceSGD(self._rt, b_u, self.lr)
        self.b_i = DeviceSGD(self._rt, b_i, self.lr)

        if self.evaluator is not None:
            self.val_metrics = []
            self.model_name = "MF"

    def _check_ids(self, pairs: DevicePairs) -> None:
        if pairs.n == 0:
            return
        if pairs.h_users.min() < 0 or pairs.h_users.max() >= self.n_users:
            raise IndexError("user id out of range")
        if pairs.h_items.min() < 0 or pairs.h_items.max() >= self.n_items:
            raise IndexError("item id out of range")

    # ------------------------------------------------------------------ fit
    def fit(self, train: dict, val: dict) -> tuple:
        """src/mf.py:68-134."""
        rt = self._rt
        # global bias: mean of the training labels (src/mf.py:84)
        self.b = np.mean(train["labels"])
        n_rows = int(np.asarray(train["features"]).shape[0])
        if self.n_epochs <= 0:
            return [], []
        # resample(..., random_state=epoch) ids (src/mf.py:88-95), sampled chunk by chunk on
        # the host while the GPU works on the chunk before
        id_stream = BatchIdStream(rt, n_rows, self.batch_size, self.n_epochs)
        try:
            return self._fit(train, val, id_stream)
        finally:
            id_stream.close()  # (its sampler thread runs from the constructor on)

    def _fit(self, train: dict, val: dict, id_stream: BatchIdStream) -> tuple:
        rt = self._rt
        self._keep_ids = []

        tr = DevicePairs(rt, train["features"])
        va = DevicePairs(rt, val["features"])
        self._check_ids(tr)
        self._check_ids(va)
        y = rt.upload(np.asarray(train["labels"]), dtype=np.float64)
        p = rt.upload(np.asarray(train["pscores"]), dtype=np.float64)
        vy = rt.upload(np.asarray(val["labels"]), dtype=np.float64)
        vp = rt.upload(np.asarray(val["pscores"]), dtype=np.float64)
        tl = rt.empty((self.n_epochs,), y.dtype)
        vl = rt.empty((self.n_epochs,), y.dtype)
        if va.n == 0:  # the reference's mean over no rows is nan (src/base.py:61)
            vl.fill_(float("nan"))
        params = (self.P.dev.data_ptr(), self.Q.dev.data_ptr(), self.b_u.dev.data_ptr(),
                  self.b_i.dev.data_ptr())
        b = float(self.b)
        h_y = np.ascontiguousarray(train["labels"], dtype=np.float64)
        h_p = np.ascontiguousarray(train["pscores"], dtype=np.float64)
        # item rows the sequential kernel may keep in LDS (rows + bias, 32 KiB)
        cache_cap = mf_cache_capacity(self.n_factors)

        ev_frame = ev_pairs = ev_loop = None
        if self.evaluator is not None:
            ev_X = self.evaluator.features[self.model_name]
            if self.device_evaluator:
                ev_frame = device_frame(rt, self.evaluator, self.estimator, int(np.asarray(ev_X).shape[0]))
            if ev_frame is not None:
                # ValEvaluator's IPS-DCG@k from scores that stay in HBM (rfm_val_dcg)
                ev_pairs = DevicePairs(rt, ev_X)
                self._check_ids(ev_pairs)
                ev_loop = EvalLoop(rt, ev_frame, self.evaluator, self.estimator, self.n_epochs)

        def after_sgd(epoch: int, ids_ptr: int) -> None:
            # train loss on the same batch with the updated parameters (src/mf.py:110-116)
            _lib.check(rt.lib.rfm_mf_predict_loss(
                rt.ctx, tr.users.data_ptr(), tr.items.data_ptr(), y.data_ptr(), p.data_ptr(),
                ids_ptr, self.batch_size, *params, b, self.n_factors, LOSS_EPS, None,
                tl.data_ptr() + epoch * 8))
            if va.n > 0:
                _lib.check(rt.lib.rfm_mf_predict_loss(
                    rt.ctx, va.users.data_ptr(), va.items.data_ptr(), vy.data_ptr(), vp.data_ptr(),
                    None, va.n, *params, b, self.n_factors, LOSS_EPS, None, vl.data_ptr() + epoch * 8))
            if ev_frame is not None:
                _lib.check(rt.lib.rfm_mf_predict(
                    rt.ctx, ev_pairs.users.data_ptr(), ev_pairs.items.data_ptr(), None, ev_pairs.n,
                    *params, b, self.n_factors, ev_loop.slot(epoch).data_ptr()))
                ev_loop.done(epoch)
            elif self.evaluator is not None:
                y_scores = self.predict(self.evaluator.features[self.model_name])
                self.val_metrics.append(
                    self.evaluator.evaluate(y_scores=y_scores, estimator=self.estimator))

        if self.hogwild:
            for epoch, rows, ids_ptr in self._epochs(id_stream):
                _lib.check(rt.lib.rfm_mf_sgd_hogwild(
                    rt.ctx, tr.users.data_ptr(), tr.items.data_ptr(), y.data_ptr(), p.data_ptr(),
                    ids_ptr, self.batch_size, *params, b, self.n_factors, float(self.lr),
                    float(self.reg)))
                after_sgd(epoch, ids_ptr)
        else:
            # exact mode: the level schedules of the coming batches are derived and staged by
            # a host thread while the GPU runs the batch before (the GPU never waits for them)
            import threading

            pipe = _SchedulePipe(rt, self.batch_size, cache_cap)

            def schedule(rows):
                return mf_schedule_ex(tr.h_users[rows], tr.h_items[rows], h_y[rows], h_p[rows],
                                      self.n_users, self.n_items, cache_cap)

            worker = threading.Thread(target=pipe.produce, args=(self._epochs(id_stream), schedule),
                                      name="rfm-mf-schedule", daemon=True)
            worker.start()
            try:
                while True:
                    got = pipe.take()
                    if got is None:
                        break
                    epoch, ids_ptr, d_ex, h_lptr, d_lptr, n_levels, d_cache, n_cached, done = got
                    _lib.check(rt.lib.rfm_mf_sgd_levels_ex(
                        rt.ctx, d_ex, h_lptr, d_lptr, n_levels, d_cache, n_cached, *params, b,
                        self.n_factors, float(self.lr), float(self.reg)))
                    done()
                    after_sgd(epoch, ids_ptr)
            finally:
                pipe.stop()
                rt.sync()
                worker.join(timeout=5.0)
        rt.sync()
        self._keep_ids = []
        if ev_frame is not None:
            self.val_metrics.extend(ev_loop.finish(self.n_epochs))
            self.evaluator_host_calls = ev_loop.host_calls
            self.evaluator_host_users = ev_loop.host_users
            ev_loop.leave_scores(self.n_epochs - 1)
        return tl.cpu().numpy().tolist(), vl.cpu().numpy().tolist()

    def _epochs(self, id_stream: BatchIdStream):
        """``(epoch, host row ids, device address of the ids)`` of every iteration."""
        for first, host_ids, dev_ids in id_stream.chunks():
            self._keep_ids.append(dev_ids)  # launches enqueued later still read these ids
            for j in range(host_ids.shape[0]):
                yield first + j, host_ids[j], dev_ids.data_ptr() + j * self.batch_size * 4

    # -------------------------------------------------------------- predict
    def predict(self, X) -> np.ndarray:
        """src/mf.py:136-152 -- scores of (user, item) pairs."""
        rt = self._rt
        if not hasattr(self, "b"):
            # the reference's global bias exists only after fit() (src/mf.py:84)
            raise AttributeError("'LogisticMatrixFactorization' object has no attribute 'b'")
        pairs = DevicePairs(rt, X)
        self._check_ids(pairs)
        out = rt.empty((pairs.n,), self.P.dev.dtype)
        _lib.check(rt.lib.rfm_mf_predict(
            rt.ctx, pairs.users.data_ptr(), pairs.items.data_ptr(), None, pairs.n,
            self.P.dev.data_ptr(), self.Q.dev.data_ptr(), self.b_u.dev.data_ptr(),
            self.b_i.dev.data_ptr(), float(self.b), self.n_factors, out.data_ptr()))
        rt.sync()
        return out.cpu().numpy()
