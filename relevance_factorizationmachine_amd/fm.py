"""``FactorizationMachines`` with the reference's surface (``src/fm.py:17-133``)
on top of the HIP kernels of librfm_hip.so.

Same dataclass fields, same ``fit(train, val) -> (train_loss, val_loss)`` and
``predict(X)`` / ``predict(X=...)``, same side attributes (``val_metrics``,
``model_name`` when an evaluator is given; ``w0``, ``w``, ``V`` callables).  The
host only prepares inputs (parameter init with the reference's NumPy calls,
mini-batch ids, uploads) and enqueues; every floating-point operation of the
loop runs in the kernels of ``csrc/rfm_fm.hip``.
"""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from .base import LOSS_EPS, PointwiseBaseRecommender
from .evaluate import EvalLoop, device_frame
from .optimizer import DeviceSGD
from .runtime import BatchIdStream, CsrCache, DeviceCSR, Runtime


class _PlanCache:
    """The last few training plans of a device, by the identity of the device copies they were
    built from (log, labels, propensities) and their shape parameters: a second model fitted on
    the same split with the same factor count and batch size takes the plan as it is (a plan
    holds no model state).  A plan that is lent out is not lent twice."""

    CAPACITY = 2

    def __init__(self):
        self._items = []  # [key, (weakrefs), plan, busy]

    def take(self, rt, csr, y, p, n_factors: int, max_batch: int, hot_min_count: int) -> "FmPlan":
        key = (id(csr), id(y), id(p), n_factors, max_batch, hot_min_count)
        for item in self._items:
            if item[0] == key and not item[3] and all(r() is o for r, o in zip(item[1], (csr, y, p))):
                item[3] = True
                return item[2]
        plan = FmPlan(rt, csr, y, p, n_factors, max_batch, hot_min_count)
        try:
            refs = tuple(weakref.ref(o) for o in (csr, y, p))
        except TypeError:
            return plan  # not remembered: give_back() closes it
        self._items.append([key, refs, plan, True])
        while len(self._items) > self.CAPACITY:
            for i, item in enumerate(self._items):
                if not item[3]:
                    self._items.pop(i)[2].close()
                    break
            else:
                break
        return plan

    def give_back(self, plan: "FmPlan") -> None:
        for item in self._items:
            if item[2] is plan:
                item[3] = False
                return
        plan.close()

    def clear(self) -> None:
        for item in [i for i in self._items if not i[3]]:
            self._items.remove(item)
            item[2].close()


def plan_cache(rt: Runtime) -> _PlanCache:
    if getattr(rt, "_plan_cache", None) is None:
        rt._plan_cache = _PlanCache()
    return rt._plan_cache


class FmPlan:
    """Owner of an ``rfm_fm_plan`` (column-major view of a training CSR)."""

    def __init__(self, rt: Runtime, csr: DeviceCSR, labels, pscores, n_factors: int, max_batch: int,
                 hot_min_count: int = 0):
        """``labels`` / ``pscores``: host arrays, or float64 device tensors the caller already
        holds.  The plan is built on the device from the device copy of the log."""
        self.rt = rt
        y = labels if hasattr(labels, "data_ptr") else rt.upload(np.asarray(labels), dtype=np.float64)
        p = pscores if hasattr(pscores, "data_ptr") else rt.upload(np.asarray(pscores), dtype=np.float64)
        if y.shape[0] != csr.shape[0] or p.shape[0] != csr.shape[0]:
            raise ValueError("labels / pscores do not match the number of rows")
        handle = C.c_void_p()
        _lib.check(rt.lib.rfm_fm_plan_create_device(
            rt.ctx, csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr(),
            y.data_ptr(), p.data_ptr(), csr.shape[0], csr.shape[1], n_factors, max_batch,
            hot_min_count, C.byref(handle)))
        self.handle = handle

    def info(self) -> dict:
        out = np.zeros(8, dtype=np.int64)
        _lib.check(self.rt.lib.rfm_fm_plan_info(self.handle, out.ctypes.data))
        return dict(zip(("tasks", "split_columns", "hot_columns", "nnz", "device_bytes",
                         "forward_workgroups", "slots", "task_words"), (int(v) for v in out)))

    def layout(self) -> dict:
        out = np.zeros(4, dtype=np.int32)
        _lib.check(self.rt.lib.rfm_fm_plan_layout(self.handle, out.ctypes.data))
        return dict(zip(("row_blocks", "row_block_bytes", "lanes_per_row", "longest_row"),
                        (int(v) for v in out)))

    def sliced(self) -> dict:
        """The sliced loss forwards of ``rfm_fm_train`` (even factor counts above 128)."""
        out = np.zeros(4, dtype=np.int32)
        _lib.check(self.rt.lib.rfm_fm_plan_sliced(self.handle, out.ctypes.data))
        return dict(zip(("slices", "factors_per_slice", "cached_columns", "records_per_row"),
                        (int(v) for v in out)))

    def hot_columns(self) -> np.ndarray:
        out = np.zeros(max(self.info()["hot_columns"], 1), dtype=np.int32)
        _lib.check(self.rt.lib.rfm_fm_plan_hot_columns(self.handle, out.ctypes.data, out.shape[0]))
        return out[: self.info()["hot_columns"]]

    def close(self) -> None:
        if self.handle is not None:
            self.rt.lib.rfm_fm_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class FactorizationMachines(PointwiseBaseRecommender):
    """Factorization Machines trained by mini-batch SGD on an MI355X.

    Args (``src/fm.py:20-29``):
    - n_features (int): number of feature columns
    - alpha (float): scale of the uniform initialisation
    - evaluator: optional object with ``.features["FM"]`` and
      ``.evaluate(y_scores=, estimator=)`` (the reference's ``ValEvaluator``)
    """

    n_features: int
    alpha: float = 2.0
    evaluator: Optional[object] = None

    # Not a constructor argument: columns with at least this many expected
    # entries per batch are summed on chip (0 = library default; -1 = never, which makes
    # every sum's order fixed and a fit bitwise reproducible; -2 = the default columns, summed
    # on chip in a fixed order: rfm_hip.h).
    hot_min_count = 0
    # Not a constructor argument: True = every sum in a fixed order, i.e. hot_min_count = -2
    # (a fit is then bitwise reproducible, at 1.1-1.25x the step time on KuaiRec-shaped logs).
    deterministic = False
    # Not a constructor argument: a ValEvaluator-like ``evaluator`` (see evaluate.py) is
    # computed on the device; False keeps the host callback for every evaluator.
    device_evaluator = True

    def __post_init__(self) -> None:
        # src/fm.py:31-53 -- the reference's NumPy calls, in its draw order
        np.random.seed(self.seed)
        w0 = np.array([0.0])
        limit = self.alpha * np.sqrt(6 / self.n_features)
        w = np.random.uniform(low=-limit, high=limit, size=self.n_features)
        limit = self.alpha * np.sqrt(6 / self.n_factors)
        V = np.random.uniform(low=-limit, high=limit, size=(self.n_features, self.n_factors))

        self._rt = Runtime.get()
        self.w0 = DeviceSGD(self._rt, w0, self.lr)
        self.w = DeviceSGD(self._rt, w, self.lr)
        self.V = DeviceSGD(self._rt, V, self.lr)
        self._csr_cache = CsrCache(self._rt)

        if self.evaluator is not None:
            self.val_metrics = []
            self.model_name = "FM"

    # ------------------------------------------------------------------ fit
    def fit(self, train: dict, val: dict) -> tuple:
        """src/fm.py:55-112.  One "epoch" is one mini-batch step; returns the
        per-iteration train and validation IPS log-loss as two lists."""
        rt = self._rt
        X = train["features"]
        n_rows = X.shape[0]
        if X.shape[1] != self.n_features:
            raise ValueError(f"train features have {X.shape[1]} columns, model has {self.n_features}")
        if self.n_epochs <= 0:
            return [], []
        # batch selection: resample(..., random_state=epoch) (src/fm.py:72-79), sampled on the
        # host chunk by chunk while the GPU trains on the chunk before
        id_stream = BatchIdStream(rt, n_rows, self.batch_size, self.n_epochs, need_host=False)
        try:
            return self._fit(train, val, id_stream)
        finally:
            id_stream.close()  # (its sampler thread runs from the constructor on)

    def _fit(self, train: dict, val: dict, id_stream: BatchIdStream) -> tuple:
        rt = self._rt
        X = train["features"]

        # a log that is already in HBM (features.assemble / load_csr_to_device) is used as it is
        # (device copies of the split are remembered per device: the drivers fit several models
        # on it; an array edited in place is uploaded again -- CsrCache._fingerprint)
        keep = rt.remember_splits
        tr = X if isinstance(X, DeviceCSR) else (rt.log_cache().get(X) if keep else DeviceCSR(rt, X))
        y = rt.upload_cached(train["labels"], np.float64)
        p = rt.upload_cached(train["pscores"], np.float64)
        va = (val["features"] if isinstance(val["features"], DeviceCSR) else
              (rt.log_cache().get(val["features"]) if keep else DeviceCSR(rt, val["features"])))
        vy = rt.upload_cached(val["labels"], np.float64)
        vp = rt.upload_cached(val["pscores"], np.float64)
        hot = -2 if self.deterministic else self.hot_min_count
        plan = (plan_cache(rt).take(rt, tr, y, p, self.n_factors, self.batch_size, hot) if keep
                else FmPlan(rt, tr, y, p, self.n_factors, self.batch_size, hot))
        self.plan_info = dict(plan.info(), **plan.layout(), **plan.sliced())  # (what the last fit trained with)
        if va.shape[0] > 0:  # (the split's device copy does not change during this fit)
            _lib.check(rt.lib.rfm_fm_plan_register_log(
                rt.ctx, plan.handle, 0, va.indptr.data_ptr(), va.indices.data_ptr(), va.values.data_ptr(),
                va.shape[0]))
        tl = rt.empty((self.n_epochs,), y.dtype)
        vl = rt.empty((self.n_epochs,), y.dtype)
        # an empty validation set: the reference's mean over no rows is nan (src/base.py:61)
        has_val = va.shape[0] > 0
        if not has_val:
            vl.fill_(float("nan"))

        chunk = {"first": 0, "ids": None}

        def run(first: int, count: int, loop=None) -> None:
            ids_ptr = chunk["ids"].data_ptr() + (first - chunk["first"]) * self.batch_size * 4
            args = (rt.ctx, plan.handle, tr.indptr.data_ptr(), tr.indices.data_ptr(), tr.values.data_ptr(),
                    y.data_ptr(), p.data_ptr(), ids_ptr, self.batch_size, count,
                    self.w0.dev.data_ptr(), self.w.dev.data_ptr(), self.V.dev.data_ptr(), float(self.lr),
                    va.indptr.data_ptr(), va.indices.data_ptr(), va.values.data_ptr(),
                    vy.data_ptr(), vp.data_ptr(), va.shape[0], LOSS_EPS,
                    tl.data_ptr() + first * 8, vl.data_ptr() + first * 8 if has_val else None)
            if loop is None:
                _lib.check(rt.lib.rfm_fm_train(*args))
                return
            fr = loop.frame
            _lib.check(rt.lib.rfm_fm_train_eval(
                *args, ev.indptr.data_ptr(), ev.indices.data_ptr(), ev.values.data_ptr(), ev.shape[0],
                fr.seg_ptr.data_ptr(), fr.rows.data_ptr(), fr.labels.data_ptr(),
                None if fr.pscores is None else fr.pscores.data_ptr(), fr.n_segments, fr.k,
                loop.scores.data_ptr(), loop.scores.shape[1], loop.users.data_ptr(), loop.users.shape[1],
                first % loop.chunk, loop.out.data_ptr() + first * 16))

        try:
            frame = loop = ev = ev_X = None
            if self.evaluator is not None:
                ev_X = self.evaluator.features[self.model_name]
                frame = (device_frame(rt, self.evaluator, self.estimator, ev_X.shape[0])
                         if self.device_evaluator else None)
                if frame is not None:
                    # ValEvaluator's IPS-DCG@k from the scores in HBM (rfm_val_dcg); only the
                    # users whose value hangs on the order of tied scores are redone on the host
                    if ev_X.shape[1] != self.n_features:
                        raise ValueError(
                            f"X has {ev_X.shape[1]} columns, model has {self.n_features}")
                    ev = self._csr_cache.get(ev_X)
                    _lib.check(rt.lib.rfm_fm_plan_register_log(
                        rt.ctx, plan.handle, 1, ev.indptr.data_ptr(), ev.indices.data_ptr(),
                        ev.values.data_ptr(), ev.shape[0]))
                    loop = EvalLoop(rt, frame, self.evaluator, self.estimator, self.n_epochs)
            for first, host_ids, dev_ids in id_stream.chunks():
                chunk["first"], chunk["ids"] = first, dev_ids
                count = dev_ids.shape[0]
                if self.evaluator is None:
                    run(first, count)
                elif frame is None:
                    # an evaluator of unknown kind is a host callback: one iteration per enqueue
                    for epoch in range(first, first + count):
                        run(epoch, 1)
                        y_scores = self.predict(X=ev_X)
                        self.val_metrics.append(
                            self.evaluator.evaluate(y_scores=y_scores, estimator=self.estimator))
                else:
                    # the recognised evaluator: scores + IPS-DCG@k inside the library's loop
                    # (rfm_fm_train_eval), one enqueue per run of iterations that fits the chunk
                    # of score slots
                    at = first
                    while at < first + count:
                        n = min(first + count - at, loop.room(at))
                        run(at, n, loop)
                        loop.ran(at, n)
                        at += n
            if loop is not None:
                self.val_metrics.extend(loop.finish(self.n_epochs))
                self.evaluator_host_calls = loop.host_calls
                self.evaluator_host_users = loop.host_users
                loop.leave_scores(self.n_epochs - 1)
            rt.sync()
        finally:
            rt.sync()
            # (the registration ends with the fit: the plan may outlive this split's device copy)
            for log_slot in (0, 1):
                rt.lib.rfm_fm_plan_register_log(rt.ctx, plan.handle, log_slot, None, None, None, 0)
            if keep:
                plan_cache(rt).give_back(plan)
            else:
                plan.close()
        return tl.cpu().numpy().tolist(), vl.cpu().numpy().tolist()

    # -------------------------------------------------------------- predict
    def predict(self, X) -> np.ndarray:
        """src/fm.py:114-133 -- scores of the rows of a sparse matrix."""
        rt = self._rt
        if X.shape[1] != self.n_features:
            raise ValueError(f"X has {X.shape[1]} columns, model has {self.n_features}")
        dev = X if isinstance(X, DeviceCSR) else self._csr_cache.get(X)
        n = dev.shape[0]
        out = rt.empty((n,), self.w.dev.dtype)
        _lib.check(rt.lib.rfm_fm_forward(
            rt.ctx, dev.indptr.data_ptr(), dev.indices.data_ptr(), dev.values.data_ptr(), None, n,
            self.w0.dev.data_ptr(), self.w.dev.data_ptr(), self.V.dev.data_ptr(),
            self.n_features, self.n_factors, out.data_ptr()))
        rt.sync()
        return out.cpu().numpy()
