"""The per-iteration validation metric of ``fit(..., evaluator=)`` on the device
(SURVEY.md 8f N1).

The reference calls ``evaluator.evaluate(y_scores=predict(...), estimator=...)``
after every iteration (``src/fm.py:104-110``, ``src/mf.py:126-132``); for its
``ValEvaluator`` that is a pandas ``groupby("user")`` plus one ``argsort`` per
user on the host (``utils/evaluate.py:183-239``).  When the object handed in is
recognisably such an evaluator -- a frame ``interaction_df`` with the columns
``user, label, pscore, ones_pscore``, a ranking depth ``k`` and
``metric_name == "DCG"`` -- the same IPS-DCG@k is computed by ``rfm_val_dcg``
from the scores that are already in HBM, and only the list of metric values
comes back when ``fit`` ends.  Any other object keeps the host callback.

Ranking ties: the device ranks equal scores later-row-first
(``argsort(kind="stable")[::-1]``).  NumPy's default sort, which the reference
calls, is not stable and its tie order changes from CPU to CPU, and saturated
sigmoid scores (exactly 0.0 / 1.0) make ties common.  ``rfm_val_dcg`` therefore
flags the users whose value depends on the tie order; for exactly those users the
value is recomputed on the host the way the reference does it -- the same
``ndarray.argsort()[::-1]`` on the same per-user score array, the same formula
(``utils/metrics.py:53-80``) -- so ``val_metrics`` is what the host callback would
have produced on this machine, while the unambiguous users (the bulk) never leave
the device (``EvalLoop``).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _lib
from .runtime import Runtime

_COLUMNS = ("user", "label", "pscore", "ones_pscore")


def group_by_user(users: np.ndarray):
    """``(order, seg_ptr)``: frame rows grouped by ascending user, rows of a
    group in frame order (``groupby("user").agg(list)``, evaluate.py:225-239)."""
    users = np.asarray(users)
    n = users.shape[0]
    order = np.argsort(users, kind="stable").astype(np.int32)
    su = users[order]
    cuts = np.flatnonzero(su[1:] != su[:-1]) + 1
    seg_ptr = (np.concatenate(([0], cuts, [n])) if n else np.zeros(1)).astype(np.int32)
    return order, seg_ptr


class ValFrame:
    """A validation frame grouped by user, on the host: what the order-dependent users are
    redone from (and all that is needed to restate ``ValEvaluator.evaluate`` without a GPU)."""

    def __init__(self, users, labels, pscores: Optional[np.ndarray], k: int):
        users = np.asarray(users)
        if users.ndim != 1:
            raise ValueError("users must be a 1-D array")
        n = users.shape[0]
        if k < 1:
            raise ValueError("k (ranking positions) must be >= 1")
        self.k, self.n_rows = int(k), int(n)
        self.h_order, self.h_seg_ptr = group_by_user(users)
        self.n_segments = int(self.h_seg_ptr.shape[0] - 1)
        labels = np.asarray(labels, dtype=np.float64)
        if labels.shape != (n,):
            raise ValueError("labels must match the frame's rows")
        self.h_labels = labels[self.h_order] if n else np.zeros(0)
        self.h_pscores = None
        if pscores is not None:
            pscores = np.asarray(pscores, dtype=np.float64)
            if pscores.shape != (n,):
                raise ValueError("pscores must match the frame's rows")
            self.h_pscores = pscores[self.h_order] if n else np.zeros(0)

    def host_user_value(self, scores: np.ndarray, g: int) -> float:
        """IPS-DCG@k of user group ``g`` exactly as the reference computes it
        (utils/evaluate.py:194-204 with utils/metrics.py:53-80), including whatever order
        NumPy's default sort leaves equal scores in."""
        lo, hi = int(self.h_seg_ptr[g]), int(self.h_seg_ptr[g + 1])
        ranked = scores[self.h_order[lo:hi]].argsort()[::-1]
        y = self.h_labels[lo:hi][ranked]
        p = np.ones(hi - lo) if self.h_pscores is None else self.h_pscores[lo:hi][ranked]
        value = 0.0
        value += y[0] / p[0]
        tail = y[1:self.k]
        positions = np.arange(1, tail.shape[0] + 1)
        value += np.sum(tail / (p[1:self.k] * np.log2(positions + 1)))
        return float(value)

    def host_user_values(self, scores: np.ndarray, groups: np.ndarray) -> np.ndarray:
        """``host_user_value`` of many groups at once: groups of equal length are stacked and
        ranked by one ``argsort(axis=1)`` -- NumPy runs the same 1-D sort on every row, so each
        row gets the order (ties included) its own ``argsort()`` call would give."""
        out = np.empty(groups.shape[0], dtype=np.float64)
        lo = self.h_seg_ptr[groups].astype(np.int64)
        lens = self.h_seg_ptr[groups + 1].astype(np.int64) - lo
        for length in np.unique(lens):
            sel = np.flatnonzero(lens == length)
            idx = lo[sel][:, None] + np.arange(length)[None, :]
            ranked = np.argsort(scores[self.h_order[idx]], axis=1)[:, ::-1]
            y = np.take_along_axis(self.h_labels[idx], ranked, axis=1)
            p = (np.ones_like(y) if self.h_pscores is None
                 else np.take_along_axis(self.h_pscores[idx], ranked, axis=1))
            value = 0.0 + y[:, 0] / p[:, 0]
            tail = y[:, 1:self.k]
            positions = np.arange(1, tail.shape[1] + 1)
            value = value + np.sum(tail / (p[:, 1:self.k] * np.log2(positions + 1)[None, :]), axis=1)
            out[sel] = value
        return out

    def resolve(self, scores: np.ndarray, user_scratch: np.ndarray) -> float:
        """The metric from the device's per-user results (``user_scratch`` = the
        ``3 * n_segments`` doubles of one ``rfm_val_dcg``) with the order-dependent users
        recomputed the reference's way on the host."""
        n = self.n_segments
        vals = user_scratch[:n].copy()
        counted = user_scratch[n: 2 * n] != 0.0
        redo = np.flatnonzero(user_scratch[2 * n: 3 * n] != 0.0)
        if redo.size:
            vals[redo] = self.host_user_values(scores, redo)
        return float(np.mean(vals[counted]))


class DeviceValFrame(ValFrame):
    """The grouped frame resident in HBM."""

    def __init__(self, rt: Runtime, users, labels, pscores: Optional[np.ndarray], k: int):
        super().__init__(users, labels, pscores, k)
        self.rt = rt
        n = self.n_rows
        self.rows = rt.upload(self.h_order if n else np.zeros(1, np.int32))
        self.seg_ptr = rt.upload(self.h_seg_ptr)
        self.labels = rt.upload(self.h_labels if n else np.zeros(1))
        self.pscores = None
        if self.h_pscores is not None:
            self.pscores = rt.upload(self.h_pscores if n else np.zeros(1))
        self.scratch = rt.empty((max(3 * self.n_segments, 1),), self.labels.dtype)

    def dcg_into(self, d_scores, out_ptr: int, scratch_ptr: Optional[int] = None) -> None:
        """Enqueue the metric of ``d_scores`` (device, frame order): the value and the
        number of users it is tie-order dependent for land at ``out_ptr`` (2 doubles), the
        per-user values / counted / order-dependent flags in ``3 * n_segments`` doubles at
        ``scratch_ptr`` (default: this frame's own scratch)."""
        rt = self.rt
        _lib.check(rt.lib.rfm_val_dcg(
            rt.ctx, d_scores.data_ptr(), self.seg_ptr.data_ptr(), self.rows.data_ptr(),
            self.labels.data_ptr(), None if self.pscores is None else self.pscores.data_ptr(),
            self.n_segments, self.k, self.scratch.data_ptr() if scratch_ptr is None else scratch_ptr,
            out_ptr))

    def dcg_checked(self, scores):
        """``(value, n_order_dependent_users)`` of host or device scores (frame order)."""
        rt = self.rt
        d = scores if hasattr(scores, "data_ptr") else rt.upload(np.asarray(scores, dtype=np.float64))
        if d.shape[0] != self.n_rows:
            raise ValueError(f"{d.shape[0]} scores for a frame of {self.n_rows} rows")
        out = rt.empty((2,), self.labels.dtype)
        self.dcg_into(d, out.data_ptr())
        rt.sync()
        o = out.cpu().numpy()
        return float(o[0]), int(o[1])

    def dcg(self, scores) -> float:
        """The metric under the device's tie rule, synchronously."""
        return self.dcg_checked(scores)[0]

    def per_user(self):
        """``(values, counted, order_dependent)`` of the last call, one entry per user group."""
        self.rt.sync()
        s = self.scratch.cpu().numpy()
        n = self.n_segments
        return s[:n].copy(), s[n: 2 * n] != 0.0, s[2 * n: 3 * n] != 0.0


class EvalLoop:
    """The evaluator hook of a ``fit()`` loop, computed on the device.

    Per iteration the caller writes the evaluator's scores into ``slot(epoch)`` and
    calls ``done(epoch)``; nothing returns to the host until a chunk of iterations is
    complete.  For an iteration with tie-order dependent users, the scores and the
    per-user results come back and those users are redone the reference's way
    (``DeviceValFrame.resolve``) -- the list ``finish()`` returns is what calling the
    evaluator every iteration would have produced on this machine."""

    CHUNK_BYTES = 1 << 30

    def __init__(self, rt: Runtime, frame: DeviceValFrame, evaluator, estimator: str, n_epochs: int):
        self.rt, self.frame, self.evaluator, self.estimator = rt, frame, evaluator, estimator
        self.n_epochs = n_epochs
        per_iter = 8 * (frame.n_rows + 3 * frame.n_segments) + 8
        self.chunk = int(max(1, min(max(n_epochs, 1), self.CHUNK_BYTES // per_iter)))
        self.scores = rt.empty((self.chunk, max(frame.n_rows, 1)), frame.labels.dtype)
        self.users = rt.empty((self.chunk, max(3 * frame.n_segments, 1)), frame.labels.dtype)
        self.out = rt.empty((max(n_epochs, 1), 2), frame.labels.dtype)
        self.values: list = []
        self.host_calls = 0   # iterations that needed host work
        self.host_users = 0   # users redone on the host, over all iterations
        self._flushed = 0

    def slot(self, epoch: int):
        """Device tensor the scores of iteration ``epoch`` go to."""
        return self.scores[epoch % self.chunk]

    def done(self, epoch: int) -> None:
        self.frame.dcg_into(self.slot(epoch), self.out.data_ptr() + epoch * 16,
                            self.users[epoch % self.chunk].data_ptr())
        if (epoch + 1) % self.chunk == 0:
            self._flush(epoch + 1)

    def room(self, epoch: int) -> int:
        """Iterations from ``epoch`` on that fit the current chunk of score slots."""
        return self.chunk - epoch % self.chunk

    def ran(self, first: int, count: int) -> None:
        """Iterations ``first .. first + count`` (inside one chunk: ``count <= room(first)``) were
        scored and measured by the library itself (``rfm_fm_train_eval``)."""
        if (first + count) % self.chunk == 0:
            self._flush(first + count)

    def _flush(self, upto: int) -> None:
        if upto <= self._flushed:
            return
        self.rt.sync()
        o = self.out[self._flushed:upto].cpu().numpy()
        for i, epoch in enumerate(range(self._flushed, upto)):
            if o[i, 1] != 0.0:
                y_scores = self.slot(epoch)[: self.frame.n_rows].cpu().numpy()
                per_user = self.users[epoch % self.chunk].cpu().numpy()
                self.values.append(self.frame.resolve(y_scores, per_user))
                self.host_calls += 1
                self.host_users += int(o[i, 1])
            else:
                self.values.append(float(o[i, 0]))
        self._flushed = upto

    def finish(self, n_done: int) -> list:
        self._flush(n_done)
        return self.values

    def leave_scores(self, epoch: int) -> None:
        """The reference's ``evaluate`` stores the scores it was given in its frame
        (``interaction_df["y_score"] = y_scores``, utils/evaluate.py:224): leave the last
        iteration's there, as the last host callback would have."""
        if epoch < 0:
            return
        try:
            self.rt.sync()
            self.evaluator.interaction_df["y_score"] = self.slot(epoch)[: self.frame.n_rows].cpu().numpy()
        except Exception:  # noqa: BLE001 -- a read-only or exotic frame: nothing to leave
            pass


def known_implementation(evaluator) -> bool:
    """True when ``evaluator.evaluate`` is known to be the reference's IPS-DCG@k: the class
    that DEFINES ``evaluate`` is ``ValEvaluator`` of a module called ``evaluate``
    (``utils/evaluate.py:160-207``; a subclass that overrides ``evaluate()`` is not), or the
    object opts in with ``rfm_device_evaluator = True``."""
    if getattr(evaluator, "rfm_device_evaluator", False) is True:
        return True
    for cls in type(evaluator).__mro__:
        if "evaluate" in vars(cls):
            return cls.__qualname__ == "ValEvaluator" and cls.__module__.split(".")[-1] == "evaluate"
    return False


def recognise(evaluator, estimator: str, any_implementation: bool = False):
    """``(users, labels, pscores, k)`` of the reference's ValEvaluator (or an object that
    opts in, see ``known_implementation``), or ``None`` when it is something else (then the
    caller keeps the host callback).  ``any_implementation`` skips the check of whose
    ``evaluate()`` it is and looks at the attributes only."""
    if not any_implementation and not known_implementation(evaluator):
        return None
    frame = getattr(evaluator, "interaction_df", None)
    k = getattr(evaluator, "k", None)
    if frame is None or not isinstance(k, (int, np.integer)) or k < 1:
        return None
    if getattr(evaluator, "metric_name", None) != "DCG":
        return None
    try:
        cols = {c: np.asarray(frame[c]) for c in _COLUMNS}
    except (KeyError, TypeError, IndexError, ValueError):
        return None
    n = cols["user"].shape[0]
    if any(v.ndim != 1 or v.shape[0] != n for v in cols.values()):
        return None
    # evaluate.py:222: pscore for IPS, ones_pscore for every other estimator
    p = cols["pscore"] if estimator == "IPS" else cols["ones_pscore"]
    return cols["user"], cols["label"], p, int(k)


def device_frame(rt: Runtime, evaluator, estimator: str, n_scores: int) -> Optional[DeviceValFrame]:
    """The device form of ``evaluator`` if it is recognised and matches the
    ``n_scores`` rows its features produce; else ``None``."""
    got = recognise(evaluator, estimator)
    if got is None or got[0].shape[0] != n_scores:
        return None
    users, labels, pscores, k = got
    return DeviceValFrame(rt, users, labels, pscores, k)


# ---------------------------------------------------------------------------
# test-set metrics: TestEvaluator on device-ranked rows
# ---------------------------------------------------------------------------
TEST_METRICS = ("Recall", "MAP", "DCG", "ME", "CatalogCoverage", "Gini")  # utils/metrics.py:169-178


class TestFrame:
    """A test frame grouped by user and the metrics of ``TestEvaluator.evaluate``
    (``utils/evaluate.py:80-127``) from the positions of every user's best ``max(K)`` rows.
    Host only: the ranking is what the device does (``DeviceTestEvaluator``); the metrics
    themselves are sums over ``n_users x max(K)`` values."""

    __test__ = False  # not a pytest class

    def __init__(self, users, items, labels, pscores, K, used_metrics, n_items: int):
        self.K = tuple(int(k) for k in K)
        if not self.K or min(self.K) < 1:
            raise ValueError("K (ranking positions) must be positive")
        self.kmax = max(self.K)
        # the reference always reports ME, then the metrics it was asked for (evaluate.py:66-78)
        self.names = ["ME"]
        for name in used_metrics:
            if name not in TEST_METRICS:
                raise ValueError(f"metric_name must be in {TEST_METRICS}. metric_name: '{name}'")
            if name not in self.names:
                self.names.append(name)
        self.n_items = int(n_items)
        users = np.asarray(users)
        n = users.shape[0]
        self.n_rows = int(n)
        self.h_order, self.h_seg_ptr = group_by_user(users)
        self.n_segments = int(self.h_seg_ptr.shape[0] - 1)
        cols = []
        for arr, dt in ((labels, np.float64), (pscores, np.float64), (items, np.int64)):
            arr = np.asarray(arr, dtype=dt)
            if arr.shape != (n,):
                raise ValueError("frame columns must match its rows")
            cols.append(arr[self.h_order] if n else arr)
        self.h_labels, self.h_pscores, self.h_items = cols
        self.h_ysum = (np.add.reduceat(self.h_labels, self.h_seg_ptr[:-1].astype(np.int64))
                       if n else np.zeros(0))

    def host_topk(self, scores: np.ndarray, g: int) -> np.ndarray:
        """Positions of user ``g``'s best rows exactly as the reference ranks them
        (``argsort()[::-1]``, whatever order NumPy leaves equal scores in)."""
        lo, hi = int(self.h_seg_ptr[g]), int(self.h_seg_ptr[g + 1])
        ranked = scores[self.h_order[lo:hi]].argsort()[::-1][: self.kmax]
        out = np.full(self.kmax, -1, dtype=np.int64)
        out[: ranked.shape[0]] = lo + ranked
        return out

    def metrics(self, pos: np.ndarray, flags: np.ndarray, scores: np.ndarray) -> dict:
        """``{metric: [value per K]}`` from the device's ``pos [n_users][kmax]`` / ``flags``
        (``rfm_topk_users``); users flagged order-dependent are ranked again on the host."""
        pos = np.asarray(pos, dtype=np.int64).reshape(self.n_segments, self.kmax).copy()
        flags = np.asarray(flags)
        for g in np.flatnonzero(flags & 2):
            pos[g] = self.host_topk(scores, int(g))
        counted = np.flatnonzero(flags & 1)
        P = pos[counted]
        valid = P >= 0
        safe = np.where(valid, P, 0)
        Y = np.where(valid, self.h_labels[safe] if self.n_rows else 0.0, 0.0)
        PS = np.where(valid, self.h_pscores[safe] if self.n_rows else 0.0, np.nan)
        IT = np.where(valid, self.h_items[safe] if self.n_rows else 0, -1)
        ysum = self.h_ysum[counted] if self.n_rows else np.zeros(0)
        out = {}
        with np.errstate(invalid="ignore", divide="ignore"):
            for name in self.names:
                vals = []
                for k in self.K:
                    if name == "ME":  # utils/metrics.py:110-126
                        vals.append(_nanmean(PS[:, k - 1]))
                    elif name == "DCG":  # utils/metrics.py:83-107
                        disc = np.log2(np.arange(1, k) + 1)
                        vals.append(_nanmean(0.0 + Y[:, 0] + np.sum(Y[:, 1:k] / disc[None, :], axis=1)))
                    elif name == "Recall":  # utils/metrics.py:32-50
                        vals.append(_nanmean(np.sum(Y[:, :k], axis=1) / ysum))
                    elif name == "MAP":  # utils/metrics.py:9-29
                        hits = (Y[:, :k] >= 1) & valid[:, :k]
                        prec = np.cumsum(Y[:, :k], axis=1) / np.arange(1, k + 1)[None, :]
                        vals.append(_nanmean(np.sum(np.where(hits, prec, 0.0), axis=1)))
                    elif name == "CatalogCoverage":  # utils/metrics.py:151-166
                        rec = IT[:, :k][valid[:, :k]]
                        vals.append(len(np.unique(rec)) / self.n_items)
                    else:  # Gini, utils/metrics.py:129-148
                        rec = IT[:, :k][valid[:, :k]]
                        rec = rec[(rec >= 0) & (rec < self.n_items)]
                        freqs = np.sort(np.bincount(rec, minlength=self.n_items), kind="merge")
                        idx = np.arange(1, self.n_items + 1)
                        vals.append(float(np.sum((2 * idx - self.n_items - 1) * freqs)
                                          / (self.n_items * np.sum(freqs))))
                out[name] = vals
        return out


def _nanmean(a: np.ndarray) -> float:
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)  # mean of no users is nan
        return float(np.nanmean(a)) if a.size else float("nan")


class DeviceTestEvaluator:
    """``TestEvaluator`` (``utils/evaluate.py:42-156``) with the per-user ranking on the
    device: same constructor fields (``interaction_df`` with the columns user / item / label /
    pscore, ``features``, ``K``, ``used_metrics``, ``n_items``), same ``evaluate(y_scores)``
    result -- ``{metric: [value per K]}`` with ``ME`` always present -- and the same side
    effect (``interaction_df["y_score"]``)."""

    def __init__(self, interaction_df, features, K, used_metrics, n_items: int, rt: Optional[Runtime] = None):
        from collections import defaultdict

        self.interaction_df, self.features = interaction_df, features
        self.K, self.used_metrics, self.n_items = K, used_metrics, n_items
        self._defaultdict = defaultdict
        self.rt = rt or Runtime.get()
        df = interaction_df
        self.frame = TestFrame(df["user"], df["item"], df["label"], df["pscore"], K, used_metrics, n_items)
        fr, up = self.frame, self.rt.upload
        n = fr.n_rows
        self._rows = up(fr.h_order if n else np.zeros(1, np.int32))
        self._seg = up(fr.h_seg_ptr)
        self._labels = up(fr.h_labels if n else np.zeros(1))
        self._pscores = up(fr.h_pscores if n else np.zeros(1))
        self._items = up((fr.h_items if n else np.zeros(1)).astype(np.int32))
        self.host_users = 0  # users ranked again on the host in the last evaluate()

    def topk(self, y_scores):
        """``(pos [n_users][max K], flags [n_users])`` of host or device scores (frame order)."""
        torch = __import__("torch")
        rt, fr = self.rt, self.frame
        d = y_scores if hasattr(y_scores, "data_ptr") else rt.upload(np.asarray(y_scores, dtype=np.float64))
        if d.shape[0] != fr.n_rows:
            raise ValueError(f"{d.shape[0]} scores for a frame of {fr.n_rows} rows")
        pos = rt.empty((max(fr.n_segments, 1), fr.kmax), torch.int32)
        flags = rt.empty((max(fr.n_segments, 1),), torch.int32)
        _lib.check(rt.lib.rfm_topk_users(
            rt.ctx, d.data_ptr(), self._seg.data_ptr(), self._rows.data_ptr(), self._labels.data_ptr(),
            self._pscores.data_ptr(), self._items.data_ptr(), fr.n_segments, fr.kmax, pos.data_ptr(),
            flags.data_ptr()))
        rt.sync()
        return pos.cpu().numpy()[: fr.n_segments], flags.cpu().numpy()[: fr.n_segments]

    def evaluate(self, y_scores):
        host_scores = y_scores.cpu().numpy() if hasattr(y_scores, "data_ptr") else np.asarray(y_scores, dtype=np.float64)
        pos, flags = self.topk(y_scores)
        self.host_users = int(np.count_nonzero(flags & 2))
        try:
            self.interaction_df["y_score"] = host_scores  # utils/evaluate.py:141
        except Exception:  # noqa: BLE001 -- a read-only frame
            pass
        results = self._defaultdict(list)
        for name, vals in self.frame.metrics(pos, flags, host_scores).items():
            results[name] = list(vals)
        return results
