"""Device plumbing: a per-device context of librfm_hip.so, torch tensors as the
device containers, and the host-side helpers (sampler, MF schedule) that need
no GPU.  PyTorch is used for allocation, copies and streams only -- no torch
operator runs on the training path."""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib


def _torch():
    import torch

    return torch


def ptr(t) -> Optional[int]:
    """Device (or host) address of a torch tensor / numpy array, or None."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    return t.data_ptr()


def content_hash(arr: np.ndarray) -> int:
    """64-bit fingerprint of EVERY byte of a host array (``rfm_hash_bytes``: multi-threaded,
    a few milliseconds per 100 MB): what the upload caches compare before they trust a device
    copy, so that any in-place edit between two calls -- a single element included -- is seen
    and the array uploaded again (the reference reads its inputs on every fit)."""
    a = np.ascontiguousarray(arr)
    out = C.c_uint64(0)
    _lib.check(_lib.load().rfm_hash_bytes(a.ctypes.data, a.nbytes, min(os.cpu_count() or 1, 16), C.byref(out)))
    return int(out.value)


class Runtime:
    """One librfm context on one GPU, bound to torch's current stream."""

    # fit() keeps the device copies of the split it was given (features, labels,
    # propensities) for the next fit on the same objects WITH THE SAME CONTENTS (every byte is
    # hashed on each call: content_hash); False = upload every time
    remember_splits = True

    _instances: Dict[int, "Runtime"] = {}

    def __init__(self, device: int):
        torch = _torch()
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.RfmError(
                "no GPU visible: relevance_factorizationmachine_amd runs its training path only "
                "as HIP kernels on an MI355X (there is no CPU fallback)")
        self.device = device
        self.torch_device = torch.device("cuda", device)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream().cuda_stream
        handle = C.c_void_p()
        _lib.check(self.lib.rfm_create(device, C.c_void_p(stream), C.byref(handle)))
        self.ctx = handle
        weakref.finalize(self, self.lib.rfm_destroy, handle)

    @classmethod
    def get(cls, device: Optional[int] = None) -> "Runtime":
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if _torch().cuda.device_count() > 1 else 0
        rt = cls._instances.get(device)
        if rt is None:
            rt = cls(device)
            cls._instances[device] = rt
        return rt

    # ---- containers -------------------------------------------------------
    def upload(self, a: np.ndarray, dtype=None):
        torch = _torch()
        arr = np.ascontiguousarray(a if dtype is None else np.asarray(a, dtype=dtype))
        return torch.from_numpy(arr).to(self.torch_device)

    def empty(self, shape, dtype):
        torch = _torch()
        return torch.empty(shape, dtype=dtype, device=self.torch_device)

    # ---- copies beside the compute stream ----------------------------------
    def copy_stream(self):
        """A second stream for host-to-device copies that must not queue behind (or stall)
        the kernels of the compute stream; consumers wait for the copy's event."""
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = _torch().cuda.Stream(device=self.torch_device)
        return self._copy_stream

    def copy_into_async(self, dst, pinned_src):
        """``dst.copy_(pinned_src)`` on the copy stream; returns the event that marks its end.
        ``dst`` must not be in use by kernels already enqueued (the caller owns it)."""
        torch = _torch()
        with torch.cuda.device(self.device), torch.cuda.stream(self.copy_stream()):
            dst.copy_(pinned_src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return ev

    def upload_cached(self, a, dtype):
        """Device copy of a host vector (labels, propensities), remembered for the next
        ``fit()`` on the same array object with the same contents (hash of the whole array, as
        ``CsrCache``): the drivers fit several models on one split."""
        if not isinstance(a, np.ndarray) or a.size == 0 or not self.remember_splits:
            return self.upload(np.asarray(a), dtype=dtype)
        cache = self.__dict__.setdefault("_vec_cache", [])
        key = (id(a), a.shape, a.dtype.str, np.dtype(dtype).str, content_hash(a))
        for ref, k, dev in cache:
            if k == key and ref() is a:
                return dev
        dev = self.upload(a, dtype=dtype)
        try:
            cache.append((weakref.ref(a), key, dev))
        except TypeError:
            return dev
        if len(cache) > 8:
            cache.pop(0)
        return dev

    def log_cache(self) -> "CsrCache":
        """Device copies of the last few training / validation matrices, shared by the models of
        this device (the drivers fit FM and MF, IPS and Naive, on one split)."""
        if getattr(self, "_log_cache", None) is None:
            self._log_cache = CsrCache(self, capacity=4)
        return self._log_cache

    def clear_caches(self) -> None:
        """Forget every remembered upload (logs, label vectors, sampled ids)."""
        self._log_cache = None
        self.__dict__.pop("_vec_cache", None)
        if getattr(self, "_plan_cache", None) is not None:
            self._plan_cache.clear()
        ID_CACHE.clear()

    def sync(self) -> None:
        _lib.check(self.lib.rfm_sync(self.ctx))


class DeviceCSR:
    """CSR matrix resident in HBM: indptr int64, indices int32, values f64."""

    def __init__(self, rt: Runtime, X):
        from scipy import sparse

        X = X.tocsr() if not sparse.isspmatrix_csr(X) else X
        if not X.has_canonical_format:
            # SciPy's X.power(2) / X.multiply sum duplicate column entries of a row before
            # squaring (src/fm.py:127 works on the de-duplicated matrix): do the same, on a
            # copy -- the caller's matrix is never modified
            X = X.copy()
            X.sum_duplicates()
        self.shape: Tuple[int, int] = X.shape
        self.nnz = int(X.nnz)
        # host copies in the ABI's dtypes (kept for the plan builder)
        self.h_indptr = np.ascontiguousarray(X.indptr, dtype=np.int64)
        self.h_indices = np.ascontiguousarray(X.indices, dtype=np.int32)
        self.h_values = np.ascontiguousarray(X.data, dtype=np.float64)
        self.indptr = rt.upload(self.h_indptr)
        self.indices = rt.upload(self.h_indices if self.nnz else np.zeros(1, np.int32))
        self.values = rt.upload(self.h_values if self.nnz else np.zeros(1, np.float64))


def _device_csr_from_device(cls, rt: Runtime, shape, nnz: int, indptr, indices, values) -> "DeviceCSR":
    """A ``DeviceCSR`` around arrays that are already in HBM (int64 / int32 / float64)."""
    self = cls.__new__(cls)
    self.shape, self.nnz = (int(shape[0]), int(shape[1])), int(nnz)
    self.indptr, self.indices, self.values = indptr, indices, values
    self.h_indptr = self.h_indices = self.h_values = None
    return self


def _device_csr_to_scipy(self):
    """Download as ``scipy.sparse.csr_matrix`` (float64 data, int32 indices)."""
    from scipy.sparse import csr_matrix

    indptr = self.indptr.cpu().numpy()
    return csr_matrix((self.values.cpu().numpy()[: self.nnz], self.indices.cpu().numpy()[: self.nnz],
                       indptr.astype(np.int32) if self.nnz < 2 ** 31 else indptr), shape=self.shape)


DeviceCSR.from_device = classmethod(_device_csr_from_device)
DeviceCSR.to_scipy = _device_csr_to_scipy


class CsrCache:
    """Remembers the device copy of the last few host matrices handed to
    predict() (the evaluator hook passes the same object every iteration)."""

    def __init__(self, rt: Runtime, capacity: int = 4):
        self.rt = rt
        self.capacity = capacity
        self._items = []  # (weakref, key, DeviceCSR)

    @staticmethod
    def _fingerprint(X) -> tuple:
        """Content check: shape, nnz and a hash of ALL bytes of the three CSR arrays, so that a
        matrix edited in place between two calls -- any element -- is uploaded again instead of
        being scored from the stale device copy."""
        nnz = int(X.nnz)
        parts = [X.shape, nnz]
        for arr in (getattr(X, "data", None), getattr(X, "indices", None), getattr(X, "indptr", None)):
            if arr is None or not len(arr):
                continue
            parts.append(content_hash(arr))
        return tuple(parts)

    def get(self, X) -> DeviceCSR:
        key = (id(X),) + self._fingerprint(X)
        for ref, k, dev in self._items:
            if k == key and ref() is X:
                return dev
        dev = DeviceCSR(self.rt, X)
        try:
            self._items.append((weakref.ref(X), key, dev))
        except TypeError:
            return dev
        if len(self._items) > self.capacity:
            self._items.pop(0)
        return dev


# ---- host-only helpers (no GPU needed) -------------------------------------
def sample_batches(n_rows: int, batch_size: int, epoch_begin: int, n_epochs: int,
                   n_threads: int = 0, out: Optional[np.ndarray] = None) -> np.ndarray:
    """Row ids of iterations ``epoch_begin .. +n_epochs`` as ``(n_epochs, B)``
    int32 -- what ``resample(..., random_state=epoch)`` selects
    (src/fm.py:72-79).  Raises ValueError if ``batch_size > n_rows``."""
    lib = _lib.load()
    if out is None:
        out = np.empty((max(n_epochs, 0), batch_size), dtype=np.int32)
    assert out.shape == (max(n_epochs, 0), batch_size) and out.dtype == np.int32 and out.flags.c_contiguous
    if n_threads <= 0:
        n_threads = min(os.cpu_count() or 1, 32)
    _lib.check(lib.rfm_sample_batches(n_rows, batch_size, epoch_begin, n_epochs,
                                      out.ctypes.data, n_threads))
    return out


class _IdCache:
    """Row ids of iteration ``epoch`` of a log of ``n_rows`` rows depend on nothing else
    (``resample(..., random_state=epoch)``), and the drivers fit several models on the same
    log (``search_params`` then the final runs; FM and MF; IPS and Naive): keep what has
    been sampled, the longest batch per (n_rows, epoch), up to ``CAP_BYTES``."""

    CAP_BYTES = 256 << 20

    def __init__(self):
        import threading

        self._lock = threading.Lock()
        self._rows: Dict[Tuple[int, int], np.ndarray] = {}
        self._bytes = 0
        self._dev: Dict[Tuple[int, int, int], object] = {}

    def clear(self) -> None:
        with self._lock:
            self._rows.clear()
            self._bytes = 0
            self._dev.clear()

    # whole fits' ids resident in HBM: (device, n_rows, batch_size) -> int32 tensor [epochs][B]
    DEV_CAP_BYTES = 1 << 30

    def get_device(self, device: int, n_rows: int, batch_size: int, n_epochs: int):
        with self._lock:
            t = self._dev.get((device, n_rows, batch_size))
        return t if t is not None and t.shape[0] >= n_epochs else None

    def put_device(self, device: int, n_rows: int, batch_size: int, ids) -> None:
        with self._lock:
            key = (device, n_rows, batch_size)
            old = self._dev.get(key)
            if old is not None and old.shape[0] >= ids.shape[0]:
                return
            self._dev[key] = ids
            total = lambda: sum(t.numel() * 4 for t in self._dev.values())  # noqa: E731
            while total() > self.DEV_CAP_BYTES and len(self._dev) > 1:
                self._dev.pop(next(k for k in self._dev if k != key))

    def get(self, n_rows: int, batch_size: int, first: int, count: int) -> Optional[np.ndarray]:
        with self._lock:
            rows = [self._rows.get((n_rows, e)) for e in range(first, first + count)]
        if any(r is None or r.shape[0] < batch_size for r in rows):
            return None
        return np.stack([r[:batch_size] for r in rows]) if rows else np.empty((0, batch_size), np.int32)

    def put(self, n_rows: int, first: int, ids: np.ndarray) -> None:
        with self._lock:
            for j in range(ids.shape[0]):
                key = (n_rows, first + j)
                old = self._rows.get(key)
                if old is not None and old.shape[0] >= ids.shape[1]:
                    continue
                if self._bytes + ids[j].nbytes > self.CAP_BYTES:
                    return
                self._bytes += ids[j].nbytes - (old.nbytes if old is not None else 0)
                self._rows[key] = ids[j].copy()


ID_CACHE = _IdCache()


class BatchIdStream:
    """The row-id lists of a whole ``fit()`` in chunks of iterations (the
    Mersenne-Twister shuffle is inherently sequential per iteration, so it stays on the host
    cores -- one iteration per thread -- and is overlapped rather than moved).  The first,
    small chunk is sampled in the constructor (a batch larger than the log raises there,
    before anything is uploaded); a background thread samples the rest, exact ``resample``
    ids, while the caller uploads the log, builds the plan and the GPU trains.

    ``need_host=False`` (FM: nothing on the host reads the ids): the fit's ids live in ONE
    device buffer; every chunk is sampled straight into pinned memory and copied on the
    runtime's copy stream by the sampler thread, and the consumer only makes the compute stream
    wait for the copy's event -- no upload blocks the host or queues behind the kernels.  The
    filled buffer is remembered (``ID_CACHE``): another fit on a log of the same length with
    the same batch size samples and uploads nothing."""

    CHUNK_IDS = 1 << 23   # most ids per chunk: 32 MiB of int32
    FIRST_ITERS = 16      # iterations of the first chunk: the GPU starts after one sampler round
    CHUNK_ITERS = 64      # iterations of the later chunks
    QUEUE_DEPTH = 4       # sampled chunks waiting for the consumer

    def __init__(self, rt: Runtime, n_rows: int, batch_size: int, n_epochs: int, need_host: bool = True):
        import queue
        import threading

        self.rt, self.n_rows, self.batch_size, self.n_epochs = rt, n_rows, batch_size, n_epochs
        most = int(max(1, self.CHUNK_IDS // max(batch_size, 1)))
        cuts, at = [], 0
        while at < n_epochs:
            size = min(self.FIRST_ITERS if at == 0 else self.CHUNK_ITERS, most, n_epochs - at)
            cuts.append((at, size))
            at += size
        self._cuts = cuts
        self._queue: "queue.Queue" = queue.Queue(maxsize=self.QUEUE_DEPTH)
        self._stop = False
        self._thread = None
        self._dev_all = None     # the fit's ids in HBM (need_host=False)
        self._resident = False   # ... found there: nothing to sample
        self._staging = []       # [pinned buffer, event of the copy that last read it]
        self._first = None
        if not cuts:
            return
        if batch_size > n_rows:
            sample_batches(n_rows, batch_size, 0, 1)  # raises the sampler's ValueError
        total = n_epochs * batch_size * 4
        if not need_host and total <= ID_CACHE.DEV_CAP_BYTES:
            got = ID_CACHE.get_device(rt.device, n_rows, batch_size, n_epochs)
            if got is not None:
                self._dev_all, self._resident = got, True
                return
            torch = _torch()
            self._dev_all = rt.empty((n_epochs, batch_size), torch.int32)
            with torch.cuda.device(rt.device):  # (the block may have had a user on the compute stream)
                rt.copy_stream().wait_stream(torch.cuda.current_stream())
            rows = max(size for _, size in cuts)
            self._staging = [[torch.empty((rows, batch_size), dtype=torch.int32, pin_memory=True), None]
                             for _ in range(2)]
            self._first = self._produce(0, *cuts[0])
        else:
            self._first = self._sample(*cuts[0])
        if len(cuts) > 1:
            self._thread = threading.Thread(target=self._work, name="rfm-sampler", daemon=True)
            self._thread.start()

    def _sample(self, first: int, count: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        got = ID_CACHE.get(self.n_rows, self.batch_size, first, count)
        if got is None:
            got = sample_batches(self.n_rows, self.batch_size, first, count, out=out)
            ID_CACHE.put(self.n_rows, first, got)
        elif out is not None:
            out[...] = got
            got = out
        return got

    def _produce(self, index: int, first: int, count: int):
        """Chunk ``index`` into its slice of the device buffer; returns the copy's event."""
        slot = self._staging[index % 2]
        if slot[1] is not None:
            slot[1].synchronize()  # the copy that read this pinned buffer two chunks ago
        pinned = slot[0][:count]
        self._sample(first, count, out=pinned.numpy())
        slot[1] = self.rt.copy_into_async(self._dev_all[first:first + count], pinned)
        return slot[1]

    def _work(self) -> None:
        try:
            for index, (first, count) in enumerate(self._cuts[1:], start=1):
                if self._stop:
                    return
                if self._dev_all is not None:
                    self._queue.put((first, self._produce(index, first, count)))
                else:
                    self._queue.put((first, self._sample(first, count)))
        except BaseException as exc:  # noqa: BLE001 -- handed to the consumer
            self._queue.put((None, exc))

    def chunks(self):
        """Yields ``(first_epoch, host_ids (count, B) or None, device_ids (count, B))`` in
        order; the device ids are valid for work enqueued on the compute stream after the
        yield."""
        if self._resident:
            if self._cuts:
                yield 0, None, self._dev_all[: self.n_epochs]
            return
        torch = _torch()
        complete = False
        try:
            for i, (first, count) in enumerate(self._cuts):
                if i == 0:
                    got = self._first
                else:
                    got_first, got = self._queue.get()
                    if got_first is None:
                        raise got
                    assert got_first == first
                if self._dev_all is not None:
                    with torch.cuda.device(self.rt.device):
                        torch.cuda.current_stream().wait_event(got)  # GPU-side: the host goes on
                    yield first, None, self._dev_all[first:first + count]
                else:
                    yield first, got, self.rt.upload(got)
            complete = True
        finally:
            self.close()
            if complete and self._dev_all is not None:
                ID_CACHE.put_device(self.rt.device, self.n_rows, self.batch_size, self._dev_all)

    def close(self) -> None:
        """Stop the sampler thread (it starts in the constructor, so a ``fit()`` that fails before
        ``chunks()`` is iterated must call this) and wait for the copies out of the pinned
        buffers.  Idempotent."""
        self._stop = True
        thread, self._thread = self._thread, None
        if thread is not None:
            while thread.is_alive():  # unblock a producer waiting on a full queue
                try:
                    self._queue.get_nowait()
                except Exception:  # noqa: BLE001 -- empty
                    thread.join(timeout=0.01)
        if self._dev_all is not None and not self._resident:
            for _, ev in self._staging:
                if ev is not None:
                    ev.synchronize()


def mf_schedule(users: np.ndarray, items: np.ndarray, n_users: int, n_items: int):
    """Level schedule of one batch: ``(order int32[B], level_ptr int32[L+1])``."""
    lib = _lib.load()
    users = np.ascontiguousarray(users, dtype=np.int32)
    items = np.ascontiguousarray(items, dtype=np.int32)
    b = users.shape[0]
    order = np.empty(b, dtype=np.int32)
    level_ptr = np.empty(b + 1, dtype=np.int32)
    n_levels = C.c_int32(0)
    _lib.check(lib.rfm_mf_schedule(users.ctypes.data, items.ctypes.data, b, n_users, n_items,
                                   order.ctypes.data, level_ptr.ctypes.data, C.byref(n_levels)))
    return order, level_ptr[: n_levels.value + 1].copy()


MF_EX_DTYPE = np.dtype([("u", np.int32), ("i", np.int32), ("cslot", np.int32), ("gap", np.int32),
                        ("ry", np.float64)])


def mf_cache_capacity(n_factors: int) -> int:
    """Item rows the sequential MF kernel may keep in LDS for a batch (``rfm_mf_cache_capacity``)."""
    out = C.c_int32(0)
    _lib.check(_lib.load().rfm_mf_cache_capacity(int(n_factors), C.byref(out)))
    return int(out.value)


def mf_schedule_ex(users: np.ndarray, items: np.ndarray, y: np.ndarray, pscore: np.ndarray,
                   n_users: int, n_items: int, cache_cap: int):
    """Level schedule of one batch as level-ordered records:
    ``(ex[B] (MF_EX_DTYPE), level_ptr int32[L+1], cache_items int32[n_cached])``."""
    lib = _lib.load()
    users = np.ascontiguousarray(users, dtype=np.int32)
    items = np.ascontiguousarray(items, dtype=np.int32)
    y = np.ascontiguousarray(y, dtype=np.float64)
    pscore = np.ascontiguousarray(pscore, dtype=np.float64)
    b = users.shape[0]
    ex = np.empty(b, dtype=MF_EX_DTYPE)
    level_ptr = np.empty(b + 1, dtype=np.int32)
    cache_items = np.empty(max(cache_cap, 1), dtype=np.int32)
    n_levels, n_cached = C.c_int32(0), C.c_int32(0)
    _lib.check(lib.rfm_mf_schedule_ex(users.ctypes.data, items.ctypes.data, y.ctypes.data,
                                      pscore.ctypes.data, b, n_users, n_items, cache_cap,
                                      ex.ctypes.data, level_ptr.ctypes.data, C.byref(n_levels),
                                      cache_items.ctypes.data, C.byref(n_cached)))
    return ex, level_ptr[: n_levels.value + 1].copy(), cache_items[: n_cached.value].copy()
