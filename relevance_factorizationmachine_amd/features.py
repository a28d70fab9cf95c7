"""The step before the training path: the FM design matrix and the
row selections the reference's loaders make, assembled on the device, and a binary CSR
cache so that large logs go from disk to HBM without pandas.

What the reference does on the host with pandas / SciPy / scikit-learn, and where:

* per-entity feature tables -- ``pd.get_dummies`` for label columns, ``StandardScaler`` +
  ``fillna(mean)`` for numeric ones, ``MultiLabelBinarizer`` for tag lists
  (``utils/dataloader/kuairec/_feature.py:90-135``).  These are small (one row per user /
  item) and stay host NumPy here: :func:`dummies`, :func:`standardise`, :func:`multi_hot`.
* the design matrix -- one-hot user (+) one-hot item (+) per-interaction columns (+) the
  user's row (+) the item's row, ``hstack``-ed (``_feature.py:54-84,201-207``; Coat's order is
  one-hot user, user row, one-hot item, item row: ``coat/_preparer.py:154-168``), and
  ``features[indices]`` for the train / val / test splits and the negatively sampled subsets
  (``kuairec/_preparer.py:117-136``, ``loader.py:104-115``).  This is the part that scales
  with the log; it is one device operation here, :func:`assemble` (``rfm_csr_assemble_*``):
  every output row is the concatenation of segments, each a one-hot of an id or a row of a
  feature table picked by an id.
* 1:1 negative sampling -- ``np.random.seed(seed); np.random.permutation(negatives)[:n_pos]``
  (``kuairec/_preparer.py:90-115``, ``coat/_preparer.py:120-131``): :func:`negative_sample`,
  bit-exact through the library's MT19937 shuffle.

The CSV readers, the click simulator and Hydra stay with the reference (out of scope).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .runtime import DeviceCSR, Runtime, sample_batches


# ---------------------------------------------------------------------------
# per-entity feature tables (host; one row per user / item)
# ---------------------------------------------------------------------------
def dummies(values) -> np.ndarray:
    """``pd.get_dummies(column, dtype=int)``: one 0/1 column per distinct value, columns in
    ascending order of the values (``_feature.py:113-114``)."""
    values = np.asarray(values)
    cats, codes = np.unique(values, return_inverse=True)
    out = np.zeros((values.shape[0], cats.shape[0]), dtype=np.float64)
    out[np.arange(values.shape[0]), codes] = 1.0
    return out


def standardise(columns) -> np.ndarray:
    """``StandardScaler().fit_transform`` (mean and population variance over the non-missing
    values of each column, a zero variance scales by 1), then missing values are filled with
    the column's mean AFTER scaling (``_feature.py:116-121``)."""
    x = np.array(columns, dtype=np.float64, ndmin=2, copy=True)
    if x.shape[0] == 1 and np.ndim(columns) == 1:
        x = x.T
    ok = ~np.isnan(x)
    cnt = ok.sum(axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = np.where(ok, x, 0.0).sum(axis=0) / cnt
        var = (np.where(ok, x - mean, 0.0) ** 2).sum(axis=0) / cnt
    scale = np.sqrt(var)
    # scikit-learn's test for a (near) constant column (_is_constant_feature): scale by 1
    eps = np.finfo(np.float64).eps
    scale[~(var > cnt * eps * var + (cnt * mean * eps) ** 2)] = 1.0
    z = (x - mean) / scale
    with np.errstate(invalid="ignore", divide="ignore"):
        fill = np.where(ok, z, 0.0).sum(axis=0) / cnt
    return np.where(ok, z, fill[None, :])


def multi_hot(lists: Sequence[Sequence]) -> np.ndarray:
    """``MultiLabelBinarizer().fit_transform``: one 0/1 column per distinct tag, ascending
    (``_feature.py:123-131``)."""
    classes = sorted({t for row in lists for t in row})
    index = {t: i for i, t in enumerate(classes)}
    out = np.zeros((len(lists), len(classes)), dtype=np.float64)
    for r, row in enumerate(lists):
        for t in row:
            out[r, index[t]] = 1.0
    return out


def table(*dense_parts):
    """``csr_matrix(np.hstack(parts))``: the reference turns every feature table into CSR from
    its dense values, so exact zeros are not stored (``_feature.py:68,81``)."""
    from scipy.sparse import csr_matrix

    parts = [np.asarray(p, dtype=np.float64).reshape(len(p), -1) for p in dense_parts]
    X = csr_matrix(np.hstack(parts) if parts else np.zeros((0, 0)))
    X.sort_indices()
    return X


# ---------------------------------------------------------------------------
# assembly on the device
# ---------------------------------------------------------------------------
class OneHot:
    """Segment: the single entry ``(first_column + ids[r], 1.0)``."""

    def __init__(self, ids, size: int):
        self.ids, self.size = np.ascontiguousarray(ids, dtype=np.int32), int(size)
        self.width = self.size


class Rows:
    """Segment: row ``ids[r]`` (``ids is None``: row ``r``) of a feature table, a scipy CSR
    matrix or a ``DeviceCSR`` already in HBM."""

    def __init__(self, block, ids=None):
        self.block = block
        self.ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.int32)
        self.width = int(block.shape[1])


def assemble(rt: Runtime, n_rows: int, segments: Sequence) -> DeviceCSR:
    """The design matrix whose row ``r`` is the concatenation of ``segments`` (each shifted
    behind the one before it), built on the device; returns a ``DeviceCSR`` (``.to_scipy()``
    brings it to the host)."""
    torch = __import__("torch")
    if not segments:
        raise ValueError("no segments")
    descs = (_lib.CsrSegment * len(segments))()
    keep = []
    col = 0
    for d, seg in zip(descs, segments):
        if isinstance(seg, OneHot):
            if seg.ids.shape[0] != n_rows:
                raise ValueError("a one-hot segment needs one id per row")
            ids = rt.upload(seg.ids if n_rows else np.zeros(1, np.int32))
            keep.append(ids)
            d.kind, d.d_ids, d.n_block_rows = 0, ids.data_ptr(), seg.size
        elif isinstance(seg, Rows):
            blk = seg.block if isinstance(seg.block, DeviceCSR) else DeviceCSR(rt, seg.block)
            keep.append(blk)
            d.kind, d.n_block_rows = 1, blk.shape[0]
            d.d_indptr, d.d_indices, d.d_values = blk.indptr.data_ptr(), blk.indices.data_ptr(), blk.values.data_ptr()
            if seg.ids is not None:
                if seg.ids.shape[0] != n_rows:
                    raise ValueError("a gathered segment needs one id per row")
                ids = rt.upload(seg.ids if n_rows else np.zeros(1, np.int32))
                keep.append(ids)
                d.d_ids = ids.data_ptr()
            elif blk.shape[0] != n_rows:
                raise ValueError("a per-row segment needs as many rows as the output")
        else:
            raise TypeError(f"unknown segment {seg!r}")
        d.col_offset = col
        col += seg.width
    indptr = rt.empty((n_rows + 1,), torch.int64)
    nnz = C.c_int64(0)
    _lib.check(rt.lib.rfm_csr_assemble_count(rt.ctx, descs, len(segments), n_rows, indptr.data_ptr(), C.byref(nnz)))
    indices = rt.empty((max(nnz.value, 1),), torch.int32)
    values = rt.empty((max(nnz.value, 1),), torch.float64)
    _lib.check(rt.lib.rfm_csr_assemble_fill(rt.ctx, descs, len(segments), n_rows, indptr.data_ptr(),
                                            indices.data_ptr(), values.data_ptr()))
    rt.sync()
    return DeviceCSR.from_device(rt, (n_rows, col), int(nnz.value), indptr, indices, values)


def fm_features_kuairec(rt: Runtime, users, items, n_users: int, n_items: int, interaction_table,
                        user_table, item_table) -> DeviceCSR:
    """``FeatureGenerator.load`` (``kuairec/_feature.py:35-88``): [one-hot user | one-hot item |
    interaction columns | user table row | video table row]."""
    return assemble(rt, len(users), [OneHot(users, n_users), OneHot(items, n_items), Rows(interaction_table),
                                     Rows(user_table, users), Rows(item_table, items)])


def fm_features_coat(rt: Runtime, users, items, user_table, item_table) -> DeviceCSR:
    """``DatasetPreparer._get_fm_features`` (``coat/_preparer.py:133-170``): [one-hot user | user
    table row | one-hot item | item table row]; the one-hots are identity matrices as wide as
    the tables are long (:51-55)."""
    return assemble(rt, len(users), [OneHot(users, user_table.shape[0]), Rows(user_table, users),
                                     OneHot(items, item_table.shape[0]), Rows(item_table, items)])


def take_rows(rt: Runtime, X, indices) -> DeviceCSR:
    """``features[indices]`` (``kuairec/_preparer.py:117-136``, ``loader.py:104-115``)."""
    return assemble(rt, len(indices), [Rows(X, indices)])


def negative_sample(labels, seed: int, negative_multiple: int = 1) -> np.ndarray:
    """Row numbers of all positives followed by as many (x ``negative_multiple``) negatives,
    drawn as ``np.random.seed(seed); np.random.permutation(negatives)[:n]``
    (``kuairec/_preparer.py:90-115``; Coat: ``coat/_preparer.py:120-131``) -- the legacy MT19937
    Fisher-Yates shuffle, bit for bit (the swap partners do not depend on the contents, so
    permuting the negatives is indexing them with the shuffled ``arange``)."""
    labels = np.asarray(labels)
    pos = np.flatnonzero(labels == 1)
    neg = np.flatnonzero(labels != 1)
    if neg.shape[0] > 1:
        neg = neg[sample_batches(neg.shape[0], neg.shape[0], int(seed), 1)[0]]
    return np.r_[pos, neg[: pos.shape[0] * int(negative_multiple)]]


# ---------------------------------------------------------------------------
# binary CSR cache
# ---------------------------------------------------------------------------
MAGIC = b"RFMCSR01"
_ALIGN = 64


def _sections(n_rows: int, nnz: int, has_labels: bool, has_pscores: bool) -> List[Tuple[str, np.dtype, int]]:
    out = [("indptr", np.dtype(np.int64), n_rows + 1), ("indices", np.dtype(np.int32), nnz),
           ("values", np.dtype(np.float64), nnz)]
    if has_labels:
        out.append(("labels", np.dtype(np.float64), n_rows))
    if has_pscores:
        out.append(("pscores", np.dtype(np.float64), n_rows))
    return out


def save_csr(path: str, X, labels=None, pscores=None) -> None:
    """Write a log as one flat file: a 64-byte header (magic, n_rows, n_cols, nnz, which of
    labels / pscores follow) and the arrays in the ABI's dtypes (indptr int64, indices int32,
    values / labels / pscores float64), each starting on a 64-byte boundary -- so
    :func:`load_csr` can map it and hand the pieces to the device as they are."""
    if isinstance(X, DeviceCSR):
        X = X.to_scipy()
    X = X.tocsr()
    n_rows, n_cols = X.shape
    arrays = {"indptr": np.ascontiguousarray(X.indptr, dtype=np.int64),
              "indices": np.ascontiguousarray(X.indices, dtype=np.int32),
              "values": np.ascontiguousarray(X.data, dtype=np.float64)}
    if labels is not None:
        arrays["labels"] = np.ascontiguousarray(labels, dtype=np.float64)
    if pscores is not None:
        arrays["pscores"] = np.ascontiguousarray(pscores, dtype=np.float64)
    for name in ("labels", "pscores"):
        if name in arrays and arrays[name].shape != (n_rows,):
            raise ValueError(f"{name} must have one value per row")
    header = np.zeros(8, dtype=np.int64)
    header[0] = int.from_bytes(MAGIC, "little")
    header[1:6] = (n_rows, n_cols, int(X.nnz), int(labels is not None), int(pscores is not None))
    tmp = path + ".tmp.%d" % os.getpid()
    with open(tmp, "wb") as fh:
        fh.write(header.tobytes())
        for name, dtype, count in _sections(n_rows, int(X.nnz), labels is not None, pscores is not None):
            fh.write(b"\0" * (-fh.tell() % _ALIGN))
            assert arrays[name].dtype == dtype and arrays[name].shape == (count,)
            fh.write(arrays[name].tobytes())
    os.replace(tmp, path)


def load_csr(path: str, mmap: bool = True) -> dict:
    """``{"shape", "indptr", "indices", "values", ["labels"], ["pscores"]}`` of a file written by
    :func:`save_csr`; the arrays are read-only views of the mapped file (``mmap=False``:
    copies in memory)."""
    header = np.fromfile(path, dtype=np.int64, count=8)
    if header.shape[0] != 8 or int(header[0]).to_bytes(8, "little") != MAGIC:
        raise ValueError(f"{path} is not an RFMCSR01 file")
    n_rows, n_cols, nnz, has_l, has_p = (int(v) for v in header[1:6])
    size = os.path.getsize(path)
    out = {"shape": (n_rows, n_cols)}
    at = 64
    for name, dtype, count in _sections(n_rows, nnz, bool(has_l), bool(has_p)):
        at += -at % _ALIGN
        end = at + count * dtype.itemsize
        if end > size:
            raise ValueError(f"{path} is truncated ({name} needs bytes up to {end}, the file has {size})")
        arr = np.memmap(path, dtype=dtype, mode="r", offset=at, shape=(count,)) if count else np.zeros(0, dtype)
        out[name] = arr if mmap else np.array(arr)
        at = end
    ip = out["indptr"]
    if ip[0] != 0 or ip[-1] != nnz:
        raise ValueError(f"{path}: indptr does not span the {nnz} entries")
    return out


def load_csr_to_device(rt: Runtime, path: str) -> Tuple[DeviceCSR, Optional[np.ndarray], Optional[np.ndarray]]:
    """The cached log in HBM (no SciPy object in between) and its labels / pscores."""
    d = load_csr(path)
    dev = DeviceCSR.from_device(rt, d["shape"], int(d["indptr"][-1]), rt.upload(np.asarray(d["indptr"])),
                                rt.upload(np.asarray(d["indices"]) if len(d["indices"]) else np.zeros(1, np.int32)),
                                rt.upload(np.asarray(d["values"]) if len(d["values"]) else np.zeros(1)))
    return dev, d.get("labels"), d.get("pscores")
