"""Host-side mirror of the reference's ``PointwiseBaseRecommender``
(``src/base.py:9-66``): the hyper-parameter record and the abstract
``fit``/``predict`` pair.  Loss and sigmoid live in the HIP kernels; the helper
below evaluates the loss on the device for callers that used the base-class
method directly."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass

import numpy as np

LOSS_EPS = 1e-8  # src/base.py:42


@dataclass
class PointwiseBaseRecommender(ABC):
    """Fields and order of ``src/base.py:22-27``."""

    estimator: str
    n_epochs: int
    n_factors: int
    lr: float
    batch_size: int
    seed: int

    @abstractmethod
    def fit(self, train, val) -> tuple:
        ...

    @abstractmethod
    def predict(self, **kwargs) -> np.ndarray:
        ...

    def _cross_entropy_loss(self, y_trues, y_scores, pscores, eps: float = LOSS_EPS) -> float:
        """IPS log-loss of given scores (src/base.py:37-61), on the device."""
        from . import _lib

        rt = self._rt
        y = rt.upload(np.asarray(y_trues), dtype=np.float64)
        s = rt.upload(np.asarray(y_scores), dtype=np.float64)
        p = rt.upload(np.asarray(pscores), dtype=np.float64)
        out = rt.empty((1,), y.dtype)
        _lib.check(rt.lib.rfm_ips_logloss(rt.ctx, y.data_ptr(), s.data_ptr(), p.data_ptr(), None,
                                          int(y.shape[0]), float(eps), out.data_ptr()))
        rt.sync()
        return float(out.cpu().numpy()[0])
