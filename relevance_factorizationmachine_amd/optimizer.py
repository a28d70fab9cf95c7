"""Parameter handles with the call surface of the reference's ``SGD``
(``utils/optimizer.py:36-64``): ``h()`` returns the parameters, ``h(index)`` one
entry, ``h.update(grad, index)`` applies ``params[index] -= lr * grad``.

The parameters live in HBM and are updated by the HIP kernels; a handle only
moves them to the host when a caller asks.  ``update`` is kept for API
compatibility (the training path never calls it)."""
from __future__ import annotations

import numpy as np


class DeviceSGD:
    def __init__(self, rt, params: np.ndarray, lr: float):
        self._rt = rt
        self.lr = lr
        self._shape = params.shape
        self.dev = rt.upload(np.ascontiguousarray(params, dtype=np.float64))

    @property
    def params(self) -> np.ndarray:
        self._rt.sync()
        return self.dev.cpu().numpy()

    def __call__(self, index=None):
        p = self.params
        return p if index is None else p[index]

    def update(self, grad, index=None) -> None:
        p = self.params
        if index is None:
            p -= self.lr * grad
        else:
            p[index] -= self.lr * grad
        self.set(p)

    def set(self, params: np.ndarray) -> None:
        host = np.ascontiguousarray(params, dtype=np.float64).reshape(self._shape)
        self.dev.copy_(self._rt.upload(host))
