"""MI355X-native FM/MF mini-batch SGD path.

Drop-in surface of the reference's ``src/fm.py`` / ``src/mf.py``::

    from relevance_factorizationmachine_amd import FactorizationMachines as FM
    from relevance_factorizationmachine_amd import LogisticMatrixFactorization as MF

Importing the package does not touch the GPU; constructing a model does, and
raises if the HIP extension or the device is missing.
"""
from .base import PointwiseBaseRecommender
from .fm import FactorizationMachines
from .mf import LogisticMatrixFactorization
from .optimizer import DeviceSGD

__all__ = [
    "PointwiseBaseRecommender",
    "FactorizationMachines",
    "LogisticMatrixFactorization",
    "DeviceSGD",
]
