from relevance_factorizationmachine_amd.mf import LogisticMatrixFactorization  # noqa: F401
