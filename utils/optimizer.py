from relevance_factorizationmachine_amd.optimizer import DeviceSGD as SGD  # noqa: F401
