from relevance_factorizationmachine_amd.fm import FactorizationMachines  # noqa: F401
