from relevance_factorizationmachine_amd.base import PointwiseBaseRecommender  # noqa: F401
