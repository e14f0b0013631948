"""Model-side mirror of the reference plugin for the hot path."""
from .blocks import AsymmetricFFN, CustomOperation, DeformableFeatureAggregation, DenseDepthNet  # noqa: F401
from .det import *  # noqa: F401,F403
from .map import *  # noqa: F401,F403
