from .blocks import EgoStatusRefinementModule, SparseEgoRefinementModule  # noqa: F401
from .instance_bank import EgoInstanceBank  # noqa: F401
