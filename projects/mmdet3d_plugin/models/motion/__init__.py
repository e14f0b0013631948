from .blocks import SparseMotionRefinementModule  # noqa: F401
