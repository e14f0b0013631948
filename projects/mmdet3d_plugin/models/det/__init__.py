from .blocks import SparseBox3DEncoder, SparseBox3DKeyPointsGenerator, SparseBox3DRefinementModule  # noqa: F401
