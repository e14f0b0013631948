from .blocks import (KeyPoint3DEncoder, SparsePoint3DEncoder, SparsePoint3DKeyPointsGenerator,  # noqa: F401
                     SparsePoint3DRefinementModule)
