from .blocks import SparsePlanAlignRefinementModule, SparsePlanRefinementModule  # noqa: F401
from .instance_bank import PlanningInstanceBank  # noqa: F401
