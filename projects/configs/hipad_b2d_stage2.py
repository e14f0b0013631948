"""HiP-AD Bench2Drive stage 2 (adds motion + 10 planning anchor groups = 480 plan queries; counterpart of
the reference's projects/configs/hipad_b2d_stage2.py -- same ``model`` dict, see _hipad_b2d_common.py)."""
from projects.configs._hipad_b2d_common import hipad_b2d as _hipad_b2d

globals().update(_hipad_b2d(stage=2))
