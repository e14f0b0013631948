"""HiP-AD Bench2Drive stage 1 (perception + 48 planning modes; counterpart of the reference's
projects/configs/hipad_b2d_stage1.py -- same ``model`` dict, see _hipad_b2d_common.py)."""
from projects.configs._hipad_b2d_common import hipad_b2d as _hipad_b2d

globals().update(_hipad_b2d(stage=1))
