"""Importable alias of the ``hip-ad_amd/`` package directory (its name has a hyphen).

``import hipad_amd.lib`` loads ``hip-ad_amd/lib.py`` etc.
"""
import os as _os
import sys as _sys

# runtime flags that must be in the environment before the HIP runtime starts (see runtime_env.py)
# FLAGS_PRESET: the process started with them (a launcher exported them) -- then the import order does not matter.
# TORCH_IMPORTED_FIRST: torch (and with it possibly the HIP runtime) was loaded before this module could set them;
# runtime_env.graph_replay_is_safe() refuses captured steps in that case unless FLAGS_PRESET.
FLAGS_PRESET = all(_os.environ.get(_k) == _v for _k, _v in (("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0"),))
TORCH_IMPORTED_FIRST = "torch" in _sys.modules
for _k, _v in (("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0"),):
    _os.environ.setdefault(_k, _v)

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "hip-ad_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _sys, _k, _v
