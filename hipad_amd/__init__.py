"""Importable alias of the ``hip-ad_amd/`` package directory (its name has a hyphen).

``import hipad_amd.lib`` loads ``hip-ad_amd/lib.py`` etc.
"""
import os as _os
import sys as _sys

# runtime flags that must be in the environment before the HIP runtime starts (see runtime_env.py)
for _k, _v in (("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0"),):
    _os.environ.setdefault(_k, _v)
TORCH_IMPORTED_FIRST = "torch" in _sys.modules

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "hip-ad_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _sys, _k, _v
