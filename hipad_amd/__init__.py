"""Importable alias of the ``hip-ad_amd/`` package directory (its name has a hyphen).

``import hipad_amd.lib`` loads ``hip-ad_amd/lib.py`` etc.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "hip-ad_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
