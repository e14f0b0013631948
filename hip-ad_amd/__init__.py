"""hip-ad_amd -- MI355X-native implementation of HiP-AD's hot path.

Holds only what the path needs: ``csrc/`` (HIP kernels + the C-ABI library
``libhipad.so``), ``lib.py`` (ctypes binding of include/hipad.h), ``synthetic.py``
(synthetic camera rig / pyramid geometry) and the host-side modules that mirror the
reference's ``projects.mmdet3d_plugin`` interface for this path.

The directory name carries a hyphen (fixed by the build contract), so it is imported
through the alias package ``hipad_amd`` at the repo root.
"""
__version__ = "0.1.0"
