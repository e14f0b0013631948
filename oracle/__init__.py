"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference algorithm for the hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker / the timed baseline -- never from the product path
(``hip-ad_amd/`` and ``projects/`` must not import it; tests/test_no_oracle_in_product.py
enforces that).

Contents
  daf_oracle.c   plain-C restatement of the two CUDA kernels (built by oracle/Makefile)
  daf.py         ctypes loader for it (numpy in / numpy out)
  blocks_ref.py  numpy/torch restatement of the Python-level functions around the op
                 (feature_maps_format, project_points, _get_weights, key-point generators,
                 the grid_sample fallback with the kernel's border mask)
"""
