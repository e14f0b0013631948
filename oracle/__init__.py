"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference algorithm for the hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker / the timed baseline -- never from the product path
(``hip-ad_amd/`` and ``projects/`` must not import it; tests/test_cabi_cpu.py
``test_product_path_never_imports_oracle`` enforces that).

Contents
  daf_oracle.c   plain-C restatement of the two CUDA kernels (built by oracle/Makefile); one thread = the
                 sequential checker, several threads (bench only) give bitwise the same result
  daf.py         ctypes loader for it (numpy in / numpy out)
  cpu_frame.py   the whole training frame on CPU tensors for bench.py's cpu_baseline: torch restatements of the
                 attention core, the sampling-weights softmax and the 3D->2D projection + the C aggregation
"""
