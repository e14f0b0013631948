"""CPU-side checks: the C-ABI library loads and exports every symbol include/hipad.h declares
(no compute calls without a GPU), the binding table is complete, host logic of the ops
package, and the product path never touches oracle/."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "hipad.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hipad_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from hipad_amd import build, lib
    so = build.build_lib()
    assert os.path.exists(so)
    dll = ctypes.CDLL(so)
    syms = header_symbols()
    assert len(syms) >= 7
    for s in syms:
        assert hasattr(dll, s), f"{s} declared in include/hipad.h but not exported"
    assert set(lib.SIGNATURES) == set(syms), "hip-ad_amd/lib.py binding table out of sync with the header"
    L = lib.load()
    assert L.hipad_abi_version() >= 1
    assert L.hipad_status_string(0) == b"ok"
    assert b"workspace" in L.hipad_status_string(-2)


def test_workspace_query_is_host_only():
    from hipad_amd import lib
    L = lib.load()
    assert L.hipad_daf_forward_workspace(1, 6, 89760, 256, 4, 8192, 13, 8) == 0  # one chunk per anchor
    assert L.hipad_daf_forward_workspace(1, 6, 89760, 256, 4, 900, 13, 8) % (900 * 256 * 4) == 0
    n = L.hipad_daf_forward_workspace(1, 6, 89760, 256, 4, 100, 300, 8)
    assert n > 0 and n % (100 * 256 * 4) == 0
    assert L.hipad_daf_forward_workspace(1, 6, 89760, 250, 4, 100, 300, 8) == 0  # invalid dims


def test_no_cpu_fallback():
    from hipad_amd import lib
    z = torch.zeros(1, 4, 256)
    with pytest.raises(lib.HipadError):
        lib.daf_forward(z, torch.zeros(1, 1, 2, dtype=torch.int32), torch.zeros(1, 1, dtype=torch.int32),
                        torch.zeros(1, 1, 1, 1, 2), torch.zeros(1, 1, 1, 1, 1, 8))


def test_feature_maps_format_matches_reference(golden):
    from projects.mmdet3d_plugin.ops import feature_maps_format
    z = golden("feature_maps_format")
    maps = [torch.from_numpy(z[f"small_map{i}"]) for i in range(4)]
    col, ss, st = feature_maps_format(maps)
    assert torch.equal(col, torch.from_numpy(z["small_col"]))
    assert ss.dtype == torch.int64 and torch.equal(ss, torch.from_numpy(z["small_spatial_shape"]))
    assert st.dtype == torch.int64 and torch.equal(st, torch.from_numpy(z["small_scale_start_index"]))
    back = feature_maps_format([col, ss, st], inverse=True)
    assert len(back) == 1 and all(torch.equal(a, b) for a, b in zip(back[0], maps))
    # a foreign triple (no host mirror attached) must invert too
    back2 = feature_maps_format([col.clone(), ss.clone(), st.clone()], inverse=True)
    assert all(torch.equal(a, b) for a, b in zip(back2[0], maps))


@pytest.mark.parametrize("tag,hw", [("704x256", (256, 704)), ("640x352", (352, 640))])
def test_index_tables_at_real_sizes(golden, tag, hw):
    from hipad_amd import synthetic as syn
    from projects.mmdet3d_plugin.ops import feature_maps_format
    z = golden("feature_maps_format")
    maps = [torch.zeros(1, 6, 1, h, w) for h, w in syn.pyramid_shapes(hw)]
    col, ss, st = feature_maps_format(maps)
    assert np.array_equal(ss.numpy(), z[f"{tag}_spatial_shape"])
    assert np.array_equal(st.numpy(), z[f"{tag}_scale_start_index"])
    assert col.shape[1] == int(z[f"{tag}_num_feat"])
    ss2, st2, F = syn.pyramid_tables(hw)
    assert np.array_equal(ss2, ss.numpy()) and np.array_equal(st2, st.numpy()) and F == col.shape[1]


def test_mixed_camera_groups_roundtrip():
    from projects.mmdet3d_plugin.ops import feature_maps_format
    g = torch.Generator().manual_seed(0)
    ga = [torch.randn(2, 2, 4, h, w, generator=g) for h, w in ((4, 6), (2, 3))]
    gb = [torch.randn(2, 3, 4, h, w, generator=g) for h, w in ((3, 5), (2, 2))]
    col, ss, st = feature_maps_format([ga, gb])
    assert ss.shape == (5, 2, 2) and st[2, 0] == 2 * (24 + 6)
    back = feature_maps_format([col, ss, st], inverse=True)
    assert len(back) == 2
    assert all(torch.equal(a, b) for a, b in zip(back[0], ga)) and all(torch.equal(a, b) for a, b in zip(back[1], gb))


def test_product_path_never_imports_oracle():
    bad = []
    for top in ("hip-ad_amd", "hipad_amd", "projects", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|oracle/_build|libhipad_oracle", txt, flags=re.M):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, f"product files reference the oracle: {bad}"


def test_library_path_override(tmp_path):
    """HIPAD_LIB selects another build of the same ABI; a path without a library fails loudly."""
    import shutil
    import subprocess
    import sys
    from hipad_amd import lib
    other = tmp_path / "libhipad_copy.so"
    shutil.copy(lib.SO_PATH, other)
    code = ("import hipad_amd; from hipad_amd import lib; lib.load(); "
            "print(lib.SO_PATH); print(lib.load().hipad_abi_version())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, HIPAD_LIB=str(other)),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == str(other)
    missing = subprocess.run([sys.executable, "-c", code], cwd=root,
                             env=dict(os.environ, HIPAD_LIB=str(tmp_path / "nope.so")),
                             capture_output=True, text=True, timeout=300)
    assert missing.returncode != 0 and "HipadError" in missing.stderr
