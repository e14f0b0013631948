"""The replay guard (hip-ad_amd/runtime_env.py): captured steps are refused when torch was imported before
hipad_amd could set the HIP runtime flags, unless the process was started with them."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, env_extra=None, drop=("DEBUG_CLR_GRAPH_PACKET_CAPTURE",)):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    env["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return out.stdout.strip().splitlines()[-1]


def test_hipad_first_is_safe():
    assert _run("import hipad_amd, torch; from hipad_amd import runtime_env as r; print(r.graph_replay_is_safe())") == "True"


def test_torch_first_is_refused():
    out = _run("import torch, hipad_amd; from hipad_amd import runtime_env as r; print(r.graph_replay_is_safe(), '|', r.why_unsafe())")
    assert out.startswith("False"), out
    assert "torch was imported before" in out


def test_torch_first_with_preset_flag_is_safe():
    out = _run("import torch, hipad_amd; from hipad_amd import runtime_env as r; print(r.graph_replay_is_safe())",
               env_extra={"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"})
    assert out == "True"


def test_wrong_value_is_refused():
    out = _run("import hipad_amd, torch; from hipad_amd import runtime_env as r; print(r.graph_replay_is_safe())",
               env_extra={"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"})
    assert out == "False"


def test_zero_arena_hands_out_zero_slices_and_clears_them_with_one_fill():
    """hipad_amd.functional._ZeroArena (partial-sum scratch of the BatchNorm kernels): slices are disjoint, zero when
    handed out, dirtied slices are zero again after reset(), an exhausted arena falls back to fresh zeros and grows."""
    import torch
    from hipad_amd.functional import _ZeroArena
    a = _ZeroArena()
    first = a.take(10, "cpu")                       # no reset() yet: plain zeros
    assert first.numel() == 64 and float(first.abs().sum()) == 0
    a.reset("cpu", capacity=256)
    s1, s2 = a.take(100, "cpu"), a.take(64, "cpu")
    assert s1.numel() == 128 and s2.numel() == 64 and s1.data_ptr() + 4 * 128 == s2.data_ptr()
    s1.fill_(3.0); s2.fill_(5.0)
    spill = a.take(200, "cpu")                      # 128 + 64 + 256 > 256: not from the arena
    assert float(spill.abs().sum()) == 0 and not (a.buf.data_ptr() <= spill.data_ptr() < a.buf.data_ptr() + 4 * a.buf.numel())
    a.reset("cpu", capacity=256)                    # clears what was handed out and grows to hold the high-water mark
    assert a.buf.numel() >= 448 and float(a.buf.abs().sum()) == 0
    t = a.take(200, "cpu")
    assert float(t.abs().sum()) == 0 and a.used == 256
