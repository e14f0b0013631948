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
