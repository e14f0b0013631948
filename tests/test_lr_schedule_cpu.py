"""Learning-rate schedule of the reference config (linear warm-up over 500 iterations from 1/3, then cosine annealing
to 1e-3; projects/configs/hipad_b2d_stage2.py lr_config): the Python closed form, the C function the optimiser kernel
shares its expression with (hipad_lr_factor, host-callable) and hand-computed values agree."""
import math

import pytest


def test_closed_form_values():
    from hipad_amd.optim import lr_factor
    cfg = dict(policy="CosineAnnealing", warmup="linear", warmup_iters=500, warmup_ratio=1.0 / 3, min_lr_ratio=1e-3)
    T = 100000
    assert lr_factor(cfg, 0, T) == pytest.approx(1.0 / 3, rel=1e-6)           # warm-up starts at warmup_ratio
    assert lr_factor(cfg, 250, T) == pytest.approx((1 - 0.5 * (2 / 3)) * (1e-3 + 0.5 * 0.999 * (1 + math.cos(math.pi * 250 / T))), rel=1e-9)
    assert lr_factor(cfg, 500, T) == pytest.approx(1e-3 + 0.5 * 0.999 * (1 + math.cos(math.pi * 500 / T)), rel=1e-12)
    assert lr_factor(cfg, T // 2, T) == pytest.approx(1e-3 + 0.5 * 0.999, rel=1e-9)
    assert lr_factor(cfg, T, T) == pytest.approx(1e-3, rel=1e-9)
    assert lr_factor(None, 7, T) == 1.0
    with pytest.raises(NotImplementedError):
        lr_factor(dict(policy="step"), 0, T)


def test_c_function_matches_closed_form():
    from hipad_amd import lib
    from hipad_amd.optim import lr_factor, schedule_struct
    cfg = dict(policy="CosineAnnealing", warmup="linear", warmup_iters=500, warmup_ratio=1.0 / 3, min_lr_ratio=1e-3)
    T = 88038
    s = schedule_struct(cfg, T)
    for it in (0, 1, 17, 499, 500, 501, 4000, T // 3, T // 2, T - 1, T, T + 5):
        assert lib.lr_factor(s, it) == pytest.approx(lr_factor(cfg, it, T), rel=2e-5, abs=1e-7), it
    assert lib.lr_factor(None, 3) == 1.0


def test_stage2_config_carries_the_schedule():
    from projects.configs._hipad_b2d_common import hipad_b2d
    cfg = hipad_b2d(stage=2)
    assert cfg["lr_config"]["policy"] == "CosineAnnealing" and cfg["lr_config"]["warmup_iters"] == 500
    assert cfg["runner"]["max_iters"] > 0
