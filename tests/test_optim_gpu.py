"""Flat clip + AdamW kernel (hip-ad_amd/csrc/optim.hip) against torch.nn.utils.clip_grad_norm_ +
torch.optim.AdamW with two learning-rate groups, several steps, with and without clipping being active."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(37, 5), (256, 256), (3,), (64, 3, 7, 7), (1,), (130,), (48, 33)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]


@pytest.mark.parametrize("max_norm", [0.5, 1e6, None])
def test_flat_adamw_matches_torch(max_norm):
    from hipad_amd.optim import FlatAdamW
    mine, ref = make_params(0), make_params(0)
    opt = FlatAdamW([(mine[:4], 2e-3), (mine[4:], 1e-3)], weight_decay=1e-2, max_norm=max_norm)
    topt = torch.optim.AdamW([dict(params=ref[:4], lr=2e-3), dict(params=ref[4:], lr=1e-3)], lr=2e-3, weight_decay=1e-2)
    assert all(p.data_ptr() >= opt.flat_p.data_ptr() for p in mine)
    assert all(p.data_ptr() % 256 == 0 and p.grad.data_ptr() % 256 == 0 for p in mine)
    assert all(torch.equal(a, b) for a, b in zip(mine, ref))  # flattening keeps the values
    g = torch.Generator().manual_seed(1)
    for step in range(5):
        grads = [torch.randn(p.shape, generator=g).cuda() * (1 + step) for p in ref]
        for p, q, gr in zip(mine, ref, grads):
            p.grad.copy_(gr)
            q.grad = gr.clone()
        expect_norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(gr) for gr in grads]))
        if max_norm is not None:
            torch.nn.utils.clip_grad_norm_(ref, max_norm)
        topt.step()
        opt.step(zero_grad=True)
        assert abs(float(opt.grad_norm) - float(expect_norm)) <= 1e-5 * float(expect_norm)
        assert int(opt.step_count) == step + 1
        for p, q in zip(mine, ref):
            assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max())), step
        assert float(opt.grads.flat.abs().max()) == 0.0  # cleared for the next step


def test_flat_adamw_rejects_bad_input():
    from hipad_amd import lib
    from hipad_amd.optim import FlatAdamW
    with pytest.raises(ValueError):
        FlatAdamW([], max_norm=1.0)
    p = [torch.nn.Parameter(torch.randn(8, 8).cuda())]
    opt = FlatAdamW([(p, 1e-3)])
    with pytest.raises(lib.HipadError):
        lib.adamw_step(opt.flat_p, opt.grads.flat[:-4], opt.exp_avg, opt.exp_avg_sq, 0, 1e-3, 1e-3, (0.9, 0.999), 1e-8, 0.0,
                       None, opt.step_count, None, opt._ws)


def test_flat_adamw_follows_the_lr_schedule_and_keeps_a_bf16_shadow():
    """Warm-up + cosine annealing evaluated inside the kernel from the device step counter == torch AdamW driven by
    the closed form (hipad_amd.optim.lr_factor) through LambdaLR; the bf16 shadow equals the rounded parameters."""
    from hipad_amd.optim import FlatAdamW, lr_factor
    lr_config = dict(policy="CosineAnnealing", warmup="linear", warmup_iters=4, warmup_ratio=1.0 / 3, min_lr_ratio=1e-3)
    max_iters = 9
    mine, ref = make_params(2), make_params(2)
    opt = FlatAdamW([(mine[:4], 2e-3), (mine[4:], 1e-3)], weight_decay=1e-2, max_norm=5.0, lr_config=lr_config,
                    max_iters=max_iters, bf16_shadow=True)
    for p in mine:
        assert torch.equal(opt.shadow_of(p), p.detach().to(torch.bfloat16))
    topt = torch.optim.AdamW([dict(params=ref[:4], lr=2e-3), dict(params=ref[4:], lr=1e-3)], lr=2e-3, weight_decay=1e-2)
    sched = torch.optim.lr_scheduler.LambdaLR(topt, lambda it: lr_factor(lr_config, it, max_iters))
    g = torch.Generator().manual_seed(3)
    for step in range(max_iters):
        grads = [torch.randn(p.shape, generator=g).cuda() for p in ref]
        for p, q, gr in zip(mine, ref, grads):
            p.grad.copy_(gr)
            q.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref, 5.0)
        topt.step()
        sched.step()
        opt.step()
        assert abs(float(opt.last_lr) - opt.lr_at(step)[0]) <= 1e-6 * opt.lr_at(step)[0], step
        for p, q in zip(mine, ref):
            assert float((p - q).abs().max()) <= 3e-6 * max(1.0, float(q.abs().max())), step
            assert torch.equal(opt.shadow_of(p), p.detach().to(torch.bfloat16)), step


def test_optimizer_keeps_the_chain_operands_current():
    """After a step the fragment-ordered bf16 pair (W, W^T) attached to every small 2-D parameter equals the packed rounded
    fp32 parameter -- the MLP-chain kernels read these copies, and the optimiser updates parameters behind torch's
    back (no version-counter bump)."""
    from hipad_amd import chain as CH
    from hipad_amd.optim import FlatAdamW
    ps = make_params(4)
    opt = FlatAdamW([(ps[:4], 2e-3), (ps[4:], 1e-3)], weight_decay=1e-2, max_norm=1.0)
    mats = [p for p in ps if p.dim() == 2 and max(p.shape) <= 256]
    assert len(mats) == 3
    for step in range(2):
        for p in ps:
            p.grad.copy_(torch.randn_like(p))
        opt.step()
        for p in mats:
            w, wt = CH.bf16_pair(p)
            assert w.data_ptr() == p._hipad_shadow[0].data_ptr()
            assert torch.equal(w, CH.pack_fragments(p.detach()))
            assert torch.equal(wt, CH.pack_fragments(p.detach().t()))


def test_bf16_copies_follow_a_checkpoint_load():
    """weights written by torch AFTER the optimiser took the parameters over (load_state_dict, checkpoint
    resume, p.copy_) must reach the bf16 operand copies the chain kernels and the encoder's convolutions read -- the
    copies are stamped with the parameters' version counters and re-derived on the first use after such a write."""
    from hipad_amd import chain as CH
    from hipad_amd.optim import FlatAdamW, shadow_is_current
    ps = make_params(6)
    opt = FlatAdamW([(ps[:4], 2e-3), (ps[4:], 1e-3)], weight_decay=0.0, max_norm=None, bf16_shadow=True)
    w = ps[1]                                         # (256, 256): has a chain operand pair and a flat bf16 shadow
    shadow = opt.shadow_of(w)
    old_pair = CH.bf16_pair(w)[0].clone()
    new = torch.randn_like(w) * 3
    with torch.no_grad():
        w.copy_(new)                                  # what load_state_dict does
    pair = CH.bf16_pair(w)                            # notices the version counter moved, has the optimiser re-derive
    assert torch.equal(pair[0], CH.pack_fragments(new)) and not torch.equal(pair[0], old_pair)
    assert torch.equal(pair[1], CH.pack_fragments(new.t()))
    assert torch.equal(shadow, new.to(torch.bfloat16))            # the flat shadow (convolution weights) as well
    assert shadow_is_current(w)
    # the optimiser's own updates (raw pointers, no version bump) keep the copies current without another refresh
    for p in ps:
        p.grad.copy_(torch.randn_like(p))
    opt.step()
    assert torch.equal(CH.bf16_pair(w)[0], CH.pack_fragments(w.detach()))
    # new storage behind the optimiser's back is an error, not a silent use of stale copies
    w.data = w.data.clone()
    with pytest.raises(RuntimeError):
        CH.bf16_pair(w)


def test_accumulate_bf16_adds_every_layout_in_one_launch():
    """hipad_accumulate_bf16: dst += src for contiguous, channels-last and lower-rank bf16 sources, more tensors than
    one table holds; equal to torch's add_ (one bf16 -> fp32 widening, one fp32 add: bit-exact)."""
    from hipad_amd import lib
    g = torch.Generator().manual_seed(4)
    shapes = [(64, 3, 7, 7), (256, 64, 1, 1), (64, 64, 3, 3), (5,), (7, 3), (2, 3, 4)] * 12      # 72 > HIPAD_ACC_MAX
    pairs, want = [], []
    for i, s in enumerate(shapes):
        dst = torch.randn(*s, generator=g).cuda()
        src = torch.randn(*s, generator=g).cuda().to(torch.bfloat16)
        if len(s) == 4 and i % 2 == 0:
            src = src.contiguous(memory_format=torch.channels_last)
        want.append(dst + src.float())
        pairs.append((dst, src))
    assert len(pairs) > lib.ACC_MAX
    lib.accumulate_bf16(pairs)
    for (dst, _), w in zip(pairs, want):
        assert torch.equal(dst, w)
    with pytest.raises(lib.HipadError):
        lib.accumulate_bf16([(torch.zeros(3, device="cuda"), torch.zeros(4, device="cuda", dtype=torch.bfloat16))])
    with pytest.raises(lib.HipadError):
        lib.accumulate_bf16([(torch.zeros(3, device="cuda"), torch.zeros(3, device="cuda"))])


def test_convolution_weight_gradients_arrive_batched(monkeypatch):
    """The encoder's convolutions under bf16 autocast with optimiser-kept bf16 weights: all weight gradients of one
    backward pass are added into the fp32 .grad buffers by one engine callback -- same numbers as one add per layer."""
    from projects.mmdet3d_plugin.models import image_encoder as IE
    from hipad_amd import functional as HF
    g = torch.Generator().manual_seed(6)
    # (layer shapes of the encoder's third stage: the library's solvers for them are the ones every other test uses)
    convs = [IE.Conv2d(64, 64, 3, padding=1, bias=False).cuda(), IE.Conv2d(64, 256, 1, bias=False).cuda()]
    for c in convs:
        c.weight._hipad_bf16 = c.weight.detach().to(torch.bfloat16).reshape(-1)
    x = torch.randn(6, 64, 16, 44, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    res = {}
    for batched in (True, False):
        monkeypatch.setattr(IE, "BATCH_WEIGHT_GRADS", batched)
        for c in convs:
            c.weight.grad = torch.full_like(c.weight, 0.5)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = convs[1](convs[0](x))
        y.float().square().sum().backward()
        assert not IE._PENDING
        res[batched] = [c.weight.grad.clone() for c in convs]
    for a, b in zip(res[True], res[False]):
        assert float((a - 0.5).abs().max()) > 0
        # the library may pick other solvers on the second pass (6.0 apart on a largest gradient of 544 seen: 1.5 bf16
        # ulps there): equal to a few bf16 roundings of the gradient, in the norm and element by element
        assert float((a - b).norm() / (b - 0.5).norm()) < 1e-2
        assert float((a - b).abs().max()) <= 2.0 ** -5 * float((b - 0.5).abs().max())
