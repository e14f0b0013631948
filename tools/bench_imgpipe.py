"""Device image pipeline: time per sample (6 frames 900 x 1600 -> 6 x 3 x 256 x 704 normalised float32) against its HBM
roofline and against the reference's host path (PIL resize/crop/flip/rotate + numpy normalise, one process).

    python tools/bench_imgpipe.py [--iters 200]
Algorithmic bytes per sample: the source rows the vertical pass reads (3 B per pixel) + the intermediate rows written and
read once (3 B per pixel each way) + 12 B per output pixel."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import hipad_amd  # noqa
import numpy as np
import torch
from hipad_amd import imgpipe as P

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=200); a = ap.parse_args()
MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
rng = np.random.default_rng(0)
imgs = rng.integers(0, 256, (6, 900, 1600, 3), dtype=np.uint8)
cfg = dict(resize=0.44, crop=(0, 140, 704, 396), flip=True, rotate=2.5)
src = torch.from_numpy(imgs).cuda()
for _ in range(5):
    out = P.transform_images(src, cfg, MEAN, STD, True, channels_last=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    out = P.transform_images(src, cfg, MEAN, STD, True, channels_last=True)
e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / a.iters
plan = P.make_plan(900, 1600, cfg)
nbytes = 6 * (plan.rows * 1600 * 3 + 2 * plan.rows * plan.res_w * 3 + 256 * 704 * 12)
from PIL import Image
t = time.perf_counter(); n = 0
while time.perf_counter() - t < 3.0:
    for im in imgs:
        p = Image.fromarray(im).resize((704, 396)).crop(cfg["crop"]).transpose(method=Image.FLIP_LEFT_RIGHT).rotate(cfg["rotate"])
        x = np.array(p).astype(np.float32)[..., ::-1]
        x = ((x - np.float32(MEAN)) * (1.0 / np.float64(STD)).astype(np.float32)).transpose(2, 0, 1)
    n += 1
cpu_ms = 1e3 * (time.perf_counter() - t) / n
print(json.dumps(dict(workload="image pipeline, 6 x 900x1600 uint8 -> 6x3x256x704 fp32 channels-last", us_per_sample=round(us, 2),
                      samples_per_s=round(1e6 / us, 1), algorithmic_MB=round(nbytes / 1e6, 2), achieved_GBs=round(nbytes / us / 1e3, 1),
                      hbm_frac=round(nbytes / us / 1e3 / 8000.0, 4), launches=2,
                      cpu_reference_ms_per_sample=round(cpu_ms, 2), cpu_cores=1, speedup=round(cpu_ms * 1e3 / us, 1))))
