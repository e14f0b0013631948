"""GridMask input augmentation, built on the device.

Behaviour of the reference module (models/grid_mask.py:75-138) for the settings the detector uses
(use_h = use_w = True, rotate = 1, offset = False, mode = 1): with probability ``prob`` pick a period
d in [2, h), a stripe length l = clamp(round(d*ratio), 1, d-1) and phases st_h, st_w in [0, d); on a
1.5x canvas zero every stripe [k*d + st, k*d + st + l) of rows and of columns, crop the centre, invert
(mode 1) and multiply.  The reference draws the mask with numpy + PIL on the host and uploads it; here
only the three random scalars come from the host RNG and the mask is two aranges on the GPU.
"""
import numpy as np
import torch
import torch.nn as nn

__all__ = ["GridMask"]


class GridMask(nn.Module):
    def __init__(self, use_h, use_w, rotate=1, offset=False, ratio=0.5, mode=0, prob=1.0):
        super().__init__()
        if rotate != 1:
            raise NotImplementedError("mask rotation is unused by the detector (rotate=1)")
        self.use_h, self.use_w, self.rotate, self.offset = use_h, use_w, rotate, offset
        self.ratio, self.mode = ratio, mode
        self.st_prob = self.prob = prob

    def set_prob(self, epoch, max_epoch):
        self.prob = self.st_prob * epoch / max_epoch

    def randomize(self, device):
        """Draw this step's mask parameters on the host (numpy RNG, like the reference) and ship them to
        a 5-float device tensor [apply, d, l, st_h, st_w] with one asynchronous copy.  forward() only
        reads that tensor, so it can sit inside a captured hipGraph while this runs outside."""
        h = getattr(self, "_last_h", None)
        if h is None:
            raise RuntimeError("GridMask.randomize() needs one forward first (image height unknown)")
        apply = float(np.random.rand() <= self.prob)
        d = np.random.randint(2, h)
        ln = min(max(int(d * self.ratio + 0.5), 1), d - 1)
        st_h, st_w = np.random.randint(d), np.random.randint(d)
        np.random.randint(self.rotate)  # keeps the host RNG stream aligned with the reference
        if getattr(self, "_dev", None) is None or self._dev.device != torch.device(device):
            self._dev = torch.empty(5, dtype=torch.float32, device=device)
        on_gpu = torch.device(device).type == "cuda"
        if getattr(self, "_ring", None) is None:
            # staging ring: the host may be frames ahead of the GPU; a slot is rewritten only after the
            # asynchronous copy that last read it has completed
            self._ring = [(torch.empty(5, dtype=torch.float32).pin_memory() if on_gpu else torch.empty(5),
                           torch.cuda.Event() if on_gpu else None) for _ in range(8)]
            self._ring_i = 0
        host, done = self._ring[self._ring_i]
        self._ring_i = (self._ring_i + 1) % len(self._ring)
        if done is not None:
            done.synchronize()
        host.copy_(torch.tensor([apply, d, ln, st_h, st_w], dtype=torch.float32))
        self._dev.copy_(host, non_blocking=True)
        if done is not None:
            done.record()

    @staticmethod
    def _stripes(length, canvas, d, ln, st, device):
        """1 outside the zeroed stripes, for the centre crop of ``length`` out of ``canvas`` (d, ln, st: 0-d tensors)."""
        pos = torch.arange(length, device=device, dtype=torch.float32) + (canvas - length) // 2
        k = torch.floor((pos - st) / d)
        inside = (pos >= st) & (k < torch.floor(canvas / d)) & ((pos - st) - k * d < ln)
        return (~inside).float()

    def forward(self, x):
        if not self.training:
            return x
        n, c, h, w = x.shape
        self._last_h = h
        if not getattr(self, "external_randomize", False):
            self.randomize(x.device)
        p = self._dev
        if (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and not self.offset and not x.requires_grad
                and self.mode in (0, 1)):
            # one launch; ``out_dtype`` / ``out_channels_last`` (set by the detector) make it hand the encoder's first
            # convolution its bf16 channels-last input directly
            from hipad_amd import lib as _lib
            return _lib.grid_mask(x, p, self.use_h, self.use_w, self.mode, getattr(self, "out_dtype", torch.float32),
                                  getattr(self, "out_channels_last", False))
        apply, d, ln, st_h, st_w = p[0], p[1], p[2], p[3], p[4]
        hh, ww = int(1.5 * h), int(1.5 * w)
        rows = self._stripes(h, hh, d, ln, st_h, x.device) if self.use_h else torch.ones(h, device=x.device)
        cols = self._stripes(w, ww, d, ln, st_w, x.device) if self.use_w else torch.ones(w, device=x.device)
        mask = rows[:, None] * cols[None, :]
        if self.mode == 1:
            mask = 1 - mask
        mask = torch.where(apply > 0, mask, torch.ones_like(mask)).to(x.dtype)  # not drawn this step: identity
        if self.offset:
            noise = torch.rand(h, w, device=x.device, dtype=x.dtype) * 2 - 1
            return x * mask + noise * (1 - mask)
        return x * mask
