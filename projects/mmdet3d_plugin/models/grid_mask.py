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

    @staticmethod
    def _stripes(length, canvas, d, ln, st, device):
        """1 outside the zeroed stripes, for the centre crop of ``length`` out of ``canvas``."""
        pos = torch.arange(length, device=device) + (canvas - length) // 2
        k = torch.div(pos - st, d, rounding_mode="floor")
        inside = (pos >= st) & (k < canvas // d) & ((pos - st) - k * d < ln)
        return (~inside).float()

    def forward(self, x):
        if np.random.rand() > self.prob or not self.training:
            return x
        n, c, h, w = x.shape
        hh, ww = int(1.5 * h), int(1.5 * w)
        d = np.random.randint(2, h)
        ln = min(max(int(d * self.ratio + 0.5), 1), d - 1)
        st_h, st_w = np.random.randint(d), np.random.randint(d)
        np.random.randint(self.rotate)  # keep the host RNG stream aligned with the reference
        rows = self._stripes(h, hh, d, ln, st_h, x.device) if self.use_h else torch.ones(h, device=x.device)
        cols = self._stripes(w, ww, d, ln, st_w, x.device) if self.use_w else torch.ones(w, device=x.device)
        mask = rows[:, None] * cols[None, :]
        if self.mode == 1:
            mask = 1 - mask
        mask = mask.to(x.dtype)
        if self.offset:
            noise = torch.from_numpy(2 * (np.random.rand(h, w) - 0.5)).to(x)
            return x * mask + noise * (1 - mask)
        return x * mask
