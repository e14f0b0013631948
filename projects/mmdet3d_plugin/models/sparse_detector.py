"""Detector shell: multi-view images -> ResNet/FPN pyramid -> flat column layout -> unified decoder.

Registered name ``SparseDetector``; constructor keywords, child names (``img_backbone``, ``img_neck``,
``head``, ``depth_branch``, ``grid_mask``) and the ``extract_feat`` / ``forward_train`` /
``simple_test`` entry points follow the reference (models/sparse_detector.py:25-174).

MI355X specifics: the encoder runs channels-last under bf16 autocast (the reference's ``fp16 =
dict(loss_scale=32.0)`` / ``@auto_fp16`` becomes bf16: same exponent range as fp32, no loss scaling),
the pyramid is cast back to fp32 for the aggregation kernels exactly as the reference does
(``out_fp32=True``), and all later aggregation calls share one feature-gradient buffer
(``shared_feature_grad``).
"""
import torch

from hipad_amd import functional as HF
from hipad_amd.compat import BACKBONES, DETECTORS, HEADS, NECKS, PLUGIN_LAYERS, BaseModule, build_from_cfg

from ..ops import feature_maps_format, level_major_tables, shared_feature_grad
from .grid_mask import GridMask
from .image_encoder import BatchNorm2d as _EncoderBN

__all__ = ["SparseDetector"]


import os as _os

FLAT_BF16 = _os.environ.get("HIPAD_FLAT_BF16", "1") == "1"   # 0: widen the flat pyramid to fp32 (round-1 behaviour)
IN_PLACE_PYRAMID = _os.environ.get("HIPAD_IN_PLACE_PYRAMID", "1") == "1"   # 0: copy the levels into the camera-major layout
FUSED_DEPTH = _os.environ.get("HIPAD_FUSED_DEPTH", "1") == "1"   # 0: depth heads as library convolutions + torch loss ops


@DETECTORS.register_module()
class SparseDetector(BaseModule):
    def __init__(self, img_backbone, head, img_neck=None, init_cfg=None, train_cfg=None, test_cfg=None,
                 pretrained=None, use_grid_mask=True, use_deformable_func=False, depth_branch=None,
                 scenes_tokenizer=None, encoder_dtype=torch.bfloat16):
        super().__init__(init_cfg)
        if scenes_tokenizer is not None:
            raise NotImplementedError("scene tokens are not used by the HiP-AD configs")
        if not use_deformable_func:
            raise ValueError("this build runs the aggregation op only (use_deformable_func=True)")
        if pretrained is not None:
            img_backbone = dict(img_backbone, pretrained=pretrained)
        self.img_backbone = build_from_cfg(img_backbone, BACKBONES)
        self.img_neck = build_from_cfg(img_neck, NECKS) if img_neck is not None else None
        self.head = build_from_cfg(head, HEADS)
        self.use_grid_mask, self.use_deformable_func = use_grid_mask, use_deformable_func
        self.depth_branch = build_from_cfg(depth_branch, PLUGIN_LAYERS) if depth_branch is not None else None
        self.scenes_tokenizer = None
        if use_grid_mask:
            self.grid_mask = GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=1, prob=0.7)
        self.encoder_dtype = encoder_dtype

    def init_weights(self):
        for m in (self.img_backbone, self.img_neck, self.head):
            if m is not None and hasattr(m, "init_weights"):
                m.init_weights()
        for m in self.modules():
            if isinstance(m, _EncoderBN):
                m.defer_counter = True          # extract_feat flushes the queued counters once per frame

    def extract_feat(self, img, return_depth=False, metas=None):
        bs = img.shape[0]
        if img.dim() == 5:
            num_cams = img.shape[1]
            img = img.flatten(end_dim=1)
        else:
            num_cams = 1
        if self.use_grid_mask:
            # in training the mask kernel writes the encoder's input format (dtype + channels-last) in the same pass
            self.grid_mask.out_dtype = self.encoder_dtype if img.is_cuda else torch.float32
            self.grid_mask.out_channels_last = True
            img = self.grid_mask(img)
        img = img.contiguous(memory_format=torch.channels_last)
        if img.is_cuda and self.training:
            HF.BN_ARENA.reset(img.device)        # one fill clears the partial-sum scratch of all norm layers
        flat = tables = None
        with torch.autocast("cuda", dtype=self.encoder_dtype, enabled=img.is_cuda and self.encoder_dtype != torch.float32):
            levels = self.img_backbone(img)
            if self.img_neck is not None:
                out_blocks = None
                if (IN_PLACE_PYRAMID and FLAT_BF16 and img.is_cuda and self.training and self.encoder_dtype == torch.bfloat16
                        and getattr(self.img_neck, "out_channels", None) == 256):
                    # the FPN's last norm layers write the levels straight into the flat pyramid the decoder reads (level by
                    # level: a channels-last level IS a block of rows) -- no copy into the "column" layout
                    hw = [tuple(int(v) for v in t.shape[-2:]) for t in levels[self.img_neck.start_level:self.img_neck.backbone_end_level]]
                    tables = level_major_tables(hw, num_cams, img.device)
                    flat = torch.empty(bs, tables[2], 256, dtype=torch.bfloat16, device=img.device)
                    out_blocks = lambda i, y: flat[:, tables[3][i][0]:tables[3][i][0] + tables[3][i][1]]   # noqa: E731
                levels = self.img_neck(levels, out_blocks) if out_blocks is not None else self.img_neck(levels)
            _EncoderBN.flush_counters()
        in_place = flat is not None and all(f.dim() == 5 for f in levels)
        # levels stay in the encoder's dtype (bf16): the depth heads and the flat-layout copy convert on read,
        # so the pyramid is written once in fp32 (as the flat tensor) instead of twice
        if not in_place:
            levels = [f.reshape((bs, num_cams) + f.shape[1:]) for f in levels]
        else:
            # an alias node between the norm layer and the level: the eager step's two-part backward takes the gradient AT
            # the level, and capturing it at a node's own output makes autograd release that node's buffers early
            levels = [f.view_as(f) for f in levels]

        depths = None
        focal = None if metas is None else metas.get("focal")
        fused_depth = (return_depth and self.depth_branch is not None and in_place and FUSED_DEPTH and torch.is_grad_enabled()
                       and hasattr(self.depth_branch, "on_pyramid")
                       and all(m.weight.dtype == torch.float32 and m.bias is not None for m in self.depth_branch.depth_layers))
        if return_depth and self.depth_branch is not None and not fused_depth:
            depths = self.depth_branch(levels, focal)
        # the flat pyramid keeps the encoder's dtype: bf16 rows go to the aggregation kernels as they are (same values as
        # the reference's fp32 copy of its fp16 pyramid would hold, half the bytes); other widths / dtypes are widened
        if in_place:
            feature_maps = [HF.flat_pyramid(flat, tables[3], levels), tables[0], tables[1]]
        else:
            keep = levels[0].dtype == torch.bfloat16 and levels[0].is_cuda and levels[0].shape[2] == 256 and FLAT_BF16
            feature_maps = feature_maps_format(levels, out_dtype=None if keep else torch.float32)
        feature_maps[0] = shared_feature_grad(feature_maps[0])
        if fused_depth:
            # the depth heads and their loss read the flat pyramid's rows in place and add their feature gradient into the
            # shared buffer of the aggregation calls (evaluated when the loss is asked for)
            depths = self.depth_branch.on_pyramid(feature_maps[0], tables[3], num_cams, levels, focal)
        # cut point of the eager step's two-part backward (hipad_amd.frame.TrainStep); rides on the flat tensor so that it
        # lives exactly as long as the forward's outputs (a persistent reference on the module kills ROCm 7.2's
        # capture_end when the step is captured)
        feature_maps[0]._hipad_levels = levels
        return (feature_maps, depths) if return_depth else feature_maps

    def forward(self, img, **data):
        return self.forward_train(img, **data) if self.training else self.forward_test(img, **data)

    def forward_train(self, img, **data):
        feature_maps, depths = self.extract_feat(img, True, data)
        model_outs = self.head(img, feature_maps, data)
        output = self.head.loss(model_outs, data)
        if depths is not None and "gt_depth" in data:
            output["loss_dense_depth"] = self.depth_branch.loss(depths, data["gt_depth"])
        return output

    def forward_test(self, img, **data):
        if isinstance(img, list):  # single "augmentation"
            data = {k: (v[0] if isinstance(v, list) else v) for k, v in data.items()}
            img = img[0]
        return self.simple_test(img, **data)

    def simple_test(self, img, **data):
        feature_maps = self.extract_feat(img)
        model_outs = self.head(img, feature_maps, data)
        results = self.head.post_process(model_outs, data)
        return [dict(img_bbox=r) for r in results]
