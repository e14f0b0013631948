"""Shared builder of the HiP-AD Bench2Drive model configs (stage 1 and stage 2).

The reference ships two ~670-line dict files (projects/configs/hipad_b2d_stage{1,2}.py) that differ in
a handful of entries; both are expressed here by one parametrised builder so the hot-path modules can
be instantiated from this repo alone.  ``tests/test_configs.py`` checks (where the reference tree is
present) that ``model`` comes out identical, key for key, to the reference files'.  Dataset /
pipeline / evaluation sections are outside the hot path: for a real
training run use the reference's own config file, which builds the same model dict.
"""
import os


def _repo_root():
    return os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def hipad_b2d(stage, project_dir=None, input_shape=(640, 352), num_gpus=8):
    if stage not in (1, 2):
        raise ValueError(stage)
    project_dir = project_dir or _repo_root()
    kmeans = f"{project_dir}/data/kmeans"
    det_class_names = ["car", "van", "truck", "bicycle", "traffic_sign", "traffic_cone", "traffic_light",
                       "pedestrian", "others"]
    map_class_names = ["Broken", "Solid", "SolidSolid", "Center"]
    num_det_classes, num_map_classes = len(det_class_names), len(map_class_names)
    map_roi_size, map_num_pts = (30, 60), 20
    fut_ts, fut_mode = 6, 6
    ego_fut_ts, ego_fut_cmd, ego_fut_mode = 6, 1, 48
    embed_dims, num_groups, num_decoder, num_single_frame_decoder = 256, 8, 6, 1
    strides = [4, 8, 16, 32]
    num_levels, num_depth_layers, drop_out, decouple_attn = len(strides), 3, 0.1, True
    use_deformable_func = True
    temporal = temporal_det = temporal_map = temporal_ego = temporal_plan = True

    query_select = ["det", "map", "plan", "ego"]
    task_select = ["det", "map", "plan", "ego"] + (["motion"] if stage == 2 else [])
    single = ["concat", "gnn", "inter_gnn", "norm", "split", "deformable", "concat", "ffn", "norm", "split", "refine"]
    temporal_layer = single[:1] + ["temp_gnn"] + single[1:]
    operation_order = single * num_single_frame_decoder + temporal_layer * (num_decoder - num_single_frame_decoder)

    anchor_paths = {"det": f"{kmeans}/b2d_det_900.npy", "map": f"{kmeans}/b2d_map_100.npy",
                    "motion": f"{kmeans}/b2d_motion_{fut_mode}.npy"}
    path_2m, path_5m = f"{kmeans}/b2d_plan_spat_6x8_2m.npy", f"{kmeans}/b2d_plan_spat_6x8_5m.npy"
    if stage == 1:
        plan_anchor_paths, plan_anchor_refer, plan_speed_refer = path_5m, ("temp", "2hz"), None
        plan_anchor_types = [("temp", "2hz")]
    else:
        speed_bins = [(0, 0.4), (0.4, 3), (3, 999)]
        plan_anchor_types = ([("temp", "5hz"), ("spat", "2m"), ("temp", "2hz"), ("spat", "5m")]
                             + [("speed", "5hz", b) for b in speed_bins] + [("speed", "2hz", b) for b in speed_bins])
        plan_anchor_paths = {t: (path_2m if t[1] in ("5hz", "2m") else path_5m) for t in plan_anchor_types}
        plan_speed_refer, plan_anchor_refer = ("temp", "5hz"), ("spat", "2m")

    def mha(width):
        return dict(type="MultiheadFlashAttention", embed_dims=width, num_heads=num_groups, batch_first=True,
                    dropout=drop_out)

    def deformable(kps):
        return dict(type="DeformableFeatureAggregation", embed_dims=embed_dims, num_groups=num_groups,
                    num_levels=num_levels, num_cams=6, attn_drop=0.15, use_deformable_func=use_deformable_func,
                    use_camera_embed=True, residual_mode="cat", kps_generator=kps)

    def line_kps(num_sample):
        return dict(type="SparsePoint3DKeyPointsGenerator", embed_dims=embed_dims, num_sample=num_sample,
                    num_learnable_pts=3, fix_height=(0, 0.5, -0.5, 1, -1), ground_height=-1.84023)

    box_offsets = [[0, 0, 0], [0.45, 0, 0], [-0.45, 0, 0], [0, 0.45, 0], [0, -0.45, 0], [0, 0, 0.45], [0, 0, -0.45]]
    fmap_scale = (input_shape[1] / strides[-1], input_shape[0] / strides[-1])
    focal = dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25)
    w_plan = 1.0 if stage == 2 else 0.0

    head = dict(
        type="SparseOneDecoder", task_select=task_select, query_select=query_select, operation_order=operation_order,
        num_single_frame_decoder=num_single_frame_decoder)
    if stage == 2:
        head["plan_speed_refer"] = plan_speed_refer
    head.update(
        plan_anchor_refer=plan_anchor_refer, with_command_embed=True, with_target_point_embed=True,
        with_supervise_ego_status=True, with_ego_instance_feature=True, with_incremental_plan_refine=True,
        motion_anchor=anchor_paths["motion"], cls_threshold_to_reg=0.05,
        det_instance_bank=dict(type="InstanceBank", num_anchor=900, embed_dims=embed_dims, anchor=anchor_paths["det"],
                               anchor_handler=dict(type="SparseBox3DKeyPointsGenerator"),
                               num_temp_instances=600 if temporal_det else -1, confidence_decay=0.6, feat_grad=False,
                               class_names=det_class_names,
                               zero_velocity_classes=["traffic_sign", "traffic_cone", "traffic_light"]),
        map_instance_bank=dict(type="InstanceBank", num_anchor=100, embed_dims=embed_dims, anchor=anchor_paths["map"],
                               anchor_handler=dict(type="SparsePoint3DKeyPointsGenerator"),
                               num_temp_instances=0 if temporal_map else -1, confidence_decay=0.6, feat_grad=True),
        ego_instance_bank=dict(type="EgoInstanceBank", anchor_type="b2d", embed_dims=embed_dims,
                               num_temp_instances=1 if temporal_ego else -1, feature_map_scale=fmap_scale),
        plan_instance_bank=dict(type="PlanningInstanceBank", embed_dims=embed_dims, ego_fut_ts=ego_fut_ts,
                                ego_fut_cmd=ego_fut_cmd, ego_fut_mode=ego_fut_mode,
                                num_temp_mode=ego_fut_mode if temporal_plan else -1, feature_map_scale=fmap_scale,
                                anchor_paths=plan_anchor_paths, anchor_types=plan_anchor_types),
        det_anchor_encoder=dict(type="SparseBox3DEncoder", vel_dims=3,
                                embed_dims=[128, 32, 32, 64] if decouple_attn else 256,
                                mode="cat" if decouple_attn else "add", output_fc=not decouple_attn, in_loops=1,
                                out_loops=4 if decouple_attn else 2),
        map_anchor_encoder=dict(type="SparsePoint3DEncoder", embed_dims=embed_dims, num_sample=map_num_pts,
                                return_points_embed=True),
        plan_anchor_encoder=dict(type="SparsePoint3DEncoder", embed_dims=embed_dims, num_sample=ego_fut_ts,
                                 return_points_embed=True),
        custom_op=dict(type="CustomOperation"),
        temp_graph_model=dict(type="TemporalSeparateAttention", query_select=query_select,
                              query_list=[["det"], ["map"], ["plan", "ego"]],
                              key_list=[["det"], ["map"], ["det", "map"]], decouple_list=[True, False, False],
                              attn=[mha(embed_dims * 2), mha(embed_dims), mha(embed_dims)]) if temporal else None,
        graph_model=dict(type="SeparateAttention", query_select=query_select, separate_list=[["det"], ["map"]],
                         decouple_list=[True, False], attn=[mha(embed_dims * 2), mha(embed_dims)]),
        inter_graph_model=dict(type="InteractiveAttention", query_select=query_select, query_list=[["plan", "ego"]],
                               key_list=[["det", "map"]], decouple_list=[False], attn=[mha(embed_dims)]),
        norm_layer=dict(type="LN", normalized_shape=embed_dims),
        ffn=dict(type="AsymmetricFFN", in_channels=embed_dims * 2, pre_norm=dict(type="LN"), embed_dims=embed_dims,
                 feedforward_channels=embed_dims * 4, num_fcs=2, ffn_drop=drop_out,
                 act_cfg=dict(type="ReLU", inplace=True)),
        det_deformable=deformable(dict(type="SparseBox3DKeyPointsGenerator", num_learnable_pts=6, fix_scale=box_offsets)),
        map_deformable=deformable(line_kps(map_num_pts)),
        ego_deformable=deformable(dict(type="SparseBox3DKeyPointsGenerator", num_learnable_pts=12,
                                       fix_scale=[[0.45, 0, 0]])),
        plan_deformable=deformable(line_kps(ego_fut_ts)),
        det_refine_layer=dict(type="SparseBox3DRefinementModule", embed_dims=embed_dims, num_cls=num_det_classes,
                              refine_yaw=True, with_quality_estimation=True),
        map_refine_layer=dict(type="SparsePoint3DRefinementModule", embed_dims=embed_dims, num_sample=map_num_pts,
                              num_cls=num_map_classes),
        ego_refine_layer=dict(type="EgoStatusRefinementModule", embed_dims=embed_dims),
        plan_refine_layer=dict(type="SparsePlanAlignRefinementModule", embed_dims=embed_dims, ego_fut_ts=ego_fut_ts,
                               ego_fut_cmd=ego_fut_cmd, ego_fut_mode=ego_fut_mode, anchor_types=plan_anchor_types),
        motion_refine_layer=dict(type="SparseMotionRefinementModule", embed_dims=embed_dims, fut_ts=fut_ts,
                                 fut_mode=fut_mode),
        det_sampler=dict(type="SparseBox3DTarget", num_dn_groups=0, num_temp_dn_groups=0,
                         dn_noise_scale=[2.0] * 3 + [0.5] * 7, max_dn_gt=32, add_neg_dn=True, cls_weight=2.0,
                         box_weight=0.25, reg_weights=[2.0] * 3 + [0.5] * 3 + [0.0] * 4,
                         cls_wise_reg_weights={det_class_names.index("traffic_cone"):
                                               [2.0, 2.0, 2.0, 1.0, 1.0, 1.0, 0.0, 0.0, 1.0, 1.0]}),
        map_sampler=dict(type="SparsePoint3DTarget",
                         assigner=dict(type="HungarianLinesAssigner",
                                       cost=dict(type="MapQueriesCost", cls_cost=dict(type="FocalLossCost", weight=1.0),
                                                 reg_cost=dict(type="LinesL1Cost", weight=10.0, beta=0.01, permute=True))),
                         num_cls=num_map_classes, num_sample=map_num_pts, roi_size=map_roi_size),
        plan_sampler=dict(type="SparsePlanTarget", ego_fut_ts=ego_fut_ts, ego_fut_cmd=ego_fut_cmd, ego_fut_mode=ego_fut_mode),
        align_sampler=dict(type="AlignPlanTarget", ego_fut_ts=ego_fut_ts, ego_fut_cmd=ego_fut_cmd, ego_fut_mode=ego_fut_mode),
        motion_sampler=dict(type="SparseMotionTarget"),
        loss_det_cls=dict(focal, loss_weight=2.0),
        loss_det_reg=dict(type="SparseBox3DLoss", loss_box=dict(type="L1Loss", loss_weight=0.25),
                          loss_centerness=dict(type="CrossEntropyLoss", use_sigmoid=True),
                          loss_yawness=dict(type="GaussianFocalLoss")),
        loss_map_cls=dict(focal, loss_weight=1.0),
        loss_map_reg=dict(type="SparseLineLoss", loss_line=dict(type="LinesL1Loss", loss_weight=10.0, beta=0.01),
                          num_sample=map_num_pts, roi_size=map_roi_size),
        loss_ego_status=dict(type="L1Loss", loss_weight=w_plan),
        loss_plan_cls=dict(focal, loss_weight=0.5 * w_plan),
        loss_plan_reg=dict(type="L1Loss", loss_weight=w_plan),
        loss_motion_cls=dict(focal, loss_weight=0.2),
        loss_motion_reg=dict(type="L1Loss", loss_weight=0.2),
        det_reg_weights=[2.0] * 3 + [1.0] * 7, map_reg_weights=[1.0] * 40,
        det_decoder=dict(type="SparseBox3DDecoder"), map_decoder=dict(type="SparsePoint3DDecoder"),
        plan_decoder=dict(type="SparsePlanDecoder", ego_fut_ts=ego_fut_ts, ego_fut_cmd=ego_fut_cmd,
                          ego_fut_mode=ego_fut_mode, ego_vehicle="b2d", anchor_types=plan_anchor_types,
                          anchor_refer=plan_anchor_refer,
                          **(dict(speed_refer=plan_speed_refer) if stage == 2 else {}), with_rescore=True),
        motion_decoder=dict(type="SparseMotionDecoder"),
    )
    model = dict(
        type="SparseDetector", use_grid_mask=True, use_deformable_func=use_deformable_func,
        img_backbone=dict(type="ResNet", depth=50, num_stages=4, frozen_stages=-1, norm_eval=False, style="pytorch",
                          with_cp=True, out_indices=(0, 1, 2, 3), norm_cfg=dict(type="BN", requires_grad=True),
                          pretrained="ckpts/resnet50-19c8e357.pth"),
        img_neck=dict(type="FPN", num_outs=num_levels, start_level=0, out_channels=embed_dims,
                      add_extra_convs="on_output", relu_before_extra_convs=True,
                      norm_cfg=dict(type="BN", requires_grad=True), no_norm_on_lateral=True,
                      in_channels=[256, 512, 1024, 2048]),
        depth_branch=dict(type="DenseDepthNet", embed_dims=embed_dims, num_depth_layers=num_depth_layers, loss_weight=0.2),
        head=dict(type="SparseHead", task_config=dict(with_onedecoder=True), evaluate_bench2dive=True,
                  onedecoder_head=head),
    )
    batch_size = 8 if stage == 1 else 6
    num_epochs = 12 if stage == 1 else 18
    iters_per_epoch = int(234769 // (num_gpus * batch_size))
    return dict(
        log_level="INFO", dist_params=dict(backend="nccl"), plugin=True, plugin_dir="projects/mmdet3d_plugin/",
        num_gpus=num_gpus, batch_size=batch_size, num_iters_per_epoch=iters_per_epoch, num_epochs=num_epochs,
        fp16=dict(loss_scale=32.0), input_shape=tuple(input_shape), num_cams=6, embed_dims=embed_dims,
        strides=strides, num_levels=num_levels, project_dir=project_dir, model=model,
        det_class_names=det_class_names, map_class_names=map_class_names, plan_anchor_types=plan_anchor_types,
        optimizer=dict(type="AdamW", lr=2e-4, weight_decay=0.001,
                       paramwise_cfg=dict(custom_keys={"img_backbone": dict(lr_mult=0.5)})),
        optimizer_config=dict(grad_clip=dict(max_norm=25, norm_type=2)),
        lr_config=dict(policy="CosineAnnealing", warmup="linear", warmup_iters=500, warmup_ratio=1.0 / 3, min_lr_ratio=1e-3),
        runner=dict(type="IterBasedRunner", max_iters=iters_per_epoch * num_epochs),
        load_from="./work_dirs/hipad_b2d_stage1/latest.pth" if stage == 2 else None,
    )
