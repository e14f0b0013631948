"""Data-pipeline side of the hot path: the image leg on the device and the host logic
that feeds it.  Same module and class names as the reference's ``projects.mmdet3d_plugin.datasets`` for what is built:

  pipelines.ResizeCropFlipImage, pipelines.NormalizeMultiviewImage   (device kernels when handed device tensors)
  pipelines.DeviceImageTransform (ours)                              the two above + HWC->CHW in two launches
  samplers.GroupInBatchSampler                                       sequence-grouped infinite sampler (host logic)
  augmentation.get_augmentation / camera_matrices / invert_pose      Bench2DriveDataset.get_augmentation, the matrix part of
                                                                     get_data_info, invert_pose

Not built (out of section 8's scope): file loading / JPEG decode, the annotation database, map vectorisation, lidar
depth maps, PhotoMetricDistortionMultiViewImage (cv2 colour-space arithmetic) and the evaluation code.
"""
from .pipelines import *  # noqa: F401,F403
from .samplers import *  # noqa: F401,F403
from .augmentation import camera_matrices, get_augmentation, invert_pose  # noqa: F401
