from .augment import ResizeCropFlipImage, compose_camera_matrices
from .transform import DeviceImageTransform, NormalizeMultiviewImage, NuScenesSparse4DAdaptor

__all__ = ["ResizeCropFlipImage", "compose_camera_matrices", "NormalizeMultiviewImage", "DeviceImageTransform",
           "NuScenesSparse4DAdaptor"]
