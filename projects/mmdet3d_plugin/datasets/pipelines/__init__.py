from .augment import BBoxRotation, ResizeCropFlipImage
from .transform import DeviceImageTransform, NormalizeMultiviewImage, NuScenesSparse4DAdaptor

__all__ = ["ResizeCropFlipImage", "BBoxRotation", "NormalizeMultiviewImage", "DeviceImageTransform", "NuScenesSparse4DAdaptor"]
