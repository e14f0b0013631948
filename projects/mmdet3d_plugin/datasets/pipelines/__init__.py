from .augment import BBoxRotation, ResizeCropFlipImage
from .transform import DeviceImageTransform, NormalizeMultiviewImage

__all__ = ["ResizeCropFlipImage", "BBoxRotation", "NormalizeMultiviewImage", "DeviceImageTransform"]
