from .augment import ResizeCropFlipImage
from .transform import DeviceImageTransform, NormalizeMultiviewImage

__all__ = ["ResizeCropFlipImage", "NormalizeMultiviewImage", "DeviceImageTransform"]
