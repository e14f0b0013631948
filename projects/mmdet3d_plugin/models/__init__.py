"""Model-side mirror of the reference plugin for the hot path."""
from .attention import FlashMHA, MultiheadFlashAttention, gen_sineembed_for_position  # noqa: F401
from .blocks import AsymmetricFFN, CustomOperation, DeformableFeatureAggregation, DenseDepthNet  # noqa: F401
from . import criterion, decode  # noqa: F401  register the loss / sampler / result-decoder classes
from .det import *  # noqa: F401,F403
from .ego import *  # noqa: F401,F403
from .grid_mask import GridMask  # noqa: F401
from .image_encoder import FPN, ResNet  # noqa: F401
from .instance_bank import InstanceBank  # noqa: F401
from .map import *  # noqa: F401,F403
from .motion import *  # noqa: F401,F403
from .plan import *  # noqa: F401,F403
from .separate_attn import InteractiveAttention, SeparateAttention, TemporalSeparateAttention  # noqa: F401
from .sparse_detector import SparseDetector  # noqa: F401
from .sparse_head import SparseHead  # noqa: F401
from .sparse_onedecoder import SparseOneDecoder  # noqa: F401
