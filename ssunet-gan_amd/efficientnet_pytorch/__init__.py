"""EfficientNet encoder, MI355X-native (mirrors the reference's vendored scripts/efficientnet_pytorch,
v0.5.1; unwired in the reference -- SURVEY.md 8a row A10: `extract_features` and `MBConvBlock` are the
per-op scope; the classifier head and `from_pretrained` are out of scope)."""
__version__ = "0.5.1"
from .model import EfficientNet, MBConvBlock
from .utils import (GlobalParams, BlockArgs, BlockDecoder, efficientnet, get_model_params, efficientnet_params,
                    round_filters, round_repeats, Conv2dStaticSamePadding, MemoryEfficientSwish, Swish, drop_connect)
