"""ssunet-gan_amd: MI355X-native hot path of ideafisher/ssUnet-GAN (segmentation-GAN training).

The package mirrors the reference's module surface (archs, normalization, models_seg_gan,
losses, metrics, srgan_utils, train_seg_gan) and runs it on hand-written gfx950 HIP kernels
behind the C-ABI in include/ssunet_hip.h.  The directory name carries a hyphen (it is the
name the build contract fixes); import it with

    import importlib; pkg = importlib.import_module('ssunet-gan_amd')

or via the root-level shim `import ssunet_gan_amd`.
"""
from . import _lib  # noqa: F401
from . import ops, bf16, blocks, archs, normalization, models_seg_gan, losses, metrics, srgan_utils, optim, dp, train_seg_gan, utils  # noqa: F401
from . import xresidualblock, spectral_norm, batchnorm, efficientnet_pytorch  # noqa: F401  (unwired per-op rows A9-A12)
from . import train  # noqa: F401  (stage-1 trainer)
from . import aerial_image_segmentation_api  # noqa: F401  (sliding-window inference)

__version__ = '0.1.0'
