"""MI355X-native (gfx950) implementation of the trimodal gesture-generation GAN hot path.

Drop-in for the reference's PoseGenerator / ConvDiscriminator / pose-mode EmbeddingNet nn.Modules and its
train_iter_gan step; all arithmetic runs in hand-written HIP kernels (csrc/, libtrimodal_hip.so) behind the C ABI of
include/trimodal_hip.h.  See DESIGN.md.
"""
from . import _lib  # noqa: F401  (does not load the .so until first use; load() raises loudly if it is missing)
from .modules import ConvDiscriminator, EmbeddingNet, PoseGenerator  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .train_gan import GanTrainer, GraphedGanStep, StepLosses  # noqa: F401
from .vocab import Vocab  # noqa: F401
from . import checkpoint, config, data, ddp, eval_metrics, fgd, layers, ops, synthesize  # noqa: F401,E402  (hip.fgd, hip.config, ... as INTEGRATION.md uses them)

__all__ = ["PoseGenerator", "ConvDiscriminator", "EmbeddingNet", "FusedAdam", "GanTrainer", "GraphedGanStep", "StepLosses",
           "Vocab"]
