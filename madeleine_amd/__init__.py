"""madeleine_amd -- MI355X-native (gfx950) implementation of MADELEINE's cross-stain SSL pretrain hot path.

Drop-in surface (same names / signatures as the reference's madeleine.models.Model, madeleine.models.abmil,
madeleine.utils.loss, madeleine.utils.trainer):

    from madeleine_amd import MADELEINE, ABMILEmbedder, BatchedABMIL, create_model
    from madeleine_amd import InfoNCE, GOT, calculate_losses, train_loop, run_inference

The numeric work runs in csrc/libmadeleine_amd.so through the C ABI of include/madeleine_amd.h.
Importing this package does not load the library (so model construction / state_dict handling works on
any box); the first forward does, and raises if it is unavailable -- there is no fallback path.
"""
from .abmil import BatchedABMIL
from .loss import GOT, InfoNCE, info_nce, init_intra_wsi_loss_function
from .model import ABMILEmbedder, MADELEINE, create_model
from .trainer import calculate_losses, train_loop
from .utils import create_model_from_pretrained, extract_slide_level_embeddings, load_checkpoint, run_inference

__all__ = ["MADELEINE", "ABMILEmbedder", "BatchedABMIL", "create_model", "InfoNCE", "info_nce", "GOT",
           "init_intra_wsi_loss_function", "calculate_losses", "train_loop", "run_inference", "extract_slide_level_embeddings",
           "load_checkpoint", "create_model_from_pretrained"]
__version__ = "0.2"
