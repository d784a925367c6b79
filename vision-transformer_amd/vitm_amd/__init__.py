"""vitm_amd -- MI355X (gfx950) forward path for the `vitm` Vision Transformer.

Same module API and state_dict layout as the reference's ``vitm.models`` (``PatchEmbed``, ``Attention``,
``MLP``/``Mlp``, ``LayerScale``, ``Block``, ``VisionTransformer``, ``VITC``); the forward of every module
runs in hand-written HIP kernels behind the C ABI of ``include/vitm_gfx950.h``.  Use it as
``import vitm_amd as vitm``.
"""
from . import models
from .models import PatchEmbed, VisionTransformer, VITC, Attention
from ._lib import VitmError, VitmLibraryError, version as library_version
from .graph import GraphedForward, GraphedTrainStep

__all__ = ["models", "PatchEmbed", "VisionTransformer", "VITC", "Attention",
           "VitmError", "VitmLibraryError", "library_version", "GraphedForward", "GraphedTrainStep"]
__version__ = "0.1.0"
