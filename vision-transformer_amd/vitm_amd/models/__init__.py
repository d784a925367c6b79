"""Public model API; the four names the reference exports (vitm/models/__init__.py:1-12) plus the
block-level classes it keeps in ``vitm.models.models``."""
from .patch_embeddings import PatchEmbed
from .models import VisionTransformer, Attention, MLP, Mlp, LayerScale, Block
from .classifier import VITC
from .parallel_block import ParallelBlock

__all__ = ["VisionTransformer", "VITC", "Attention", "PatchEmbed", "MLP", "Mlp", "LayerScale", "Block", "ParallelBlock"]
