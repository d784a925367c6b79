"""Classification wrapper: backbone -> class-token row -> linear head.

API mirror of the reference's ``VITC`` (vitm/models/classifier.py:6-39): ``VITC(vit_model)`` with
attributes ``vit_model`` and ``head = nn.Linear(embed_dim, num_classes)`` and state_dict keys
``vit_model.*``, ``head.{weight,bias}``.  With autograd enabled, fp32 or bf16 images and fp32 parameters that require
gradients, ``forward`` returns logits carrying a graph (``vitm_amd.training``); otherwise it is inference.  When the backbone is this package's ``VisionTransformer`` the final
LayerNorm is applied to the class-token rows only (folded into the head GEMM on the bf16 path).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _ops
from .models import VisionTransformer
from .patch_embeddings import layernorm_params


class VITC(nn.Module):
    def __init__(self, vit_model: VisionTransformer):
        super().__init__()
        self.vit_model = vit_model
        self.head = nn.Linear(self.vit_model.embed_dim, self.vit_model.num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, C, H, W) images -> (B, num_classes) logits, in the dtype of ``x``."""
        _ops.require_cuda(x, "VITC.forward")
        if x.shape[0] == 0:                       # an empty batch is an empty result, as in the reference
            self.vit_model._empty_tokens(x)       # (its shape checks)
            return x.new_empty((0, self.head.out_features), dtype=_ops.compute_dtype(x))
        if (torch.is_grad_enabled() and x.dtype in (torch.float32, torch.bfloat16)
                and any(p.requires_grad for p in self.parameters())):
            # training: the forward that keeps its activations, with a hand-written backward behind it (fp32 images:
            # fp32 arithmetic; bf16 images: bf16 activations and matrix cores over the fp32 master parameters)
            from .. import training
            if training.supported(self) is None and all(p.dtype == torch.float32 for p in self.parameters()):
                return training.vitc_forward_with_grad(self, x)
        vt = self.vit_model
        C = self.head.out_features
        with torch.cuda.device(x.device):
            if isinstance(vt, VisionTransformer):
                x2d, B, N, D = vt._run_blocks(x)              # pre-final-norm tokens, rows (b, n)
                cls_rows = x2d.view(B, N, x2d.shape[1])[:, 0]  # (B, ld) view with row stride N*ld
                ln = layernorm_params(vt.norm_final, "VisionTransformer.norm_final")
            else:                                             # foreign backbone: take its output as is
                emb = vt(x)
                B, N, D = emb.shape
                _ops.compute_dtype(emb)
                cls_rows = emb[:, 0].contiguous()
                kp = _ops.round_up(D, _ops.k_multiple())
                if kp != D:
                    pad = torch.zeros((B, kp), dtype=emb.dtype, device=x.device)
                    pad[:, :D] = cls_rows
                    cls_rows = pad
                ln = None
            w, b = self.head.weight, self.head.bias
            dt = cls_rows.dtype
            f32 = dt == torch.float32
            tag = "f32" if f32 else "bf16"
            if ln is not None and f32:                        # fp32: explicit LayerNorm of the class rows
                normed = torch.zeros((B, cls_rows.shape[1]), dtype=dt, device=x.device)
                _ops.layernorm(cls_rows, normed, B, D, _ops.f32vec(ln[0]), _ops.f32vec(ln[1]), ln[2])
                cls_rows, ln = normed, None
            # logits: bf16 for bf16 callers, fp32 for fp32 callers
            if ln is not None:
                pk = _ops.cached_pack(self, "head_ln_" + tag, (w, b, ln[0], ln[1]),
                                      lambda: _ops.pack_linear(w, b, ln[0], ln[1], dtype=dt))
                stats = _ops.ln_stats(cls_rows, B, D, ln[2])
                logits = torch.empty((B, pk.npad), dtype=dt, device=x.device)
                _ops.gemm(cls_rows, pk, logits, _ops.EPI_LN, B, stats=stats)
            else:
                pk = _ops.cached_pack(self, "head_" + tag, (w, b), lambda: _ops.pack_linear(w, b, dtype=dt))
                logits = torch.empty((B, pk.npad), dtype=dt, device=x.device)
                _ops.gemm(cls_rows, pk, logits, _ops.EPI_BIAS, B)
            return logits[:, :C]
