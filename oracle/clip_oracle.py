"""CPU restatement of the OpenCLIP text tower as `FrozenOpenCLIPEmbedder` drives it
(sgm/modules/encoders/modules.py:559-632: token + positional embedding -> the first `layers - layer_idx` residual
attention blocks under the causal attn_mask -> ln_final).

TEST INFRASTRUCTURE ONLY (used by tests/; never imported by panacea_amd).

**Parity unpinned.**  The arithmetic of this class is the third-party package `open_clip_torch` (pinned as
open-clip-torch==2.20.0 in the reference's panacea.yml), which is neither under /root/reference nor installed in this image,
and the reference holds no test or golden vector for it.  What is restated here is that package's published architecture
(open_clip/transformer.py, `ResidualAttentionBlock`): x = x + MHA(ln_1(x), attn_mask); x = x + c_proj(GELU(c_fc(ln_2(x)))) with
`torch.nn.MultiheadAttention` semantics (packed in_proj [3W, W], heads of W / H, scaled dot product, additive -inf mask above
the diagonal, out_proj) — evaluated with PyTorch's own `multi_head_attention_forward`, i.e. the very function open_clip calls.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def text_tower(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, heads: int, layer: str = "penultimate",
               prefix: str = "model.") -> torch.Tensor:
    """tokens [B, L] int64 -> [B, L, W] fp32"""
    g = lambda k: sd[prefix + k].float()   # noqa: E731
    x = g("token_embedding.weight")[tokens] + g("positional_embedding")          # modules.py:605-606
    L = x.shape[1]
    mask = torch.full((L, L), float("-inf")).triu_(1)                             # open_clip CLIP.build_attention_mask
    rb = prefix + "transformer.resblocks."
    n_layers = 1 + max(int(k[len(rb):].split(".")[0]) for k in sd if k.startswith(rb))
    n_run = n_layers - (1 if layer == "penultimate" else 0)                       # modules.py:613-616
    x = x.permute(1, 0, 2)                                                        # NLD -> LND
    W = x.shape[-1]
    for i in range(n_run):
        b = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (W,), g(b + "ln_1.weight"), g(b + "ln_1.bias"), 1e-5)
        a, _ = F.multi_head_attention_forward(
            h, h, h, W, heads, g(b + "attn.in_proj_weight"), g(b + "attn.in_proj_bias"), None, None, False, 0.0,
            g(b + "attn.out_proj.weight"), g(b + "attn.out_proj.bias"), training=False, need_weights=False, attn_mask=mask)
        x = x + a
        h = F.layer_norm(x, (W,), g(b + "ln_2.weight"), g(b + "ln_2.bias"), 1e-5)
        h = F.linear(F.gelu(F.linear(h, g(b + "mlp.c_fc.weight"), g(b + "mlp.c_fc.bias"))),
                     g(b + "mlp.c_proj.weight"), g(b + "mlp.c_proj.bias"))
        x = x + h
    x = x.permute(1, 0, 2)
    return F.layer_norm(x, (W,), g("ln_final.weight"), g("ln_final.bias"), 1e-5)    # modules.py:610
