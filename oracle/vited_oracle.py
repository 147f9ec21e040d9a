"""CPU fp32 oracle for the ViT encoder-decoder (ViT-ED) hot path.

TEST INFRASTRUCTURE - NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
(``vit-ed_amd``) never does and fails loudly when its HIP library is missing.

It is an independent restatement, in plain eager PyTorch fp32, of what the reference computes on
the path named by BASELINE.json ``north_star``:

* models/vision_transformer.py:13-80    Attention        -> :func:`self_attention`
* models/vision_transformer.py:83-127   Block            -> :func:`encoder_block`
* models/vision_transformer.py:130-200  CrossAttention   -> :func:`cross_attention`
* models/vision_transformer.py:213-272  CrossBlock       -> :func:`decoder_block`
* models/vision_transformer.py:378-420  forward_first_part / prepare_x2 / cross_part /
  forward_second_part / forward_features / forward -> :class:`OracleViTED`
* timm==0.9.2 (requirements.txt:2; NOT vendored, NOT installed here): PatchEmbed, Mlp,
  VisionTransformer.{_pos_embed, forward_head, init_weights}.  Restated from that release's
  documented behaviour: conv k=s=p patch projection, fc1->GELU(erf)->fc2, ``cat(cls) + pos_embed``,
  ``x[:, 0] -> head``; LayerNorm eps 1e-6.

Pinning status: ``oracle/pin_against_reference.py`` runs the reference's OWN classes (loaded by
file path from /root/reference, with ``oracle/_timm_standin`` supplying the five absent timm
names) against this module and writes the fixtures in ``tests/golden``.  So the parts of the
algorithm that live in /root/reference are pinned against the reference itself; the timm pieces
are "parity unpinned" (the reference ships no numeric test or golden vector at all -
tests/hisfrag_evaluation_test.py:143 only pins two-stage == one-shot).

Inactive reference features (LayerScale, DropPath, dropout, q/k-norm) are Identity for every
shipped config (models/build.py:19-32 never forwards them) and are not restated.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

LN_EPS = 1e-6


@dataclass(frozen=True)
class ViTEDShape:
    """The constructor arguments models/build.py:19-32 forwards for MODEL.TYPE == 'pjs'."""
    img_size: int = 64
    patch_size: int = 8
    in_chans: int = 3
    num_classes: int = 4
    embed_dim: int = 384
    depth: int = 8
    c_depth: int = 8
    num_heads: int = 12
    mlp_ratio: float = 4.0
    qkv_bias: bool = True

    @property
    def n1(self):
        return (self.img_size // self.patch_size) ** 2

    @property
    def n2(self):
        return self.n1 + 1

    @property
    def hidden(self):
        return int(self.embed_dim * self.mlp_ratio)

    @property
    def head_dim(self):
        return self.embed_dim // self.num_heads


SHAPE_A = ViTEDShape()  # configs/puzzle/div2k_erosion7_4bin_patch8_64.yaml + config.py:68-79 defaults
SHAPE_H = ViTEDShape(img_size=512, patch_size=16, num_classes=1, num_heads=6, depth=12, c_depth=12)
SHAPE_T = ViTEDShape(img_size=64, patch_size=32, num_classes=1, embed_dim=32, num_heads=1, depth=1, c_depth=1)


# ----------------------------------------------------------------------------------------------
# functional core
# ----------------------------------------------------------------------------------------------
def _heads(t, num_heads):
    b, n, d = t.shape
    return t.view(b, n, num_heads, d // num_heads).transpose(1, 2)  # [B, h, N, hd]


def _sdpa(q, k, v):
    """softmax(q k^T / sqrt(hd)) v, no mask, no dropout (vision_transformer.py:63-75,183-195)."""
    scale = q.shape[-1] ** -0.5
    p = torch.softmax((q * scale) @ k.transpose(-2, -1), dim=-1)
    o = p @ v
    b, h, n, hd = o.shape
    return o.transpose(1, 2).reshape(b, n, h * hd)


def self_attention(m, x, num_heads):
    """qkv columns are ordered [3][h][hd] (vision_transformer.py:58)."""
    q, k, v = F.linear(x, m.qkv.weight, m.qkv.bias).chunk(3, dim=-1)
    o = _sdpa(_heads(q, num_heads), _heads(k, num_heads), _heads(v, num_heads))
    return F.linear(o, m.proj.weight, m.proj.bias)


def cross_attention(m, x, context, num_heads):
    """q from the image-2 tokens, k/v from image-1 features; kv columns [2][h][hd] (:177-178)."""
    q = F.linear(x, m.q.weight, m.q.bias)
    k, v = F.linear(context, m.kv.weight, m.kv.bias).chunk(2, dim=-1)
    o = _sdpa(_heads(q, num_heads), _heads(k, num_heads), _heads(v, num_heads))
    return F.linear(o, m.proj.weight, m.proj.bias)


def _ln(m, x):
    return F.layer_norm(x, (x.shape[-1],), m.weight, m.bias, LN_EPS)


def _mlp(m, x):
    return F.linear(F.gelu(F.linear(x, m.fc1.weight, m.fc1.bias)), m.fc2.weight, m.fc2.bias)


def encoder_block(m, x, num_heads):
    x = x + self_attention(m.attn, _ln(m.norm1, x), num_heads)
    return x + _mlp(m.mlp, _ln(m.norm2, x))


def decoder_block(m, x, context, num_heads):
    x = x + self_attention(m.attn, _ln(m.norm1, x), num_heads)
    x = x + cross_attention(m.cross_attn, _ln(m.norm_cross, x), _ln(m.norm_context, context), num_heads)
    return x + _mlp(m.mlp, _ln(m.norm2, x))


# ----------------------------------------------------------------------------------------------
# module tree with the reference's state_dict layout
# ----------------------------------------------------------------------------------------------
class _Bag(nn.Module):
    """Plain container; children/parameters are attached by attribute."""


def _linear(out_f, in_f, bias=True):
    return nn.Linear(in_f, out_f, bias=bias)


def _norm(d):
    return nn.LayerNorm(d, eps=LN_EPS)


def _mlp_bag(d, hidden):
    m = _Bag()
    m.fc1 = _linear(hidden, d)
    m.fc2 = _linear(d, hidden)
    return m


def _self_attn_bag(d, qkv_bias):
    m = _Bag()
    m.qkv = _linear(3 * d, d, qkv_bias)
    m.proj = _linear(d, d)
    return m


def _encoder_bag(s: ViTEDShape):
    m = _Bag()
    m.norm1 = _norm(s.embed_dim)
    m.attn = _self_attn_bag(s.embed_dim, s.qkv_bias)
    m.norm2 = _norm(s.embed_dim)
    m.mlp = _mlp_bag(s.embed_dim, s.hidden)
    return m


def _decoder_bag(s: ViTEDShape):
    m = _Bag()
    m.norm1 = _norm(s.embed_dim)
    m.attn = _self_attn_bag(s.embed_dim, s.qkv_bias)
    m.norm_cross = _norm(s.embed_dim)
    m.norm_context = _norm(s.embed_dim)
    ca = _Bag()
    ca.q = _linear(s.embed_dim, s.embed_dim, s.qkv_bias)
    ca.kv = _linear(2 * s.embed_dim, s.embed_dim, s.qkv_bias)
    ca.proj = _linear(s.embed_dim, s.embed_dim)
    m.cross_attn = ca
    m.norm2 = _norm(s.embed_dim)
    m.mlp = _mlp_bag(s.embed_dim, s.hidden)
    return m


class OracleViTED(nn.Module):
    """fp32 CPU restatement of VisionTransformerCustom (vision_transformer.py:275-420)."""

    def __init__(self, shape: ViTEDShape = SHAPE_A):
        super().__init__()
        s = self.shape = shape
        d = s.embed_dim
        self.cls_token = nn.Parameter(torch.zeros(1, 1, d))
        self.pos_embed = nn.Parameter(torch.zeros(1, s.n1 + 1, d))
        pe = _Bag()
        pe.proj = nn.Conv2d(s.in_chans, d, kernel_size=s.patch_size, stride=s.patch_size, bias=True)
        self.patch_embed = pe
        self.blocks = nn.ModuleList([_encoder_bag(s) for _ in range(s.depth)])
        self.norm = _norm(d)
        self.head = _linear(s.num_classes, d)
        self._init_encoder_side()  # timm init runs inside the base ctor, BEFORE the decoder exists
        self.cross_blocks = nn.ModuleList([_decoder_bag(s) for _ in range(s.c_depth)])

    def _init_encoder_side(self):
        """timm 0.9.2 default init: Linear trunc-normal(.02)/zero bias, pos_embed
        trunc-normal(.02), cls_token normal(1e-6); conv and LayerNorm keep PyTorch defaults.  The
        decoder Linears are created afterwards and keep PyTorch's default kaiming-uniform init."""
        nn.init.trunc_normal_(self.pos_embed, std=.02, a=-2., b=2.)
        nn.init.normal_(self.cls_token, std=1e-6)
        for mod in self.modules():
            if isinstance(mod, nn.Linear):
                nn.init.trunc_normal_(mod.weight, std=.02, a=-2., b=2.)
                if mod.bias is not None:
                    nn.init.zeros_(mod.bias)

    # -- pieces (names follow the reference's methods) ---------------------------------------
    def _patch_tokens(self, img):
        s = self.shape
        assert img.shape[-2:] == (s.img_size, s.img_size), "input size does not match the model"
        return F.conv2d(img, self.patch_embed.proj.weight, self.patch_embed.proj.bias,
                        stride=s.patch_size).flatten(2).transpose(1, 2)

    def forward_first_part(self, x1):
        x = self._patch_tokens(x1) + self.pos_embed[:, 1:]          # :378-384, no cls token
        for blk in self.blocks:
            x = encoder_block(blk, x, self.shape.num_heads)
        return x

    def prepare_x2(self, x2):
        x = self._patch_tokens(x2)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)  # timm _pos_embed
        return x + self.pos_embed

    def cross_part(self, x1, x2):
        for blk in self.cross_blocks:
            x2 = decoder_block(blk, x2, x1, self.shape.num_heads)
        return _ln(self.norm, x2)

    def forward_second_part(self, x1, x2):
        return self.cross_part(x1, self.prepare_x2(x2))

    def forward_head(self, x):
        return F.linear(x[:, 0], self.head.weight, self.head.bias)

    def forward(self, x, x2=None, forward_first_part=False):
        if forward_first_part:
            return self.forward_first_part(x)
        if x2 is not None:
            return self.forward_head(self.forward_second_part(x, x2))
        x1, x2 = torch.unbind(x, 1)
        return self.forward_head(self.forward_second_part(self.forward_first_part(x1), x2))


# ----------------------------------------------------------------------------------------------
# deterministic closed-form fillers (fixtures must not depend on any RNG implementation)
# ----------------------------------------------------------------------------------------------
def closed_form(shape, salt: int, scale: float, offset: float = 0.0):
    """value[i] = offset + scale * sin(0.37*i*(1+salt%7) + 1.3*salt) * cos(0.011*i + salt)."""
    n = int(math.prod(shape))
    i = torch.arange(n, dtype=torch.float64)
    v = torch.sin(0.37 * i * (1 + salt % 7) + 1.3 * salt) * torch.cos(0.011 * i + salt)
    return (offset + scale * v).to(torch.float32).reshape(shape)


def fill_closed_form_(module: nn.Module):
    """Overwrite every parameter with the closed-form filler (salt = crc32 of the tensor's name,
    so the reference's module and the oracle get identical values whatever their key order)."""
    import zlib
    with torch.no_grad():
        for name, p in module.state_dict().items():
            salt = zlib.crc32(name.encode()) % 9973
            parts = name.split('.')
            if len(parts) >= 2 and parts[-2].startswith('norm') and parts[-1] == 'weight':
                p.copy_(closed_form(p.shape, salt, 0.2, 1.0))
            elif name.endswith('bias') or name in ('cls_token',):
                p.copy_(closed_form(p.shape, salt, 0.05))
            elif name == 'pos_embed':
                p.copy_(closed_form(p.shape, salt, 0.1))
            else:
                fan_in = p[0].numel() if p.ndim > 1 else p.numel()
                p.copy_(closed_form(p.shape, salt, 1.5 / math.sqrt(fan_in)))
    return module


def closed_form_pairs(batch: int, s: ViTEDShape, salt: int = 1000):
    """Synthetic stacked pair tensor [B, 2, C, S, S] in [-1, 1] (data/transforms.py:14-18 range)."""
    return closed_form((batch, 2, s.in_chans, s.img_size, s.img_size), salt, 1.0)


def state_dict_spec(s: ViTEDShape) -> "OrderedDict[str, tuple]":
    """name -> shape of every tensor the reference's state_dict holds."""
    return OrderedDict((k, tuple(v.shape)) for k, v in OracleViTED(s).state_dict().items())
