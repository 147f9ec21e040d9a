"""``VisionTransformerCustom`` on MI355X: the module contract of the reference
(models/vision_transformer.py:275-420) with every FLOP executed by the HIP kernels in ``csrc/``.

Drop-in surface kept:
  * ctor keyword names that ``models/build.py:19-32`` forwards;
  * ``forward(x)``, ``forward(x, forward_first_part=True)``, ``forward(feats, x2)`` and the
    ``forward_first_part / prepare_x2-less forward_second_part / forward_head``-level methods;
  * parameter names, shapes and ranks == the reference ``state_dict`` (checkpoints load unchanged;
    ``misc/optimizer.py:36-46`` puts ``ndim == 1`` / ``*.bias`` in the no-decay group; ``.head`` is
    re-initialised by ``misc/utils.py:110-119``);
  * raw fp32 logits out.

Precision follows the caller exactly like the reference follows ``torch.cuda.amp.autocast``
(misc/engine.py:208): inside an autocast region the bf16 MFMA kernels run, outside it the fp32
kernels run; ``model.compute_dtype = torch.bfloat16 | torch.float32`` pins it.

There is no PyTorch fallback: calling the model with CPU tensors or without the built
``libvited_hip.so`` raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functions as F_
from .functions import DEC_BLOCK_KEYS, DEC_SHARED_KEYS, ENC_BLOCK_KEYS, ENC_SHARED_KEYS, Runtime

LN_EPS = F_.LN_EPS


class _Holder(nn.Module):
    """Parameter container (its forward is never used: the Functions read the parameters)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError('sub-modules of the HIP ViT-ED are parameter holders; call the model itself')


class _Linear(_Holder):
    def __init__(self, in_f, out_f, bias=True):
        super().__init__()
        ref = nn.Linear(in_f, out_f, bias=bias)  # PyTorch default init (decoder side keeps it)
        self.in_features, self.out_features = in_f, out_f
        self.weight = ref.weight
        self.bias = ref.bias

    def extra_repr(self):
        return f'in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}'


class _Norm(_Holder):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))
        self.eps = LN_EPS

    def extra_repr(self):
        return f'({self.weight.numel()},), eps={self.eps}'


class _SelfAttn(_Holder):
    def __init__(self, d, heads, qkv_bias):
        super().__init__()
        self.num_heads, self.head_dim, self.scale = heads, d // heads, (d // heads) ** -0.5
        self.qkv = _Linear(d, 3 * d, qkv_bias)
        self.proj = _Linear(d, d)
        self.keep_attn = False

    _store, _key = None, None     # set by VisionTransformerCustom: the runtime's attention store and this module's key

    def _lookup(self, what):
        ent = (self._store() or {}).get(self._key) if self._store is not None else None
        if ent is None or what not in ent:
            raise RuntimeError('no attention map recorded: build the model with MODEL.PJS.KEEP_ATTN True (keep_attn=True) and '
                               'run a forward' + (' and a backward' if what == 'grad' else '') + ' first')
        return ent[what]

    def get_attn(self):
        """softmax(q k^T / sqrt(hd)) [B, h, Nq, Nk] of the last forward (vision_transformer.py:46-47,166-167)."""
        return self._lookup('attn')

    def get_attn_gradients(self):
        """d loss / d attn of the last backward (vision_transformer.py:49-50,169-170)."""
        return self._lookup('grad')


class _CrossAttn(_SelfAttn):
    def __init__(self, d, heads, qkv_bias):
        _Holder.__init__(self)
        self.num_heads, self.head_dim, self.scale = heads, d // heads, (d // heads) ** -0.5
        self.q = _Linear(d, d, qkv_bias)
        self.kv = _Linear(d, 2 * d, qkv_bias)
        self.proj = _Linear(d, d)
        self.keep_attn = False


class _Mlp(_Holder):
    def __init__(self, d, hidden):
        super().__init__()
        self.fc1 = _Linear(d, hidden)
        self.fc2 = _Linear(hidden, d)


class Block(_Holder):
    """Encoder block parameters (vision_transformer.py:83-127)."""

    def __init__(self, d, heads, hidden, qkv_bias):
        super().__init__()
        self.norm1 = _Norm(d)
        self.attn = _SelfAttn(d, heads, qkv_bias)
        self.norm2 = _Norm(d)
        self.mlp = _Mlp(d, hidden)


class CrossBlock(_Holder):
    """Decoder block parameters (vision_transformer.py:213-272)."""

    def __init__(self, d, heads, hidden, qkv_bias):
        super().__init__()
        self.norm1 = _Norm(d)
        self.attn = _SelfAttn(d, heads, qkv_bias)
        self.norm_cross = _Norm(d)
        self.norm_context = _Norm(d)
        self.cross_attn = _CrossAttn(d, heads, qkv_bias)
        self.norm2 = _Norm(d)
        self.mlp = _Mlp(d, hidden)


class _PatchEmbed(_Holder):
    def __init__(self, img_size, patch, in_chans, d):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch, patch)
        self.grid_size = (img_size // patch, img_size // patch)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, d, kernel_size=patch, stride=patch, bias=True)  # holder only


def _get(module, dotted):
    for part in dotted.split('.'):
        module = getattr(module, part)
    return module


class VisionTransformerCustom(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, c_depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, keep_attn=False, arch_version='v1', compute_dtype=None,
                 **unsupported):
        super().__init__()
        live = {k: v for k, v in unsupported.items() if v not in (None, False, 0, 0., '', 'token')}
        if live:
            raise NotImplementedError(f'options outside the shipped pjs configs are not on the HIP hot path: {sorted(live)}')
        if embed_dim % num_heads:
            raise AssertionError('dim should be divisible by num_heads')
        if img_size % patch_size:
            raise AssertionError('image size must be a multiple of the patch size')
        # limits of the HIP kernels (they would return VITED_ERR_UNSUPPORTED at the first launch): say so here
        if embed_dim % 4 or embed_dim > 1024:
            raise NotImplementedError(f'EMBED_DIM={embed_dim}: the LayerNorm kernels cover multiples of 4 up to 1024 '
                                      '(csrc/layernorm.hip, LN_MAX_VPL)')
        if embed_dim // num_heads not in (32, 64):
            raise NotImplementedError(f'head_dim={embed_dim // num_heads}: the attention kernels cover head_dim 32 and 64 '
                                      '(12 x 32 at configs/puzzle, 6 x 64 at configs/hisfrag)')
        self.img_size, self.patch_size, self.in_chans = img_size, patch_size, in_chans
        self.num_classes, self.embed_dim = num_classes, embed_dim
        self.num_features = embed_dim
        self.depth, self.c_depth, self.num_heads = depth, c_depth, num_heads
        self.keep_attn = bool(keep_attn)    # visualisation slow path: attention maps are ALSO materialised (PyTorch ops)
        self.arch_version = arch_version.lower()
        self.compute_dtype = compute_dtype
        # uint8 inputs are normalised inside the patch-embedding kernel: ToTensor + Normalize(0.5, 0.5) of data/transforms.py:14-18
        self.input_mean, self.input_std = (0.5,) * in_chans, (0.5,) * in_chans
        hidden = int(embed_dim * mlp_ratio)
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        n1 = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n1 + 1, embed_dim))
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, hidden, qkv_bias) for _ in range(depth)])
        self.norm = _Norm(embed_dim)
        self.head = _Linear(embed_dim, num_classes)
        self._init_like_timm()  # runs before the decoder exists, exactly as in the reference ctor (:344-347)
        self.cross_blocks = nn.ModuleList([CrossBlock(embed_dim, num_heads, hidden, qkv_bias) for _ in range(c_depth)])
        self._runtimes = {}
        import weakref
        me = weakref.ref(self)
        self._attn_store = {}
        store = lambda: (me()._attn_store if me() is not None else None)
        for i, blk in enumerate(self.blocks):
            blk.attn._store, blk.attn._key, blk.attn.keep_attn = store, ('blocks', i, 'attn'), self.keep_attn
        for i, blk in enumerate(self.cross_blocks):
            blk.attn._store, blk.attn._key, blk.attn.keep_attn = store, ('cross_blocks', i, 'attn'), self.keep_attn
            blk.cross_attn._store, blk.cross_attn._key, blk.cross_attn.keep_attn = store, ('cross_blocks', i, 'cross_attn'), self.keep_attn
        print(f'Using {arch_version} Arch!')

    def _init_like_timm(self):
        nn.init.trunc_normal_(self.pos_embed, std=.02, a=-2., b=2.)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, _Linear):
                nn.init.trunc_normal_(m.weight, std=.02, a=-2., b=2.)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    # -- launch context ----------------------------------------------------------------------
    def _act_dtype(self):
        if self.compute_dtype is not None:
            return self.compute_dtype
        return torch.bfloat16 if torch.is_autocast_enabled() else torch.float32

    def runtime(self, act_dtype=None) -> Runtime:
        dt = act_dtype or self._act_dtype()
        rt = self._runtimes.get(dt)
        if rt is None:
            rt = Runtime(img_size=self.img_size, patch_size=self.patch_size, in_chans=self.in_chans,
                         num_classes=self.num_classes, embed_dim=self.embed_dim, depth=self.depth, c_depth=self.c_depth,
                         num_heads=self.num_heads, act_dtype=dt)
            self._runtimes[dt] = rt
        rt.direct_grads = bool(getattr(self, 'direct_param_grads', False))
        rt.input_mean, rt.input_std = self.input_mean, self.input_std
        rt.keep_attn, rt.attn_store = self.keep_attn, self._attn_store
        return rt

    def _encoder_params(self):
        ps = [_get(self, k) for k in ENC_SHARED_KEYS]
        for blk in self.blocks:
            ps += [_get(blk, k) for k in ENC_BLOCK_KEYS]
        return ps

    def _decoder_params(self):
        ps = [_get(self, k) for k in DEC_SHARED_KEYS]
        for blk in self.cross_blocks:
            ps += [_get(blk, k) for k in DEC_BLOCK_KEYS]
        return ps

    def _check_images(self, img):
        if img.dim() != 4 or img.shape[1] != self.in_chans or img.shape[2] != self.img_size or img.shape[3] != self.img_size:
            raise AssertionError(f'input {tuple(img.shape)} does not match the model '
                                 f'([B, {self.in_chans}, {self.img_size}, {self.img_size}])')
        if not img.is_cuda:
            raise RuntimeError('the HIP ViT-ED runs on MI355X only: got a CPU tensor (no CPU fallback exists)')

    # -- the reference's forward surface -----------------------------------------------------
    def forward_first_part(self, x1):
        self._check_images(x1)
        return F_.EncoderFn.apply(self.runtime(), x1, *self._encoder_params())

    supports_x2_index = True   # engine.pairwise_similarity gathers image-2 rows inside the patch-embed kernel

    def forward_second_part_head(self, x1_feats, x2, x2_index=None):
        self._check_images(x2)
        if x2_index is not None:
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                raise NotImplementedError('x2_index (gather-in-kernel) is an inference path: call it under torch.no_grad()')
            x2_index = x2_index.to(device=x2.device, dtype=torch.int64).contiguous()
        return F_.DecoderFn.apply(self.runtime(), x1_feats, x2, x2_index, *self._decoder_params())

    # -- pair-cached inference (engine.pairwise_similarity) ---------------------------------
    supports_pair_cache = True

    @torch.no_grad()
    def cache_image2_tokens(self, images):
        """Everything of the decoder that depends on image 2 alone, once per image: prepare_x2 (vision_transformer.py:390-395),
        the first CrossBlock's self-attention branch and its cross-attention queries.  Returns (tokens [n, N2, D] fp32, q0 | None)."""
        self._check_images(images)
        p = self._decoder_params()
        nb = len(DEC_BLOCK_KEYS)
        block0 = p[8: 8 + nb] if self.c_depth else None
        return F_.image2_tokens(self.runtime(), images, p[0], p[1], p[2], p[3], block0)

    @torch.no_grad()
    def cache_context_kv(self, feats):
        """Cross-attention keys / values of every decoder block for a block of image-1 features (:177-179), once per block."""
        rt = self.runtime()
        p = self._decoder_params()
        nb = len(DEC_BLOCK_KEYS)
        return F_.context_kv(rt, feats, [p[8 + l * nb: 8 + (l + 1) * nb] for l in range(self.c_depth)])

    @torch.no_grad()
    def forward_pairs_cached(self, tokens2, j_idx, kvs, i_idx, q0=None):
        """== self(feats[i_idx], images[j_idx]) (hisfrag.py:226-229) from the caches."""
        dev = tokens2.device
        j_idx = j_idx.to(device=dev, dtype=torch.int64).contiguous()
        i_idx = i_idx.to(device=dev, dtype=torch.int64).contiguous()
        return F_.decoder_cached(self.runtime(), tokens2, j_idx, kvs, i_idx, self._decoder_params(), q0)

    def forward(self, x, x2=None, forward_first_part=False, x2_index=None):
        if forward_first_part:
            return self.forward_first_part(x)
        if x2 is not None:
            return self.forward_second_part_head(x, x2, x2_index)
        if x.dim() != 5 or x.shape[1] != 2:
            raise AssertionError(f'expected stacked pairs [B, 2, C, S, S], got {tuple(x.shape)}')
        feats = self.forward_first_part(x[:, 0])      # strided views: the kernels take a batch stride
        return self.forward_second_part_head(feats, x[:, 1])

    def flops_parts(self):
        """Algorithmic forward FLOPs (2MNK per contraction) of (the encoder on ONE image incl. its
        patch embedding, the decoder + head on ONE pair incl. image 2's patch embedding)."""
        d, n1 = self.embed_dim, self.patch_embed.num_patches
        n2 = n1 + 1
        kp = self.in_chans * self.patch_size ** 2
        enc = self.depth * (24 * n1 * d * d + 4 * n1 * n1 * d)
        dec = self.c_depth * (28 * n2 * d * d + 4 * n1 * d * d + 4 * n2 * n2 * d + 4 * n2 * n1 * d)
        return 2 * n1 * kp * d + enc, 2 * n1 * kp * d + dec + 2 * d * self.num_classes

    def flops(self, batch=1):
        """Algorithmic forward FLOPs per pair (one-shot forward on a stacked pair)."""
        enc, dec = self.flops_parts()
        return batch * (enc + dec)
