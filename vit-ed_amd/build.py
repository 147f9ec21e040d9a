"""``build_model(config)`` - the reference's model factory (models/build.py:15-95), pjs branch.

Reads exactly the keys the reference reads for ``MODEL.TYPE == 'pjs'`` (models/build.py:19-32) and
returns the HIP-backed module.  The other model types (timm ViT, SimSiam, ResNet baselines) are not
part of the ViT-ED hot path and are rejected with a clear message.
"""
from .model import VisionTransformerCustom


def build_model(config, is_pretrain=False):
    model_type = config.MODEL.TYPE
    if model_type != 'pjs':
        raise NotImplementedError(
            f"MODEL.TYPE={model_type!r}: only the 'pjs' ViT encoder-decoder is implemented on MI355X "
            f"(the other families are baselines outside the accelerated path)")
    pjs = config.MODEL.PJS
    return VisionTransformerCustom(
        img_size=config.DATA.IMG_SIZE,
        patch_size=pjs.PATCH_SIZE,
        in_chans=pjs.IN_CHANS,
        num_classes=config.MODEL.NUM_CLASSES,
        embed_dim=pjs.EMBED_DIM,
        depth=pjs.DEPTH,
        c_depth=pjs.C_DEPTH,
        num_heads=pjs.NUM_HEADS,
        mlp_ratio=pjs.MLP_RATIO,
        qkv_bias=pjs.QKV_BIAS,
        keep_attn=pjs.KEEP_ATTN,
        arch_version=pjs.ARCH_VERSION,
    )
