"""vit-ed_amd: the ViT encoder-decoder hot path of glmanhtu/vit-ed on MI355X (gfx950).

The directory name is not a Python identifier; import it with
``importlib.import_module('vit-ed_amd')`` or through the ``vited_amd`` alias module at the repo
root.  Public surface: ``build_model``, ``get_config`` / ``config_from_yaml``,
``VisionTransformerCustom``, the ``ops`` (functional C-ABI wrappers) and ``torch.ops.vited.*`` (the same
kernels as PyTorch custom operators, ``custom_ops.py``).
"""
from . import _lib, config, custom_ops, engine, functions, ops, optim  # noqa: F401  (custom_ops registers torch.ops.vited.*)
from .build import build_model  # noqa: F401
from .config import config_from_yaml, get_config  # noqa: F401
from .model import VisionTransformerCustom  # noqa: F401

__all__ = ['build_model', 'get_config', 'config_from_yaml', 'VisionTransformerCustom', 'ops', 'functions', 'config']
