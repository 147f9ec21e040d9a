"""ctypes binding of libvited_hip.so (the C ABI declared in include/vited.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent,
importing the ops raises, and every op raises ``RuntimeError`` for tensors that are not on a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VITED_LIB') or os.path.join(_HERE, 'libvited_hip.so')  # VITED_LIB: kernel-experiment builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'vited.h')

F32, BF16 = 0, 1
EPI_STORE, EPI_GELU, EPI_RESIDUAL, EPI_MUL_GELU_GRAD, EPI_STORE_F32, EPI_MUL, EPI_GELU_GRAD = 0, 1, 2, 3, 4, 5, 6
B_NK, B_KN = 0, 1

_p, _i64, _i, _f = C.c_void_p, C.c_int64, C.c_int, C.c_float

# name -> (restype, argtypes); mirrors include/vited.h one to one (tests/test_abi.py checks that the
# header declares exactly these names and that the library exports every one of them)
SIGNATURES = {
    'vited_abi_version': (_i, []),
    'vited_strerror': (C.c_char_p, [_i]),
    'vited_last_gemm_path': (_i, []),
    'vited_last_attention_path': (_i, []),
    'vited_cast': (_i, [_p, _i, _p, _i, _i64, _p]),
    'vited_cast_transpose': (_i, [_p, _p, _i, _i64, _i64, _p]),
    'vited_cast_weights': (_i, [_p, _i, _i64, _p]),
    'vited_patchify': (_i, [_p, _i64, _p, _p, _i, _i64, _i, _i, _i, _p]),
    'vited_patchify_u8': (_i, [_p, _i64, _p, _p, _i, _i64, _i, _i, _i, _p, _p, _p]),
    'vited_crop_pairs_u8': (_i, [_p, _i64, _p, _p, _p, _i64, _i, _i, _p]),
    'vited_slice_rows_cast': (_i, [_p, _p, _i, _i64, _i64, _i64, _i64, _i64, _p]),
    'vited_write_cls_row': (_i, [_p, _p, _p, _i64, _i64, _i64, _p]),
    'vited_sum_rows_workspace_bytes': (_i64, [_i64, _i64]),
    'vited_sum_rows': (_i, [_p, _i, _i64, _p, _i64, _i64, _p, _i64, _p]),
    'vited_layernorm_fwd': (_i, [_p, _i64, _p, _p, _p, _i, _i64, _p, _p, _i64, _i64, _f, _p]),
    'vited_layernorm_bwd_workspace_bytes': (_i64, [_i64, _i64]),
    'vited_layernorm_bwd': (_i, [_p, _i, _i64, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _p, _i, _i64, _p, _p, _i,
                                 _i64, _i64, _p, _i64, _p]),
    'vited_gemm': (_i, [_p, _i64, _p, _i64, _i, _i, _i64, _i64, _i64, _i, _p, _p, _p, _p, _p, _i64, _i64, _i64,
                        _i64, _i, _p]),
    'vited_linear_bwd_weight_workspace_bytes': (_i64, [_i64, _i64, _i64]),
    'vited_linear_bwd_weight': (_i, [_p, _i64, _p, _i64, _i, _i64, _i64, _i64, _p, _p, _i, _p, _i64, _p]),
    'vited_linear_bwd_weight_batched_supported': (_i, [_i, _p, _p, _p, _i]),
    'vited_linear_bwd_weight_batched_workspace_bytes': (_i64, [_i, _p, _p, _p]),
    'vited_linear_bwd_weight_batched': (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _i64, _p]),
    'vited_linear_layernorm_supported': (_i, [_i64, _i64, _i64]),
    'vited_linear_residual_layernorm_fwd': (_i, [_p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _p, _f, _p, _i64, _p, _p, _i64, _i64, _i64, _p]),
    'vited_linear_layernorm_bwd_segmented': (_i, [_p, _i64, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64, _p, _p,
                                                  _i, _i64, _i64, _p, _i64, _p]),
    'vited_linear_layernorm_bwd_partial_rows': (_i64, [_i64]),
    'vited_layernorm_bwd_finish_batched': (_i, [_i, _p, _p, _p, _p, _p, _i64, _p]),
    'vited_linear_layernorm_bwd_workspace_bytes': (_i64, [_i64, _i64]),
    'vited_linear_layernorm_bwd': (_i, [_p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i, _i64, _i64, _i64,
                                        _p, _i64, _p]),
    'vited_fold_context_weights': (_i, [_i, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p]),
    'vited_unfold_context_grads': (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i, _p]),
    'vited_mlp_fwd': (_i, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f, _p]),
    'vited_block_workspace_bytes': (_i64, [_i64, _i64, _i64, _i64, _i]),
    'vited_block_fwd': (_i, [_p, _p, _i64, _i64, _i64, _i, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i64, _p]),
    'vited_cross_block_workspace_bytes': (_i64, [_i64, _i64, _i64, _i64, _i64, _i]),
    'vited_cross_block_fwd': (_i, [_p, _p, _p, _i64, _i64, _i64, _i64, _i, _i64] + [_p] * 22 + [_f, _p, _i64, _p]),
    'vited_adamw_workspace_bytes': (_i64, []),
    'vited_adamw_step': (_i, [_p, _i, _i64, _p, _i64, _p, _f, _i, _p, _p, _i64, _p]),
    'vited_attention_fwd': (_i, [_p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _i, _i64, _i,
                                 _i64, _i64, _i, _f, _p]),
    'vited_attention_fwd_indexed': (_i, [_p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _i64, _i64, _p, _i, _i64, _i,
                                         _i64, _i64, _i, _f, _p]),
    'vited_attention_bwd': (_i, [_p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _i64, _i64, _p, _p,
                                 _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _i, _i64, _i, _i64, _i64, _i, _f, _p]),
}

_lib = None


class VitedLibraryError(RuntimeError):
    pass


def header_declared_functions():
    """Names of every function include/vited.h declares (used by the ABI test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vited_[a-z0-9_]+)\s*\(', text)))


def load():
    """Load the shared library (once) and attach the signatures.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VitedLibraryError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C vit-ed_amd/csrc`.  There is no CPU or PyTorch fallback for the ViT-ED hot path.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VitedLibraryError(f'{LIB_PATH} does not export {name} (stale build?)') from e
        fn.restype = res
        fn.argtypes = args
    if lib.vited_abi_version() != 1:
        raise VitedLibraryError(f'ABI version mismatch: library {lib.vited_abi_version()}, binding 1')
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().vited_strerror(code).decode()
        raise RuntimeError(f'{what} failed: {msg} (vited error {code})')
