"""Config surface of the reference, re-implemented without yacs (not installed here).

Keeps what callers of the hot path rely on (config.py:12-336 of the reference): the same key tree
and defaults, YAML files with ``BASE`` parents, the argparse overrides ``main.py`` / ``hisfrag.py``
pass, ``--opts KEY VALUE ...`` last, ``OUTPUT = <output>/<MODEL.NAME>/<TAG>``, and a frozen
attribute-style node (``config.MODEL.PJS.EMBED_DIM``) with ``defrost() / freeze() / clone() /
dump() / merge_from_file() / merge_from_list()``.  The shipped YAMLs under ``configs/`` parse to
the same values as with the reference's loader (tests/test_config_build.py).
"""
from __future__ import annotations

import ast
import copy
import os

import yaml


class CfgNode(dict):
    """Attribute-access dict with the subset of the yacs API the reference uses."""

    _FROZEN = '__frozen__'

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(f'Attempted to set {name} to {value}, but CfgNode is immutable')
        self[name] = value

    def is_frozen(self):
        return object.__getattribute__(self, CfgNode._FROZEN)

    def _set_frozen(self, flag):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        object.__setattr__(out, CfgNode._FROZEN, self.is_frozen())
        return out

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kw):
        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            return list(v) if isinstance(v, tuple) else v
        return yaml.safe_dump(plain(self.to_dict()), **kw)

    # -- merging -----------------------------------------------------------------------------
    def _merge(self, other: dict, path=()):
        for k, v in other.items():
            where = '.'.join(path + (k,))
            if k not in self:
                raise KeyError(f'Non-existent config key: {where}')
            cur = self[k]
            if isinstance(cur, CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f'{where}: expected a mapping')
                cur._merge(v, path + (k,))
            else:
                dict.__setitem__(self, k, _coerce(v, cur, where))

    def merge_from_file(self, cfg_file):
        with open(cfg_file) as f:
            loaded = yaml.safe_load(f) or {}
        self._merge(loaded)

    def merge_from_list(self, opts):
        if len(opts) % 2:
            raise ValueError(f'Override list has odd length: {opts}; it must be a list of pairs')
        for key, raw in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f'Non-existent config key: {key}')
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f'Non-existent config key: {key}')
            dict.__setitem__(node, parts[-1], _coerce(_literal(raw), node[parts[-1]], key))


def _literal(v):
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, cur, where):
    """yacs-style type check: same type, or int->float, list<->tuple, anything over None."""
    if cur is None or new is None or type(new) is type(cur):
        return new
    if isinstance(cur, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(cur, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(cur, list) and isinstance(new, tuple):
        return list(new)
    raise ValueError(f'Type mismatch ({type(cur)} vs. {type(new)}) with values ({cur} vs. {new}) for config key: {where}')


def default_tree() -> dict:
    """Key tree + defaults of the reference (config.py:12-238)."""
    return {
        'BASE': [''],
        'DATA': dict(BATCH_SIZE=128, TEST_BATCH_SIZE=128, DATA_PATH='', DATASET='imagenet', IMG_SIZE=224,
                     INTERPOLATION='bicubic', ZIP_MODE=False, CACHE_MODE='part', PIN_MEMORY=True, NUM_WORKERS=8,
                     EROSION_RATIO=0.07, EVAL_N_ITEMS_PER_CATEGORY=5),
        'MODEL': dict(
            TYPE='pjs', NAME='div2k_erosion7_4bin_patch8_64', PRETRAINED='', RESUME='', NUM_CLASSES=1, DROP_RATE=0.0,
            DROP_PATH_RATE=0.1, LABEL_SMOOTHING=0.1,
            PJS=dict(PATCH_SIZE=16, IN_CHANS=3, EMBED_DIM=768, DEPTH=8, C_DEPTH=8, NUM_HEADS=12, MLP_RATIO=4.,
                     QKV_BIAS=True, QK_SCALE=None, KEEP_ATTN=False, ARCH_VERSION='v1'),
            VIT=dict(PATCH_SIZE=16, IN_CHANS=3, EMBED_DIM=768, DEPTH=12, NUM_HEADS=12, MLP_RATIO=4., QKV_BIAS=True,
                     QK_SCALE=None),
            SS=dict(ARCH='resnet34', PRETRAINED='', EMBED_DIM=2048, PRED_DIM=512, DROPOUT=0., N_CLASSES=0),
            RES=dict(ARCH='resnet18', PRETRAINED='', LAYERS_FREEZE=-1),
            MIXCONV=dict(ARCH='resnet18', PRETRAINED='', MIX_DEPTH=4, OUT_ROWS=1, OUT_CHANNELS=512, LAYERS_FREEZE=-1)),
        'PCA': dict(DIM=256),
        'TRAIN': dict(
            START_EPOCH=0, EPOCHS=300, WARMUP_EPOCHS=20, WEIGHT_DECAY=0.05, BASE_LR=1e-4, WARMUP_LR=5e-7, MIN_LR=5e-6,
            CLIP_GRAD=5.0, AUTO_RESUME=True, ACCUMULATION_STEPS=1, USE_CHECKPOINT=False, LOAD_LR_SCHEDULER=True,
            LR_SCHEDULER=dict(NAME='cosine', DECAY_EPOCHS=30, DECAY_RATE=0.1, WARMUP_PREFIX=True, GAMMA=0.1, MULTISTEPS=[]),
            OPTIMIZER=dict(NAME='adamw', EPS=1e-8, BETAS=(0.9, 0.999), MOMENTUM=0.9),
            LAYER_DECAY=1.0, MOE=dict(SAVE_MASTER=False)),
        'AUG': dict(COLOR_JITTER=0.4, AUTO_AUGMENT='rand-m9-mstd0.5-inc1', REPROB=0.25, REMODE='pixel', RECOUNT=1,
                    MIXUP=0., CUTMIX=0, CUTMIX_MINMAX=None, MIXUP_PROB=1.0, MIXUP_SWITCH_PROB=0.5, MIXUP_MODE='batch'),
        'TEST': dict(CROP=True, SEQUENTIAL=False, SHUFFLE=False),
        'ENABLE_AMP': False, 'AMP_ENABLE': True, 'AMP_OPT_LEVEL': '', 'OUTPUT': '', 'TAG': 'default', 'SAVE_FREQ': 1,
        'SAVE_TMP_FREQ': 5, 'PRINT_FREQ': 50, 'SEED': 0, 'EVAL_MODE': False, 'THROUGHPUT_MODE': False, 'LOCAL_RANK': 0,
        'FUSED_WINDOW_PROCESS': False, 'FUSED_LAYERNORM': False,
    }


_C = CfgNode(default_tree())

# CLI attribute -> (config path, value to store or None for "the argument's own value")
_ARG_MAP = (
    ('batch_size', ('DATA.BATCH_SIZE', 'DATA.TEST_BATCH_SIZE'), None),
    ('eval_n_items_per_category', ('DATA.EVAL_N_ITEMS_PER_CATEGORY',), None),
    ('data_path', ('DATA.DATA_PATH',), None),
    ('zip', ('DATA.ZIP_MODE',), True),
    ('cache_mode', ('DATA.CACHE_MODE',), None),
    ('pretrained', ('MODEL.PRETRAINED',), None),
    ('resume', ('MODEL.RESUME',), None),
    ('keep_attn', ('MODEL.PJS.KEEP_ATTN',), None),
    ('accumulation_steps', ('TRAIN.ACCUMULATION_STEPS',), None),
    ('use_checkpoint', ('TRAIN.USE_CHECKPOINT',), True),
    ('disable_amp', ('AMP_ENABLE',), False),
    ('output', ('OUTPUT',), None),
    ('tag', ('TAG',), None),
    ('eval', ('EVAL_MODE',), True),
    ('throughput', ('THROUGHPUT_MODE',), True),
    ('enable_amp', ('ENABLE_AMP',), None),
    ('fused_window_process', ('FUSED_WINDOW_PROCESS',), True),
    ('fused_layernorm', ('FUSED_LAYERNORM',), True),
    ('optim', ('TRAIN.OPTIMIZER.NAME',), None),
)


def _set_path(cfg, path, value):
    node = cfg
    parts = path.split('.')
    for p in parts[:-1]:
        node = node[p]
    dict.__setitem__(node, parts[-1], value)


def _update_config_from_file(config: CfgNode, cfg_file: str):
    """YAML merge with ``BASE`` parents resolved relative to the including file (config.py:241-253)."""
    config.defrost()
    with open(cfg_file) as f:
        loaded = yaml.safe_load(f) or {}
    for parent in loaded.get('BASE', ['']):
        if parent:
            _update_config_from_file(config, os.path.join(os.path.dirname(cfg_file), parent))
    print(f'=> merge config from {cfg_file}')
    config.defrost()
    config.merge_from_file(cfg_file)
    config.freeze()


def update_config(config: CfgNode, args):
    _update_config_from_file(config, args.cfg)
    config.defrost()
    for attr, paths, fixed in _ARG_MAP:
        val = getattr(args, attr, None)
        if not val:
            continue
        for path in paths:
            _set_path(config, path, val if fixed is None else fixed)
    if getattr(args, 'amp_opt_level', None):
        print('[warning] Apex amp has been deprecated, please use pytorch amp instead!')
        if args.amp_opt_level == 'O0':
            config.AMP_ENABLE = False
    if 'LOCAL_RANK' in os.environ:
        config.LOCAL_RANK = int(os.environ['LOCAL_RANK'])
    config.OUTPUT = os.path.join(config.OUTPUT, config.MODEL.NAME, config.TAG)
    if getattr(args, 'opts', None):
        config.merge_from_list(args.opts)
    config.freeze()


def get_config(args) -> CfgNode:
    """Defaults -> YAML (+BASE) -> CLI arguments -> --opts, frozen (config.py:329-336)."""
    config = _C.clone()
    update_config(config, args)
    return config


def config_from_yaml(cfg_file: str, opts=None) -> CfgNode:
    """Convenience for tests / bench: defaults + one YAML (+ optional KEY VALUE overrides)."""
    class _A:
        pass
    a = _A()
    a.cfg, a.opts = cfg_file, list(opts) if opts else None
    return get_config(a)
