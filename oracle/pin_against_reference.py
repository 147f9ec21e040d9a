#!/usr/bin/env python3
"""Pin the oracle against the reference's OWN classes and (re)generate tests/golden/*.npz.

TEST INFRASTRUCTURE.  Runs ONLY in the build container, where /root/reference exists; it exits
cleanly (code 0, message) anywhere else.  Nothing from /root/reference is copied: the reference's
``models/vision_transformer.py`` is loaded *by file path* (bypassing ``models/__init__`` which
needs torchvision) with ``oracle/_timm_standin`` on ``sys.path`` to supply the five timm names
that are absent from this image.  The fixtures written are DATA: logits,
feature slices, per-parameter gradient norms and a few gradient slices produced by the reference's
classes on closed-form inputs/weights (``vited_oracle.closed_form``) - inputs and weights are
re-derivable from the closed form, so only outputs are stored.

    python oracle/pin_against_reference.py            # verify + rewrite fixtures
    python oracle/pin_against_reference.py --check    # verify only
"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF_FILE = '/root/reference/models/vision_transformer.py'
GOLDEN = os.path.join(REPO, 'tests', 'golden')

sys.path.insert(0, REPO)
from oracle import vited_oracle as vo  # noqa: E402

CASES = {
    # name: (shape, batch, with_backward)
    'T': (vo.SHAPE_T, 3, True),
    'A_1x1': (vo.ViTEDShape(depth=1, c_depth=1), 4, True),
    'A_2x2': (vo.ViTEDShape(depth=2, c_depth=2), 3, True),
    'A_full': (vo.SHAPE_A, 2, True),
    'H_1x1_128': (vo.ViTEDShape(img_size=128, patch_size=16, num_classes=1, num_heads=6, depth=1, c_depth=1), 2, True),
    'H_full': (vo.SHAPE_H, 1, False),
    # config H's token counts (1024 / 1025: the flash attention forward AND backward kernels) at 1 + 1 blocks, with backward
    'H_1x1_512': (vo.ViTEDShape(img_size=512, patch_size=16, num_classes=1, num_heads=6, depth=1, c_depth=1), 2, True),
    # ... and at 4 + 4 blocks with backward: gradient flow through stacked flash-attention blocks, d(features) accumulated over
    # four cross-attention blocks, the cls-only last decoder block behind three full ones
    'H_4x4_512': (vo.ViTEDShape(img_size=512, patch_size=16, num_classes=1, num_heads=6, depth=4, c_depth=4), 2, True),
}


def load_reference_module():
    sys.path.insert(0, os.path.join(HERE, '_timm_standin'))
    spec = importlib.util.spec_from_file_location('_ref_vision_transformer', REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_reference(ref, s: vo.ViTEDShape):
    # the kwargs models/build.py:19-32 forwards for MODEL.TYPE == 'pjs'
    return ref.VisionTransformerCustom(
        img_size=s.img_size, patch_size=s.patch_size, in_chans=s.in_chans, num_classes=s.num_classes,
        embed_dim=s.embed_dim, depth=s.depth, c_depth=s.c_depth, num_heads=s.num_heads,
        mlp_ratio=s.mlp_ratio, qkv_bias=s.qkv_bias, keep_attn=False, arch_version='v1')


def targets_for(batch, s):
    return (vo.closed_form((batch, s.num_classes), 77, 1.0) > 0.2).float()


def run(model, x, y, with_backward):
    out = {}
    model.zero_grad(set_to_none=True)
    x1, x2 = torch.unbind(x, 1)
    with torch.set_grad_enabled(with_backward):
        feats = model(x1, forward_first_part=True)
        logits = model(x)
        two_stage = model(feats, x2)
        out['logits'] = logits.detach()
        out['two_stage_logits'] = two_stage.detach()
        out['feats'] = feats.detach()
        if with_backward:
            loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, y)
            loss.backward()
            out['loss'] = loss.detach()
            out['grads'] = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    return out


def rel_err(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--check', action='store_true')
    args = ap.parse_args()
    if not os.path.exists(REF_FILE):
        print('reference not present here - nothing to pin (fixtures in tests/golden stay as committed)')
        return 0
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = load_reference_module()
    os.makedirs(GOLDEN, exist_ok=True)
    worst = 0.0
    for name, (s, batch, bwd) in CASES.items():
        rm = vo.fill_closed_form_(build_reference(ref, s)).eval()
        om = vo.fill_closed_form_(vo.OracleViTED(s)).eval()
        assert list(rm.state_dict().keys()) == list(om.state_dict().keys()), 'state_dict key order differs'
        for (k, a), (_, b) in zip(rm.state_dict().items(), om.state_dict().items()):
            assert a.shape == b.shape and torch.equal(a, b), k
        x = vo.closed_form_pairs(batch, s)
        y = targets_for(batch, s)
        r = run(rm, x, y, bwd)
        o = run(om, x, y, bwd)
        # structural invariants the reference's only test relies on (tests/hisfrag_evaluation_test.py:143)
        assert rel_err(r['two_stage_logits'], r['logits']) < 1e-6
        # train() == eval() (no live stochastic op)
        with torch.no_grad():
            assert torch.equal(rm.train()(x), rm.eval()(x))
        errs = {'logits': rel_err(o['logits'], r['logits']), 'feats': rel_err(o['feats'], r['feats'])}
        if bwd:
            errs['loss'] = rel_err(o['loss'], r['loss'])
            errs['grad'] = max(rel_err(o['grads'][n], g) for n, g in r['grads'].items())
        # fp64 run of BOTH: algorithmic identity shows as ~1e-13 agreement, which separates a real
        # difference from fp32 summation-order noise (SDPA vs explicit softmax, 16 blocks deep)
        r64 = run(rm.double(), x.double(), y.double(), bwd)
        o64 = run(om.double(), x.double(), y.double(), bwd)
        errs['logits64'] = rel_err(o64['logits'], r64['logits'])
        if bwd:
            errs['grad64'] = max(rel_err(o64['grads'][n], g) for n, g in r64['grads'].items())
        worst = max(worst, errs['logits64'], errs.get('grad64', 0.0))
        print(f'{name:10s} B={batch} ' + ' '.join(f'{k}={v:.2e}' for k, v in errs.items()))
        assert max(errs['logits64'], errs.get('grad64', 0.0)) < 1e-10, f'oracle != reference (fp64) on {name}: {errs}'
        assert max(errs['logits'], errs['feats'], errs.get('loss', 0.0)) < 2e-5, f'fp32 drift on {name}: {errs}'
        assert errs.get('grad', 0.0) < 2e-3, f'fp32 grad drift on {name}: {errs}'
        if not args.check:
            blob = {
                'shape': np.array([s.img_size, s.patch_size, s.in_chans, s.num_classes, s.embed_dim, s.depth,
                                   s.c_depth, s.num_heads], dtype=np.int64),
                'batch': np.array(batch),
                'logits': r['logits'].numpy(),
                'logits_f64': r64['logits'].numpy(),
                'feats_slice': r['feats'][:, :4, :].numpy().copy(),
                'feats_abs_mean': r['feats'].abs().mean().numpy(),
            }
            if bwd:
                names = list(r['grads'].keys())
                blob['loss'] = r['loss'].numpy()
                blob['grad_names'] = np.array(names)
                blob['grad_norms'] = np.array([float(r['grads'][n].norm()) for n in names], dtype=np.float64)
                blob['grad_head_weight'] = r['grads']['head.weight'].numpy()
                blob['grad_pos_embed_slice'] = r['grads']['pos_embed'][0, :3, :16].numpy().copy()
                blob['grad_cls_token'] = r['grads']['cls_token'].numpy()
                blob['grad_qkv0_slice'] = r['grads']['blocks.0.attn.qkv.weight'][:8, :8].numpy().copy()
                blob['grad_kv0_slice'] = r['grads']['cross_blocks.0.cross_attn.kv.weight'][:8, :8].numpy().copy()
            np.savez_compressed(os.path.join(GOLDEN, f'vited_{name}.npz'), **blob)
    print(f'oracle == reference classes on all cases (worst fp64 rel err {worst:.2e})')
    return 0


if __name__ == '__main__':
    sys.exit(main())
