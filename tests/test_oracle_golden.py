"""CPU: the oracle against the golden fixtures (produced by the reference's own classes, see
oracle/pin_against_reference.py) and the structural invariants the reference's only test relies on."""
import os

import numpy as np
import pytest
import torch

from oracle import vited_oracle as vo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['T', 'A_1x1', 'A_2x2', 'H_1x1_128', 'A_full', 'H_1x1_512', 'H_4x4_512']


def _load(name):
    fx = np.load(os.path.join(GOLDEN, f'vited_{name}.npz'))   # allow_pickle stays False
    img, p, c, ncls, d, depth, cdepth, heads = [int(v) for v in fx['shape']]
    s = vo.ViTEDShape(img_size=img, patch_size=p, in_chans=c, num_classes=ncls, embed_dim=d, depth=depth, c_depth=cdepth,
                      num_heads=heads)
    return fx, s, int(fx['batch'])


@pytest.mark.parametrize('name', CASES)
def test_oracle_reproduces_reference_fixture(name):
    fx, s, batch = _load(name)
    torch.set_num_threads(8)
    m = vo.fill_closed_form_(vo.OracleViTED(s))
    x = vo.closed_form_pairs(batch, s)
    y = (vo.closed_form((batch, s.num_classes), 77, 1.0) > 0.2).float()
    feats = m(x[:, 0], forward_first_part=True)
    logits = m(x)
    two_stage = m(feats, x[:, 1])
    np.testing.assert_allclose(logits.detach().numpy(), fx['logits'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(logits.detach().numpy(), fx['logits_f64'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(feats[:, :4].detach().numpy(), fx['feats_slice'], rtol=1e-4, atol=1e-5)
    assert torch.allclose(two_stage, logits, rtol=0, atol=1e-6)   # two-stage == one-shot (tests/hisfrag_evaluation_test.py:143)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, y)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(fx['loss']), rtol=1e-5)
    names = [str(n) for n in fx['grad_names']]
    assert names == [n for n, _ in m.named_parameters()]
    got = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
    np.testing.assert_allclose(got, fx['grad_norms'], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(m.head.weight.grad.numpy(), fx['grad_head_weight'], rtol=1e-3, atol=1e-5 * float(fx['grad_norms'].max()))
    np.testing.assert_allclose(m.cls_token.grad.numpy(), fx['grad_cls_token'], rtol=1e-3, atol=1e-5 * float(fx['grad_norms'].max()))


def test_h_full_forward_fixture():
    fx, s, batch = _load('H_full')
    torch.set_num_threads(8)
    m = vo.fill_closed_form_(vo.OracleViTED(s)).eval()
    with torch.no_grad():
        logits = m(vo.closed_form_pairs(batch, s))
    np.testing.assert_allclose(logits.numpy(), fx['logits'], rtol=5e-5, atol=5e-6)


def test_train_equals_eval_and_param_counts():
    s = vo.ViTEDShape(depth=1, c_depth=1)
    m = vo.fill_closed_form_(vo.OracleViTED(s))
    x = vo.closed_form_pairs(2, s)
    with torch.no_grad():
        assert torch.equal(m.train()(x), m.eval()(x))     # no live stochastic op
    assert sum(p.numel() for p in vo.OracleViTED(vo.SHAPE_A).parameters()) == 33_236_356
    assert sum(p.numel() for p in vo.OracleViTED(vo.SHAPE_H).parameters()) == 50_392_321


def test_reference_style_init_statistics():
    torch.manual_seed(0)
    m = vo.OracleViTED(vo.ViTEDShape(depth=2, c_depth=2))
    assert abs(float(m.blocks[0].attn.qkv.weight.std()) - 0.02) < 2e-3 and float(m.blocks[0].attn.qkv.bias.abs().max()) == 0
    assert float(m.cls_token.abs().max()) < 1e-4 and abs(float(m.pos_embed.std()) - 0.02) < 2e-3
    # decoder Linears are created after timm's init ran: PyTorch default (kaiming-uniform, bound 1/sqrt(fan_in))
    w = m.cross_blocks[0].attn.qkv.weight
    assert float(w.abs().max()) <= 1 / 384 ** 0.5 + 1e-6 and float(w.std()) > 0.025
    assert float(m.cross_blocks[0].attn.qkv.bias.abs().max()) > 0
