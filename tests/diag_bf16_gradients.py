#!/usr/bin/env python3
"""Where does the bf16 MFMA path lose gradient accuracy?  

Per block, the relative error against the fp32 CPU oracle of (a) the block's output in forward and (b) the gradient with
respect to the block's input in backward, for two bf16 implementations of the same network on the same inputs:
the HIP path (taps in functions.py) and PyTorch's own CPU bf16 autocast of the oracle.  Run on the GPU box:

    python3 tests/diag_bf16_gradients.py [A_full|A_2x2|A_1x1|T|rand8]

Test infrastructure (imports the oracle); tests/test_gpu_model.py::test_bf16_error_growth_per_block asserts the bounds
this script established.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vited_oracle as vo  # noqa: E402

CASES = {'T': (vo.SHAPE_T, 3), 'A_1x1': (vo.ViTEDShape(depth=1, c_depth=1), 4), 'A_2x2': (vo.ViTEDShape(depth=2, c_depth=2), 3),
         'A_full': (vo.SHAPE_A, 2), 'rand8': (vo.SHAPE_A, 8),
         'H_1x1_512': (vo.ViTEDShape(img_size=512, patch_size=16, num_classes=1, num_heads=6, depth=1, c_depth=1), 2),
         'H_4x4_512': (vo.ViTEDShape(img_size=512, patch_size=16, num_classes=1, num_heads=6, depth=4, c_depth=4), 2)}


def oracle_taps(m, x, y, autocast):
    """logits, per-parameter grads and per-block taps of the oracle (fp32 or CPU bf16 autocast)."""
    s = m.shape
    m.zero_grad(set_to_none=True)
    taps = {}
    with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
        x1, x2 = torch.unbind(x, 1)
        t = (m._patch_tokens(x1) + m.pos_embed[:, 1:]).float()
        enc_in = []
        for i, blk in enumerate(m.blocks):
            t.retain_grad()
            enc_in.append(t)
            t = vo.encoder_block(blk, t, s.num_heads)
            taps[f'enc.x.{i}'] = t.detach().float()
        feats = t
        u = m.prepare_x2(x2).float()
        dec_in = []
        cross = []
        for i, blk in enumerate(m.cross_blocks):
            u.retain_grad()
            dec_in.append(u)
            # vo.decoder_block, opened up at the cross attention (vision_transformer.py:174-200) to tap q, kv, o and their gradients
            u1 = u + vo.self_attention(blk.attn, vo._ln(blk.norm1, u), s.num_heads)
            ca = blk.cross_attn
            qq = torch.nn.functional.linear(vo._ln(blk.norm_cross, u1), ca.q.weight, ca.q.bias)
            kvv = torch.nn.functional.linear(vo._ln(blk.norm_context, feats), ca.kv.weight, ca.kv.bias)
            kk, vv = kvv.chunk(2, dim=-1)
            oo = vo._sdpa(vo._heads(qq, s.num_heads), vo._heads(kk, s.num_heads), vo._heads(vv, s.num_heads))
            for t_ in (qq, kvv, oo):
                t_.retain_grad()
            cross.append((qq, kvv, oo))
            u2 = u1 + torch.nn.functional.linear(oo, ca.proj.weight, ca.proj.bias)
            u = u2 + vo._mlp(blk.mlp, vo._ln(blk.norm2, u2))
            taps[f'dec.x.{i}'] = u.detach().float()
        logits = m.forward_head(vo._ln(m.norm, u))
    torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), y).backward()
    for i, t_ in enumerate(enc_in):
        taps[f'enc.dx.{i}'] = t_.grad.detach().float()
    for i, t_ in enumerate(dec_in):
        taps[f'dec.dx.{i}'] = t_.grad.detach().float()
    for i, (qq, kvv, oo) in enumerate(cross):
        taps[f'dec.q.{i}'], taps[f'dec.kv.{i}'], taps[f'dec.oc.{i}'] = qq.detach().float(), kvv.detach().float(), oo.detach().float()
        taps[f'dec.dq.{i}'], taps[f'dec.dkv.{i}'], taps[f'dec.doc.{i}'] = qq.grad.float(), kvv.grad.float(), oo.grad.float()
    return logits.detach().float(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}, taps


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def like(tap, ref, batch):
    """The last decoder block of the HIP path runs on the cls rows only (functions._dec_block_fwd): compare those rows."""
    if tap.numel() == ref.numel():
        return ref.reshape(-1)
    return ref.reshape(batch, -1, ref.shape[-1])[:, 0, :].reshape(-1)


def anatomy(case, dev):
    import vited_amd as V
    s, batch = CASES[case]
    if case == 'rand8':
        torch.manual_seed(0)
        om = vo.OracleViTED(s)
        x = torch.randn(batch, 2, 3, s.img_size, s.img_size).clamp(-1, 1)
        y = (torch.rand(batch, s.num_classes) > 0.75).float()
    else:
        om = vo.fill_closed_form_(vo.OracleViTED(s))
        x = vo.closed_form_pairs(batch, s)
        y = (vo.closed_form((batch, s.num_classes), 77, 1.0) > 0.2).float()
    l32, g32, t32 = oracle_taps(om, x, y, False)
    lac, gac, tac = oracle_taps(om, x, y, True)
    hm = V.VisionTransformerCustom(img_size=s.img_size, patch_size=s.patch_size, in_chans=s.in_chans, num_classes=s.num_classes,
                                   embed_dim=s.embed_dim, depth=s.depth, c_depth=s.c_depth, num_heads=s.num_heads).to(dev)
    hm.load_state_dict(om.state_dict())
    hm.compute_dtype = torch.bfloat16
    rt = hm.runtime()
    rt.tap = {}
    lh = hm(x.to(dev))
    torch.nn.functional.binary_cross_entropy_with_logits(lh, y.to(dev)).backward()
    th = {k: v.float().cpu() for k, v in rt.tap.items()}
    rt.tap = None
    gh = {n: p.grad.detach().cpu() for n, p in hm.named_parameters()}
    rows = []
    for kind in ('enc', 'dec'):
        n_blk = s.depth if kind == 'enc' else s.c_depth
        for i in range(n_blk):
            fx, fd = f'{kind}.x.{i}', f'{kind}.dx.{i}'
            ref_x, ref_d = like(th[fx], t32[fx], batch), like(th[fd], t32[fd], batch)
            rows.append((f'{kind}{i}', rel(th[fx].reshape(-1), ref_x), rel(like(th[fx], tac[fx], batch), ref_x),
                         rel(th[fd].reshape(-1), ref_d), rel(like(th[fd], tac[fd], batch), ref_d)))
    cross_rows = []
    for i in range(s.c_depth):
        cross_rows.append((i,) + tuple(v for k in ('q', 'kv', 'oc', 'doc', 'dq', 'dkv')
                                       for v in (rel(th[f'dec.{k}.{i}'].reshape(-1), like(th[f'dec.{k}.{i}'], t32[f'dec.{k}.{i}'], batch)),
                                                 rel(like(th[f'dec.{k}.{i}'], tac[f'dec.{k}.{i}'], batch), like(th[f'dec.{k}.{i}'], t32[f'dec.{k}.{i}'], batch)))))
    tot = lambda g: (sum(float((g[n].double() - g32[n].double()).norm() ** 2) for n in g32) / sum(float(g32[n].double().norm() ** 2) for n in g32)) ** 0.5
    return {'rows': rows, 'cross': cross_rows, 'logits': (rel(lh.detach().cpu(), l32), rel(lac, l32)), 'grads': (tot(gh), tot(gac)),
            'per_param': {n: (rel(gh[n], g32[n]), rel(gac[n], g32[n]), float(g32[n].norm())) for n in g32}}


def main():
    cases = sys.argv[1:] or ['A_full']
    dev = torch.device('cuda:0')
    for case in cases:
        r = anatomy(case, dev)
        print(f'== {case}: relative error vs the fp32 oracle (HIP bf16 | CPU bf16 autocast)')
        print(f'   logits {r["logits"][0]:.3e} | {r["logits"][1]:.3e}    all parameter gradients {r["grads"][0]:.3e} | {r["grads"][1]:.3e}')
        print('   block   fwd out: hip      autocast    d(input): hip     autocast')
        for name, fh, fa, dh, da in r['rows']:
            print(f'   {name:6s}  {fh:.3e}  {fa:.3e}     {dh:.3e}  {da:.3e}')
        print('   cross attention (hip|autocast): q            kv           o            dO           dq           dkv')
        for row in r['cross']:
            print(f'   dec{row[0]}   ' + '  '.join(f'{row[1 + 2 * j]:.1e}|{row[2 + 2 * j]:.1e}' for j in range(6)))
        worst = sorted(r['per_param'].items(), key=lambda kv: -kv[1][0] / (kv[1][1] + 1e-3))[:8]
        print('   worst parameters by hip/autocast error ratio: name, hip, autocast, |g|')
        for n, (eh, ea, gn) in worst:
            print(f'     {n:44s} {eh:.3e} {ea:.3e} {gn:.3e}')


if __name__ == '__main__':
    main()
