import sys, torch
sys.path.insert(0, '.')
from oracle import vited_oracle as vo
for name, s, B in [('T', vo.SHAPE_T, 3), ('A_1x1', vo.ViTEDShape(depth=1, c_depth=1), 4), ('A_2x2', vo.ViTEDShape(depth=2, c_depth=2), 3)]:
    res = {}
    for mode in ('fp32', 'bf16'):
        m = vo.fill_closed_form_(vo.OracleViTED(s))
        x = vo.closed_form_pairs(B, s)
        y = (vo.closed_form((B, s.num_classes), 77, 1.0) > 0.2).float()
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=(mode == 'bf16')):
            out = m(x)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(out.float(), y)
        loss.backward()
        res[mode] = (out.detach().float(), {n: p.grad.clone() for n, p in m.named_parameters()})
    o32, g32 = res['fp32']; o16, g16 = res['bf16']
    rel = {n: ((g32[n] - g16[n]).norm() / g32[n].norm()).item() for n in g32}
    nrel = {n: abs(g32[n].norm() - g16[n].norm()).item() / g32[n].norm().item() for n in g32}
    worst = sorted(rel.items(), key=lambda kv: -kv[1])[:4]
    print(name, 'logit maxabs diff %.3e' % (o32 - o16).abs().max().item(), 'worst grad rel', [(n, round(v, 3)) for n, v in worst],
          'worst norm rel %.3f' % max(nrel.values()), 'patch_w', round(rel['patch_embed.proj.weight'], 3))
