import sys, torch
sys.path.insert(0, '.')
import vited_amd as v
from oracle import vited_oracle as vo
dev = torch.device('cuda:0')
torch.manual_seed(0)
s = vo.SHAPE_A
oracle = vo.OracleViTED(s)
B = 8
x = torch.randn(B, 2, 3, 64, 64).clamp(-1, 1)
y = (torch.rand(B, 4) > 0.75).float()
def ograds(autocast):
    oracle.zero_grad()
    with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
        out = oracle(x)
    torch.nn.functional.binary_cross_entropy_with_logits(out.float(), y).backward()
    return out.detach().float(), {n: p.grad.clone() for n, p in oracle.named_parameters()}
o32, g32 = ograds(False)
oac, gac = ograds(True)
def hip(dt):
    m = v.VisionTransformerCustom(img_size=64, patch_size=8, num_classes=4, embed_dim=384, depth=8, c_depth=8, num_heads=12)
    m.compute_dtype = dt
    m = m.to(dev); m.load_state_dict(oracle.state_dict())
    out = m(x.to(dev))
    torch.nn.functional.binary_cross_entropy_with_logits(out, y.to(dev)).backward()
    return out.detach().cpu(), {n: p.grad.cpu() for n, p in m.named_parameters()}
oh32, gh32 = hip(torch.float32)
oh16, gh16 = hip(torch.bfloat16)
def total(g):
    num = sum(float((g[n].double() - g32[n].double()).norm() ** 2) for n in g32)
    den = sum(float(g32[n].double().norm() ** 2) for n in g32)
    return (num / den) ** 0.5
print('logits: hip32 %.2e  hip16 %.2e  torch-autocast %.2e (max abs vs oracle fp32; |logit| max %.3f)' % (
    (oh32 - o32).abs().max(), (oh16 - o32).abs().max(), (oac - o32).abs().max(), o32.abs().max()))
print('global grad err: hip32 %.3e hip16 %.3e torch-autocast %.3e' % (total(gh32), total(gh16), total(gac)))
worst = sorted(((float((gh16[n] - g32[n]).norm() / g32[n].norm()), float((gac[n] - g32[n]).norm() / g32[n].norm()), n) for n in g32), reverse=True)[:8]
for e, ea, n in worst: print('  %-45s hip16 %.3e autocast %.3e' % (n, e, ea))
