import sys, torch
sys.path.insert(0, '.')
import vited_amd as v
from oracle import vited_oracle as vo
dev = torch.device('cuda:0')
s = vo.SHAPE_T
def run(dtype):
    m = v.VisionTransformerCustom(img_size=s.img_size, patch_size=s.patch_size, num_classes=1, embed_dim=32, depth=1, c_depth=1, num_heads=1)
    m.compute_dtype = dtype
    m = vo.fill_closed_form_(m.to(dev))
    x = vo.closed_form_pairs(3, s).to(dev)
    y = (vo.closed_form((3, 1), 77, 1.0) > 0.2).float().to(dev)
    out = m(x)
    torch.nn.functional.binary_cross_entropy_with_logits(out, y).backward()
    return out.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}
o32, g32 = run(torch.float32)
o16, g16 = run(torch.bfloat16)
print('logits', o32.view(-1).tolist(), o16.view(-1).tolist())
for n in g32:
    a, b = g32[n], g16[n]
    print(f'{n:45s} |g32|={a.norm():.4e} |g16|={b.norm():.4e} rel={(a-b).norm()/a.norm():.3e}')
