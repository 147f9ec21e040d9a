import sys, torch
sys.path.insert(0, '.')
import vited_amd as v
from oracle import vited_oracle as vo
dev = torch.device('cuda:0')
s = vo.SHAPE_T
ops = v.ops
orig = ops.linear_bwd_weight
calls = []
def checked(dy, x, want_bias=True):
    calls.append((dy.float().clone(), x.float().clone()))
    return orig(dy, x, want_bias)
v.functions.ops.linear_bwd_weight = checked
def run(dt):
    calls.clear()
    m = v.VisionTransformerCustom(img_size=s.img_size, patch_size=s.patch_size, num_classes=1, embed_dim=32, depth=1, c_depth=1, num_heads=1)
    m.compute_dtype = dt
    m = vo.fill_closed_form_(m.to(dev))
    x = vo.closed_form_pairs(3, s).to(dev)
    y = (vo.closed_form((3, 1), 77, 1.0) > 0.2).float().to(dev)
    out = m(x)
    torch.nn.functional.binary_cross_entropy_with_logits(out, y).backward()
    return list(calls)
c32 = run(torch.float32)
c16 = run(torch.bfloat16)
for i, ((dy32, x32), (dy16, x16)) in enumerate(zip(c32, c16)):
    print(i, tuple(dy32.shape), tuple(x32.shape), 'dy rel %.3e' % ((dy32-dy16).norm()/dy32.norm()).item(), 'x rel %.3e' % ((x32-x16).norm()/x32.norm()).item(),
          '|dW32| %.4f |dW16| %.4f' % ((dy32.t()@x32).norm().item(), (dy16.t()@x16).norm().item()))
dy32, x32 = c32[-1]; dy16, x16 = c16[-1]
print('dy32 row norms', dy32.norm(dim=1).tolist())
print('dy16 row norms', dy16.norm(dim=1).tolist())
print('dy32[0,:8]', dy32[0,:8].tolist()); print('dy16[0,:8]', dy16[0,:8].tolist())
# how collinear are rows of patches?
print('x32 gram', (x32 @ x32.t())[:4,:4].tolist())
