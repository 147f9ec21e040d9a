import sys, torch
sys.path.insert(0, '.')
import vited_amd as v
from oracle import vited_oracle as vo
dev = torch.device('cuda:0')
s = vo.SHAPE_T
ops = v.ops
orig = ops.linear_bwd_weight
def checked(dy, x, want_bias=True):
    dw, db = orig(dy, x, want_bias)
    ref = dy.double().t() @ x.double()
    err = (dw.double() - ref).norm() / ref.norm()
    print('bwd_weight', tuple(dy.shape), tuple(x.shape), dy.dtype, 'path', ops.last_paths()[0], 'rel err %.3e' % err.item(),
          'ptr%16', dy.data_ptr() % 16, x.data_ptr() % 16, 'strides', dy.stride(), x.stride(), '|ref|', ref.norm().item())
    return dw, db
ops.linear_bwd_weight = checked
v.functions.ops.linear_bwd_weight = checked
m = v.VisionTransformerCustom(img_size=s.img_size, patch_size=s.patch_size, num_classes=1, embed_dim=32, depth=1, c_depth=1, num_heads=1)
m.compute_dtype = torch.bfloat16
m = vo.fill_closed_form_(m.to(dev))
x = vo.closed_form_pairs(3, s).to(dev)
y = (vo.closed_form((3, 1), 77, 1.0) > 0.2).float().to(dev)
out = m(x)
torch.nn.functional.binary_cross_entropy_with_logits(out, y).backward()
