"""Patch-pair assembly (data/datasets/div2k_patch.py:108-162): the CPU restatement against Pillow-made fixtures, the batch-level
plan against the restated per-sample logic, and (GPU) the ``vited_crop_pairs_u8`` kernel bit for bit against both."""
import os

import numpy as np
import pytest
import torch

from oracle import pair_crops as pc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pair_crops.npz')


def _cases():
    fx = np.load(GOLDEN)
    i = 0
    while f'meta{i}' in fx:
        size, e, c1, c2 = [int(v) for v in fx[f'meta{i}']]
        yield size, e, (c1, c2), fx[f'region{i}'], fx[f'pair{i}']
        i += 1


def test_restatement_matches_the_pillow_fixtures():
    n = 0
    for size, e, cells, region, want in _cases():
        assert np.array_equal(pc.assemble_pair(region, cells, e, size), want)
        n += 1
    assert n >= 4
    # torchvision's center_crop offset rounds half to even; the erosion size is a ceiling
    assert [pc.center_crop_offset(64, e) for e in (64, 61, 59, 58)] == [0, 2, 2, 3]
    assert pc.erosion_size(64, 0.07) == 60 and pc.erosion_size(64, 0.14) == 56


def test_restatement_matches_pillow_itself_when_installed():
    pytest.importorskip('PIL')
    assert pc.pin(write=False) >= 5


class _Feed:
    def __init__(self, vals):
        self.vals = list(vals)

    def random(self):
        return self.vals.pop(0)


def test_batch_plan_equals_the_per_sample_logic(vited):
    """engine.div2k_pair_plan (vectorised) == the reference's per-sample draws (div2k_patch.py:121-153) fed the same numbers."""
    g = torch.Generator().manual_seed(3)
    u = torch.rand(4096, 4, generator=g)
    cells, labels, erode = vited.engine.div2k_pair_plan(u, 64, 0.07, with_negative=True, train=True)
    for k in range(0, 4096, 7):
        (c1, c2), lab = pc.choose_pair(_Feed(u[k, :3].tolist()), with_negative=True)
        assert (int(cells[k, 0]), int(cells[k, 1])) == (c1, c2) and labels[k].tolist() == lab, k
        assert int(erode[k]) == pc.erosion_size(64, 0.07 * (1.0 + float(u[k, 3].double())))
    assert 0.27 < float((labels.sum(1) == 0).float().mean()) < 0.33            # 30 % negatives
    assert set(cells[:, 0].tolist()) <= {0, 1, 2, 3, 4} and int(erode.min()) >= pc.erosion_size(64, 0.14)
    # evaluation: fixed erosion; without negatives the first draw is never consulted
    cells_e, labels_e, erode_e = vited.engine.div2k_pair_plan(u, 64, 0.07, with_negative=False, train=False)
    assert bool((labels_e.sum(1) == 1).all()) and set(erode_e.tolist()) == {60}
    for k in range(0, 512, 5):
        (c1, c2), lab = pc.choose_pair(_Feed(u[k, 1:3].tolist()), with_negative=False)
        assert (int(cells_e[k, 0]), int(cells_e[k, 1])) == (c1, c2) and labels_e[k].tolist() == lab


@pytest.mark.gpu
def test_crop_pairs_kernel_is_bit_exact(vited, gpu):
    for size, e, cells, region, want in _cases():
        r = torch.from_numpy(region).unsqueeze(0).to(gpu)
        got = vited.ops.crop_pairs_u8(r, torch.tensor([cells], dtype=torch.int32, device=gpu), torch.tensor([e], dtype=torch.int32, device=gpu), size)
        assert torch.equal(got[0].cpu(), torch.from_numpy(want)), (size, e, cells)
    # a whole batch with per-sample cells / erosion, strided regions, against the restatement
    g = torch.Generator().manual_seed(5)
    B, S = 33, 64
    big = torch.randint(0, 256, (B, 2, 3, 2 * S, 3 * S), generator=g, dtype=torch.uint8)
    regions = big[:, 1]                                          # batch stride = 2 regions
    u = torch.rand(B, 4, generator=g)
    cells, labels, erode = vited.engine.div2k_pair_plan(u.to(gpu), S, 0.07)
    got = vited.engine.assemble_pairs(regions.to(gpu)[:], cells, erode, S).cpu().numpy()
    for k in range(B):
        want = pc.assemble_pair(regions[k].numpy(), cells[k].tolist(), int(erode[k]), S)
        assert np.array_equal(got[k], want), k
    # and it feeds the model: uint8 pairs in, logits out
    from oracle import vited_oracle as vo
    s = vo.ViTEDShape(depth=1, c_depth=1)
    model = vited.VisionTransformerCustom(img_size=64, patch_size=8, num_classes=4, embed_dim=384, depth=1, c_depth=1, num_heads=12).to(gpu)
    model.compute_dtype = torch.float32
    oracle = vo.OracleViTED(s)
    oracle.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    pairs = torch.from_numpy(got[:4]).to(gpu)
    with torch.no_grad():
        lh = model(pairs).cpu()
        lo = oracle((torch.from_numpy(got[:4]).float() / 255.0 - 0.5) / 0.5)
    torch.testing.assert_close(lh, lo, rtol=1e-3, atol=1e-5)
