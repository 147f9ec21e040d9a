"""CPU restatement of how the reference assembles one DIV2K patch pair (TEST INFRASTRUCTURE - not product code).

Follows data/datasets/div2k_patch.py:108-162 and data/transforms.py:14-18,121-129:

  * a (2 S) x (3 S) region of the augmented image is cut into a grid of 3 columns x 2 rows of S x S cells
    (``transforms.crop(patch, 3, 2)``: row-major, cells 0..5);
  * every cell used is eroded: ``CenterCrop(e)`` with e = ceil(S (1 - erosion_ratio)) (torchvision: top = left =
    int(round((S - e) / 2)), Python's round-half-to-even);
  * the pair (first, second) and the 4-bin label are chosen by the random swaps of :131-153;
  * ``Resize(S)`` back to S x S (PIL bilinear on the PIL image), then ToTensor + Normalize(0.5, 0.5).

``resize_bilinear_u8`` restates Pillow's 8-bit bilinear resample (two passes, horizontal then vertical, 22-bit fixed-point
coefficients, uint8 intermediate) in integer numpy; ``pin()`` checks it against Pillow itself and writes
tests/golden/pair_crops.npz (inputs + Pillow's outputs: data only).  The product's ``vited_crop_pairs_u8`` kernel is tested
bit-for-bit against this module and that fixture.
"""
from __future__ import annotations

import math
import os

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Pillow: libImaging/Resample.c


def _coeffs(in_size: int, out_size: int):
    """Pillow's precompute_coeffs for the bilinear filter: per output index the first tap and the fixed-point weights."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, kk = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = sum(w)
        w = [v / ww for v in w] + [0.0] * (ksize - xmax)
        bounds.append((xmin, xmax))
        kk.append([int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w])
    return bounds, kk


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    src = np.moveaxis(img, axis, -1).astype(np.int64)
    bounds, kk = _coeffs(src.shape[-1], out_size)
    out = np.empty(src.shape[:-1] + (out_size,), dtype=np.uint8)
    for xx, ((xmin, xmax), k) in enumerate(zip(bounds, kk)):
        acc = np.full(src.shape[:-1], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(xmax):
            acc = acc + src[..., xmin + x] * k[x]
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, -1, axis)


def resize_bilinear_u8(img: np.ndarray, size: int) -> np.ndarray:
    """uint8 [..., H, W] -> [..., size, size]: Pillow's Image.resize((size, size), BILINEAR) per channel."""
    h, w = img.shape[-2:]
    out = img
    if w != size:
        out = _pass(out, size, -1)      # horizontal pass first, as Pillow does
    if h != size:
        out = _pass(out, size, -2)
    return out


def center_crop_offset(size: int, e: int) -> int:
    return int(round((size - e) / 2.0))     # torchvision.transforms.functional.center_crop (round half to even)


def erosion_size(size: int, erosion_ratio: float) -> int:
    return math.ceil(size * (1 - erosion_ratio))     # div2k_patch.py:116-119


def cell(region: np.ndarray, index: int, size: int) -> np.ndarray:
    """region uint8 [C, 2 S, 3 S] -> cell `index` of the 3-column x 2-row grid (transforms.crop(patch, 3, 2))."""
    r, c = divmod(index, 3)
    return region[:, r * size:(r + 1) * size, c * size:(c + 1) * size]


def assemble_pair(region: np.ndarray, cells, e: int, size: int) -> np.ndarray:
    """uint8 [2, C, S, S]: the two chosen cells, eroded to e x e and resized back to S x S."""
    out = []
    for idx in cells:
        cl = cell(region, int(idx), size)
        o = center_crop_offset(size, e)
        out.append(resize_bilinear_u8(cl[:, o:o + e, o:o + e], size))
    return np.stack(out)


def choose_pair(rng, with_negative: bool = True):
    """The random pair / label logic of div2k_patch.py:121-153 given a ``rng`` with ``.random()`` in [0, 1).
    Cells: first = 0, second = 1 (right of first), third = 4 (below second), fourth = 3 (below first), spare = 2.
    Returns ((cell of image 1, cell of image 2), label[4])."""
    first, second, third, fourth = 0, 1, 4, 3
    label = [1., 0., 0., 0.]
    if with_negative and 0.3 > rng.random():
        if 0.5 < rng.random():
            second, third = third, second
        else:
            second = 2
        if 0.5 < rng.random():
            second, first = first, second
        label = [0., 0., 0., 0.]
    else:
        if 0.5 < rng.random():
            second, fourth = fourth, second
            label = [0., 1., 0., 0.]
        if 0.5 < rng.random():
            first, second = second, first
            label = [0., 0., 1., 0.] if label[0] == 1 else [0., 0., 0., 1.]
    return (first, second), label


def pin(write: bool = True):
    """Check resize_bilinear_u8 against Pillow (bit for bit) and write tests/golden/pair_crops.npz."""
    from PIL import Image
    rng = np.random.default_rng(7)
    cases = []
    for size, e in ((64, 59), (64, 60), (64, 55), (64, 64), (32, 29), (512, 476), (64, 33)):
        region = rng.integers(0, 256, size=(3, 2 * size, 3 * size), dtype=np.uint8)
        cells = (int(rng.integers(0, 6)), int(rng.integers(0, 6)))
        mine = assemble_pair(region, cells, e, size)
        ref = []
        for idx in cells:
            cl = cell(region, idx, size)
            o = center_crop_offset(size, e)
            pil = Image.fromarray(np.ascontiguousarray(cl[:, o:o + e, o:o + e].transpose(1, 2, 0)))
            ref.append(np.asarray(pil.resize((size, size), Image.BILINEAR)).transpose(2, 0, 1))
        ref = np.stack(ref)
        assert np.array_equal(mine, ref), f'resize restatement differs from Pillow at size {size}, e {e}: max |d| = {np.abs(mine.astype(int) - ref.astype(int)).max()}'
        cases.append((size, e, cells, region, ref))
    if write:
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        blob = {}
        for i, (size, e, cells, region, ref) in enumerate(cases[:5]):      # the small ones travel as fixtures
            blob[f'meta{i}'] = np.array([size, e, cells[0], cells[1]], dtype=np.int64)
            blob[f'region{i}'] = region
            blob[f'pair{i}'] = ref
        np.savez_compressed(os.path.join(here, 'tests', 'golden', 'pair_crops.npz'), **blob)
    return len(cases)


if __name__ == '__main__':
    print(f'resize restatement == Pillow {__import__("PIL").__version__} on {pin()} cases; wrote tests/golden/pair_crops.npz')
