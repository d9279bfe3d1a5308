"""The 16-bit corner-block grid tables the 16-bit head kernels read (round 4; csrc/grid_device.h: level_block_issue / level_block_finish,
gfpp_head_model.pos_grid_blk / amb_grid_blk, frame_pipeline.corner_block_table) against the oracle's grid encoder.  The lookup is restated in numpy with the kernel's index arithmetic -- one row per
z plane, neighbours baked into the row -- so a wrong neighbour, stride, mask or row layout shows up here, without a GPU.  Tolerance: the table and the corner
weights are rounded to fp16 (the reference's autocast path reads a half table as well, grid.py:43-47)."""
import ctypes

import numpy as np
import pytest
import torch

from genefaceplusplus_amd._lib import call
from genefaceplusplus_amd.radnerfs import frame_pipeline as fp
from oracle import oracle as orc

f32 = np.float32


def _levels(D, offsets, S, H, gridtype=1, align_corners=False):
    L = len(offsets) - 1
    lv = (fp.GridLevel * L)()
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    call("gfpp_grid_levels_fill", D, L, float(S), H, gridtype, int(align_corners), off.ctypes.data, 0, lv)
    return lv


def _block_lookup(u, table, lv, D):
    """level_block_issue + level_block_finish for all levels: u [B, D] in [0, 1] -> [L, B, 2]."""
    B = u.shape[0]
    out = np.zeros((len(lv), B, 2), np.float64)
    tab = table.astype(np.float64)
    for l, d in enumerate(lv):
        pos = (u * f32(d.scale) + f32(0.5)).astype(f32)                 # fmaf in the kernel: one rounding; the test points keep away from cell borders
        base = np.floor(pos).astype(np.int64)
        frac = (pos - np.floor(pos)).astype(f32)
        row = base[:, 0] + base[:, 1] * int(d.sy)
        if D == 3:
            row = row + base[:, 2] * int(d.sz)
        row &= int(d.mask)
        rows = [row] if D == 2 else [row, (row + int(d.sz)) & int(d.mask)]
        wx = np.stack([1 - frac[:, 0], frac[:, 0]], 1).astype(f32)
        wy = [wx * (1 - frac[:, 1:2]), wx * frac[:, 1:2]]
        for z, rz in enumerate(rows):
            v = tab[int(d.offset) + rz]                                  # [B, 8]
            for y in range(2):
                w = wy[y].astype(f32)
                if D == 3:
                    w = w * (frac[:, 2:3] if z else 1 - frac[:, 2:3])
                wh = w.astype(np.float16).astype(np.float64)
                out[l, :, 0] += (v[:, 4 * y:4 * y + 2] * wh).sum(1)
                out[l, :, 1] += (v[:, 4 * y + 2:4 * y + 4] * wh).sum(1)
    return out


@pytest.mark.parametrize("D", [3, 2])
def test_corner_block_lookup_matches_the_grid_encoder(D):
    offsets, pls = orc.grid_offsets(D, 16, 2, 2, 16, 16, desired_resolution=2048)
    S, H = np.log2(pls), 16
    rng = np.random.default_rng(D)
    emb = (rng.random((int(offsets[-1]), 2)) * 2 - 1).astype(f32)
    lv = _levels(D, offsets, S, H)
    assert not any(d.flags for d in lv), "the shipped tiled grids need no hash / true modulo"
    table = fp.corner_block_table(torch.from_numpy(emb), offsets, lv).numpy()
    assert table.shape == (int(offsets[-1]), 8) and table.dtype == np.float16
    u = rng.random((4000, D)).astype(f32)
    u[:8] = [[0.0] * D, [1.0] * D, [0.5] * D, [1.0, 0.0, 1.0][:D], [0.0, 1.0, 0.0][:D], [0.999999] * D, [1e-7] * D, [0.25, 0.75, 0.5][:D]]   # lattice edges
    ref = orc.grid_encode_raw(u, emb, offsets, S, H, 1, False, 0)     # [L, B, 2] fp32 tables, fp32 arithmetic
    got = _block_lookup(u, table, lv, D)
    # fp16 table values and fp16 corner weights: ~2^-11 relative per term of a sum of <= 8 terms of magnitude <= 1
    np.testing.assert_allclose(got, ref, rtol=0, atol=2.5e-3)
    assert np.sqrt(((got - ref) ** 2).mean()) < 2.5e-4


def test_block_rows_hold_the_neighbours_of_the_plain_table():
    offsets, pls = orc.grid_offsets(3, 16, 2, 2, 16, 16, desired_resolution=2048)
    lv = _levels(3, offsets, np.log2(pls), 16)
    emb = torch.arange(int(offsets[-1]) * 2, dtype=torch.float32).reshape(-1, 2) % 1024        # exactly representable in fp16
    table = fp.corner_block_table(emb, offsets, lv).float()
    for l in (0, 2, 5, 15):
        d = lv[l]
        size, sy, mask = int(d.size), int(d.sy), int(d.mask)
        T = emb[int(offsets[l]):int(offsets[l + 1])]
        for r in (0, 1, size // 2, size - 2, size - 1):
            wrap = (lambda k: k & mask) if mask != 0xFFFFFFFF else (lambda k: k % size)
            want = [T[r, 0], T[wrap(r + 1), 0], T[r, 1], T[wrap(r + 1), 1], T[wrap(r + sy), 0], T[wrap(r + sy + 1), 0], T[wrap(r + sy), 1], T[wrap(r + sy + 1), 1]]
            assert table[int(offsets[l]) + r].tolist() == [float(x) for x in want]
