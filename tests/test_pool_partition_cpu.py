"""The work partition of the pooled trip kernels (k_head_trip_pool / k_head_trip_wp, csrc/frame_head_lp.hip, frame_head.hip), restated in Python:
for every alive-ray count, step count and grid size each alive ray must be taken by exactly one (round, workgroup, wavefront, lane, sub) and
every sample slot must stay inside the workgroup's pool.  The kernels themselves are compared with their tile-per-wavefront predecessors on the
GPU (tests/test_render_gpu.py::test_pooled_trips_equal_per_wavefront_trips); this pins the index arithmetic for sizes no GPU test renders."""
import numpy as np
import pytest

K_SLOTS = 128          # kLpSlots / kWSlots: sample slots of one wavefront tile


def partition(n_alive, n_step, grid, waves):
    """(rounds, rw, n_tiles, mult) exactly as the kernels compute them (uint32 arithmetic)."""
    tiles_per_round = grid * waves
    rw_max = K_SLOTS // n_step
    rounds = (n_alive + tiles_per_round * rw_max - 1) // (tiles_per_round * rw_max)
    rw = (n_alive + tiles_per_round * rounds - 1) // (tiles_per_round * rounds)
    n_tiles = (n_alive + rw - 1) // rw
    mult = 1237 if n_tiles % 1237 else 1
    return rounds, rw, n_tiles, mult


def rays_taken(n_alive, n_step, grid, waves):
    rounds, rw, n_tiles, mult = partition(n_alive, n_step, grid, waves)
    assert 1 <= rw <= K_SLOTS // n_step, (rw, n_step)
    assert rw * n_step * waves <= K_SLOTS * waves               # the pool of one workgroup round
    assert rw <= 128, "at most two rays per lane (sub 0 / 1)"
    assert n_tiles <= rounds * grid * waves, "every tile has a (round, workgroup, wavefront)"
    q = np.arange(n_tiles, dtype=np.uint64)
    tile = (q * np.uint64(mult)) % np.uint64(n_tiles)
    assert len(np.unique(tile)) == n_tiles, "q -> q * mult mod n_tiles must be a permutation"
    n = (tile[:, None] * np.uint64(rw) + np.arange(rw, dtype=np.uint64)[None, :]).reshape(-1)
    n = n[n < n_alive]
    # slot of the last sample of the last local ray of a workgroup
    assert ((waves - 1) * rw + (rw - 1)) * n_step + (n_step - 1) < K_SLOTS * waves
    assert max(int(tile.max()) * rw + rw - 1, 0) < 2 ** 32, "uint32 ray index"
    return n


@pytest.mark.parametrize("grid,waves", [(256, 8), (512, 4), (32, 8), (64, 4), (304, 8), (1, 8)])
def test_every_alive_ray_is_taken_exactly_once(grid, waves):
    rng = np.random.default_rng(grid * 17 + waves)
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 1023, 2048, 4096, 12326, 65536, 262144, 262145, 409600, 1048576, 589824]
    sizes += [int(v) for v in rng.integers(1, 1 << 21, size=12)]
    if grid < 32:
        sizes = [s for s in sizes if s <= 70000]            # (a tiny grid takes many rounds: keep the test quick)
    for n_alive in sizes:
        for n_step in range(1, 9):
            n = rays_taken(n_alive, n_step, grid, waves)
            assert len(n) == n_alive and len(np.unique(n)) == n_alive, (n_alive, n_step, grid)


def test_full_frame_is_one_round_of_full_tiles():
    """512 x 512 rays, trip 0 on 256 CUs: one round, 128 rays per wavefront (two per lane), 2048 tiles -- the case the kernel was sized for."""
    assert partition(262144, 1, 256, 8) == (1, 128, 2048, 1237)
    assert partition(262144, 1, 512, 4) == (1, 128, 2048, 1237)
    # the small grid of the calibrated tail launch takes the same rays in eight rounds
    assert partition(262144, 1, 32, 8)[0] == 8


def test_trip_schedule_that_ends_by_step_budget():
    """renderer.py:364-384 with the alive counts of the bench frame: n_step 1, 2, 2, 2, 4, 8 -> the sixth trip uses up max_steps = 16, which is why
    the multi-trip launch that starts with it returns without a barrier (lp_separate_trips() == 5)."""
    N, max_steps = 262144, 16
    alive = [262144, 112146, 104459, 88893, 63151, 12326, 400]
    step, steps = 0, []
    for n_alive in alive:
        if step >= max_steps:
            break
        n_step = max(min(N // n_alive, 8), 1)
        steps.append(n_step)
        step += n_step
    assert steps == [1, 2, 2, 2, 4, 8] and step == 19
