"""The arithmetic behind gfpp_head_frame_persist_lp / gfpp_head_frame_resolve (csrc/frame_head_lp.hip), restated in numpy and checked against a
direct simulation of the reference's loop (renderer.py:354-384 + raymarching.cu:978-1022) on random rays:

  * a ray composites its first min(c, e + 1, B) samples whatever way the loop cuts them into trips (c = occupied samples it owns, e = first sample
    whose pre-sample transmittance is below T_thresh, B = sum of n_step over the trips that run);
  * B and the n_alive sequence are functions of the histogram of m = min(c, e) alone (alive after the window ending at S  <=>  m >= S);
  * the ownership tiles (32 rays, multiplicative permutation, slots b, b + G, ... per workgroup) cover every ray exactly once;
  * a workgroup round never overflows its pool (8 x rw x n_step <= 1024 slots)."""
import numpy as np
import pytest


def reference_loop(c, e, N, max_steps):
    """renderer.py:354-384 on per-ray (c, e): returns (samples composited per ray, [n_alive per trip], B)."""
    done = np.zeros(N, np.int64)
    alive = np.arange(N)
    step, trace = 0, []
    while step < max_steps:
        n_alive = alive.size
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        trace.append(n_alive)
        keep = []
        for r in alive:
            s = 0
            while s < n_step:
                idx = step + s
                if idx >= c[r]:                     # deltas == 0: no sample left
                    break
                done[r] = idx + 1                   # composited
                if idx == e[r]:                     # pre-sample T < T_thresh: break after compositing, step not advanced
                    break
                s += 1
            if s == n_step:
                keep.append(r)
        alive = np.array(keep, dtype=np.int64)
        step += n_step
    return done, trace, step, alive.size


def budget_from_hist(hist, N, max_steps):
    """k_head_budget_resolve / budget_from_hist restated."""
    S, gone, alive, counters = 0, 0, N, []
    while S < max_steps and alive > 0:
        counters.append(alive)
        n = max(min(N // alive, 8), 1)
        gone += int(hist[S:S + n].sum())
        S += n
        alive = N - gone
    counters.append(alive)
    return S, counters


@pytest.mark.parametrize("seed,max_steps,thin", [(0, 16, False), (1, 16, True), (2, 7, False), (3, 24, True), (4, 1, False), (5, 16, None)])
def test_budget_and_alive_counts_follow_from_the_histogram(seed, max_steps, thin):
    rng = np.random.default_rng(seed)
    N = int(rng.integers(200, 3000))
    cap = max_steps + 7
    c = rng.integers(0, cap + 1, N)
    c[rng.random(N) < 0.5] = 0                                      # half the rays never meet an occupied cell
    if thin is None:
        c[:] = 0                                                     # empty bitfield
    e = np.where(rng.random(N) < (0.1 if thin else 0.7), rng.integers(0, cap + 1, N), 10 ** 6)   # thin scenes rarely terminate by transmittance
    done_ref, trace, B_ref, left = reference_loop(c, e, N, max_steps)
    # the persistent launch: every ray runs to ITS end, whatever the local schedule
    d = np.minimum(c, e + 1)
    m = np.minimum(np.minimum(c, e), cap)
    hist = np.bincount(m, minlength=32)
    assert hist.sum() == N
    B, counters = budget_from_hist(hist, N, max_steps)
    assert B == B_ref
    assert counters[:len(trace)] == trace and counters[len(trace)] == left
    np.testing.assert_array_equal(np.minimum(d, B), done_ref)       # snapshot selection: min(d, B) samples
    assert ((d > B) <= (B >= max_steps)).all() and (d <= cap).all()   # a snapshot index B - max_steps in [0, 6] exists whenever one is needed
    assert (B - max_steps <= 6) or not (d > B).any()


@pytest.mark.parametrize("kind,HW,max_steps", [("shell", 48, 16), ("speckle", 48, 16), ("speckle", 40, 7), ("shell", 33, 24)])
def test_budget_on_rays_with_holes(oracle_mod, kind, HW, max_steps):
    """The same identity on the ray population of a NON-CONVEX occupancy (rays that go occupied -> empty -> occupied, tests/helpers.nonconvex_occupancy):
    c = the occupied samples the reference's own marcher finds along each ray -- in as many separate runs as the ray crosses -- and e from a field whose
    opacity differs per run (a thin first run, an opaque second one: the ray survives the hole and ends inside the second run).  Holes change nothing in
    the arithmetic -- a trip takes the ray's next n_step OCCUPIED samples wherever they lie (raymarching.cu:878-927) -- and this pins that."""
    from helpers import frame_case, nonconvex_occupancy
    case = nonconvex_occupancy(frame_case("may_torso", HW), kind)
    hp, sd = case["hp"], case["sd"]
    rays = oracle_mod.get_rays(case["pose"], case["intr"], HW, HW)
    ro, rd = rays["rays_o"][0], rays["rays_d"][0]
    N = HW * HW
    cap = max_steps + 7
    nears, fars = oracle_mod.near_far_from_aabb(ro, rd, sd["aabb_infer"], 0.05)
    _, _, deltas = oracle_mod.march_rays(N, cap, np.arange(N, dtype=np.int32), nears.copy(), ro, rd, float(hp["bound"]), sd["density_bitfield"], 1, 128, nears, fars,
                                         -1, False, hp["dt_gamma"], 1024)
    d = deltas[:N * cap].reshape(N, cap, 2)
    occupied = d[:, :, 0] > 0
    c = occupied.sum(1)
    # a run starts where the step from the previous sample's end exceeds its own dt (the marcher skipped empty cells in between)
    t_end, dt = d[:, :, 1], d[:, :, 0]
    new_run = np.zeros_like(occupied)
    new_run[:, 1:] = occupied[:, 1:] & ((t_end[:, 1:] - t_end[:, :-1]) > 1.5 * dt[:, 1:])
    run_id = np.cumsum(new_run, axis=1)
    assert (run_id.max(axis=1) >= 1).mean() > 0.2                       # a good share of the rays has at least one hole
    rng = np.random.default_rng(HW)
    # opacity per run: first run thin (alpha 0.05), later runs opaque (alpha 0.5..0.9); e = first sample whose PRE-sample transmittance < T_thresh
    alpha = np.where(run_id == 0, 0.05, rng.uniform(0.5, 0.9, size=(N, 1))) * occupied
    T_pre = np.concatenate([np.ones((N, 1)), np.cumprod(1 - alpha, axis=1)[:, :-1]], axis=1)
    below = (T_pre < 0.01) & occupied
    e = np.where(below.any(1), below.argmax(1), 10 ** 6)
    ended_in_later_run = (e < 10 ** 6) & (run_id[np.arange(N), np.minimum(e, cap - 1)] >= 1)
    assert ended_in_later_run.sum() > 0
    done_ref, trace, B_ref, left = reference_loop(c, e, N, max_steps)
    dd = np.minimum(c, e + 1)
    m = np.minimum(np.minimum(c, e), cap)
    hist = np.bincount(m, minlength=32)
    B, counters = budget_from_hist(hist, N, max_steps)
    assert B == B_ref and counters[:len(trace)] == trace and counters[len(trace)] == left
    np.testing.assert_array_equal(np.minimum(dd, B), done_ref)
    assert len(trace) >= 4                                               # a real multi-trip schedule


TILE = 8        # kPTile


def _ownership(N, G):
    """frame_head_lp.hip: tile slots [q0, q0 + my_tiles) per workgroup (consecutive), tile of slot q = (q * mult) % n_tiles."""
    n_tiles = -(-N // TILE)
    mult = next((m for m in (1237, 251, 61, 7) if n_tiles % m and n_tiles * m < 2 ** 32), 1)
    G = min(G, n_tiles)
    per_wg, extra = divmod(n_tiles, G)
    b = np.arange(G)
    my_tiles = per_wg + (b < extra)
    q0 = b * per_wg + np.minimum(b, extra)
    return n_tiles, mult, G, q0, my_tiles


@pytest.mark.parametrize("N,G", [(512 * 512, 256), (256 * 256, 256), (37 * 37, 256), (1, 256), (64 * 64, 5), (640 * 640, 256), (1237 * 8, 256), (2048 * 2048, 256)])
def test_ownership_tiles_cover_every_ray_once(N, G):
    n_tiles, mult, G, q0, my_tiles = _ownership(N, G)
    q = np.arange(n_tiles, dtype=np.uint64)
    tiles = (q * np.uint64(mult)) % np.uint64(n_tiles)
    assert np.array_equal(np.sort(tiles), np.arange(n_tiles, dtype=np.uint64))          # a permutation
    assert int(q.max()) * mult < 2 ** 32                                                   # the kernel multiplies in 32 bits
    assert my_tiles.sum() == n_tiles and q0[0] == 0 and np.array_equal(q0[1:], np.cumsum(my_tiles)[:-1])   # the slot ranges tile [0, n_tiles)
    assert my_tiles.max() - my_tiles.min() <= 1


@pytest.mark.parametrize("HW,bar", [(512, 1.12), (256, 1.2)])
def test_workgroup_shares_are_balanced_on_the_bench_scene(oracle_mod, HW, bar):
    """Occupied samples per workgroup share (an upper bound of its evaluations) on the bench frame: the busiest of the 256 workgroups stays within
    12 % (512^2) / 20 % (256^2) of the mean.  (Slots strided by the grid size put every tile of a workgroup into one column block: 2.2 x the mean.)"""
    from helpers import frame_case
    case = frame_case("may_torso", HW)
    hp, sd = case["hp"], case["sd"]
    rays = oracle_mod.get_rays(case["pose"], case["intr"], HW, HW)
    ro, rd = rays["rays_o"][0], rays["rays_d"][0]
    N = HW * HW
    nears, fars = oracle_mod.near_far_from_aabb(ro, rd, sd["aabb_infer"], 0.05)
    _, _, deltas = oracle_mod.march_rays(N, 23, np.arange(N, dtype=np.int32), nears.copy(), ro, rd, float(hp["bound"]), sd["density_bitfield"], 1, 128, nears, fars,
                                         -1, False, hp["dt_gamma"], hp["max_steps"])
    c = (deltas[:N * 23, 0].reshape(N, 23) > 0).sum(1).astype(np.float64)
    n_tiles, mult, G, q0, my_tiles = _ownership(N, 256)
    per_tile = np.add.reduceat(np.pad(c, (0, n_tiles * TILE - N)), np.arange(0, n_tiles * TILE, TILE))
    in_slot_order = per_tile[(np.arange(n_tiles, dtype=np.int64) * mult) % n_tiles]
    share = np.array([in_slot_order[s:s + n].sum() for s, n in zip(q0, my_tiles)])
    assert share.max() / share.mean() <= bar, share.max() / share.mean()
    strided = np.array([in_slot_order[b::G].sum() for b in range(G)])
    assert strided.max() / strided.mean() > 1.5                                            # what the consecutive slots avoid


def _ownership_xcd(N, W, frames, G=256):
    """XCD-local ownership (round 5; frame_head_lp.hip, gfpp_frame_ws.row_rays): workgroup b belongs to XCD b % 8, which owns the tile columns c with c % 8 == b % 8;
    the G / 8 workgroups of an XCD share its n_tiles / 8 tiles like all G shared all tiles: consecutive slots of the XCD's own numbering, permuted, and
    tile = 8 * (XCD's tile number) + XCD.  Returns owner [n_tiles] or None where the library does not take this path."""
    if W % 64 or W // 64 < 8 or N % W or G % 8:
        return None
    n_tiles = frames * (N // TILE)
    nt, Gs = n_tiles // 8, G // 8
    if nt < G:
        return None
    mult = next((m for m in (1237, 251, 61, 7) if nt % m and nt * m < 2 ** 32), 1)
    per_wg, extra = divmod(nt, Gs)
    owner = np.full(n_tiles, -1, np.int64)
    for b in range(G):
        bs, x = b >> 3, b & 7
        my = per_wg + (bs < extra)
        q0 = bs * per_wg + min(bs, extra)
        local = (np.arange(q0, q0 + my, dtype=np.uint64) * np.uint64(mult)) % np.uint64(nt)
        tile = (local * np.uint64(8) + np.uint64(x)).astype(np.int64)
        assert (owner[tile] == -1).all()
        owner[tile] = b
    return owner


@pytest.mark.parametrize("HW,frames", [(512, 1), (512, 4), (1024, 2), (640, 3)])
def test_xcd_local_ownership_covers_every_tile_once_and_keeps_columns_apart(HW, frames):
    owner = _ownership_xcd(HW * HW, HW, frames)
    assert owner is not None and (owner >= 0).all()
    tiles_per_row = HW // TILE
    col = np.arange(owner.size) % tiles_per_row
    assert np.array_equal(owner % 8, col % 8)                                             # XCD x renders the tile columns x, x + 8, ...: a comb of 8-pixel columns
    counts = np.bincount(owner, minlength=256)
    assert counts.max() - counts.min() <= 1
    assert _ownership_xcd(256 * 256, 256, 4) is None and _ownership_xcd(37 * 37, 37, 4) is None     # too few columns per XCD / ragged rows: the image-wide permutation


def test_xcd_local_shares_are_balanced_on_the_bench_scene(oracle_mod):
    """The comb gives every XCD the same share of every part of the image: four consecutive 512^2 frames of the bench clip, occupied samples per workgroup -- the
    busiest workgroup stays within 5 % of the mean (the image-wide permutation: 2 %), the busiest XCD within 2 %."""
    from helpers import frame_case
    HW, frames = 512, 4
    per_tile = []
    for i in range(frames):
        case = frame_case("may_torso", HW, frame_idx=i)
        hp, sd = case["hp"], case["sd"]
        rays = oracle_mod.get_rays(case["pose"], case["intr"], HW, HW)
        ro, rd = rays["rays_o"][0], rays["rays_d"][0]
        N = HW * HW
        nears, fars = oracle_mod.near_far_from_aabb(ro, rd, sd["aabb_infer"], 0.05)
        _, _, deltas = oracle_mod.march_rays(N, 23, np.arange(N, dtype=np.int32), nears.copy(), ro, rd, float(hp["bound"]), sd["density_bitfield"], 1, 128, nears, fars,
                                             -1, False, hp["dt_gamma"], hp["max_steps"])
        c = (deltas[:N * 23, 0].reshape(N, 23) > 0).sum(1).astype(np.float64)
        per_tile.append(c.reshape(-1, TILE).sum(1))
    per_tile = np.concatenate(per_tile)
    owner = _ownership_xcd(HW * HW, HW, frames)
    share = np.bincount(owner, weights=per_tile, minlength=256)
    per_xcd = share.reshape(32, 8).sum(0)
    assert share.max() / share.mean() <= 1.05, share.max() / share.mean()
    assert per_xcd.max() / per_xcd.mean() <= 1.02, per_xcd.max() / per_xcd.mean()


def test_round_geometry_never_overflows_the_pool():
    for A in range(1, 1025):
        rw = (A + 7) // 8
        for cap in (1, 2, 4, 8):
            n_step = max(1, min(128 // rw, cap, 8))
            assert 8 * rw * n_step <= 1024 and rw <= 128
            assert (8 * rw - 1) * n_step + n_step - 1 < 1024                               # the last slot index


def _ingest_loop(counts, exists, room):
    """The ingest step as the kernel ran it until round 4: candidate tiles in order, the first that does not exist or does not fit ends the step."""
    acc, accepted, offsets = 0, 0, []
    for t, n in enumerate(counts):
        if not exists[t] or acc + n > room:
            break
        offsets.append(acc)
        acc += n
        accepted = t + 1
    return accepted, acc, offsets


def _ingest_scan(counts, exists, room):
    """The same decision as the kernel takes it now (k_head_frame_persist, frame_head_lp.hip): inclusive prefix sums of the 128 candidates' occupied-ray counts
    (two wavefront scans), population count of "exists and fits", offsets = exclusive prefix sums."""
    inc = np.cumsum(counts)
    fits = exists & (inc <= room)
    accepted = int(fits.sum())
    acc = int(inc[accepted - 1]) if accepted else 0
    return accepted, acc, [int(v) for v in (inc - counts)[:accepted]]


@pytest.mark.parametrize("seed", range(6))
def test_ingest_step_accepts_the_same_tiles_by_scan_as_by_loop(seed):
    """Tiles are accepted in order and a tile that does not fit ends the step, so the accepted tiles are a prefix of the candidates -- which is what lets the
    kernel replace its 128-iteration loop by prefix sums and a population count.  Random occupancies (0..8 occupied rays per 8-ray tile: image borders, holes),
    rooms from 256 (the smallest at which a step runs) to the whole pool, and workgroups whose tile share ends inside the step."""
    rng = np.random.default_rng(seed)
    for _ in range(400):
        kind = rng.integers(0, 4)
        counts = rng.integers(0, 9, 128) if kind else np.full(128, 8)
        if kind == 2:
            counts[rng.random(128) < 0.6] = 0                                   # mostly empty tiles (image border)
        left = int(rng.integers(1, 200))                                        # tiles this workgroup still owns
        exists = np.arange(128) < left
        room = int(rng.integers(256, 1025))
        a = _ingest_loop(counts, exists, room)
        b = _ingest_scan(counts, exists, room)
        assert a == b, (counts.tolist(), left, room, a, b)
        # monotone: nothing after the first rejected tile could have been taken
        assert b[0] == 128 or not exists[b[0]] or np.cumsum(counts)[b[0]] > room
        assert b[1] <= room
