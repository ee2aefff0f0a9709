"""CPU: the stream-K work decomposition of paella_amd/csrc/gemm.hip, restated in Python and checked exhaustively on small cases.

The kernel gives workgroup g of G the unit range [g*q + min(g, r), +q + (g < r)) (U = G*q + r) of the U = tiles*KT (tile, K-step) units.  Partial tiles are
published to slab slot 2g ("head": the segment starts the workgroup's range) or 2g+1 (a later, necessarily last, segment) and the
last arriver recomputes from (tile, G, U) alone which workgroups contributed and in which slot.  This test proves, for every
(tiles, KT, G) in a grid, that reader and writers agree: same part set, same slots, ticket target = parts - 1, every K step of
every tile covered exactly once.  (No GPU: pure integer logic mirrored from the kernel source.)"""
import itertools

import pytest


def start(g, U, G):
    q, r = divmod(U, G)  # workgroup g owns q + (g < r) units
    return g * q + min(g, r)


def owner(x, U, G):
    q, r = divmod(U, G)
    big = r * (q + 1)
    return x // (q + 1) if x < big else r + (x - big) // q


def writer_segments(g, U, G, KT):
    """What workgroup g does: list of (tile, k0, k1, first_segment)."""
    u0, u1 = start(g, U, G), start(g + 1, U, G)
    segs, first = [], True
    tile, kt, i, n = u0 // KT, u0 % KT, 0, u1 - u0
    while i < n:
        seg = min(KT - kt, n - i)
        segs.append((tile, kt, kt + seg, first))
        first = False
        i += seg
        kt += seg
        if kt == KT:
            kt = 0
            tile += 1
    return segs


def reader_parts(tile, U, G, KT):
    """What the last arriver of `tile` computes: [(workgroup, slot)] in summation order."""
    tb = tile * KT
    g_first = owner(tb, U, G)
    g_last = owner(tb + KT - 1, U, G)
    parts = []
    for gp in range(g_first, g_last + 1):
        tail = gp == g_first and start(g_first, U, G) < tb
        parts.append((gp, 2 * gp + (1 if tail else 0)))
    return parts


@pytest.mark.parametrize("tiles,KT", [(1, 1), (1, 7), (3, 5), (7, 4), (10, 40), (80, 40), (5, 13), (2, 160)])
def test_reader_and_writers_agree(tiles, KT):
    U = tiles * KT
    for G in sorted(set(list(range(1, min(U, 70) + 1)) + [U, max(1, U // 2), max(1, U - 1), min(U, 256), min(U, 512)])):
        if G > U:
            continue
        cover = {t: [0] * KT for t in range(tiles)}
        published = {}  # slot -> (tile, k0, k1)
        full = set()
        for g in range(G):
            segs = writer_segments(g, U, G, KT)
            assert segs, "every workgroup owns at least one unit when G <= U"
            for (tile, k0, k1, first) in segs:
                for k in range(k0, k1):
                    cover[tile][k] += 1
                if k0 == 0 and k1 == KT:
                    full.add(tile)
                else:
                    slot = 2 * g + (0 if first else 1)
                    assert slot not in published, "a slab slot is written twice in one launch"
                    published[slot] = (tile, k0, k1)
        assert all(all(c == 1 for c in v) for v in cover.values()), (tiles, KT, G)
        for tile in range(tiles):
            parts = reader_parts(tile, U, G, KT)
            if tile in full:
                assert len(parts) == 1  # no ticket is ever taken for this tile
                continue
            ks = []
            for (gp, slot) in parts:
                assert slot in published and published[slot][0] == tile, (tiles, KT, G, tile, parts)
                ks.append(published[slot][1:])
            # the parts tile the K range in ascending order: fixed, launch-independent summation order
            assert ks[0][0] == 0 and ks[-1][1] == KT and all(a[1] == b[0] for a, b in zip(ks, ks[1:]))
            n_writers = sum(1 for v in published.values() if v[0] == tile)
            assert n_writers == len(parts)  # ticket target = parts - 1 arrivals before the last


def test_classic_cases_are_special_cases():
    # G == tiles: one whole tile per workgroup, nothing published
    for tiles, KT in [(12, 8), (1000, 20)]:
        U = tiles * KT
        for g in range(tiles):
            assert writer_segments(g, U, tiles, KT) == [(g, 0, KT, True)]
    # G == tiles * S with S | KT: classic split-K, S equal slices per tile
    tiles, KT, S = 6, 40, 4
    U, G = tiles * KT, tiles * S
    for g in range(G):
        (tile, k0, k1, first), = writer_segments(g, U, G, KT)
        assert tile == g // S and k0 == (g % S) * (KT // S) and k1 == k0 + KT // S and first


def tile_coords(tile, tiles_m, tiles_n, gm):
    """sk_tile_coords of gemm.hip: groups of gm tile rows, m fastest inside a group, then n, then the next group (gm >= tiles_m:
    plain m-fastest order)."""
    if gm >= tiles_m:
        return tile % tiles_m, tile // tiles_m
    width = gm * tiles_n
    grp, rem = divmod(tile, width)
    first_m = grp * gm
    gsz = min(tiles_m - first_m, gm)
    tile_n, dm = divmod(rem, gsz)
    return first_m + dm, tile_n


@pytest.mark.parametrize("tiles_m,tiles_n,gm", [(32, 4, 8), (33, 5, 8), (39, 7, 8), (40, 4, 16), (64, 40, 8), (7, 9, 8), (100, 3, 8), (32, 6, 4), (35, 4, 32)])
def test_grouped_rasterisation_is_a_bijection_with_block_locality(tiles_m, tiles_n, gm):
    """Every (tile_m, tile_n) is produced exactly once (ragged last group included), and any window of gm * 4 consecutive tiles inside
    a full group touches at most gm tile rows and 5 tile columns -- the L2-sharing property the order exists for."""
    T = tiles_m * tiles_n
    coords = [tile_coords(t, tiles_m, tiles_n, gm) for t in range(T)]
    assert sorted(coords) == sorted(itertools.product(range(tiles_m), range(tiles_n)))
    if gm < tiles_m:
        full_groups = tiles_m // gm
        for t0 in range(0, full_groups * gm * tiles_n - gm * 4, 7):
            win = coords[t0:t0 + gm * 4]
            if win[0][0] // gm != win[-1][0] // gm:
                continue  # window straddles two row groups
            assert len({m for m, _ in win}) <= gm and len({n for _, n in win}) <= 5


def _fast_div_of(d):
    """Mirror of paella_amd/csrc/gemm.hip fast_div_of / fast_div: n // d for n < 2^31 as one multiply-high and one shift."""
    if d <= 1:
        return 0, 0, 0xffffffff
    l = 0
    while (1 << l) < d:
        l += 1
    mul = ((1 << (31 + l)) + d - 1) // d
    assert mul < (1 << 32)
    return mul, l - 1, 0


def _fast_div(n, f):
    mul, shr, pas = f
    return (((n * mul) >> 32) + (n & pas)) >> shr


def test_fast_div_is_exact_below_2_pow_31():
    """The GEMM kernel divides unit / tile indices (< 2^31, enforced by the launcher) by launch constants with a precomputed multiplier: exact for every
    divisor class (1, powers of two, 2^k +- 1, large, random) at the range's edges and at random points."""
    import random
    rng = random.Random(5)
    ds = [1, 2, 3, 5, 7, 10, 40, 41, 160, 641, 1280, 5120, 65535, 65536, 65537, (1 << 20) - 1, (1 << 30) + 1, (1 << 31) - 1] + [rng.randrange(1, 1 << 31) for _ in range(300)]
    for d in ds:
        f = _fast_div_of(d)
        ns = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 31) - 1, (1 << 31) - d, ((1 << 31) - 1) // d * d, ((1 << 31) - 1) // d * d - 1]
        ns += [rng.randrange(0, 1 << 31) for _ in range(200)]
        for n in ns:
            if 0 <= n < (1 << 31):
                assert _fast_div(n, f) == n // d, (n, d)
