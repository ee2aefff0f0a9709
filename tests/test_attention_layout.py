"""CPU mirror of attention_lds_kernel's unpadded direct-to-LDS layout (attention.hip, STG 2: odd head_dim / 16, e.g. head_dim 80 of the 1B model).

The kernel's tiles are exactly [32 keys][D] floats so that a workgroup's two stages are 40 960 B at head_dim 80 and FOUR workgroups fit the CU's 160 KB; bank conflicts are
avoided by where the DMA puts things: K tile -- LDS chunk position p of row r holds source chunk p ^ f((r >> 2) & 3), f = {0, 3, 2, 1}; V tile -- LDS row r holds key
r ^ ((r >> 2) & 1).  This file restates that arithmetic and checks, with the LDS banking rules of MI355X_MICROARCH.md (LDS section), that
  * the DMA mapping is a bijection: every (key, 16-byte chunk) of the 32 x D tile lands exactly once and the fragment reads find it where the kernel looks;
  * every ds_read_b128 lane group of a K fragment read covers 16 distinct 16-byte slots of the 256-byte bank row (conflict-free);
  * every ds_read_b32 lane group of a V fragment read covers 32 distinct banks (conflict-free);
and that the padded layout (STG 1: pitch D + 4) has the 2-way K-read conflicts the round-5 counters showed."""
import pytest

KTILE = 32
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
B32_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def f(r):
    return (4 - ((r >> 2) & 3)) & 3          # {0, 3, 2, 1}: the kernel's (4 - ((row >> 2) & 3)) & 3


def dma_map(D):
    """LDS chunk g (16 bytes) of the K / V tile <- (key row, source chunk), as dma_stage issues it."""
    rch = D // 4
    k_src, v_src = {}, {}
    for g in range(KTILE * rch):
        row, pos = g // rch, g % rch
        k_src[g] = (row, pos ^ f(row))
        v_src[g] = (row ^ ((row >> 2) & 1), pos)
    return k_src, v_src


@pytest.mark.parametrize("D", [16, 48, 80, 112])
def test_unpadded_dma_mapping_is_a_bijection_and_the_fragment_reads_find_their_data(D):
    rch = D // 4
    k_src, v_src = dma_map(D)
    assert sorted(k_src.values()) == sorted(v_src.values()) == [(r, c) for r in range(KTILE) for c in range(rch)]
    for lane in range(64):
        r16, kq = lane & 15, lane >> 4
        for half in (0, 16):
            for j in range(D // 16):
                # K fragment: the kernel reads 16 bytes at floats (r16 + half) * D + (kq ^ f(r16)) * 4 + 16 j and expects K[key r16 + half][4 kq + 16 j .. + 3]
                word = (r16 + half) * D + (kq ^ f(r16)) * 4 + 16 * j
                assert word % 4 == 0 and k_src[word // 4] == (r16 + half, kq + 4 * j)
                # V fragment: float at ((4 kq + e) ^ (kq & 1) + half) * D + r16 + 16 j must be V[key 4 kq + e + half][r16 + 16 j]
                for e in range(4):
                    word = (((4 * kq + e) ^ (kq & 1)) + half) * D + r16 + 16 * j
                    assert v_src[word // 4] == (4 * kq + e + half, (r16 + 16 * j) // 4)


def k_read_conflicts(D, pitch, swizzled):
    worst = 1
    for j in range(D // 16):
        for group in B128_GROUPS:
            slots = {}
            for lane in group:
                r16, kq = lane & 15, lane >> 4
                word = r16 * pitch + ((kq ^ f(r16)) if swizzled else kq) * 4 + 16 * j
                slots.setdefault((word // 4) % 16, set()).add(word)
            worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def v_read_conflicts(D, pitch, permuted):
    worst = 1
    for j in range(D // 16):
        for e in range(4):
            for group in B32_GROUPS:
                banks = {}
                for lane in group:
                    r16, kq = lane & 15, lane >> 4
                    row = ((4 * kq + e) ^ (kq & 1)) if permuted else 4 * kq + e
                    word = row * pitch + r16 + 16 * j
                    banks.setdefault(word % 32, set()).add(word)
                worst = max(worst, max(len(v) for v in banks.values()))
    return worst


@pytest.mark.parametrize("D", [16, 48, 80, 112])
def test_unpadded_layout_is_bank_conflict_free(D):
    assert k_read_conflicts(D, D, True) == 1
    assert v_read_conflicts(D, D, True) == 1
    # the same tiles without the swizzle / the row permutation conflict: the layout is doing the work, not the pitch
    assert k_read_conflicts(D, D, False) >= 2
    assert v_read_conflicts(D, D, False) == 2


@pytest.mark.parametrize("D", [32, 64, 80, 96, 128])
def test_padded_layout_has_two_way_k_read_conflicts(D):
    """Pitch D + 4 (STG 1, the even head_dim / 16 path; rounds 2-5 at every head_dim): V reads are conflict-free; K reads are 2-way -- the odd slot pitch spreads 16
    CONSECUTIVE rows over 16 slots, but a ds_read_b128 lane group is 8 rows of one chunk column and 8 OTHER rows of the next (SQ_LDS_BANK_CONFLICT != 0 in
    profiles/r05_attention_pmc_and_probe.txt)."""
    assert v_read_conflicts(D, D + 4, False) == 1
    assert k_read_conflicts(D, D + 4, False) == 2


def test_unpadded_stage_fits_four_workgroups_per_cu_at_head_dim_80():
    D = 80
    stage_bytes = 2 * KTILE * D * 4            # K tile + V tile
    assert (KTILE * D // 4) % 64 == 0          # a tile is a whole number of 64-lane DMA instructions: no rounding waste
    assert 2 * stage_bytes == 40960 and 4 * 2 * stage_bytes == 160 * 1024
