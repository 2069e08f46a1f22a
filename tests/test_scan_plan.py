"""CPU: the phases of the scan launch (topk256.hip scan256_plan, host logic) through the experiment library's
revo_debug_scan_plan -- no device needed.

With 8 query tiles and more, query tiles are pinned to XCDs and the slices of an XCD's query tiles have to line up, so the
launch takes them in phases of 8 a query tiles (a = 1, 2, 4 ... 32 per XCD, 32 / a slices each: 256 workgroups, every CU
busy), largest first, then the < 8 left over with the slices of a small launch.  Checked here: every query tile is in
exactly one phase, pinned phases start at a multiple of 8 blocks and fill an XCD's 32 CUs exactly, the segment-slot count
is the largest slice count, the block count adds up; plus the cases DESIGN.md quotes (8 192 and 9 984 queries)."""
import ctypes as C

import pytest

from reverso_amd import _lib


def _plan(Q, rows):
    lib = _lib.load_exp()
    out = (C.c_int64 * 64)()
    blocks = lib.revo_debug_scan_plan(Q, rows, out, 64)
    assert blocks > 0, (Q, rows)
    nph, splits = int(out[0]), int(out[1])
    phases = [tuple(int(out[2 + 4 * i + j]) for j in range(4)) for i in range(nph)]     # first block, q0, qn, ns
    return int(blocks), splits, phases


@pytest.mark.parametrize("rows", [16384, 60000, 116808, 991808, 10_000_000])
@pytest.mark.parametrize("Q", [1, 64, 255, 256, 257, 1000, 1792, 2048, 2049, 2304, 4096, 8192, 9984, 10000, 10240, 65536,
                               65537, 100_000, 1_000_000])
def test_phases_cover_every_query_tile_once(Q, rows):
    blocks, splits, phases = _plan(Q, rows)
    qtiles, tiles = (Q + 255) // 256, (rows + 255) // 256
    assert 1 <= len(phases) <= 8
    nxt, used = 0, 0
    for i, (first, q0, qn, ns) in enumerate(phases):
        assert q0 == nxt and qn >= 1 and 1 <= ns <= max(tiles, 1), (i, phases)
        assert first == used, (i, phases)                       # phases follow each other (padded to 8 blocks in between)
        nxt += qn
        used += qn * ns
        last = i == len(phases) - 1
        if not last:
            assert first % 8 == 0 and qn % 8 == 0, phases       # pinned: block b of the phase -> XCD b % 8 = its query tile % 8
            used = (used + 7) // 8 * 8
        if qtiles >= 8 and qn % 8 == 0 and (not last or qn >= 8):
            per_xcd = qn // 8 if qn <= 256 else 32              # query tiles an XCD holds at a time
            assert per_xcd in (1, 2, 4, 8, 16, 32), phases
            assert per_xcd * ns <= 32, phases                   # ... times their slices: at most its 32 CUs
            if tiles >= 3 * 32:
                assert per_xcd * ns == 32, phases               # and exactly 32 once the gallery has tiles to slice
    assert nxt == qtiles and used == blocks
    assert splits == max(p[3] for p in phases)
    if qtiles < 8:
        assert len(phases) == 1


def test_the_cases_the_design_quotes():
    tiles_rows = 991808                                         # 1 M rows minus the 8192-row pre-pass
    blocks, splits, phases = _plan(8192, tiles_rows)            # 32 query tiles: four per XCD x eight slices, nothing left over
    assert phases == [(0, 0, 32, 8)] and blocks == 256 and splits == 8
    blocks, splits, phases = _plan(9984, tiles_rows)            # 39 = 32 + 7
    assert phases[0] == (0, 0, 32, 8) and phases[1][1:3] == (32, 7) and len(phases) == 2
    assert 7 * phases[1][3] <= 256 and splits == phases[1][3]
    blocks, splits, phases = _plan(2048, tiles_rows)            # eight query tiles: one per XCD x 32 slices
    assert phases == [(0, 0, 8, 32)]
    blocks, splits, phases = _plan(64, tiles_rows)              # one query tile: a small launch, one round of workgroups
    assert len(phases) == 1 and phases[0][2] == 1 and 128 <= phases[0][3] <= 256
