"""The XOR swizzles of the fused ResBlock kernels (respair.hip, respair_wide.hip, reschain.hip) against the hardware's LDS
lane groups, by enumeration (tools/lds_conflicts.py): the operand read of the 16x16x32 MFMA must be conflict-free for
every first row (a tap is an arbitrary row offset), the staging stores too, and -- on 64- / 128-byte rows -- the epilogue's
16-byte stores of 8 consecutive rows as well.  The formulas below are the ones in the kernels."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import lds_conflicts as L          # noqa: E402

USED = {64: lambda r: (r >> 1) & 3,            # rc_swz / rpn_swz<64>
        128: lambda r: r & 7,                   # rpn_swz<128>
        256: lambda r: (r & 7) << 1,            # swz16<256>
        512: lambda r: (r & 7) << 1}            # swz16<512>


def _costs(rowb, f):
    cpr = rowb // 16
    sw = lambda r: (f(r) % cpr) << 4
    rd = max(L.cycles(L.RD128, lambda l: (first + (l & 15)) * rowb + (((4 * ks + (l >> 4)) << 4) ^ sw(first + (l & 15))), 64, 16)
             for first in range(32) for ks in range(max(1, cpr // 4)))
    ws = max(L.cycles(L.WR128, lambda l: (first + (l & 15) + 16 * ((l >> 4) & 1)) * rowb + ((c << 4) ^ sw(first + (l & 15) + 16 * ((l >> 4) & 1))), 32, 16)
             for first in (0, 5, 64) for c in range(cpr))
    st = max(L.cycles(L.WR128, lambda l: (first + l // cpr) * rowb + (((l % cpr) << 4) ^ sw(first + l // cpr)), 32, 16) for first in (0, 3))
    return rd, ws, st


def test_operand_reads_and_staging_stores_are_conflict_free_for_every_row_size():
    for rowb, f in USED.items():
        rd, _, st = _costs(rowb, f)
        assert rd == 4, (rowb, rd)             # 4 lane groups, one LDS cycle each
        assert st == 8, (rowb, st)             # 8 lane groups, one LDS cycle each


def test_epilogue_stores_are_conflict_free_on_the_narrow_rows():
    for rowb in (64, 128):
        assert _costs(rowb, USED[rowb])[1] == 8, rowb
    # the round's first swizzles kept chunk bit 0 fixed and were 2-way conflicted there: the check would have caught it
    assert _costs(64, lambda r: ((r >> 2) & 1) << 1)[1] == 16
    assert _costs(128, lambda r: ((r >> 1) & 3) << 1)[1] == 16
