#!/usr/bin/env python3
"""The LDS bank model of MI355X_MICROARCH.md (section LDS), as a checker for fragment-read layouts.

A wave64 `ds_read_b128` is served in four fixed groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 —, one LDS
cycle per group when its lanes touch 64 distinct 4-byte banks (bank = (byte address / 4) mod 64; identical addresses broadcast); every
further distinct address on a busy bank costs the group one more cycle.  `ds_read_b128_cycles(addr)` returns the LDS cycles of one
instruction given each lane's FLOAT index; 4 is conflict-free.

Round 5 found the Winograd kernel's patch-row reads at 8 cycles with this model (and `SQ_LDS_BANK_CONFLICT`, tools/pmc_lds_conflicts.py):
lane -> tile row r = lane[3], tile column c = lane[2:0] (adjacent tiles are two staged pixels apart), channel quad kg = lane[5:4]; address
= row(r) * row_pitch + 2 c * pixel_pitch + 4 kg.  With a pixel pitch of 36 floats the read is conflict-free iff two rows' pitch is a
multiple of 64 floats.

    python tools/lds_bank_model.py            # the library's layout, the round-1 layout, and a search over row pitches
"""

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def ds_read_b128_cycles(addr):
    """addr: 64 float indices (one per lane, each the start of 4 consecutive floats) -> LDS cycles of the instruction."""
    total = 0
    for grp in B128_GROUPS:
        banks = {}
        for lane in grp:
            for d in range(4):
                banks.setdefault((addr[lane] + d) % 64, set()).add(addr[lane] + d)
        total += max(len(v) for v in banks.values())
    return total


def wino_patch_row_read(pixel_pitch, row_pitch, row0, skew=lambda row: 0):
    """Float index per lane of the Winograd kernel's patch-row read (kernels.hip, lread): first tile row of the fragment at staged
    row `row0`, the second one two rows below."""
    out = []
    for lane in range(64):
        r, c, kg = (lane & 15) >> 3, lane & 7, lane >> 4
        row = row0 + 2 * r
        out.append(row * row_pitch + skew(row) + 2 * c * pixel_pitch + 4 * kg)
    return out


def wino_layout_cycles(pixel_pitch, row_pitch, skew=lambda row: 0):
    """Worst case over the eight staged rows a fragment can start at."""
    return max(ds_read_b128_cycles(wino_patch_row_read(pixel_pitch, row_pitch, row0, skew)) for row0 in range(8))


if __name__ == "__main__":
    print("library layout (pixel pitch 36, row pitch 672):", wino_layout_cycles(36, 672), "LDS cycles per ds_read_b128")
    print("round-1 layout (pixel pitch 36, row pitch 648 + 4-float skew per row pair):",
          wino_layout_cycles(36, 648, lambda row: 4 * ((row >> 1) & 1)), "LDS cycles")
    for pp in (36, 40, 44, 52, 68):
        ok = [rp for rp in range(18 * pp, 18 * pp + 68, 4) if wino_layout_cycles(pp, rp) == 4]
        print("pixel pitch %d: conflict-free row pitches %s" % (pp, ok or "none within +64 floats"))
