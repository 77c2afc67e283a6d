"""CPU: the Winograd kernel's LDS layout constants (kernels.hip: WPSTR, WPITCH) against the LDS bank model of MI355X_MICROARCH.md
(tools/lds_bank_model.py): the patch-row reads must stay at 4 LDS cycles per ds_read_b128.  Round 1's layout paid 8 (two-way
conflicts in every lane group) for four rounds; the PMC side of the same fact is tools/pmc_lds_conflicts.py."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import lds_bank_model as M  # noqa: E402


def _constants():
    src = open(os.path.join(ROOT, "deepcut-cnn_amd", "csrc", "kernels.hip")).read()
    wkc = int(re.search(r"constexpr int WBTY = \d+, WBTX = \d+, WBN = \d+, WKC = (\d+);", src).group(1))
    pad = int(re.search(r"constexpr int WPSTR = WKC \+ (\d+);", src).group(1))
    pitch = int(re.search(r"constexpr int WPITCH = (\d+);", src).group(1))
    return wkc + pad, pitch


def test_the_winograd_patch_row_reads_are_conflict_free():
    pixel_pitch, row_pitch = _constants()
    assert row_pitch >= 18 * pixel_pitch and row_pitch % 4 == 0 and pixel_pitch % 4 == 0
    assert M.wino_layout_cycles(pixel_pitch, row_pitch) == 4


def test_the_model_sees_the_round_1_layout_as_two_way_conflicts():
    assert M.wino_layout_cycles(36, 648, lambda row: 4 * ((row >> 1) & 1)) == 8
    assert M.wino_layout_cycles(36, 648) == 8  # the skew was never the issue: the pitch of two rows must be a multiple of 64 floats
    # the gather-GEMM's register-ring stage (rows of BK + 4 floats, lane -> row lane & 31, k offset 4 * (lane >> 5)): conflict-free
    for bk in (32, 64):
        assert M.ds_read_b128_cycles([(lane & 31) * (bk + 4) + 4 * (lane >> 5) for lane in range(64)]) == 4


def test_the_lds_dma_swizzles_are_conflict_free_in_the_model_too():
    """kernels.hip, LDS-DMA stages: 16-byte chunk c of row r sits at position c ^ swz(r), swz(r) = (r >> 1) & 7 for 128-byte rows and
    r & 15 for 256-byte rows; the fragment reader (lane -> row lane & 31, chunk 2 q + (lane >> 5)) XORs the same term back.  The PMC
    counters read 0.0 % conflicts for every such tile (profiles/r05_pmc_lds_conflicts*.txt): model and counters agree."""
    for q in range(4):
        addr = [((lane & 31) * 128 + ((2 * q + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) * 16) // 4 for lane in range(64)]
        assert M.ds_read_b128_cycles(addr) == 4, q
    for q in range(8):
        addr = [((lane & 31) * 256 + ((2 * q + (lane >> 5)) ^ ((lane & 31) & 15)) * 16) // 4 for lane in range(64)]
        assert M.ds_read_b128_cycles(addr) == 4, q
    # without the swizzle a 128-byte-pitch stage collides: 32 rows on 2 distinct bank offsets
    assert M.ds_read_b128_cycles([((lane & 31) * 128 + (lane >> 5) * 16) // 4 for lane in range(64)]) > 4


def _ws1x1f_read(K, j):
    """Float index per lane of ws1x1f's operand read j (csrc/stream1x1_f32.hip, frag0 + 16 j): lane = 16 q + p reads run q of pixel p's row;
    RPR rows share a 1 KiB (+ 16 bytes) block, row r of a block keeps its chunks rotated by r runs."""
    rpr = 1 if K >= 256 else 256 // K
    blk = (max(K * 4, 1024) + 16) // 4  # floats between padded blocks
    return [(p // rpr) * blk + (p % rpr) * K + ((q + p % rpr) & 3) * (K // 4) + 4 * j for q in range(4) for p in range(16)]


def test_the_float32_stream_reads_its_pixel_rows_without_conflicts_for_every_k():
    for K in (64, 128, 256, 512):
        for j in range(K // 16):
            assert M.ds_read_b128_cycles(_ws1x1f_read(K, j)) == 4, (K, j)
    # what the rotation inside a block is for: without it two (K = 128) / four (K = 64) lanes of a group share a bank
    for K, want in ((128, 8), (64, 16)):
        rpr, blk = 256 // K, (1024 + 16) // 4
        plain = [(p // rpr) * blk + (p % rpr) * K + q * (K // 4) for q in range(4) for p in range(16)]
        assert M.ds_read_b128_cycles(plain) == want
    # the epilogue's 16 x 16 tile in the matrix view (lane (p, q) at 16-byte slot 4 p + q: one write and one shortcut read per step) is two-way
    # conflicted in the model, the memory view of the same tile (slot = lane) is not: part of what SQ_LDS_BANK_CONFLICT reads for the kernel
    # (profiles/r06_pmc_lds_conflicts_f32.txt: 28.6 % of its LDS cycles — a few dozen cycles of a step beside 512-4 096 cycles of products)
    assert M.ds_read_b128_cycles([(4 * p + q) * 4 for q in range(4) for p in range(16)]) == 8
    assert M.ds_read_b128_cycles([lane * 4 for lane in range(64)]) == 4
