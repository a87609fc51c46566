"""The LDS layouts of the MFMA kernels against the gfx950 bank model of `ds_read_b128` (tools/lds_bank_model.py): four fixed
16-lane groups, 64 banks of 4 bytes.  The model reproduced the measured conflict share of conv_halo256_kernel's row-index
patch swizzle exactly (40 %: SQ_LDS_BANK_CONFLICT 59.97M of SQ_LDS_IDX_ACTIVE 149.4M), so it is used here to pin the
layouts the kernels rely on: every fragment read costs the conflict-free 4 LDS cycles."""
import importlib.util
import os

import pytest

_spec = importlib.util.spec_from_file_location(
    "lds_bank_model", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lds_bank_model.py"))
bank = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(bank)


def test_halo_patch_column_swizzle_is_conflict_free_and_row_swizzle_was_not():
    col, row = bank.SWIZZLES["patch column (qx>>1)&7"], bank.SWIZZLES["row index   (q>>1)&7"]
    assert bank.average(4, 2, col) == 4.0 and bank.average(8, 4, col) == 4.0   # conv_halo_kernel, conv_halo256_kernel
    assert bank.average(8, 4, row) == 8.0                                      # what round 1 shipped: 2-way on every read
    share = (16 * 8.0 + 8 * 4 - 96) / (16 * 8.0 + 8 * 4)
    assert abs(share - 59965440.0 / 149422080.0) < 5e-3                        # the PMC pass (profiles/r02_pmc_conv_attn_summary.txt)


@pytest.mark.parametrize("row0", [0, 32, 64, 96, 128, 192])
@pytest.mark.parametrize("kk", range(4))
def test_gemm_and_attention_fragment_reads_are_conflict_free(row0, kk):
    """128-byte rows, 16-byte chunk c of row r at slot c ^ ((r>>1)&7), a wave reads rows row0 .. row0+31 (lanes 0-31 and
    32-63 take the two chunks of a k-step): the A / B tiles of every GEMM kernel, K and (since round 2) V^T of self_attn."""
    def addr(lane):
        l31, lhi = lane & 31, lane >> 5
        r = row0 + l31
        return r * 128 + (((kk * 2 + lhi) ^ ((r >> 1) & 7)) << 4)
    assert bank.read_cycles(addr) == 4


def test_unswizzled_rows_would_be_eight_way():
    """Why the swizzle exists: linear 128-byte rows put a 16-lane group on two slots of the bank row."""
    def addr(lane):
        return ((lane & 31)) * 128 + ((lane >> 5) << 4)
    assert bank.read_cycles(addr) == 32     # 8 cycles per group
