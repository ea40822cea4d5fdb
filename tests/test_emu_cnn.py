"""CPU (fiber-emulator) runs of the CNN-block parity cases (K6): forward, BN running stats, and every gradient,
against torch autograd on the CPU oracle graph.  Small shapes; the GPU tests run the production shapes."""
import pytest

from tests import parity_cases as P
from tests.emu_support import emu  # noqa: F401

# (layer, B, T, F): F follows the 2023 recipe's per-layer mel width where cheap enough
CASES = [(0, 2, 6, 16), (1, 2, 5, 64), (2, 2, 6, 32), (3, 1, 10, 16), (4, 2, 9, 8), (5, 2, 20, 4), (6, 2, 35, 2)]


@pytest.mark.parametrize("layer,B,T,F", CASES)
def test_block_train(emu, layer, B, T, F):
    P.case_cnn_block("cpu", layer, B, T, F, training=True, dropout_p=0.5)


@pytest.mark.parametrize("layer,B,T,F", [(0, 1, 5, 8), (2, 1, 4, 32), (6, 1, 70, 2)])
def test_block_eval_nodrop(emu, layer, B, T, F):
    P.case_cnn_block("cpu", layer, B, T, F, training=False, dropout_p=0.0)


@pytest.mark.parametrize("layer,B,T,F", [c for c in CASES if c[0] > 0])
def test_block_train_split_bf16(emu, layer, B, T, F):
    """Same blocks with the split-bf16 (bf16x3) convolution: fp32-level accuracy is part of the contract."""
    P.case_cnn_block("cpu", layer, B, T, F, training=True, dropout_p=0.5, precision="bf16x3", tol=1e-4)


@pytest.mark.parametrize("layer,B,T,F", [(2, 2, 6, 32), (6, 2, 35, 2)])
def test_block_persistent_tiles(emu, layer, B, T, F):
    """Wide GLU kernels with several tiles per workgroup: exercises the register prefetch of the next tile."""
    from desed_task_amd import _lib
    _lib.set_tuning("glu_grid_cap", 3)
    try:
        P.case_cnn_block("cpu", layer, B, T, F, training=True, dropout_p=0.5)
    finally:
        _lib.set_tuning("glu_grid_cap", 0)


@pytest.mark.parametrize("B,T,F,fused", [(2, 37, 32, True), (1, 18, 8, True), (2, 6, 16, False), (3, 33, 16, True)])
def test_first_block_fused_and_unfused(emu, B, T, F, fused):
    """Block 0 through sed_block0_fwd / sed_block0_bwd (conv output recomputed, never stored) and through the unfused kernels:
    odd frame counts, several 16-row tiles per clip, tiles shared by persistent workgroups."""
    P.case_cnn_block("cpu", 0, B, T, F, training=True, dropout_p=0.5, block0_fused=fused)


def test_first_block_persistent_workgroups(emu):
    """Several 16-row tiles per persistent workgroup (grid capped at 2)."""
    from desed_task_amd import _lib
    for key, v in (("glu_grid_cap", 2),):
        _lib.set_tuning(key, v)
        try:
            P.case_cnn_block("cpu", 0, 3, 33, 16, training=True, dropout_p=0.5, block0_fused=True)
        finally:
            _lib.set_tuning(key, 0)


@pytest.mark.parametrize("layer,B,T,F", [(1, 2, 21, 32), (2, 3, 13, 32), (1, 1, 25, 32)])     # (1, 25): 7 tiles -> unequal shares
def test_conv_persistent_workgroups(emu, layer, B, T, F):
    """Split-bf16 convolutions of the single-chunk layers with several tiles per persistent workgroup (the next tile's halo patch and
    first weight row are prefetched during the last kernel row of the current one): forward, BN-folded data gradient, statistics."""
    from desed_task_amd import _lib
    for tpw in (2, 5):
        _lib.set_tuning("convb_tpw", tpw)
        try:
            P.case_cnn_block("cpu", layer, B, T, F, training=True, dropout_p=0.5, precision="bf16x3", tol=1e-4)
        finally:
            _lib.set_tuning("convb_tpw", 0)


def test_first_block_fused_eval(emu):
    import torch
    with torch.no_grad():
        pass
    P.case_cnn_block("cpu", 0, 2, 21, 16, training=False, dropout_p=0.0)          # gradients through eval-mode BN: unfused path


def test_glu128_other_variants(emu):
    """The selectable alternatives of the 128-channel GLU kernels: backward on 32x32x16 / exact f32, forward on 32x32x16."""
    from desed_task_amd import _lib
    for key, v in (("glu_bwd128_split", 1), ("glu_bwd128_split", 3), ("glu_fwd128", 1)):
        _lib.set_tuning(key, v)
        try:
            P.case_cnn_block("cpu", 4, 2, 9, 8, training=True, dropout_p=0.5, precision="bf16x3", tol=1e-4)
        finally:
            _lib.set_tuning(key, 0)


@pytest.mark.parametrize("layer,B,T,F,cap", [(3, 1, 10, 16, 0), (4, 2, 9, 8, 0), (5, 2, 37, 4, 2), (6, 2, 35, 2, 0), (3, 3, 13, 16, 3)])
def test_glu128_bwd_16x16x32(emu, layer, B, T, F, cap):
    """128-channel GLU backward on the 16x16x32 split-bf16 tiling (wave = 16 columns x 64 rows): ragged last tile, several
    tiles per workgroup (register prefetch of the next tile), partial reduce with one vector slab."""
    from desed_task_amd import _lib
    _lib.set_tuning("glu_bwd128_split", 0)          # 0 = the default 16x16x32 kernel (1 = 32x32x16 tiling, 3 = exact f32)
    _lib.set_tuning("glu_grid_cap", cap)
    try:
        P.case_cnn_block("cpu", layer, B, T, F, training=True, dropout_p=0.5, precision="bf16x3", tol=1e-4)
    finally:
        _lib.set_tuning("glu_bwd128_split", 0)
        _lib.set_tuning("glu_grid_cap", 0)


@pytest.mark.parametrize("layer,B,T,F,cap,narrow", [(1, 2, 9, 64, 3, 0), (2, 3, 7, 32, 2, 0), (1, 1, 4, 32, 0, 0), (2, 2, 6, 32, 0, 1)])
def test_narrow_wgrad_split_bf16(emu, layer, B, T, F, cap, narrow):
    """All-taps weight gradient of the 16->32 / 32->64 layers on the split-bf16 MFMA (three column-shifted transposed copies of the
    halo patch): ragged last frame tile, F edges, several tiles per workgroup (register prefetch); narrow=1 = the exact-f32 kernel."""
    from desed_task_amd import _lib
    _lib.set_tuning("wgrad_cap", cap)
    _lib.set_tuning("wgrad_narrow", narrow)
    try:
        P.case_cnn_block("cpu", layer, B, T, F, training=True, dropout_p=0.5, precision="bf16x3", tol=1e-4)
    finally:
        _lib.set_tuning("wgrad_cap", 0)
        _lib.set_tuning("wgrad_narrow", 0)


@pytest.mark.parametrize("layer,B,T,F,onetap", [(3, 2, 121, 16, 0), (4, 3, 85, 8, 0), (3, 1, 10, 16, 1), (4, 1, 9, 8, 1)])
def test_wide_wgrad_kernel_row(emu, layer, B, T, F, onetap):
    """Wide weight gradients: one kernel row (three taps, column shifts by v_alignbit on the operand octets) per workgroup, several
    tiles per workgroup (B T F > 64 x the split count), ragged last tile; onetap = 1: the one-tap kernel."""
    from desed_task_amd import _lib
    _lib.set_tuning("wgrad_wide", onetap)
    try:
        P.case_cnn_block("cpu", layer, B, T, F, training=True, dropout_p=0.5, precision="bf16x3", tol=1e-4)
    finally:
        _lib.set_tuning("wgrad_wide", 0)
