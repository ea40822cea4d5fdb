"""CPU (fiber-emulator) runs of the K1-K5 parity cases.  Same cases run on the GPU in test_gpu_parity.py."""
from tests import parity_cases as P
from tests.emu_support import emu  # noqa: F401


def test_mfma_maps(emu):
    P.case_mfma_selftest("cpu")


def test_mel_matches_oracle(emu):
    P.case_mel("cpu")
    P.case_mel("cpu", batch=3, n_samples=256 * 21 + 100)       # ragged clip length, a batch the XCD walk does not divide
    P.case_mel("cpu", batch=8, n_samples=256 * 6)               # one clip per XCD (S = 1)
    P.case_mel("cpu", batch=9, n_samples=256 * 5 + 17)          # 9 clips in 8 segments each


def test_mel_walk_batch_independent(emu):
    P.case_mel_walk_batch_independent("cpu", batch=13, n_samples=256 * 9 + 40, probe=(0, 5, 12))


def test_mel_workgroup_kernel_matches_oracle(emu):
    """The round-1..4 kernel (`sed_mel_fwd`, one frame per workgroup; `mel_wave` = 2).  The default is the wave-per-frame kernel."""
    from desed_task_amd import _lib
    _lib.set_tuning("mel_wave", 2)
    try:
        P.case_mel("cpu")
        P.case_mel("cpu", batch=3, n_samples=256 * 21 + 100)
    finally:
        _lib.set_tuning("mel_wave", 0)


def test_logscale_generic(emu):
    P.case_logscale_generic("cpu")


def test_mixup_and_specaug(emu):
    P.case_mixup_specaug("cpu")


def test_backward_entries_whole_and_split(emu):
    """sed_head_bwd / sed_gru_bwd in one call == kernel + sed_head_bwd_reduce / sed_gru_bias_reduce (what ops.py launches)."""
    P.case_backward_entries_whole_and_split("cpu")


def test_cnn_prologue_equals_its_three_launches(emu):
    """sed_cnn_prologue_bf16: weight packs + seeded SpecAugment bands + the private copy of the input in one launch."""
    P.case_cnn_prologue("cpu")


def test_postprocess_median_threshold_events(emu):
    """K13: batched median filter / thresholds / event regions, bit-exact vs scipy and the oracle."""
    P.case_postprocess("cpu")


def test_dataset_scaler(emu, tmp_path):
    """statistic "dataset" of SEDTask4._init_scaler: fit over the training loader through the mel kernel, save, reload."""
    P.case_dataset_scaler("cpu", tmp_path)


def test_mt_loss_modes(emu):
    """K9 losses, self_sup_loss mse and bce, against torch."""
    P.case_mt_loss("cpu")


def test_round4_entry_points_on_degenerate_arguments(emu):
    P.case_edge_round4_entries("cpu")
