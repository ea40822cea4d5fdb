"""CPU (fiber-emulator) runs of the GEMM and BiGRU parity cases (K7)."""
from tests import parity_cases as P
from tests.emu_support import emu  # noqa: F401


def test_gemm(emu):
    P.case_gemm("cpu")
    P.case_gemm("cpu", entry="sed_gemm_bf16x3")


def test_linear_packed(emu):
    """The BEATs encoder's packed-weight Linear (round 5)."""
    P.case_linear_packed("cpu", shapes=((300, 128, 64, 0), (257, 256, 32, 1)))


def test_linear_tiles(emu):
    """Round 6: K-tiled plane images + the all-DMA 256 x 256 kernel."""
    P.case_linear_tiles("cpu", shapes=((300, 256, 64, 0), (257, 512, 16, 1), (520, 256, 48, 0)))
    P.case_linear_tiles("cpu", shapes=((300, 256, 64, 0), (257, 512, 16, 1), (520, 256, 48, 0)), form=5)


def test_layernorm_tiles(emu):
    P.case_layernorm_tiles("cpu", shapes=((300, 256), (70, 768)))


def test_linear_n96_tile(emu):
    """The 128 x 96 tile of the split-bf16 GEMM (BEATs N = 768 layers)."""
    P.case_linear_n96_tile("cpu", shapes=((300, 192, 64, 0), (130, 96, 96, 1), (2100, 192, 32, 0)))     # last: 17 row panels -> the XCD-aware walk


def test_bigru_layer0(emu):
    P.case_bigru("cpu", B=2, T=7, I=128)


def test_bigru_layer1(emu):
    P.case_bigru("cpu", B=1, T=5, I=256)


def test_bigru_multi_chunk(emu):
    P.case_bigru("cpu", B=1, T=19, I=128)        # 2 full chunks of 8 steps + a partial one


def test_bigru_192_units(emu):
    """n_RNN_cell = 192 (recipes/dcase2024_task4_baseline/confs/pretrained.yaml:92): 768-thread recurrences, 4-step chunks; layer 0
    (I = 128) and layer 1 (I = 384), partial last chunk."""
    P.case_bigru("cpu", B=2, T=7, I=128, H=192)
    P.case_bigru("cpu", B=1, T=10, I=384, H=192)
