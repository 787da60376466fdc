"""A slice of the seeded fuzzers (tests/fuzz_conv.py, fuzz_model.py, fuzz_tiled.py, fuzz_post.py; profiles/r06_fuzz.txt holds the long runs) in the
GPU suite: random single convs against the fp32 conv of the same operands, random model family / width / batch / input size through
the fp32 verification path (== oracle head maps and post-processing) and the bf16 path, random slides / tilings / masks through the
slide loop against the oracle's stitching of the same per-tile detections (exact).  Seeds differ from the recorded long runs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')


def test_fuzz_single_convs():
    import fuzz_conv
    assert fuzz_conv.run(80, 101) == 0
    assert fuzz_conv.run_fp8(30, 105) == 0


def test_fuzz_models_and_input_sizes():
    import fuzz_model
    assert fuzz_model.run(12, 102) == 0


def test_fuzz_slide_loop():
    import fuzz_tiled
    assert fuzz_tiled.run(40, 103) == 0


def test_fuzz_exact_post_processing_kernels():
    import fuzz_post
    assert fuzz_post.run(40, 104) == 0
