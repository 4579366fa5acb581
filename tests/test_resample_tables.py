"""host half of the GPU resize: the vectorised coefficient tables must equal the oracle's loop restatement (which
tests/test_oracle_resize.py pins against PIL) exactly, for every filter and a spread of up/down-scaling sizes"""
import numpy as np
import pytest

from domain_rag_amd import resample
from oracle import resize as R


@pytest.mark.parametrize("filt", ["bicubic", "bilinear", "lanczos"])
def test_tables_equal_oracle(filt):
    rng = np.random.default_rng(0)
    sizes = [(640, 299), (480, 224), (375, 384), (500, 384), (61, 224), (1024, 224), (2, 5), (7, 3), (333, 332), (224, 224)]
    sizes += [(int(a), int(b)) for a, b in rng.integers(1, 900, (25, 2))]
    for n_in, n_out in sizes:
        b0, k0 = R.precompute_coeffs(n_in, n_out, filt)
        b1, k1 = resample.coeff_tables(n_in, n_out, filt)
        assert np.array_equal(b0, b1) and np.array_equal(k0, k1), (n_in, n_out)


def test_clip_resize_plan_matches_torchvision_rule():
    # torchvision Resize(int): shorter side -> size, longer = int(size * long / short) ... openai-CLIP then center-crops
    assert resample.clip_resize_plan(640, 480) == (299, 224, (38, 0, 262, 224))
    assert resample.clip_resize_plan(480, 640) == (224, 299, (0, 38, 224, 262))
    assert resample.clip_resize_plan(224, 224) == (224, 224, (0, 0, 224, 224))
    assert resample.clip_resize_plan(500, 375)[:2] == (299, 224)
