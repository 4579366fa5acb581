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
    # torchvision _compute_resized_output_size: new_long = int(size * long / short); CenterCrop offsets = int(round(d / 2.0))
    assert resample.clip_resize_plan(640, 480) == (298, 224, (37, 0, 261, 224))          # int(298.67) = 298
    assert resample.clip_resize_plan(480, 640) == (224, 298, (0, 37, 224, 261))
    assert resample.clip_resize_plan(224, 224) == (224, 224, (0, 0, 224, 224))
    assert resample.clip_resize_plan(500, 375)[:2] == (298, 224)                         # int(298.67)
    assert resample.clip_resize_plan(640, 427) == (335, 224, (56, 0, 280, 224))          # int(335.74) = 335; round(55.5) = 56
