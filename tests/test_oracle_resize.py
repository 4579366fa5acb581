"""oracle/resize.py against PIL.Image.resize itself, bit for bit (the one place where the reference's own dependency
is importable here, so this oracle is PINNED, see its header)."""
import numpy as np
import pytest
from PIL import Image

from oracle import resize as R

PIL_FILTER = {"bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC, "lanczos": Image.LANCZOS}


@pytest.mark.parametrize("filt", ["bicubic", "bilinear", "lanczos"])
@pytest.mark.parametrize("src,dst", [((640, 480), (299, 224)), ((480, 640), (224, 299)), ((500, 375), (384, 384)),
                                     ((97, 61), (224, 224)), ((1024, 1024), (224, 224)), ((333, 500), (333, 250)),
                                     ((64, 48), (64, 48)), ((3, 2), (7, 5)), ((1365, 1024), (1360, 1024))])
def test_resize_matches_pil(filt, src, dst):
    rng = np.random.default_rng(hash((filt, src, dst)) & 0xffff)
    img = rng.integers(0, 256, (src[1], src[0], 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize(dst, PIL_FILTER[filt]))
    got = R.resize_u8(img, dst[0], dst[1], filt)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_resize_extremes_and_gray():
    img = np.zeros((50, 70, 3), np.uint8); img[::2] = 255                       # ringing must clamp, not wrap
    assert np.array_equal(R.resize_u8(img, 31, 23, "lanczos"), np.asarray(Image.fromarray(img).resize((31, 23), Image.LANCZOS)))
    g = np.random.default_rng(1).integers(0, 256, (40, 33), dtype=np.uint8)
    assert np.array_equal(R.resize_u8(g, 16, 64, "bicubic"), np.asarray(Image.fromarray(g, "L").resize((16, 64), Image.BICUBIC)))
