"""drag_resample_u8 against PIL.Image.resize itself — bit-exact (byte work).  PIL is the dependency that does this
resize in the reference (clip `preprocess`, SiglipImageProcessor), so this parity is pinned, not restated."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
PIL_FILTER = {"bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC, "lanczos": Image.LANCZOS}


def _img(seed, w, h, c=3):
    return np.random.default_rng(seed).integers(0, 256, (h, w, c), dtype=np.uint8)


@pytest.mark.parametrize("filt", ["bicubic", "bilinear", "lanczos"])
@pytest.mark.parametrize("src,dst", [((640, 480), (299, 224)), ((480, 640), (224, 299)), ((500, 375), (384, 384)),
                                     ((97, 61), (224, 224)), ((1024, 1024), (224, 224)), ((333, 500), (333, 250)),
                                     ((500, 333), (250, 333)), ((64, 48), (64, 48)), ((3, 2), (7, 5)), ((1365, 1024), (1360, 1024))])
def test_resize_equals_pil(gpu, filt, src, dst):
    from domain_rag_amd import resample
    img = _img(hash((filt, src, dst)) & 0xffff, *src)
    ref = np.asarray(Image.fromarray(img).resize(dst, PIL_FILTER[filt]))
    got = resample.resize_u8(torch.from_numpy(img).to(gpu), dst[0], dst[1], filt).cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_batch_gray_and_extremes(gpu):
    from domain_rag_amd import resample
    batch = np.stack([_img(i, 320, 200) for i in range(5)])
    got = resample.resize_u8(torch.from_numpy(batch).to(gpu), 111, 77, "bicubic").cpu().numpy()
    for i in range(5):
        assert np.array_equal(got[i], np.asarray(Image.fromarray(batch[i]).resize((111, 77), Image.BICUBIC)))
    stripes = np.zeros((50, 70, 3), np.uint8); stripes[::2] = 255                      # ringing must clamp, not wrap
    assert np.array_equal(resample.resize_u8(torch.from_numpy(stripes).to(gpu), 31, 23, "lanczos").cpu().numpy(),
                          np.asarray(Image.fromarray(stripes).resize((31, 23), Image.LANCZOS)))
    g = _img(9, 33, 40, 1)
    assert np.array_equal(resample.resize_u8(torch.from_numpy(g).to(gpu), 16, 64, "bicubic").cpu().numpy()[:, :, 0],
                          np.asarray(Image.fromarray(g[:, :, 0], "L").resize((16, 64), Image.BICUBIC)))
    with pytest.raises(ValueError):
        resample.resize_u8(torch.from_numpy(g), 16, 64)          # host tensor: no CPU fallback


@pytest.mark.parametrize("wh", [(640, 480), (480, 640), (500, 375), (224, 224), (224, 500), (1000, 224), (100, 80)])
def test_clip_preprocess_equals_reference_transform(gpu, wh):
    """Resize(224, BICUBIC) + CenterCrop(224) as openai-CLIP's `preprocess` does on a PIL image; the embedding of the
    device-preprocessed uint8 equals the embedding of the host-preprocessed float tensor's uint8 source"""
    from domain_rag_amd import resample
    from domain_rag_amd.retrieval import clip_preprocess, CLIP_MEAN, CLIP_STD
    img = _img(wh[0] * 7 + wh[1], *wh)
    dev_u8 = resample.clip_preprocess_u8(torch.from_numpy(img).to(gpu)).cpu()
    host = clip_preprocess(Image.fromarray(img))                         # float CHW, normalised
    back = (host.permute(1, 2, 0) * torch.tensor(CLIP_STD) + torch.tensor(CLIP_MEAN)) * 255.0
    assert dev_u8.shape == (224, 224, 3)
    assert torch.equal(dev_u8, back.round().clamp(0, 255).to(torch.uint8))


def test_siglip_resize(gpu):
    from domain_rag_amd import resample
    from domain_rag_amd.engine import siglip_input
    pil = [Image.fromarray(_img(3, 640, 427)), Image.fromarray(_img(4, 1024, 1024))]
    ref = siglip_input(pil, 384)
    for i, im in enumerate(pil):
        got = resample.siglip_resize_u8(torch.from_numpy(np.asarray(im).copy()).to(gpu)).cpu()
        assert torch.equal(got, ref[i])


def test_embedding_independent_of_where_the_resize_ran(gpu):
    """encode_image(GPU-resized uint8) == encode_image(host `preprocess` float tensor), bit for bit: the uint8 front
    end applies ToTensor / Normalize with torch's two divisions"""
    from domain_rag_amd import retrieval as R
    model, preprocess = R.load_clip("ViT-B/32", device=gpu)
    pil = [Image.fromarray(_img(20 + i, w, h)) for i, (w, h) in enumerate([(640, 480), (375, 500), (224, 224), (90, 300)])]
    host = model.encode_image(torch.stack([preprocess(im) for im in pil]))
    dev = model.encode_image(torch.stack([R.clip_preprocess_device(im, gpu) for im in pil]))
    assert torch.equal(host, dev)


def test_resample_argument_errors(gpu):
    """the C entry point refuses inconsistent descriptors instead of reading out of bounds"""
    import ctypes
    from domain_rag_amd import _lib
    lib = _lib.load()
    src = torch.zeros((1, 8, 8, 3), dtype=torch.uint8, device=gpu)
    dst = torch.zeros((1, 4, 4, 3), dtype=torch.uint8, device=gpu)
    tab = torch.zeros((4, 2), dtype=torch.int32, device=gpu)

    def call(**kw):
        a = _lib.ResampleArgs()
        a.src, a.dst, a.batch, a.channels = src.data_ptr(), dst.data_ptr(), 1, 3
        a.src_h, a.src_w, a.src_image_stride, a.src_row_stride = 8, 8, 192, 24
        a.out_h, a.out_w, a.dst_image_stride, a.dst_row_stride = 4, 4, 48, 12
        for k, v in kw.items():
            setattr(a, k, v)
        return lib.drag_resample_u8(ctypes.byref(a), torch.cuda.current_stream().cuda_stream)
    assert call(channels=5) != 0 and b"channels" in lib.drag_last_error()
    assert call(kx=tab.data_ptr()) != 0                                   # weights without bounds
    assert call(kx=tab.data_ptr(), bx=tab.data_ptr(), ksize_x=1, ky=tab.data_ptr(), by=tab.data_ptr(), ksize_y=1) != 0   # both passes, no tmp
    assert call(src_col0=6) != 0 and b"window" in lib.drag_last_error()   # copy window 6..9 leaves the 8-wide source
    assert call() == 0                                                    # plain 4x4 window copy is fine
