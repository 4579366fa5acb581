"""Stage-1 retrieval on the HIP path: CLIP ViT-B/32 embedding, exact inner-product top-k over a
corpus kept resident in HBM, ResNet50-stem style re-rank.  Objects mirror what
``retrieval/clip100_resnet_style_all_shots.py`` calls so its call sites work unchanged:

* ``load_clip("ViT-B/32", device) -> (model, preprocess)``; ``model.encode_image(x)``   (ref :206-211,:171)
* ``IndexFlatIP(d)``, ``.add(x)``, ``.search(q, k) -> (D, I)``                            (ref :425-434)
* ``StemStyle`` = ``ResNetEncoder`` + ``calc_mean_std``                                  (ref :51-74,:180-203)
* ``clip_first_stage_retrieval`` / ``resnet_second_stage_rerank``                        (ref :396-497)

Differences that are deliberate: the index is built once, not per query; corpus
embedding is batched instead of batch-1 with a host sync per image; ``CUDA_VISIBLE_DEVICES`` is
honoured.  Multi-GPU: each rank embeds a contiguous shard of the sorted path list and ONE
all-gather (RCCL over xGMI with backend "nccl") makes the whole corpus resident on every GPU, in
the global row order, so top-k indices equal the single-GPU result bit-for-bit.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import ops
from .vit import ClipVitF32HIP, VitConfig, VitHIP, init_generic_params, openai_clip_to_generic

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ------------------------------------------------------------------ host helpers (reference semantics)
def clean_image_path(path):
    """ref :77-86 — strip the stray ``pipeline/`` prefix, then map ../../datasets/coco -> ./coco"""
    if isinstance(path, str):
        if "../../pipeline/datasets" in path:
            return path.replace("../../pipeline/datasets", "../../datasets")
        if "../../datasets/coco" in path:
            return path.replace("../../datasets/coco", "./coco")
    return path


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """contiguous shards, the first n % world ranks take one extra (split_samples_for_gpus rule,
    outpainting_updown_sampling_redux.py:157-177)"""
    base, rem = divmod(n, max(world, 1))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def clip_preprocess(pil_image, size: int = 224) -> torch.Tensor:
    """openai clip ``_transform``: Resize(size, bicubic) -> CenterCrop(size) -> RGB -> ToTensor -> Normalize.
    Host-side (PIL), exactly where the reference does it."""
    from PIL import Image
    img = pil_image.convert("RGB")
    w, h = img.size
    # torchvision Resize(int): the short side becomes `size`, the long one int(size * long / short) — TRUNCATED, not rounded
    nw, nh = (size, int(size * h / w)) if w <= h else (int(size * w / h), size)
    img = img.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    img = img.crop((left, top, left + size, top + size))
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).float().div_(255.0)
    x = (x - torch.tensor(CLIP_MEAN)) / torch.tensor(CLIP_STD)
    return x.permute(2, 0, 1).contiguous()


def clip_preprocess_device(pil_image, device="cuda", size: int = 224) -> torch.Tensor:
    """same transform with the resize on the GPU: decoded RGB bytes are uploaded as they are and
    ``drag_resample_u8`` (bit-identical to PIL's BICUBIC) resizes + centre-crops -> uint8 [size, size, 3] on device.
    ToTensor / Normalize run inside ``drag_patchify_u8`` with torch's arithmetic, so ``encode_image`` of this equals
    ``encode_image(clip_preprocess(pil_image))`` bit for bit."""
    from . import resample
    raw = torch.from_numpy(np.array(pil_image.convert("RGB"), dtype=np.uint8, copy=True)).to(device)
    return resample.clip_preprocess_u8(raw, size)


def cv2_linear_tables(h: int, w: int, out_w: int, out_h: int):
    """OpenCV's 8-bit INTER_LINEAR tap tables (imgproc/resize.cpp): half-pixel centres, two taps per axis, 11-bit fixed-point
    weights.  Returns (sx, sx1, a0, a1) per output column and (y0, y1, b0, b1) per output row, int64 arrays."""
    def axis(n_in, n_out):
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        return s, f

    sx, fx = axis(w, out_w)
    lo = sx < 0
    fx[lo], sx[lo] = 0.0, 0
    hi = sx >= w - 1
    fx[hi], sx[hi] = 0.0, w - 1
    a1 = np.rint(fx * 2048.0).astype(np.int64)
    a0 = np.rint((1.0 - fx) * 2048.0).astype(np.int64)
    sx1 = np.minimum(sx + 1, w - 1)
    sy, fy = axis(h, out_h)
    b1 = np.rint(fy * 2048.0).astype(np.int64)
    b0 = np.rint((1.0 - fy) * 2048.0).astype(np.int64)
    y0, y1 = np.clip(sy, 0, h - 1), np.clip(sy + 1, 0, h - 1)
    return (sx, sx1, a0, a1), (y0, y1, b0, b1)


def cv2_resize_linear_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``cv2.resize(img, (out_w, out_h))`` (INTER_LINEAR) for uint8 HWC, restated from OpenCV's published algorithm
    (imgproc/resize.cpp): horizontal pass into int32, vertical pass ``((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2``.
    PARITY UNPINNED: cv2 is not importable in the build container (and opencv-python may route this call through IPP); what
    matters for the style statistics is that, like cv2 and unlike PIL's BILINEAR, it does not low-pass when shrinking.
    ``drag_cv_resize_linear_u8_f32`` applies the same tables on the GPU (``StemStyle.features_from_files``)."""
    h, w = img.shape[:2]
    if (w, h) == (out_w, out_h):
        return img.copy()
    (sx, sx1, a0, a1), (y0, y1, b0, b1) = cv2_linear_tables(h, w, out_w, out_h)
    src = img.astype(np.int64)
    rows = src[:, sx] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]            # [h, out_w, c] int
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ CLIP
class ClipImageModel:
    """``model`` half of ``clip.load``; only the image tower is on Domain-RAG's path."""

    def __init__(self, vit: "VitHIP | ClipVitF32HIP"):
        self.visual = vit
        self.device = vit.dev

    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        """float [B,3,224,224] (``preprocess`` output, any device) or uint8 [B,224,224,3] -> fp32 [B,512] on device"""
        return self.visual(image.to(self.device))

    def embed_normalized(self, image: torch.Tensor) -> torch.Tensor:
        """encode_image followed by ``x / x.norm(dim=-1, keepdim=True)`` (ref :171-172)"""
        return ops.l2_normalize_(self.encode_image(image).clone())


def weights_fingerprint(g: dict) -> str:
    """identity of what an embedding depends on (stage 1 stores it next to its feature caches).  Every tensor takes part
    (a fine-tuned tower that shares its projection and patch embedding with the stock one must not pass for it): small
    tensors whole, large ones through ~4096 evenly strided elements plus their float64 sum"""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(g):
        if not torch.is_tensor(g[name]):
            continue
        a = g[name].detach().to("cpu", torch.float32).contiguous().reshape(-1)
        h.update(name.encode()); h.update(str(tuple(g[name].shape)).encode())
        if a.numel() <= 65536:
            h.update(a.numpy().tobytes())
        else:
            h.update(a[:: a.numel() // 4096].contiguous().numpy().tobytes())
            h.update(a.double().sum().numpy().tobytes())
    return h.hexdigest()[:16]


def load_clip(name: str = "ViT-B/32", device="cuda", weights: str | dict | None = None, seed: int = 0, precision: str | None = None):
    """(model, preprocess).  ``weights``: an openai-CLIP state_dict (or a path to one saved with torch.save);
    None -> seeded synthetic weights of the ViT-B/32 architecture (no checkpoints offline).
    ``precision``: "fp32" (default; $DRAG_CLIP_PRECISION overrides) — the arithmetic openai-CLIP uses on its CPU path and at
    least the precision of its fp16 CUDA path; "bf16" — the bf16 MFMA tower (≈4x the throughput, embeddings ≈1e-2 away from
    the fp32 values, so near-ties among the top-k can order differently than the reference's)."""
    precision = precision or os.environ.get("DRAG_CLIP_PRECISION", "fp32")
    if precision not in ("fp32", "bf16"):
        raise ValueError("precision must be fp32 or bf16")
    if name != "ViT-B/32":
        raise ValueError("Domain-RAG uses CLIP ViT-B/32 only")
    cfg = VitConfig.clip_vit_b32()
    if isinstance(weights, str):
        try:                       # openai distributes ViT-B-32.pt as a TorchScript archive
            weights = torch.jit.load(weights, map_location="cpu").state_dict()
        except RuntimeError:
            weights = torch.load(weights, map_location="cpu")
    if weights is not None:
        weights = {k: v for k, v in weights.items() if torch.is_tensor(v)}
        cfg = VitConfig.from_openai_state_dict(weights)
        g = openai_clip_to_generic(weights, cfg)
    else:
        g = init_generic_params(cfg, seed, device=device if str(device) != "cpu" else "cpu")
    tower = ClipVitF32HIP(cfg, g, device) if precision == "fp32" else VitHIP(cfg, g, device)
    model = ClipImageModel(tower)
    h = weights_fingerprint(g)
    model.fingerprint = ("synthetic-seed%d-" % seed if weights is None else "") + h
    model.precision = precision
    return model, clip_preprocess


def load_clip_device_preprocess(device="cuda"):
    """the ``preprocess`` to pass around when the resize should run on the GPU (corpus embedding)"""
    import functools
    return functools.partial(clip_preprocess_device, device=device)


# ------------------------------------------------------------------ exact inner-product index
class IndexFlatIP:
    """faiss.IndexFlatIP look-alike with the corpus resident in HBM.  Scores follow the order pinned in
    oracle/topk.c; ties resolve to the lower index."""

    def __init__(self, d: int, device="cuda"):
        self.d, self.device = d, torch.device(device)
        self._chunks: list[torch.Tensor] = []
        self._corpus: torch.Tensor | None = None

    @property
    def ntotal(self) -> int:
        return sum(c.shape[0] for c in self._chunks) if self._corpus is None else self._corpus.shape[0]

    def add(self, x) -> None:
        t = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)) if not torch.is_tensor(x) else x.float()
        if t.dim() != 2 or t.shape[1] != self.d:
            raise ValueError(f"expected [n, {self.d}]")
        if self._corpus is not None:
            self._chunks, self._corpus = [self._corpus], None
        self._chunks.append(t.to(self.device).contiguous())

    def _resident(self) -> torch.Tensor:
        if self._corpus is None:
            self._corpus = self._chunks[0] if len(self._chunks) == 1 else torch.cat(self._chunks, 0).contiguous()
            self._chunks = []
        return self._corpus

    def search_device(self, q: torch.Tensor, k: int):
        return ops.cosine_topk(self._resident(), q.to(self.device).float().contiguous(), k)

    def search(self, q, k: int):
        """(D float32 [Q,k] descending, I int64 [Q,k]) as numpy, like faiss"""
        qt = torch.as_tensor(np.ascontiguousarray(q, dtype=np.float32))
        D, I = self.search_device(qt, k)
        return D.cpu().numpy(), I.cpu().numpy()


# ------------------------------------------------------------------ ResNet stem style
class StemStyle:
    """ResNet50 stem (conv1+bn1+relu+maxpool, eval) + channel mean / unbiased std -> 128-d style vector."""

    def __init__(self, state: dict | None = None, device="cuda", seed: int = 0):
        dev = torch.device(device)
        if state is None:   # seeded synthetic stem (torchvision IMAGENET1K_V1 weights are not available offline)
            g = torch.Generator().manual_seed(seed)
            state = {"conv1.weight": torch.randn(64, 3, 7, 7, generator=g) * math.sqrt(2.0 / 147),
                     "bn1.weight": 1 + 0.1 * torch.randn(64, generator=g), "bn1.bias": 0.1 * torch.randn(64, generator=g),
                     "bn1.running_mean": 0.1 * torch.randn(64, generator=g), "bn1.running_var": 1 + 0.1 * torch.rand(64, generator=g)}
        self.state = {k: v.float() for k, v in state.items() if k.split(".")[0] in ("conv1", "bn1")}
        scale = self.state["bn1.weight"] / torch.sqrt(self.state["bn1.running_var"] + 1e-5)
        shift = self.state["bn1.bias"] - self.state["bn1.running_mean"] * scale
        self.w = self.state["conv1.weight"].contiguous().to(dev)
        self.scale, self.shift = scale.contiguous().to(dev), shift.contiguous().to(dev)
        self.device = dev
        self.gpu_files = False      # set by stage 1 (--decode gpu): candidate files are decoded / resized on the device in batches
        self.force_restated = False  # set by stage 1 (--style-resize restated): never call cv2, also on the host route

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        """fp32 [B,3,H,W] in [0,1] -> fp32 [B,128] = cat(mean, std)   (ref :197-200)"""
        return ops.resnet_stem_style(img.to(self.device).float().contiguous(), self.w, self.scale, self.shift, 1e-5)

    def features_from_files(self, image_paths: list, chunk: int = 2048) -> list:
        """style vectors of many files with everything on the GPU: native file reads -> JPEG decode (``jpeg.decode_files``,
        PIL's bytes) -> OpenCV-linear resize to 256x256 + /255 (``drag_cv_resize_linear_u8_f32``) -> stem.  Returns one float32
        [128] vector (numpy) or None per path, in order.  Files the device decoder declines (PNG, CMYK, damaged, EXIF-
        rotated, ...) take ``features_from_path`` on the host with the SAME restated resize, so a cache never mixes algorithms."""
        import ctypes
        from . import _lib, jpeg
        lib = _lib.load()
        out: list = [None] * len(image_paths)
        paths = [clean_image_path(p) for p in image_paths]
        for c0 in range(0, len(paths), chunk):
            sub = paths[c0: c0 + chunk]
            staged = jpeg.stage_paths(sub, self.device, slot=2)
            host = staged.buf.numpy()
            rotated = set()
            for k in range(len(sub)):          # cv2.imread applies the EXIF orientation: such files go the host way
                o0, o1 = int(staged.offsets[k]), int(staged.offsets[k + 1])
                if o1 > o0 and host[o0: min(o1, o0 + 65536)].tobytes().find(b"Exif\x00\x00") >= 0:
                    rotated.add(k)
            dec = jpeg.decode_files(staged, self.device)
            idx = [k for k in range(len(sub)) if dec.status[k] == 0 and k not in rotated and k not in staged.errors]
            if idx:
                n = len(idx)
                L = 256
                tab = np.zeros((n, 8, L), dtype=np.int32)
                hw = np.zeros((n, 2), dtype=np.int32)
                off = np.zeros(n, dtype=np.int64)
                base = dec._out.data_ptr()
                for r, k in enumerate(idx):
                    h, w = int(dec.height[k]), int(dec.width[k])
                    (sx, sx1, a0, a1), (y0, y1, b0, b1) = cv2_linear_tables(h, w, L, L)
                    tab[r] = np.stack([sx, sx1, a0, a1, y0, y1, b0, b1]).astype(np.int32)
                    hw[r] = (h, w)
                    off[r] = int(dec._off[k])
                d_tab, d_hw, d_off = (torch.from_numpy(a).to(self.device) for a in (tab, hw, off))
                x = torch.empty((n, 3, L, L), dtype=torch.float32, device=self.device)
                ops.check(lib.drag_cv_resize_linear_u8_f32(ctypes.c_void_p(base), ops._p(d_off), ops._p(d_hw), ops._p(d_tab), ops._p(x),
                                                           n, L, L, ops._stream()), "drag_cv_resize_linear_u8_f32")
                vec = torch.cat([self(x[b: b + 256]) for b in range(0, n, 256)], 0).cpu().numpy()
                for r, k in enumerate(idx):
                    out[c0 + k] = vec[r]
            done = set(idx)
            for k in range(len(sub)):
                if k not in done:
                    out[c0 + k] = self.features_from_path(sub[k], restated_resize=True)
        return out

    def features_from_path(self, image_path: str, restated_resize: bool = False):
        """compute_resnet_features (ref :180-203): imread -> RGB -> resize 256x256 (bilinear) -> /255"""
        image_path = clean_image_path(image_path)
        try:
            try:
                if restated_resize or self.force_restated:
                    raise ImportError("restated resize requested")
                import cv2
                img = cv2.imread(image_path)
                if img is None:
                    print(f"警告：无法读取图像 {image_path}")
                    return None
                img = cv2.resize(cv2.cvtColor(img, cv2.COLOR_BGR2RGB), (256, 256))
            except ImportError:   # no OpenCV: same decode via PIL, and OpenCV's INTER_LINEAR restated (2 taps, NO antialiasing —
                from PIL import Image   # PIL's BILINEAR widens its support when shrinking, which changes texture statistics)
                from PIL import ImageOps                      # cv2.imread applies the EXIF orientation
                img = cv2_resize_linear_u8(np.asarray(ImageOps.exif_transpose(Image.open(image_path)).convert("RGB")), 256, 256)
            x = torch.from_numpy(np.array(img, copy=True)).float().permute(2, 0, 1).unsqueeze(0) / 255.0
            return self(x)[0].cpu().numpy()
        except Exception as e:  # reference behaviour: log and skip
            print(f"计算ResNet特征时出错: {e}, 图像: {image_path}")
            return None


# ------------------------------------------------------------------ two-stage retrieval (reference call shapes)
_index_cache: dict = {}


def _index_for(dataset_features: dict, device) -> IndexFlatIP:
    """build the resident index ONCE per corpus (the reference rebuilds it for every query, :419-430).  The cache keeps
    the arrays it was built from alive and compares by identity: an ``id()`` alone can be reused by a new array."""
    used = [(name, f) for name, f in dataset_features.items() if f is not None and len(f) > 0]
    ent = _index_cache.get("entry")
    if ent is not None and ent["device"] == str(device) and len(ent["used"]) == len(used) and \
            all(n0 == n1 and f0 is f1 for (n0, f0), (n1, f1) in zip(ent["used"], used)):
        return ent["index"]
    feats = [np.asarray(f, dtype=np.float32) for _, f in used]
    idx = IndexFlatIP(feats[0].shape[1], device)
    idx.add(np.vstack(feats))
    _index_cache["entry"] = dict(used=used, index=idx, device=str(device))
    return idx


def clip_first_stage_retrieval(query_feature, dataset_features: dict, dataset_paths: dict, top_k: int = 100, device="cuda"):
    """ref :396-451 — list of {similarity, image_path, source_dataset, index}, descending inner product"""
    all_paths, all_sources = [], []
    for name, feats in dataset_features.items():
        if feats is not None and len(feats) > 0:
            all_paths.extend(dataset_paths[name])
            all_sources.extend([name] * len(dataset_paths[name]))
    if not all_paths:
        print("错误：没有可用的数据集特征")
        return []
    index = _index_for(dataset_features, device)
    D, I = index.search(np.asarray([query_feature], dtype=np.float32), min(top_k, index.ntotal))
    return [{"similarity": float(D[0][i]), "image_path": all_paths[idx], "source_dataset": all_sources[idx], "index": int(idx)}
            for i, idx in enumerate(I[0]) if 0 <= idx < len(all_paths)]


def resnet_second_stage_rerank(query_image_path, first_stage_results, stem: StemStyle, style_cache: dict | None = None):
    """ref :454-497 — L2 distance between style vectors, stable ascending sort, similarity = 1/(1+d), rank = i+1.
    Candidates whose image cannot be read are dropped; ``style_cache`` (path -> vector) avoids re-reading candidates."""
    query_image_path = clean_image_path(query_image_path)
    gpu_files = getattr(stem, "gpu_files", False)
    qf = stem.features_from_path(query_image_path, restated_resize=True) if gpu_files else stem.features_from_path(query_image_path)
    if qf is None:
        print(f"警告：无法计算查询图像的ResNet特征: {query_image_path}")
        return first_stage_results
    rer = []
    batch: dict = {}
    if gpu_files:                                # the uncached candidates of this query in ONE device batch
        need = [clean_image_path(r["image_path"]) for r in first_stage_results]
        need = [p for p in dict.fromkeys(need) if style_cache is None or style_cache.get(p) is None]
        if need:
            batch = dict(zip(need, stem.features_from_files(need)))
            if style_cache is not None:
                style_cache.update({p: f for p, f in batch.items() if f is not None})
    for r in first_stage_results:
        path = clean_image_path(r["image_path"])
        f = style_cache.get(path) if style_cache is not None else None
        if f is None:
            f = batch[path] if path in batch else stem.features_from_path(path)
            if style_cache is not None and f is not None:
                style_cache[path] = f
        if f is not None:
            rer.append({"clip_similarity": r["similarity"], "resnet_distance": float(np.linalg.norm(qf - f)),
                        "image_path": path, "source_dataset": r.get("source_dataset", "unknown")})
    rer.sort(key=lambda x: x["resnet_distance"])          # list.sort is stable, like the reference
    return [{"rank": i + 1, "similarity": float(1.0 / (1.0 + r["resnet_distance"])), "image_path": r["image_path"],
             "source_dataset": r["source_dataset"]} for i, r in enumerate(rer)]


# ------------------------------------------------------------------ corpus embedding (+ all-gather)
def embed_images(model: ClipImageModel, tensors: torch.Tensor, batch: int = 256) -> torch.Tensor:
    """L2-normalised fp32 embeddings [n,512] on device for a stack of preprocessed images"""
    outs = []
    for i in range(0, tensors.shape[0], batch):
        outs.append(model.embed_normalized(tensors[i:i + batch]))
    return torch.cat(outs, 0) if outs else torch.empty((0, 512), device=model.device)


def pack_shard(local: torch.Tensor, n_total: int, world: int, rank: int) -> torch.Tensor:
    """this rank's rows padded with zero rows to the largest shard's row count (all-gather needs equal sizes)"""
    cap = shard_bounds(n_total, world, 0)[1]                      # largest shard
    s, e = shard_bounds(n_total, world, rank)
    if local.shape[0] != e - s:
        raise ValueError(f"rank {rank}: local shard has {local.shape[0]} rows, shard_bounds() says {e - s}")
    send = torch.zeros((cap, local.shape[1]), dtype=local.dtype, device=local.device)
    send[: e - s] = local
    return send


def unpack_shards(recv: torch.Tensor, n_total: int) -> torch.Tensor:
    """[world, cap, d] gathered padded shards -> [n_total, d] in the GLOBAL row order (rank 0's rows first), so an index
    into the result is the row's position in the sorted corpus path list whatever the world size"""
    world = recv.shape[0]
    parts = []
    for r in range(world):
        s, e = shard_bounds(n_total, world, r)
        parts.append(recv[r, : e - s])
    return torch.cat(parts, 0)


def allgather_rows(local: torch.Tensor, n_total: int, group=None, force: bool = False) -> torch.Tensor:
    """Concatenate per-rank row shards (split by ``shard_bounds``) into the global [n_total, d] matrix on every
    rank with ONE all_gather of equal-sized (padded) shards: ``all_gather_into_tensor`` on RCCL (backend "nccl",
    GPU tensors), list-form ``all_gather`` on gloo.  A single rank returns ``local`` itself unless ``force`` (which
    sends the one shard through the collective anyway: the RCCL call path is then exercised on a 1-GPU box)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1 and not force:
        return local
    send = pack_shard(local, n_total, world, rank)
    if local.is_cuda and dist.get_backend(group) == "gloo":       # gloo gathers host tensors only (bench.py --debug-share-gpu)
        host = send.cpu()
        recv = torch.empty((world,) + tuple(host.shape), dtype=local.dtype)
        dist.all_gather(list(recv.unbind(0)), host, group=group)
        return unpack_shards(recv.to(local.device), n_total)
    recv = torch.empty((world,) + tuple(send.shape), dtype=local.dtype, device=local.device)
    if local.is_cuda and hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(recv, send, group=group)
    else:
        dist.all_gather(list(recv.unbind(0)), send, group=group)
    return unpack_shards(recv, n_total)


def _embed_files_gpu_decode(model: ClipImageModel, mine: list[str], feats: torch.Tensor, ok: torch.Tensor, batch: int,
                            files_per_batch: int = 16384, readers: int = 32, max_pixels: int | None = None) -> dict:
    """corpus rows of this rank with the JPEG decode on the GPU.  Reader threads put the FILES of a chunk straight into a pinned
    buffer (``jpeg.stage_paths``), the chunk is uploaded as one blob, decoded (``jpeg.decode_files``: byte-identical to
    ``Image.open(f).convert("RGB")``), resized + centre-cropped per size class by the PIL-exact resample kernel and embedded.
    The entropy decoder runs one image per LANE (~0.15 s for a 125-KiB file whatever the batch), so its throughput is its batch:
    16384 files are 256 waves = every CU once; the reads of chunk c + 1 overlap the GPU work of chunk c.
    A chunk's decode buffers take about 7.5 bytes per pixel, so a chunk whose headers add up to more than ``max_pixels`` (default
    2^33 = 64 GB of buffers; 16384 files of 640x480 are 5e9 pixels = 38 GB and stay one piece; $DRAG_JPEG_MAX_PIXELS overrides)
    is decoded in consecutive pieces that each stay within it — a corpus of multi-megapixel files (mini-ImageNet has some) costs more launches, never an
    out-of-memory abort (the reference decodes one file at a time and takes any size).
    Files outside the device decoder's coverage (CMYK, arithmetic-coded, PNG, damaged, ...) go through PIL on the host, one by one, like
    the reference does for every file; what PIL cannot open is skipped with the reference's message (:290-292).  Returns counters."""
    import concurrent.futures as cf
    import io
    from PIL import Image
    from . import jpeg, resample
    dev = model.device
    stats = {"gpu_decoded": 0, "host_decoded": 0, "failed": 0, "decode_pieces": 0}
    chunks = [list(range(i, min(i + files_per_batch, len(mine)))) for i in range(0, len(mine), files_per_batch)]
    if max_pixels is None:
        max_pixels = int(os.environ.get("DRAG_JPEG_MAX_PIXELS", str(1 << 33)))

    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)       # decode + resize of chunk c + 1 run here, under the tower of chunk c on `main`

    def decode_chunk(rows, staged):
        """files -> uint8 crops on the side stream (the host blocks in the decoder's two read-backs; the GPU does not)"""
        for k, ex in sorted(staged.errors.items()):
            print(f"处理图像 {mine[rows[k]]} 时出错: {ex}")
            stats["failed"] += 1
        n = len(rows)
        with torch.cuda.stream(side):
            crops = torch.empty((n, 224, 224, 3), dtype=torch.uint8, device=dev)
            have = np.zeros(n, dtype=bool)
            pieces = [(0, n)]
            px = jpeg.parse_pixels(staged, dev)
            if int(px.sum()) > max_pixels:
                pieces = jpeg.budget_bounds(px, max_pixels)
            stats["decode_pieces"] += len(pieces)
            for a, b in pieces:
                dec = jpeg.decode_files(staged if (a, b) == (0, n) else staged.slice(a, b), dev)
                for (_h, _w), idx, imgs in dec.groups():
                    crops[torch.from_numpy(idx + a).to(dev)] = resample.clip_preprocess_u8(imgs)
                    have[idx + a] = True
                del dec
            stats["gpu_decoded"] += int(have.sum())
            for g in np.nonzero(~have)[0].tolist():         # the device decoder declined: PIL decides (same bytes by definition)
                if g in staged.errors:
                    continue
                try:
                    img = Image.open(io.BytesIO(staged.file_bytes(g))).convert("RGB")
                    crops[g] = clip_preprocess_device(img, dev)
                    have[g] = True
                    stats["host_decoded"] += 1
                except Exception as ex:
                    print(f"处理图像 {mine[rows[g]]} 时出错: {ex}")
                    stats["failed"] += 1
            keep = np.nonzero(have)[0]
            if len(keep) == 0:
                return None
            if len(keep) < n:
                crops = crops[torch.from_numpy(keep).to(dev)]
            done = torch.cuda.Event()
            done.record(side)
        crops.record_stream(main)                           # allocated on `side`, consumed on `main`
        return crops, [rows[g] for g in keep.tolist()], done

    def embed_chunk(job):
        crops, grows, done = job
        main.wait_event(done)
        emb = embed_images(model, crops, batch)             # asynchronous: the next chunk's decode starts while this runs
        ii = torch.tensor(grows, device=dev)
        feats[ii] = emb
        ok[ii] = 1.0

    with cf.ThreadPoolExecutor(max_workers=1) as stager:       # one Python thread drives the library's native reader threads
        def stage(ci):
            return jpeg.stage_paths([clean_image_path(mine[j]) for j in chunks[ci]], dev, slot=ci & 1, threads=readers)
        nxt = stager.submit(stage, 0) if chunks else None
        for ci, rows in enumerate(chunks):
            staged = nxt.result()
            # slot (ci + 1) & 1 was uploaded one chunk ago and that upload was waited for (descriptor read-back): free to refill
            nxt = stager.submit(stage, ci + 1) if ci + 1 < len(chunks) else None
            job = decode_chunk(rows, staged)
            if job is not None:
                embed_chunk(job)
    torch.cuda.synchronize(dev)
    return stats


def compute_corpus_features(model: ClipImageModel, preprocess, image_paths: list[str], batch: int = 256, decode_workers: int | None = None,
                            decode_procs: int = 0, gpu_decode: bool = False):
    """compute_coco_clip_features (ref :236-298), batched and rank-sharded.  Returns (features float32 [n,512]
    numpy in path order, valid_paths).  Unreadable images are skipped like the reference (:290-292).
    ``gpu_decode``: the files are decoded on the GPU (``jpeg.py``; the host only reads them) — same pixels as PIL, hence the
    same embeddings as every other route.
    ``decode_procs`` > 0: JPEG decode + CLIP's bicubic resize / centre crop run in that many worker processes
    (io_pool.ClipDecodePool; PIL's own resize — the bytes the GPU resample kernel reproduces, so the embeddings are the same)
    and this thread only stacks uint8 batches.  Otherwise file decoding runs ``decode_workers`` images ahead on a thread pool
    and ``preprocess`` (host PIL, or the device resize) is applied here, in path order."""
    import collections
    import concurrent.futures as cf
    import os as _os
    import torch.distributed as dist
    from PIL import Image
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    s, e = shard_bounds(len(image_paths), world, rank)
    feats = torch.zeros((e - s, 512), dtype=torch.float32, device=model.device)
    ok = torch.zeros((e - s, 1), dtype=torch.float32, device=model.device)
    buf, idxs = [], []

    def flush():
        if buf:
            emb = model.embed_normalized(torch.stack(buf))
            ii = torch.tensor(idxs, device=model.device)
            feats[ii] = emb
            ok[ii] = 1.0
            buf.clear(); idxs.clear()

    def decode(p):
        img = Image.open(clean_image_path(p)).convert("RGB")
        img.load()
        return img

    mine = image_paths[s:e]
    if gpu_decode:
        st = _embed_files_gpu_decode(model, mine, feats, ok, batch)
        print(f"GPU JPEG decode: {st['gpu_decoded']} on the device, {st['host_decoded']} through PIL (outside the device decoder's coverage), "
              f"{st['failed']} unreadable")
    elif decode_procs > 0:
        from .io_pool import ClipDecodePool
        # batches are assembled with plain memcpys into two pinned staging buffers (torch.stack of 256 CPU tensors spins up
        # the whole OpenMP pool on a many-core host and starves the decoders: 6.4 k -> 1.1 k img/s) and uploaded asynchronously
        pin = [torch.empty((batch, 224, 224, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
        pin_np = [t.numpy() for t in pin]
        busy = [None, None]
        cur, fill, rows = 0, 0, []

        def launch(n):
            x = pin[cur][:n].to(model.device, non_blocking=True)
            emb = model.embed_normalized(x)
            ii = torch.tensor(rows, device=model.device)
            feats[ii] = emb
            ok[ii] = 1.0
            busy[cur] = torch.cuda.Event()
            busy[cur].record()

        pool = ClipDecodePool(decode_procs, 224)
        try:
            for j, good, payload in pool.run([clean_image_path(p) for p in mine]):
                if not good:
                    print(f"处理图像 {mine[j]} 时出错: {payload}")
                    continue
                pin_np[cur][fill] = np.frombuffer(payload, dtype=np.uint8).reshape(224, 224, 3)
                rows.append(j)
                fill += 1
                if fill == batch:
                    launch(fill)
                    cur, fill, rows = cur ^ 1, 0, []
                    if busy[cur] is not None:
                        busy[cur].synchronize()          # the upload that last used this staging buffer has finished
        finally:
            pool.close()
        if fill:
            launch(fill)
        torch.cuda.synchronize()
    else:
        workers = decode_workers if decode_workers is not None else min(16, _os.cpu_count() or 1)
        with cf.ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
            pending: collections.deque = collections.deque()
            nxt = 0
            for j, p in enumerate(mine):
                while nxt < len(mine) and len(pending) < 4 * max(1, workers):
                    pending.append(pool.submit(decode, mine[nxt])); nxt += 1
                fut = pending.popleft()
                try:
                    buf.append(preprocess(fut.result()))
                    idxs.append(j)
                except Exception as ex:
                    print(f"处理图像 {p} 时出错: {ex}")
                if len(buf) == batch:
                    flush()
        flush()
    allf = allgather_rows(torch.cat([feats, ok], 1), len(image_paths))
    keep = allf[:, 512] > 0.5
    valid = [p for p, k in zip(image_paths, keep.cpu().tolist()) if k]
    return allf[keep][:, :512].cpu().numpy(), valid
