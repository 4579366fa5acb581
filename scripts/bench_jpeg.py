"""GPU JPEG decode throughput (csrc/jpeg.hip): synthetic 640x480 4:2:0 files, batch sweep; per-kernel split from events.
Also the whole route files -> decode -> PIL-exact resize/crop -> float32 CLIP tower -> unit-norm embeddings."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
import __graft_entry__ as ge
ge.build()
from domain_rag_amd import jpeg, resample, retrieval as R

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def natural(h, w):
    base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    a = np.asarray(Image.fromarray(base).resize((w, h), Image.BICUBIC)).astype(np.int16) + rng.integers(-12, 12, (h, w, 3))
    return Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))


files = []
for i in range(256):
    bio = io.BytesIO(); natural(480, 640).save(bio, "JPEG", quality=90, subsampling=2); files.append(bio.getvalue())
print(f"{len(files)} distinct files, mean {np.mean([len(f) for f in files]) / 1024:.1f} KiB", flush=True)
t0 = time.perf_counter()
for f in files[:64]:
    Image.open(io.BytesIO(f)).convert("RGB").load()
print(f"PIL decode on one host core: {64 / (time.perf_counter() - t0):.0f} img/s", flush=True)
for n in (64, 512, 2048, 4096):
    blobs = [files[i % len(files)] for i in range(n)]
    jpeg.decode_files(blobs, dev); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        b = jpeg.decode_files(blobs, dev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"decode_files n={n}: {dt * 1e3:.1f} ms  = {n / dt:.0f} img/s (incl. host concat + upload + descriptor read-back)", flush=True)

model, _ = R.load_clip("ViT-B/32", dev, weights=None)
n = 4096
blobs = [files[i % len(files)] for i in range(n)]
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = jpeg.decode_files(blobs, dev)
    crops = torch.empty((n, 224, 224, 3), dtype=torch.uint8, device=dev)
    for (h, w), idx, imgs in b.groups():
        crops[torch.from_numpy(idx).to(dev)] = resample.clip_preprocess_u8(imgs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    emb = R.embed_images(model, crops, 1024)
    torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"files -> crops: {n / (t1 - t0):.0f} img/s; crops -> embeddings: {n / (t2 - t1):.0f} img/s; end to end {n / (t2 - t0):.0f} img/s", flush=True)
