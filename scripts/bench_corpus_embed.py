"""stage-1 corpus embedding from JPEG files: the reference's loop shape (one image at a time: host PIL preprocess,
encode, .cpu() — retrieval/clip100_resnet_style_all_shots.py:270-287) vs compute_corpus_features (decode thread pool,
PIL-exact resize on the GPU, batches of 256).  Same model, same files; embeddings must agree bit for bit."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from domain_rag_amd import retrieval as R
dev = torch.device("cuda:0")
N = int(os.environ.get("N", 2000))
d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
paths = []
import shutil
DISTINCT = min(N, 2048)          # encoding is the slow part of the set-up: the files beyond DISTINCT are copies
for i in range(N):
    p = os.path.join(d, f"{i:06d}.jpg"); paths.append(p)
    if i >= DISTINCT:
        shutil.copyfile(paths[i % DISTINCT], p)
        continue
    arr = np.kron(np.roll(base, i, axis=1), np.ones((8, 8, 1), dtype=np.uint8))            # 640x480, compressible like a photo
    arr = (arr.astype(np.int16) + rng.integers(-8, 9, arr.shape)).clip(0, 255).astype(np.uint8)
    Image.fromarray(arr).save(p, quality=90)
model, host_pre = R.load_clip("ViT-B/32", device=dev)
# reference loop shape
def ref_loop(paths):
    out = []
    for p in paths:
        x = host_pre(Image.open(p).convert("RGB")).unsqueeze(0).to(dev)
        out.append(model.embed_normalized(x).cpu().numpy()[0])       # (the reference normalises with torch ops: same values to 1 ulp)
    return np.stack(out)
ref_loop(paths[:32])
t0 = time.perf_counter(); a = ref_loop(paths[:500]); t_ref = (time.perf_counter() - t0) / 500
for procs in ((8, 32) if not os.environ.get("GPU_ONLY") else ()):
    R.compute_corpus_features(model, None, paths[:64], 256, decode_procs=procs)
    t0 = time.perf_counter(); b, valid = R.compute_corpus_features(model, None, paths, 256, decode_procs=procs); t_new = (time.perf_counter() - t0) / N
    print(f"decode_procs={procs}: {1/t_new:.0f} img/s ({t_new*1e3:.2f} ms/img) vs reference-shaped loop {1/t_ref:.0f} img/s: {t_ref/t_new:.1f}x; "
          f"embeddings bit-identical: {np.array_equal(a, b[:500])}", flush=True)
for workers in ((1, 16) if not os.environ.get("GPU_ONLY") else ()):
    R.compute_corpus_features(model, R.load_clip_device_preprocess(dev), paths[:64], 256, decode_workers=workers)
    t0 = time.perf_counter(); b, valid = R.compute_corpus_features(model, R.load_clip_device_preprocess(dev), paths, 256, decode_workers=workers); t_new = (time.perf_counter() - t0) / N
    print(f"decode_workers={workers}: {1/t_new:.0f} img/s ({t_new*1e3:.2f} ms/img) vs reference-shaped loop {1/t_ref:.0f} img/s ({t_ref*1e3:.2f} ms/img): {t_ref/t_new:.1f}x; "
          f"embeddings bit-identical: {np.array_equal(a, b[:500])}", flush=True)
R.compute_corpus_features(model, None, paths[:256], 256, gpu_decode=True)
t0 = time.perf_counter(); b, valid = R.compute_corpus_features(model, None, paths, 1024, gpu_decode=True); t_new = (time.perf_counter() - t0) / N
print(f"gpu_decode: {1/t_new:.0f} img/s ({t_new*1e3:.3f} ms/img) vs reference-shaped loop {1/t_ref:.0f} img/s: {t_ref/t_new:.1f}x; "
      f"embeddings bit-identical: {np.array_equal(a, b[:500])}", flush=True)
