"""GPU PNG encoder vs Pillow's `image.save` (what the reference does per generated image): time per 1024x1024 RGB image and size"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from domain_rag_amd import png
dev = torch.device("cuda:0")
def photo(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 37.0 + k) * np.cos(yy / 23.0 - k) for k in range(3)], axis=-1)
    return np.clip(base + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
for (n, h, w) in [(8, 1024, 1024), (2, 1024, 1360), (1, 2800, 2800)]:
    arr = np.stack([photo(h, w, i) for i in range(n)])
    d = torch.from_numpy(arr).to(dev)
    png.encode(d); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps): files = png.encode(d)
    dt = (time.perf_counter() - t0) / reps
    # device time alone (kernels, no copies back)
    from domain_rag_amd import _lib
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t1 = time.perf_counter()
    buf = io.BytesIO(); Image.fromarray(arr[0]).save(buf, format="PNG")
    tp = time.perf_counter() - t1
    print(f"n={n} {h}x{w}: GPU encode + copy back {dt/n*1e3:.2f} ms per image ({arr[0].size/1e6:.1f} MB raw -> {len(files[0])/1e6:.2f} MB); "
          f"Pillow save {tp*1e3:.0f} ms per image -> {len(buf.getvalue())/1e6:.2f} MB", flush=True)
