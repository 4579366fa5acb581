"""one workload for rocprofv3 --kernel-trace: GPU JPEG decode of 64 and of 4096 synthetic 640x480 files (per-kernel split)"""
import io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from domain_rag_amd import jpeg
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
files = []
for i in range(64):
    base = rng.integers(0, 256, (62, 82, 3), dtype=np.uint8)
    a = np.asarray(Image.fromarray(base).resize((640, 480), Image.BICUBIC)).astype(np.int16) + rng.integers(-12, 12, (480, 640, 3))
    bio = io.BytesIO(); Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(bio, "JPEG", quality=int(os.environ.get("Q", 90)), subsampling=2); files.append(bio.getvalue())
print("mean KiB", np.mean([len(f) for f in files]) / 1024)
for n in (64, 64, 4096):
    jpeg.decode_files([files[i % 64] for i in range(n)], dev)
torch.cuda.synchronize()
