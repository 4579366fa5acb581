"""one workload for rocprofv3: the float32 CLIP tower at B=1024 (a few iterations)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import retrieval as R
dev = torch.device("cuda:0")
model, _ = R.load_clip("ViT-B/32", dev, precision="fp32")
img = torch.randint(0, 256, (int(os.environ.get("B", "1024")), 224, 224, 3), device=dev, dtype=torch.uint8)
for _ in range(4):
    model.embed_normalized(img)
torch.cuda.synchronize()
