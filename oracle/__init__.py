"""oracle/ — CPU restatements of the reference's algorithm for the retrieve-then-generate hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg
may import anything from here, and only as the checker — never as the thing measured or shipped.
The product path (domain-rag_amd/) never imports this package and has no CPU fallback.

PARITY UNPINNED for the numeric core: the reference (LiYu0524/Domain-RAG) ships no tests or golden vectors, and the
wheels that hold its arithmetic (diffusers 0.33.1, transformers 4.46.3, openai/CLIP@dcba3cb,
faiss 1.10.0, torchvision 0.22.0 — requirements.txt:4,7,8,9,60,62) are neither vendored under
/root/reference nor installable offline; simple-lama-inpainting (un-pinned) and its big-lama.pt weights likewise.  Those
restatements (flux.py, vae.py, redux.py, fill.py, stem.py, topk.c, lama.py) are anchored on the reference's call sites
(cited per function).
PINNED parts: resize.py is checked bit-for-bit against PIL itself (tests/test_oracle_resize.py); vit.py drives the
`transformers` CLIP / SigLIP vision modules that ARE importable here (version 5.x, not the pinned 4.46.3) with shared
weights; stem.py's conv/bn/relu/pool equals transformers' ResNetEmbeddings (the same ResNet-50 stem) bit for bit; the host logic is checked against goldens captured from the imported reference scripts
(tests/golden/make_host_goldens.py, make_stage2_goldens.py, make_stage1_goldens.py, make_lama_goldens.py).  DESIGN.md lists the status row by row.
"""
