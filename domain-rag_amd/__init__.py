"""domain-rag_amd — MI355X-native (gfx950) implementation of Domain-RAG's retrieve-then-generate
hot path (CLIP/ResNet retrieval -> Flux-Redux DiT denoise -> VAE decode).

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every FLOP of
the hot path runs in hand-written HIP kernels behind the C ABI in ``include/domainrag_hip.h``.
There is no CPU or eager-PyTorch fallback: importing :mod:`domain_rag_amd.ops` fails loudly when
``libdomainrag_hip.so`` is missing.
"""
__version__ = "0.1.0"

# RCCL's platform environment has to be in place before this process first touches the GPU (rccl.py says why)
import os as _os_env
_os_env.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
