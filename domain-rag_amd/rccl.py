"""Joining the RCCL communicator (torch.distributed backend "nccl" IS RCCL on ROCm) — one process per GPU, as
`domainrag.sh:4-31` / `outpainting_updown_sampling_redux.py:1605-1715` fan the reference's stages out, plus the one exchange
north_star adds (the embedding all-gather before top-k, retrieval/clip100_resnet_style_all_shots.py:298,419).

Whatever started the ranks (bench.py's own launcher, the driver's ``python -m torch.distributed.run``, srun ...), the
environment RCCL needs is put in place HERE, before the process group exists, not only by one launcher:

* ``HSA_ENABLE_IPC_MODE_LEGACY=0``: this driver stack only has dmabuf IPC; with the legacy mode RCCL's intra-node
  transport setup dies in ``hipIpcGetMemHandle: invalid argument``.  The ROCr runtime reads it when the process first
  touches the GPU, so ``prepare_env()`` also runs at package import (``domain_rag_amd/__init__``) and at the top of
  bench.py, ahead of the first HIP call.
"""
from __future__ import annotations

import os

_ENV = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def prepare_env() -> dict:
    """set (without overriding the user's choice) what RCCL needs on this platform; returns what is now in force"""
    for k, v in _ENV.items():
        os.environ.setdefault(k, v)
    return {k: os.environ[k] for k in _ENV}


def init_rccl(device, **kw):
    """``init_process_group("nccl", device_id=device)`` with the environment prepared and the device bound first (an
    unbound rank would open its communicator on GPU 0).  Extra keywords (init_method, rank, world_size) pass through."""
    import torch
    import torch.distributed as dist
    prepare_env()
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("init_rccl: RCCL communicators live on GPUs (use gloo for CPU self-tests)")
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", device_id=device, **kw)
    return dist
