"""Stage 1 never runs on random weights silently, and its caches are tied to the weights they were computed from
(retrieval/clip100_resnet_style_all_shots.py:54,209 load pretrained CLIP / IMAGENET1K_V1 weights by default)."""
import json
import os

import numpy as np
import pytest

from domain_rag_amd.cli import stage1_retrieval as S1
from domain_rag_amd.cli import stage2_generate as S2


def _args(*extra):
    return S1.build_parser().parse_args(list(extra))


def test_weights_must_be_named_or_synthetic_requested(tmp_path, monkeypatch):
    monkeypatch.delenv("DRAG_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.delenv("DRAG_CLIP_WEIGHTS", raising=False)
    monkeypatch.delenv("DRAG_RESNET_WEIGHTS", raising=False)
    monkeypatch.setenv("HOME", str(tmp_path))                       # no library download caches here
    with pytest.raises(SystemExit) as e:
        S1.resolve_weights(_args())
    assert "--clip-weights" in str(e.value) and "--synthetic-weights" in str(e.value)
    assert S1.resolve_weights(_args("--synthetic-weights")) == (None, None)
    monkeypatch.setenv("DRAG_SYNTHETIC_WEIGHTS", "1")
    assert S1.resolve_weights(_args()) == (None, None)
    monkeypatch.delenv("DRAG_SYNTHETIC_WEIGHTS")
    clip, rn = tmp_path / "ViT-B-32.pt", tmp_path / "rn50.pth"
    clip.write_bytes(b"x"); rn.write_bytes(b"y")
    with pytest.raises(SystemExit) as e:                            # one of the two is not enough
        S1.resolve_weights(_args("--clip-weights", str(clip)))
    assert "resnet50" in str(e.value)
    assert S1.resolve_weights(_args("--clip-weights", str(clip), "--resnet-weights", str(rn))) == (str(clip), str(rn))
    monkeypatch.setenv("DRAG_CLIP_WEIGHTS", str(clip)); monkeypatch.setenv("DRAG_RESNET_WEIGHTS", str(rn))
    assert S1.resolve_weights(_args()) == (str(clip), str(rn))
    with pytest.raises(SystemExit):
        S1.resolve_weights(_args("--synthetic-weights", "--clip-weights", str(clip)))
    # the libraries' own download caches are the last default (what the reference's clip.load / torchvision leave behind)
    monkeypatch.delenv("DRAG_CLIP_WEIGHTS"); monkeypatch.delenv("DRAG_RESNET_WEIGHTS")
    c2 = tmp_path / ".cache" / "clip" / "ViT-B-32.pt"
    r2 = tmp_path / ".cache" / "torch" / "hub" / "checkpoints" / "resnet50-0676ba61.pth"
    c2.parent.mkdir(parents=True); r2.parent.mkdir(parents=True); c2.write_bytes(b"x"); r2.write_bytes(b"y")
    assert S1.resolve_weights(_args()) == (str(c2), str(r2))


def test_local_feature_cache_is_dropped_when_weights_or_precision_change(tmp_path):
    cache_f = str(tmp_path / "coco_clip_features.npy")
    np.save(cache_f, np.zeros((2, 512), np.float32))
    want = {"clip": "abc", "precision": "fp32", "preprocess": "pil-bicubic-224-centercrop"}
    assert not S1.local_cache_is_stale(cache_f, want)               # no side file: a foreign cache, taken as the reference takes it
    with open(S1._meta_path(cache_f), "w") as f:
        json.dump(want, f)
    assert not S1.local_cache_is_stale(cache_f, want)
    assert S1.local_cache_is_stale(cache_f, dict(want, clip="other-weights"))
    assert S1.local_cache_is_stale(cache_f, dict(want, precision="bf16"))
    with open(S1._meta_path(cache_f), "w") as f:
        f.write("{not json")
    assert S1.local_cache_is_stale(cache_f, want)


def test_tensor_fingerprint_sees_values_and_shapes():
    import torch
    a = torch.arange(12, dtype=torch.float32)
    assert S1.tensor_fingerprint([a]) == S1.tensor_fingerprint([a.clone()])
    assert S1.tensor_fingerprint([a]) != S1.tensor_fingerprint([a.view(3, 4)])
    b = a.clone(); b[5] += 1e-3
    assert S1.tensor_fingerprint([a]) != S1.tensor_fingerprint([b])


def test_stage2_ranks_share_one_timestamp(monkeypatch):
    monkeypatch.delenv("DRAG_TIMESTAMP", raising=False)
    a, b = S2.run_timestamp(4), S2.run_timestamp(4)                 # derived from the common parent process, not from now()
    assert a == b and len(a) == 15 and a[8] == "_"
    monkeypatch.setenv("DRAG_TIMESTAMP", "20240101_000000")
    assert S2.run_timestamp(4) == S2.run_timestamp(1) == "20240101_000000"


def test_clip_fingerprint_sees_every_tensor():
    """a tower that differs from another only in one transformer-block weight (or one element of a big matrix the strided
    sample misses) has a different fingerprint: a feature cache computed with other weights is stale"""
    import torch
    from domain_rag_amd.retrieval import weights_fingerprint
    from domain_rag_amd.vit import VitConfig, init_generic_params
    cfg = VitConfig.clip_vit_b32()
    g = init_generic_params(cfg, seed=0)
    base = weights_fingerprint(g)
    assert base == weights_fingerprint({k: v.clone() for k, v in g.items()})
    for name, pos in (("l7.fc1.weight", 5), ("l0.q.bias", 3), ("l11.fc2.weight", 1234567), ("pos", 11)):
        h = {k: v.clone() for k, v in g.items()}
        flat = h[name].view(-1)
        flat[pos] = flat[pos] + 0.5
        assert weights_fingerprint(h) != base, name


def test_style_resize_prefers_real_opencv_when_it_exists(monkeypatch):
    """with OpenCV installed the style re-rank must keep calling cv2 (the reference's pixels) unless the restated,
    parity-unpinned resize — the only one the GPU batch route can use — is asked for; without OpenCV the restated one is the default"""
    import builtins
    import sys
    import types
    import pytest
    real_import = builtins.__import__

    def no_cv2(name, *a, **k):
        if name == "cv2":
            raise ImportError("No module named 'cv2'")
        return real_import(name, *a, **k)
    monkeypatch.delitem(sys.modules, "cv2", raising=False)
    monkeypatch.setattr(builtins, "__import__", no_cv2)
    assert S1.style_resize_mode("auto") == "restated" and S1.style_resize_mode("restated") == "restated"
    with pytest.raises(SystemExit):
        S1.style_resize_mode("cv2")
    monkeypatch.setattr(builtins, "__import__", real_import)
    monkeypatch.setitem(sys.modules, "cv2", types.ModuleType("cv2"))
    assert S1.style_resize_mode("auto") == "cv2" and S1.style_resize_mode("cv2") == "cv2" and S1.style_resize_mode("restated") == "restated"
