"""scripts/accept_real_weights.py — the comparison arithmetic and the "reference stack not available" path (the only path that
can run in the build container: no diffusers / clip / faiss, no checkpoints)."""
import importlib.util
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("accept_real_weights", os.path.join(ROOT, "scripts", "accept_real_weights.py"))
acc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(acc)


def test_topk_comparison_reports_equality_and_first_mismatch():
    I = np.arange(12).reshape(3, 4)
    D = np.linspace(1, 0, 12).reshape(3, 4)
    r = acc.compare_topk(D, I, D, I)
    assert r["indices_equal"] and r["rows_equal"] == 3 and "first_mismatch" not in r
    J = I.copy(); J[1, [1, 2]] = J[1, [2, 1]]                 # a swap inside one row: same set, different order
    r = acc.compare_topk(D, I, D, J)
    assert not r["indices_equal"] and r["rows_equal"] == 2 and r["same_sets"] == 3
    assert r["first_mismatch"] == {"query": 1, "rank": 1, "ref_index": 5, "hip_index": 6, "ref_score": D[1, 1], "hip_score": D[1, 1]}


def test_latent_and_pixel_comparisons_apply_the_stated_tolerances():
    rng = np.random.default_rng(0)
    ref = [rng.standard_normal((1, 16, 64)) for _ in range(4)]
    close = [a + 1e-3 * np.abs(a).max() for a in ref]
    far = [a + 0.1 * np.abs(a).max() for a in ref]
    assert acc.compare_latents(ref, close)["pass"]
    r = acc.compare_latents(ref, far)
    assert not r["pass"] and r["steps_compared"] == 4 and abs(r["worst"] - 0.1) < 1e-9
    assert not acc.compare_latents(ref, close[:3])["pass"]           # a different number of steps is a failure, not a shorter comparison
    img = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    ok = img.astype(np.int32); ok[0, 0, 0] = min(255, ok[0, 0, 0] + 3)
    assert acc.compare_pixels(img, ok.astype(np.uint8))["pass"]      # 3 levels = ceil(1e-2 * 255)
    bad = img.astype(np.int32); bad[1, 1, 1] = (bad[1, 1, 1] + 40) % 256
    r = acc.compare_pixels(img, bad.astype(np.uint8))
    assert not r["pass"] and r["max_abs_levels"] >= 4
    assert not acc.compare_pixels(img, img[:4])["pass"]


def test_without_the_reference_stack_it_says_what_is_missing_and_exits_2(tmp_path):
    out = tmp_path / "report.json"
    rc = acc.main(["--model-root", str(tmp_path / "model"), "--out", str(out)])
    assert rc == 2
    rep = json.load(open(out))
    assert rep["verdict"].startswith("not-run")
    assert any("FLUX.1-Fill-dev" in m for m in rep["missing"]) and any("--target" in m for m in rep["missing"])


def test_checkpoint_key_sets_are_compared_before_any_arithmetic(tmp_path):
    """real-checkpoint loading is checked on KEY SETS + shapes first (safetensors headers only): a directory written from this
    repository's own expected layout passes; a renamed tensor, a missing one and a transposed one are each reported"""
    import torch
    from safetensors.torch import save_file
    from domain_rag_amd import flux_params, redux, vit
    cfg = flux_params.FluxConfig(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32)
    exp = flux_params.param_shapes(cfg)
    d = tmp_path / "transformer"; d.mkdir()
    names = sorted(exp)
    half = len(names) // 2                                # two shards, like the real 3-shard transformer directory
    save_file({k: torch.zeros(exp[k], dtype=torch.bfloat16) for k in names[:half]}, str(d / "a.safetensors"))
    save_file({k: torch.zeros(exp[k], dtype=torch.bfloat16) for k in names[half:]}, str(d / "b.safetensors"))
    found = acc.safetensors_dir_shapes(str(d))
    r = acc.compare_key_sets(found, exp)
    assert r["pass"] and r["tensors_in_file"] == len(exp) == r["tensors_expected"]
    bad = dict(found)
    bad["transformer_blocks.0.attn.to_q.weight_renamed"] = bad.pop("transformer_blocks.0.attn.to_q.weight")
    del bad["proj_out.bias"]
    bad["x_embedder.weight"] = tuple(reversed(bad["x_embedder.weight"]))
    r = acc.compare_key_sets(bad, exp)
    assert not r["pass"] and r["n_missing"] == 2 and r["n_unexpected"] == 1 and list(r["shape_mismatch"]) == ["x_embedder.weight"]
    assert "proj_out.bias" in r["missing_from_file"] and r["unexpected_in_file"] == ["transformer_blocks.0.attn.to_q.weight_renamed"]
    # extras the loaders do not consume are ignored only when named
    extra = dict(found); extra["vision_model.head.probe"] = (1, 1, 8)
    assert not acc.compare_key_sets(extra, exp)["pass"] and acc.compare_key_sets(extra, exp, ignore_prefixes=("vision_model.head.",))["pass"]
    # the SigLIP name table matches what siglip_to_generic reads
    vc = vit.VitConfig(image_size=56, patch_size=14, hidden=64, heads=2, layers=2, intermediate=96)
    sd = {k: torch.zeros(s) for k, s in acc.expected_siglip_shapes(vc).items()}
    g = vit.siglip_to_generic(sd, vc)
    assert g["patch.weight"].shape == (64, 3 * 14 * 14) and g["l1.fc2.weight"].shape == (64, 96) and g["pos"].shape == (vc.tokens, 64)
    assert set(redux.param_shapes(64, 32)) == set(redux.init_redux_params(64, 32))
