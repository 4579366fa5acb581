"""Flux DiT forward on the HIP path vs the CPU oracle (seeded synthetic weights, reduced depth)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
SLACK = 1.3      # HIP may sit at most this factor further from float32 than the reference-dtype (bf16) oracle does


def _setup(cfg_kw, B, St, h, w, seed=0):
    from domain_rag_amd.flux import latent_image_ids
    from domain_rag_amd.flux_params import FluxConfig, init_params
    cfg = FluxConfig(**cfg_kw)
    params = init_params(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    hidden = torch.randn(B, h * w, cfg.in_channels, generator=g).bfloat16()
    enc = torch.randn(B, St, cfg.joint_attention_dim, generator=g).bfloat16()
    pooled = torch.randn(B, cfg.pooled_projection_dim, generator=g).bfloat16()
    t = torch.linspace(0.9, 0.3, B)
    gd = torch.full((B,), 30.0)
    return cfg, params, hidden, enc, pooled, t, gd, latent_image_ids(h, w), torch.zeros(St, 3)


@pytest.mark.parametrize("B,St,h,w,nl,ns,inch", [(1, 24, 6, 8, 1, 1, 64), (2, 77, 12, 10, 2, 2, 384)])
def test_flux_forward_vs_oracle(gpu, B, St, h, w, nl, ns, inch):
    from domain_rag_amd.flux import FluxTransformerHIP
    from oracle import flux as oflux
    cfg, params, hidden, enc, pooled, t, gd, img_ids, txt_ids = _setup(
        dict(in_channels=inch, num_layers=nl, num_single_layers=ns, num_attention_heads=2, joint_attention_dim=128,
             pooled_projection_dim=64), B, St, h, w)
    ocfg = oflux.FluxConfig(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    taps_ref, taps = {}, {}
    ref = oflux.flux_forward(params, ocfg, hidden, enc, pooled, t, img_ids, txt_ids, gd, taps=taps_ref)
    p32 = {k: v.float() for k, v in params.items()}
    ref32 = oflux.flux_forward(p32, ocfg, hidden.float(), enc.float(), pooled.float(), t, img_ids, txt_ids, gd,
                               time_dtype=torch.bfloat16)
    model = FluxTransformerHIP(cfg, params, gpu)
    out = model(hidden.to(gpu), enc.to(gpu), pooled.to(gpu), t, img_ids, txt_ids, gd, taps=taps)
    torch.cuda.synchronize()

    def rel(a, b):
        return ((a.double().cpu() - b.double()).abs().max() / (b.double().abs().max() + 1e-9)).item()

    # per-stage taps localise a failure
    assert rel(taps["temb"], taps_ref["temb"]) < 2e-2
    assert rel(taps["x_embed"], taps_ref["x_embed"]) < 2e-2
    assert rel(taps["ctx_embed"], taps_ref["ctx_embed"]) < 2e-2
    for i in range(nl):
        assert rel(taps[f"double.{i}"], taps_ref[f"double.{i}"]) < 3e-2, f"double block {i}"
    for i in range(ns):
        assert rel(taps[f"single.{i}"], taps_ref[f"single.{i}"]) < 3e-2, f"single block {i}"
    # stated tolerance: 1e-2 relative (north_star) against the fp32 oracle, bf16 pipeline
    e_bf16, e_f32 = rel(out, ref), rel(out, ref32)
    e_oracle = rel(ref, ref32)  # how far the bf16 oracle itself sits from fp32
    assert e_bf16 < 2e-2, (e_bf16, e_f32, e_oracle)
    assert e_f32 < max(1e-2, SLACK * e_oracle), f"HIP vs f32 {e_f32:.4e}, bf16 oracle vs f32 {e_oracle:.4e}, ratio {e_f32 / max(e_oracle, 1e-30):.2f} (bar {SLACK}); HIP vs bf16 oracle {e_bf16:.4e}"


def test_flux_depth_error_growth(gpu):
    """Deeper stack (6 double + 12 single blocks, reduced width): rounding differences compound with depth, so the
    meaningful statement is that the HIP bf16 path stays as close to the fp32 yardstick as the reference-dtype (bf16)
    oracle itself does."""
    from domain_rag_amd.flux import FluxTransformerHIP
    from oracle import flux as oflux
    cfg, params, hidden, enc, pooled, t, gd, img_ids, txt_ids = _setup(
        dict(in_channels=64, num_layers=6, num_single_layers=12, num_attention_heads=2, joint_attention_dim=128,
             pooled_projection_dim=64), 1, 40, 10, 12, seed=3)
    ocfg = oflux.FluxConfig(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    ref = oflux.flux_forward(params, ocfg, hidden, enc, pooled, t, img_ids, txt_ids, gd).float()
    p32 = {k: v.float() for k, v in params.items()}
    ref32 = oflux.flux_forward(p32, ocfg, hidden.float(), enc.float(), pooled.float(), t, img_ids, txt_ids, gd, time_dtype=torch.bfloat16)
    out = FluxTransformerHIP(cfg, params, gpu)(hidden.to(gpu), enc.to(gpu), pooled.to(gpu), t, img_ids, txt_ids, gd).float().cpu()
    scale = ref32.abs().max().item()
    e_hip = (out - ref32).abs().max().item() / scale
    e_or = (ref - ref32).abs().max().item() / scale
    rms_hip = (out - ref32).pow(2).mean().sqrt().item() / ref32.pow(2).mean().sqrt().item()
    rms_or = (ref - ref32).pow(2).mean().sqrt().item() / ref32.pow(2).mean().sqrt().item()
    print(f"depth 18 blocks: max-rel hip {e_hip:.4f} oracle-bf16 {e_or:.4f} | rms-rel hip {rms_hip:.4f} oracle-bf16 {rms_or:.4f}")
    assert e_hip < max(1e-2, 2.0 * e_or) and rms_hip < max(5e-3, 2.0 * rms_or), (e_hip, e_or, rms_hip, rms_or)


def test_graph_replay_is_bit_identical(gpu):
    """hipGraph capture/replay of the forward: same kernels in the same order -> identical bits, for changing timesteps"""
    from domain_rag_amd.flux import FluxTransformerHIP
    cfg, params, hidden, enc, pooled, t, gd, img_ids, txt_ids = _setup(
        dict(in_channels=64, num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=128,
             pooled_projection_dim=64), 2, 40, 8, 8, seed=4)
    m = FluxTransformerHIP(cfg, params, gpu)
    h, e, p = hidden.to(gpu), enc.to(gpu), pooled.to(gpu)
    for tt in (torch.tensor([0.9, 0.9]), torch.tensor([0.4, 0.4]), torch.tensor([0.05, 0.7])):
        eager = m.forward(h, e, p, tt, img_ids, txt_ids, gd).clone()
        graphed = m.forward_graphed(h, e, p, tt, img_ids, txt_ids, gd).clone()
        assert torch.equal(eager, graphed)
    assert len(m._graphs) == 1
