"""BASELINE configs[2] at FULL size, end to end, against the CPU oracle: the complete FLUX.1-Fill-dev architecture (19 double + 38
single blocks, D = 3072, S = 1241 + 4096), the full VAE, SigLIP-so400m + Redux, one 1024x1024 image + mask, 2 denoise steps at
strength 1.0, identical seeds -> composited uint8 pixels.

Default (round 3: part of every `-m gpu` run, about two minutes of host time): the oracle runs ONCE, in the reference's dtype (bfloat16);
bar: the HIP path's uint8 pixels differ from the bf16 oracle's by at most 1.3 levels on average and 12 levels anywhere (two bf16
evaluations of the same graph with different summation orders; measured 0.87 / 7 — a kernel regression that doubled the HIP path's
error would put the mean near 1.7).  With DRAG_FULLSIZE_E2E=1 the float32 yardstick runs as well (+4 min) and the ratio bar of the
other pipeline tests applies: HIP within max(1e-2, 1.3 x the bf16 oracle's own distance from float32) of full scale, max and mean.
The round's log is committed under profiles/."""
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


RES, STEPS, STRENGTH = 1024, 2, 1.0


def _inputs(gpu):
    """everything the two routes share, from seeds (private generators: safe to draw from two threads)"""
    from domain_rag_amd import redux, vae, vit
    from domain_rag_amd.flux_params import FluxConfig, init_params
    res = RES
    cfg = FluxConfig(in_channels=384)
    assert (cfg.num_layers, cfg.num_single_layers, cfg.num_attention_heads) == (19, 38, 24)
    tp_dev = init_params(cfg, seed=0, device=gpu)              # 11.9 G parameters: drawn on the GPU, the oracle gets a host copy
    vcfg = vae.VaeConfig()
    vp = vae.init_params(vcfg, seed=1)
    vitcfg = vit.VitConfig.siglip_so400m()
    vitp = vit.init_generic_params(vitcfg, 2)
    rp = redux.init_redux_params(seed=3)
    g = torch.Generator().manual_seed(4)
    # a smooth picture + noise (flat random pixels make the VAE encoder's statistics degenerate), one keep-box in the mask
    yy, xx = torch.meshgrid(torch.arange(res), torch.arange(res), indexing="ij")
    base = torch.stack([128 + 90 * torch.sin(xx / 41.0 + c) * torch.cos(yy / 29.0 - c) for c in range(3)], -1)
    image = (base + 8 * torch.randn(res, res, 3, generator=g)).clamp(0, 255).to(torch.uint8)[None]
    mask = torch.full((1, res, res), 255, dtype=torch.uint8); mask[:, 362:662, 362:662] = 0
    bg = torch.randint(0, 256, (1, 384, 384, 3), generator=g, dtype=torch.uint8)
    t5 = torch.randn(512, 4096, generator=g).bfloat16(); pooled = torch.randn(768, generator=g).bfloat16()
    en = torch.randn(1, 16, res // 8, res // 8, generator=g).bfloat16()
    mn = torch.randn(1, 16, res // 8, res // 8, generator=g).bfloat16()
    nt = torch.randn(1, (res // 16) ** 2, 64, generator=g).bfloat16()
    return dict(cfg=cfg, tp_dev=tp_dev, vcfg=vcfg, vp=vp, vitcfg=vitcfg, vitp=vitp, rp=rp, image=image, mask=mask, bg=bg, t5=t5, pooled=pooled,
                en=en, mn=mn, nt=nt)


def _oracle(x, with_f32, t0):
    """the CPU oracle's composite(s) for the inputs `x`: {"bf16": (uint8 image, float image)[, "f32": ...]}"""
    from oracle import fill as ofill, flux as oflux, redux as ored, vit as ovit
    cfg, vcfg, vitcfg = x["cfg"], x["vcfg"], x["vitcfg"]
    tp = {k: v.cpu() for k, v in x["tp_dev"].items()}
    res_or = {}
    ocfg = oflux.FluxConfig(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    for name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32))[: 2 if with_f32 else 1]:
        cast = (lambda d: {k: v.to(dt) for k, v in d.items()})
        with torch.no_grad():
            lat = ovit.siglip_last_hidden_state(x["vitp"], 384, 14, 1152, 16, 27, 4304, ovit.normalize_u8(x["bg"], vitcfg.mean, vitcfg.std), dt)
            pes, pps = ored.redux_prior(lat, cast(x["rp"]), x["t5"].to(dt), x["pooled"].to(dt), [1.0], [1.0])
            u8, img = ofill.fill_pipeline(cast(tp), ocfg, cast(x["vp"]), dict(block_out=vcfg.block_out_channels, layers=vcfg.layers_per_block),
                                          x["image"], x["mask"], pes, pps, 30.0, STEPS, STRENGTH, x["en"], x["mn"], x["nt"], dtype=dt)
        res_or[name] = (u8, img.float())
        print(f"[e2e] oracle {name} done {time.time() - t0:.0f} s", flush=True)
    return res_or


# The oracle is two minutes of HOST time and needs nothing from the HIP path: tests/conftest.py starts it in a CHILD PROCESS (this file run as
# a script, 48 host threads) as soon as the collection is known to hold this test; it runs under the GPU tests in front of this one and
# the test joins it — same inputs, same oracle, same bars, ~110 s less wall clock for the `-m gpu` run (the
# suite must stay well inside the driver's limit).  (A thread in the pytest process was tried first: two torch CPU workloads in one
# process oversubscribe the cores — the other oracle-bound tests ran 3 x slower and the suite 200 s LONGER.)
# DRAG_ORACLE_PREFETCH=0 computes it inline as before; so does a child that fails.
_prefetch = {}
PREFETCH_THREADS = 32      # (the bf16 oracle's best thread count on the GPU box: tests/conftest.py oracle_threads)


def start_oracle_prefetch():
    import subprocess
    import sys
    import tempfile
    if "proc" in _prefetch:
        return
    out = tempfile.NamedTemporaryFile(prefix="drag_e2e_oracle_", suffix=".pt", delete=False)
    out.close()
    env = dict(os.environ, OMP_NUM_THREADS=str(PREFETCH_THREADS), MKL_NUM_THREADS=str(PREFETCH_THREADS))
    log = open(out.name + ".log", "w")
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--oracle", out.name], env=env, stdout=log, stderr=subprocess.STDOUT,
                            cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _prefetch.update(proc=proc, path=out.name, log=log)


def _join_prefetch():
    """the child's result, or None (the test then computes the oracle itself)"""
    proc, path = _prefetch["proc"], _prefetch["path"]
    rc = proc.wait()
    _prefetch["log"].close()
    try:
        if rc != 0:
            print(f"[e2e] oracle child exited with {rc}: computing inline\n" + open(path + ".log").read()[-2000:], flush=True)
            return None
        return torch.load(path)
    finally:
        for f in (path, path + ".log"):
            try:
                os.remove(f)
            except OSError:
                pass


def test_fullsize_fill_pipeline_vs_oracle(gpu):
    from domain_rag_amd import fill_pipeline as fp, redux, vae
    from domain_rag_amd.flux import FluxTransformerHIP
    from conftest import oracle_threads
    oracle_threads(torch.bfloat16)
    t0 = time.time()
    res, steps, strength = RES, STEPS, STRENGTH
    x = _inputs(gpu)
    cfg, vcfg, vitcfg = x["cfg"], x["vcfg"], x["vitcfg"]
    image, mask, bg, t5, pooled, en, mn, nt = (x[k] for k in ("image", "mask", "bg", "t5", "pooled", "en", "mn", "nt"))
    print(f"[e2e] parameters ready {time.time() - t0:.0f} s", flush=True)

    prior = redux.ReduxPriorHIP(vitcfg, x["vitp"], x["rp"], gpu)
    fill = fp.FluxFillHIP(FluxTransformerHIP(cfg, x["tp_dev"], gpu), vae.FluxVaeHIP(vcfg, x["vp"], gpu))
    pe, pp = prior(bg.to(gpu), t5.to(gpu), pooled.to(gpu), [1.0], [1.0], group=1)
    assert pe.shape == (1, 1241, 4096)
    out = fill(image.to(gpu), mask.to(gpu), pe, pp, guidance_scale=30.0, num_inference_steps=steps, strength=strength,
               enc_noise=en.to(gpu), masked_enc_noise=mn.to(gpu), noise_tokens=nt.to(gpu)).cpu()
    torch.cuda.synchronize()
    del fill, prior
    print(f"[e2e] HIP path done {time.time() - t0:.0f} s", flush=True)

    with_f32 = os.environ.get("DRAG_FULLSIZE_E2E") == "1"
    res_or = _join_prefetch() if "proc" in _prefetch else None
    if res_or is not None:
        print(f"[e2e] oracle joined from the child process {time.time() - t0:.0f} s", flush=True)
    else:
        res_or = _oracle(x, with_f32, t0)
    del x
    torch.cuda.empty_cache()
    assert out.shape == (1, res, res, 3) and out.dtype == torch.uint8
    hip = out.float() / 255.0
    lv = (out.int() - res_or["bf16"][0].int()).abs()
    lv_max, lv_mean, same = lv.max().item(), lv.float().mean().item(), 100 * (lv == 0).float().mean().item()
    e_vs_bf = (hip - res_or["bf16"][1].permute(0, 2, 3, 1)).abs().max().item()
    print(f"[e2e] full-size Fill pipeline, {steps} steps @ {res}^2: HIP vs bf16 oracle max {e_vs_bf:.4f} of full scale; uint8 levels vs bf16 "
          f"oracle: max {lv_max}, mean {lv_mean:.3f}, identical {same:.1f} %", flush=True)
    assert lv_mean <= 1.3 and lv_max <= 12, f"uint8 levels vs the bf16 oracle: mean {lv_mean:.3f} (bar 1.3), max {lv_max} (bar 12), identical {same:.1f} %"
    if with_f32:
        ref_img = res_or["f32"][1]
        e_or = (res_or["bf16"][1] - ref_img).abs().max().item()
        e = (hip - ref_img.permute(0, 2, 3, 1)).abs().max().item()
        m = (hip - ref_img.permute(0, 2, 3, 1)).abs().mean().item()
        m_or = (res_or["bf16"][1] - ref_img).abs().mean().item()
        print(f"[e2e] max |HIP - f32 oracle| {e:.4f} of full scale (mean {m:.5f}); bf16 oracle vs f32 oracle max {e_or:.4f} (mean {m_or:.5f}); "
              f"ratios {e / max(e_or, 1e-30):.2f} / {m / max(m_or, 1e-30):.2f}", flush=True)
        assert e < max(1e-2 + 0.5 / 255, 1.3 * e_or), f"max: HIP vs f32 {e:.4e}, bf16 oracle vs f32 {e_or:.4e}, ratio {e / max(e_or, 1e-30):.2f} (bar 1.3)"
        assert m < max(2e-3, 1.3 * m_or), f"mean: HIP vs f32 {m:.4e}, bf16 oracle vs f32 {m_or:.4e}, ratio {m / max(m_or, 1e-30):.2f} (bar 1.3)"


if __name__ == "__main__":      # the oracle child (start_oracle_prefetch)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert sys.argv[1] == "--oracle"
    torch.set_num_threads(PREFETCH_THREADS)
    import __graft_entry__ as ge
    ge.build()
    dev = torch.device("cuda:0")
    t_start = time.time()
    res = _oracle(_inputs(dev), os.environ.get("DRAG_FULLSIZE_E2E") == "1", t_start)
    torch.save(res, sys.argv[2])
