"""bench.py — composited images/sec @1024^2, 30 Flux-Redux (Fill) steps, on N MI355X (default workload), and the
retrieval side of the path (``--workload retrieval``: sharded CLIP corpus embedding -> ONE RCCL all-gather -> exact top-100).

Launching.  ``python bench.py --gpus N``:
  * under torch.distributed.run (WORLD_SIZE in the environment, as the driver starts it) this process is one rank;
    ``--gpus`` must equal WORLD_SIZE or the run aborts;
  * from a bare shell with N > 1 the script starts the N ranks ITSELF (re-executes under
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1``), one rank per GPU,
    and refuses to start when fewer than N GPUs are visible — it never reports ``n_gpus`` for ranks it did not run.
  ``rccl_ranks`` in the JSON line is the sum of a GPU all-reduce of ones over backend "nccl" (= RCCL): the number of
  ranks that really joined the communicator.

generate (default): one "step" = one pass of the hot path over one batch of B=8 synthetic 1024x1024 composites per GPU:
Redux prior -> VAE encodes -> mask prep -> 30 x (Flux-Fill DiT forward + flow-Euler step) -> VAE decode.
Data-parallel over ranks (independent images, no collective on the data path): weak scaling.

retrieval: one "step" = each rank embeds ITS contiguous shard of a synthetic N=118 287-image corpus (uint8 224^2 crops
resident in HBM) with the float32 CLIP ViT-B/32 tower, ONE all-gather of the [N/W, 512] fp32 shards makes the corpus
resident on every GPU in the global row order, then exact top-100 for the rank's shard of the queries.  Total work is
fixed (BASELINE configs[4]'s corpus): strong scaling.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel and
`cpu_baseline` (the oracle timed on this box's host cores; N = 1 only).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# dmabuf IPC only on this driver: RCCL's intra-node transports need it, and the ROCr runtime reads it at its first GPU call —
# so it is set in EVERY rank, whoever launched it (domain_rag_amd/rccl.py), before torch touches the device
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured-achievable)
HBM_MEASURED_GBS = 6290.0
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16
MFMA_F32_PEAK_TF = 157.3       # f32-input matrix core = the f32 vector rate
XGMI_LINK_GBS, XGMI_LINKS = 153.0, 7


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("generate", "retrieval"), default="generate")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--denoise-steps", type=int, default=30)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the BASELINE configs[1] side measurement (N = 1 only)")
    ap.add_argument("--roofline-every", type=int, default=5, help="bracket the GEMMs of every n-th denoise step with events")
    ap.add_argument("--corpus", type=int, default=118287, help="retrieval: corpus images in total (BASELINE configs[4])")
    ap.add_argument("--queries", type=int, default=64, help="retrieval: queries in total (sharded over ranks)")
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--embed-batch", type=int, default=1024)
    ap.add_argument("--clip-precision", choices=("fp32", "bf16"), default="fp32")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU check of the launcher itself: ranks rendezvous over gloo, shard units, all-gather rows; no GPU work")
    ap.add_argument("--debug-share-gpu", action="store_true",
                    help="NOT a measurement: all N ranks run on GPU 0 and talk over gloo, so the N > 1 control flow of a workload "
                         "(sharding, collectives, rank-0-only work) can be exercised on a 1-GPU box; the line says so")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------- launching
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch_if_needed(args) -> None:
    """Returns in a process that IS a rank (or the single process of an N=1 run); otherwise spawns the ranks and exits."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks; refusing to report a "
                     f"number for a rank count that did not run")
        return
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus == 1:
        return
    if not args.selftest_launcher and not args.debug_share_gpu:
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} requested but {n} GPU(s) are visible on this node: one rank per GPU is "
                     f"required (no oversubscription, no silent fall-back to fewer ranks)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    sys.exit(subprocess.call(cmd, env=env))


class Dist:
    """the process group of this run (RCCL via backend "nccl" on GPUs; gloo only for the launcher self-test)"""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.cpu = args.selftest_launcher
        self.shared = bool(getattr(args, "debug_share_gpu", False)) and not self.cpu
        self.dist = None
        if self.cpu:
            self.dev = torch.device("cpu")
        else:
            if not torch.cuda.is_available():
                sys.exit("bench.py: no GPU visible (domain-rag_amd has no CPU path)")
            if self.shared:
                self.local = 0
            if self.local >= torch.cuda.device_count():
                sys.exit(f"bench.py: local rank {self.local} has no GPU of its own ({torch.cuda.device_count()} visible)")
            torch.cuda.set_device(self.local)
            self.dev = torch.device("cuda", self.local)
        self.rccl_ranks = None
        if self.world > 1 or (self.cpu and "MASTER_ADDR" in os.environ):
            import torch.distributed as dist
            self.dist = dist
            if self.cpu or self.shared:
                dist.init_process_group("gloo")
            else:
                from domain_rag_amd.rccl import init_rccl
                init_rccl(self.dev)
            ones = torch.ones(1, device="cpu" if self.shared else self.dev)
            dist.all_reduce(ones)                       # every rank that joined adds 1
            self.rccl_ranks = int(round(ones.item()))
            if self.rccl_ranks != self.world:
                sys.exit(f"bench.py: communicator has {self.rccl_ranks} ranks, expected {self.world}")

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return x
        t = torch.tensor([x], device="cpu" if self.shared else self.dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.item()

    def all_ranks(self, x: float) -> list:
        """every rank's own value, in rank order (a straggler shows up here; the reported clock is the max)"""
        if self.dist is None:
            return [x]
        t = torch.tensor([x], device="cpu" if self.shared else self.dev, dtype=torch.float64)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.item() for o in out]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def launcher_selftest(args, d: Dist) -> None:
    """what the N>1 paths do around the GPU work, on CPU tensors over gloo: unit sharding with the reference's rule, the one
    all-gather of row shards into the global order, the max-over-ranks clock."""
    from domain_rag_amd.hostlogic import split_samples_for_gpus
    from domain_rag_amd.retrieval import allgather_rows, shard_bounds
    n_total = 37
    full = torch.arange(n_total * 4, dtype=torch.float32).view(n_total, 4)
    s, e = shard_bounds(n_total, d.world, d.rank)
    gathered = allgather_rows(full[s:e].clone(), n_total)
    units = list(range(n_total))
    mine = split_samples_for_gpus(units, d.world)[d.rank] if d.world > 1 else units
    ok = torch.equal(gathered, full) and mine == list(range(s, e))
    dt = d.max_over_ranks(float(d.rank + 1))
    ok = ok and dt == float(d.world)
    allok = d.max_over_ranks(0.0 if ok else 1.0) == 0.0
    if d.rank == 0:
        print(json.dumps({"selftest": "launcher", "n_ranks": d.world, "joined_ranks": d.rccl_ranks if d.rccl_ranks is not None else 1,
                          "backend": "gloo", "ok": bool(allok)}), flush=True)
    if not allok:
        sys.exit(1)


# ----------------------------------------------------------------------------------------------- CPU baselines
def _pick_threads() -> int:
    """torch's CPU kernels stop scaling (and then slow down: 256 threads ran the CLIP tower 4x slower than 64) well before a
    256-core host is full; probe one GEMM at a few thread counts and keep the fastest.  `cores` in the JSON = this number."""
    cores = os.cpu_count() or 1
    a, b = torch.randn(2048, 3072), torch.randn(3072, 3072)
    best, best_t = cores, float("inf")
    for n in sorted({cores, max(1, cores // 2), max(1, cores // 4), min(cores, 32)}, reverse=True):
        torch.set_num_threads(n)
        a @ b
        t0 = time.perf_counter()
        for _ in range(3):
            a @ b
        t = time.perf_counter() - t0
        if t < best_t:
            best, best_t = n, t
    torch.set_num_threads(best)
    return best


def _host_cpu() -> dict:
    """what the CPU legs ran on: logical CPUs the OS reports, physical cores and the model name from /proc/cpuinfo, and the torch thread
    count the probe in _pick_threads chose (north_star asks for the core count next to the CPU number)"""
    model, phys = None, set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name") and model is None:
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    pid = ln.split(":", 1)[1].strip()
                elif ln.startswith("core id"):
                    cid = ln.split(":", 1)[1].strip()
                elif not ln.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return {"host_cpus": logical, "cores": len(phys) or logical, "cpu_model": model, "threads": torch.get_num_threads()}


def _median_time(fn, repeats=3):
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def cpu_baseline_generate(res: int, denoise_steps: int):
    """Oracle (CPU restatement, bf16 like the reference) on this box's host cores on a BOUNDED sample of the same workload:
    one double-stream + one single-stream Flux block at full sequence length (B=1), one VAE decode at 1/4 of the latent side
    and the Redux prior on one image — median of 3 repeats each — extrapolated to a full image:
    denoise_steps * (19 double + 38 single) + 2 encodes + decode + prior.  Labelled extrapolated."""
    from oracle import flux as oflux
    from oracle import redux as oredux
    from oracle import vae as ovae
    from oracle import vit as ovit
    from domain_rag_amd import redux as redux_mod, vae as vae_mod, vit as vit_mod
    from domain_rag_amd.flux_params import FluxConfig, init_params
    _pick_threads()
    cfg = FluxConfig(in_channels=384, num_layers=1, num_single_layers=1)
    p = init_params(cfg, seed=0)
    ocfg = oflux.FluxConfig(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    Si, St = (res // 16) ** 2, 512 + 729
    g = torch.Generator().manual_seed(0)
    D = cfg.dim
    hs = torch.randn(1, Si, D, generator=g).bfloat16()
    ehs = torch.randn(1, St, D, generator=g).bfloat16()
    temb = torch.randn(1, D, generator=g).bfloat16()
    cos, sin = oflux.rope_tables(torch.cat([torch.zeros(St, 3), oflux.latent_image_ids(res // 16, res // 16)]))
    with torch.no_grad():
        ehs2, hs2 = oflux.double_block(p, "transformer_blocks.0.", ocfg, hs, ehs, temb, cos, sin)
        joint = torch.cat([ehs2, hs2], 1)
        td, tds = _median_time(lambda: oflux.double_block(p, "transformer_blocks.0.", ocfg, hs, ehs, temb, cos, sin))
        ts, tss = _median_time(lambda: oflux.single_block(p, "single_transformer_blocks.0.", ocfg, joint, temb, cos, sin))
        # VAE decode: latent side res/32 (1/16 of the pixels of the full frame; convolution cost is linear in pixels, the
        # mid-block attention is quadratic and is 4 % of the decode at full size, so x16 slightly under-states the CPU time)
        vcfg = vae_mod.VaeConfig()
        vp = {k: v.bfloat16() for k, v in vae_mod.init_params(vcfg, seed=1).items()}
        lat = max(res // 32, 8)
        z = torch.randn(1, 16, lat, lat, generator=g).bfloat16()
        tv, tvs = _median_time(lambda: ovae.decode(vp, z))
        vae_scale = (res // 8) ** 2 / float(lat * lat)
        # Redux prior: SigLIP-so400m (upstream transformers code) + Redux MLP on one image
        vc = vit_mod.VitConfig.siglip_so400m()
        gp = vit_mod.init_generic_params(vc, 2)
        rp = redux_mod.init_redux_params(vc.hidden, cfg.joint_attention_dim, seed=3)
        img = torch.randint(0, 256, (1, vc.image_size, vc.image_size, 3), generator=g, dtype=torch.uint8)
        t5 = torch.randn(512, cfg.joint_attention_dim, generator=g).bfloat16()
        pooled = torch.randn(768, generator=g).bfloat16()

        def prior():
            lat_ = ovit.siglip_last_hidden_state(gp, vc.image_size, vc.patch_size, vc.hidden, vc.heads, vc.layers, vc.intermediate,
                                                 ovit.normalize_u8(img, vc.mean, vc.std), torch.bfloat16)
            oredux.redux_prior(lat_, rp, t5, pooled, [1.0], [1.0])
        tp, tps = _median_time(prior)
    # encode ~ 0.48 x decode in FLOPs (dec 10.5, 2 x enc ~ 10 TFLOP)
    sec_vae = tv * vae_scale * (1.0 + 2 * 0.48)
    sec_per_img = denoise_steps * (19 * td + 38 * ts) + sec_vae + tp
    return {"value": 1.0 / sec_per_img, "unit": "images/s", **_host_cpu(), "kind": "port",
            "extrapolated": True, "seconds_per_image": sec_per_img,
            "legs_s": {"double_block": tds, "single_block": tss, "vae_decode_sample": tvs, "redux_prior": tps},
            "sample": f"oracle bf16 on CPU, median of 3: 1 double ({td:.2f}s) + 1 single ({ts:.2f}s) Flux block at S={St + Si}, B=1; "
                      f"VAE decode of a {lat}x{lat} latent ({tv:.2f}s, x{vae_scale:.0f} to {res // 8}^2, x1.96 for the two encodes); "
                      f"Redux prior on one 384^2 image ({tp:.2f}s); extrapolated x{denoise_steps} steps x(19+38) blocks + VAE + prior"}


def cpu_baseline_retrieval(topk: int, budget_s: float = 60.0):
    """BASELINE configs[0] timed IN FULL on the host cores: 1000 synthetic 640x480 images -> PIL bicubic resize + centre crop
    (the reference's preprocess) -> CLIP ViT-B/32 in fp32 (upstream transformers code = what openai-CLIP computes on CPU) ->
    unit-norm -> exact top-k of 16 queries over the 1000 rows (oracle/topk.c).  If the embedding alone would exceed
    ``budget_s`` it stops after the batches that fit and says so."""
    import numpy as np
    from PIL import Image
    from oracle import retrieval as oret
    from oracle import vit as ovit
    from domain_rag_amd import vit as vit_mod
    _pick_threads()
    n_img, bs = 1000, 50
    rng = np.random.default_rng(0)
    vc = vit_mod.VitConfig.clip_vit_b32()
    gp = vit_mod.init_generic_params(vc, 0)
    feats, done = [], 0
    t_pre = t_emb = 0.0
    t_start = time.perf_counter()
    for b0 in range(0, n_img, bs):
        t0 = time.perf_counter()
        batch = []
        for _ in range(bs):
            raw = Image.fromarray(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8))
            im = raw.resize((298, 224), Image.BICUBIC).crop((37, 0, 261, 224))
            batch.append(np.asarray(im))
        x = ovit.normalize_u8(torch.from_numpy(np.stack(batch)), vc.mean, vc.std)
        t1 = time.perf_counter()
        e = ovit.clip_image_embeds(gp, vc.image_size, vc.patch_size, vc.hidden, vc.heads, vc.layers, vc.intermediate, vc.proj_dim,
                                   x, torch.float32)
        feats.append((e / e.norm(dim=-1, keepdim=True)).numpy())
        t2 = time.perf_counter()
        t_pre += t1 - t0; t_emb += t2 - t1
        done += bs
        if time.perf_counter() - t_start > budget_s:
            break
    corpus = np.concatenate(feats, 0)
    q = corpus[:16] + 0.05 * rng.standard_normal((16, 512)).astype(np.float32)
    t0 = time.perf_counter()
    oret.cosine_topk(corpus, q, min(topk, corpus.shape[0]))
    t_top = time.perf_counter() - t0
    total = t_pre + t_emb + t_top
    return {"value": done / total, "unit": "corpus images/s", **_host_cpu(), "kind": "port",
            "legs_s": {"pil_resize_crop": t_pre, "clip_vit_b32_fp32": t_emb, "top%d_16_queries" % topk: t_top},
            "sample": f"BASELINE configs[0] on the host cores: {done} of 1000 synthetic 640x480 images (PIL bicubic preprocess + upstream "
                      f"transformers CLIP ViT-B/32 fp32 in batches of {bs} + oracle/topk.c for 16 queries)"
                      + ("" if done == n_img else f"; stopped at the {budget_s:.0f}s budget")}


# ----------------------------------------------------------------------------------------------- PMC traffic
def _pmc_traffic(kernel_substr: str):
    """HBM/fabric bytes per launch of a kernel from the NEWEST committed rocprofv3 PMC summary (profiles/rNN_pmc_traffic*.json,
    written by scripts/rocpd_pmc.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per the gfx950 guide)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic*.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                pm = json.load(f)
            key = next(k for k in pm if kernel_substr in k)
            return pm[key]["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def _pmc_mfma_util():
    """counter-based matrix-pipe utilisation of the two dominant kernels from the NEWEST committed profiles/rNN_pmc_mfma_util.json
    (scripts/pmc_mfma_util.sh: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x CUs), both from one --pmc pass over this bench's default
    workload; the effective clock = GRBM_GUI_ACTIVE / launch duration of the same pass)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_mfma_util.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                pm = json.load(f)
            out = {"source": os.path.relpath(path, ROOT)}
            for label, subs in (("gemm", ("gemm_bf16_w4p", "gemm_bf16_t256ILi0")), ("attention", ("attention_q64g_kernel", "attention_q64_kernel", "attention_d128_kernelILi8"))):
                key = next(k for sub in subs for k in pm if sub in k)          # (the 64-query kernel since round 4)
                r = pm[key]
                out[label] = {"kernel": key.split("(")[0][-60:], "mfma_util": r["mfma_util"], "clock_ghz": r["clock_ghz"], "avg_us_profiled": r["avg_us_profiled"],
                              "wave_parked_frac": r.get("sq_wait_any_per_wave_cycle"), "wave_issue_stall_frac": r.get("sq_wait_inst_any_per_wave_cycle"),
                              "lds_bank_conflict_frac": r.get("sq_lds_bank_conflict_per_wave_cycle")}
            return out
        except Exception:
            continue
    return None


class PowerSampler:
    """socket power and shader clock of one GPU, sampled with rocm-smi from a side thread while the timed region runs.  The dominant
    kernels of this workload are ended by the socket's power cap (scripts/power_probe.py, profiles/r05_power_probe_socket_power_cap.log:
    the GEMM draws 1379 W of 1400 W at 1.81 GHz on N(0, 1) operands and runs 1633 TFLOP/s at 2.39 GHz and 1023 W on zeros), so a
    throughput figure only compares across boxes next to the clock and the power it was measured at."""

    def __init__(self, device_index: int, period_s: float = 0.5, enabled: bool = True):
        self.smi = "/opt/rocm/bin/rocm-smi" if enabled and os.path.exists("/opt/rocm/bin/rocm-smi") else None
        self.idx, self.period, self.samples, self._stop, self._th = device_index, period_s, [], False, None

    def _loop(self):
        import re
        import subprocess
        while not self._stop:
            try:
                t = subprocess.run([self.smi, "-d", str(self.idx), "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True,
                                   timeout=10).stdout
                pw = re.search(r"Graphics Package Power \(W\):\s*([\d.]+)", t.split("Power Consumption")[-1])
                cap = re.search(r"Max Graphics Package Power \(W\):\s*([\d.]+)", t)
                ck = re.search(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)", t)
                self.samples.append((float(pw.group(1)) if pw else None, float(cap.group(1)) if cap else None, int(ck.group(1)) if ck else None))
            except Exception:
                pass
            time.sleep(self.period)

    def __enter__(self):
        if self.smi:
            import threading
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._th is not None:
            self._th.join(timeout=15)

    def summary(self):
        pw = [x[0] for x in self.samples if x[0] is not None]
        ck = [x[2] for x in self.samples if x[2] is not None]
        cap = [x[1] for x in self.samples if x[1] is not None]
        if not pw:
            return None
        return {"socket_w_mean": sum(pw) / len(pw), "socket_w_max": max(pw), "cap_w": cap[0] if cap else None,
                "sclk_mhz_mean": sum(ck) / len(ck) if ck else None, "sclk_mhz_min": min(ck) if ck else None, "samples": len(pw),
                "source": "rocm-smi --showpower --showclocks every 0.5 s over the timed region (whole pipeline, rank 0's GPU)"}


# ----------------------------------------------------------------------------------------------- workloads
def run_generate(args, d: Dist):
    from domain_rag_amd import ops
    from domain_rag_amd.fill_pipeline import SyntheticFillJob
    job = SyntheticFillJob(batch=args.batch, res=args.res, denoise_steps=args.denoise_steps, device=d.dev, seed=1234 + d.rank)
    for _ in range(args.warmup):
        job.run_batch()
    torch.cuda.synchronize()
    d.barrier()
    # live per-launch timing of the dominant kernel (GEMM) with events on the launch stream: every GEMM / conv launch of
    # the prior, the VAE encodes and the decode, and of every 5th denoise step (all 30 launch the same shapes) of the
    # first timed batch; everything else runs exactly as the product path does (hipGraph replay of the DiT forward)
    rec = ops.GemmRecorder(every=args.roofline_every)
    power = PowerSampler(d.local if not d.shared else 0, enabled=d.rank == 0)
    t0 = time.perf_counter()
    with power:
        for i in range(args.steps):
            job.run_batch(recorder=rec if i == 0 else None)
        torch.cuda.synchronize()
        own = time.perf_counter() - t0                  # this rank's own clock, before the closing barrier
    d.barrier()
    dt = d.max_over_ranks(time.perf_counter() - t0)
    per_rank = d.all_ranks(own)
    images = args.steps * args.batch * d.world
    if d.rank != 0:
        return None
    flops, ms, launches = rec.totals()
    all_gemm = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    bk = rec.by_kernel()                      # the dominant kernel by time: every large Linear (gemm_bf16_w4p since round 5, gemm_bf16_t256<0> before)
    dom = max(bk, key=lambda k: bk[k][1]) if bk else "gemm_bf16_w4p"
    dn, dms, dfl = bk.get(dom, (0, 0.0, 0.0))
    traffic, traffic_src = _pmc_traffic({"gemm_bf16_w4p": "gemm_bf16_w4p", "gemm_bf16_t256<0>": "gemm_bf16_t256ILi0"}.get(dom, dom))
    achieved = dfl / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
    # the second-hottest kernel (a quarter of the step): the attention launches of the same sampled steps, bracketed the same way
    ak = rec.attention_by_kernel()
    adom = max(ak, key=lambda k: ak[k][1]) if ak else None
    attention = None
    if adom is not None:
        an, ams, afl = ak[adom]
        a_ach = afl / (ams * 1e-3) / 1e12 if ams > 0 else 0.0
        attention = {"kernel": adom, "launches_timed": an, "avg_launch_ms": ams / max(an, 1), "achieved": a_ach, "peak": MFMA_BF16_PEAK_TF,
                     "unit": "TFLOP/s", "frac": a_ach / MFMA_BF16_PEAK_TF,
                     "flops_per_launch": afl / max(an, 1),
                     "note": "algorithmic 4 S^2 128 flops per (batch, head) / event-bracketed launch time on the launch stream; the fused q "
                             "preparation (RMSNorm + RoPE of the query rows) runs inside the same launch and is not counted as flops",
                     "all_kernels": {k: {"launches": v[0], "avg_launch_ms": v[1] / max(v[0], 1),
                                         "achieved": v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0} for k, v in ak.items()}}
    pw = power.summary()
    if pw is not None and pw.get("socket_w_mean"):
        # energy of the timed region on rank 0's GPU: mean socket power x the region's seconds / the images this GPU produced
        pw["joules_per_image"] = pw["socket_w_mean"] * dt / (args.steps * args.batch)
        pw["joules_note"] = "mean sampled socket W x timed seconds / images of this GPU (rank 0); whole pipeline, idle gaps included"
    out = {
        "metric": f"composited images/sec @{args.res}^2, {args.denoise_steps} Flux-Redux steps", "value": images / dt, "unit": "images/s",
        "n_gpus": d.world, "rccl_ranks": d.rccl_ranks if d.rccl_ranks is not None else 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "per_rank": {"seconds": per_rank, "images_per_s": [args.steps * args.batch / t for t in per_rank],
                     "clock": "value uses the max-over-ranks time between the two barriers; these are each rank's own "
                              "seconds to finish its batches (before the closing barrier)"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"Flux-Redux outpaint (Fill) {args.res}x{args.res}, {args.denoise_steps} steps, "
                               f"batch={args.batch} per GPU (BASELINE configs[2])",
                   "stages": job.stages(), "global_batch": args.batch * d.world, "parallelism": f"dp{d.world}",
                   "weights": "seeded random init of the FLUX.1-Fill-dev architecture"},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": MFMA_BF16_PEAK_TF,
                     "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_PEAK_TF, "traffic": traffic,
                     "traffic_note": f"bytes/launch of {dom} from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                     f"({traffic_src}); includes Infinity-Cache hits",
                     "launches_timed": dn, "sampled": f"all non-DiT stages + every {rec.every}th denoise step of the first timed batch",
                     "avg_launch_ms": dms / max(dn, 1),
                     "kernel_time_share_of_gemm": dms / ms if ms > 0 else None,
                     "all_gemm_conv_launches": {"launches": launches, "achieved": all_gemm, "avg_launch_ms": ms / max(launches, 1),
                                                "by_kernel": {k: {"launches": v[0], "avg_launch_ms": v[1] / max(v[0], 1),
                                                                  "achieved": v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0}
                                                              for k, v in bk.items()}},
                     "e2e_mfma_frac": job.flops_per_image() * images / d.world / dt / (MFMA_BF16_PEAK_TF * 1e12),
                     "attention": attention,
                     "mfma_util_pmc": _pmc_mfma_util(), "power": pw},
    }
    if not args.no_side_configs and d.world == 1:
        del job
        torch.cuda.empty_cache()
        out["side_configs"] = {"configs1": side_config1(d.dev)}
        torch.cuda.empty_cache()
        out["side_configs"]["stage3_2048"] = side_stage3(d.dev)
        torch.cuda.empty_cache()
        out["side_configs"]["retrieval"] = side_retrieval(d.dev, cpu_budget_s=0.0 if args.no_cpu_baseline else 10.0)
    if not args.no_cpu_baseline and d.world == 1:
        out["cpu_baseline"] = cpu_baseline_generate(args.res, args.denoise_steps)
    return out


def side_config1(dev) -> dict:
    """BASELINE configs[1] as a side field (a parity-test configuration, not the bench line): Flux-schnell shape (no guidance
    embedding, 512 T5 tokens), 512x512, 4 denoise steps, batch 1, VAE decode included; outside the timed region."""
    from domain_rag_amd import vae as vae_mod
    from domain_rag_amd.engine import FluxTxt2ImgHIP, generator_noise, pack_noise
    from domain_rag_amd.flux import FluxTransformerHIP
    from domain_rag_amd.flux_params import FluxConfig, init_params
    cfg = FluxConfig(in_channels=64, guidance_embeds=False)
    tr = FluxTransformerHIP(cfg, init_params(cfg, seed=0, device=dev), dev)
    vcfg = vae_mod.VaeConfig()
    pipe = FluxTxt2ImgHIP(tr, vae_mod.FluxVaeHIP(vcfg, vae_mod.init_params(vcfg, seed=1, device=dev), dev))
    g = torch.Generator(device=dev).manual_seed(2)
    pe = torch.randn(1, 512, 4096, device=dev, generator=g).bfloat16()
    pp = torch.randn(1, 768, device=dev, generator=g).bfloat16()
    noise = pack_noise(generator_noise(0, 1, 512, 512, 1)[0])
    run = lambda: pipe(pe, pp, height=512, width=512, guidance_scale=0.0, num_inference_steps=4, noise_tokens=noise)   # noqa: E731
    run(); run()
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    S = 512 + 1024
    flops = 4 * (57 * (2 * 12 * 3072 * 3072 * S + 2 * 2 * S * S * 3072)) + 2.6e12
    return {"workload": "BASELINE configs[1]: Flux-schnell shape 512x512, 4 steps, batch=1 (latency case: 1536 joint rows)",
            "value": 1.0 / dt, "unit": "images/s", "ms_per_image": dt * 1e3, "achieved_tflops": flops / dt / 1e12,
            "mfma_frac": flops / dt / (MFMA_BF16_PEAK_TF * 1e12)}


def side_stage3(dev, res: int = 2048, batch: int = 2, steps: int = 3) -> dict:
    """the reference's stage 3 composites at sides in [1024, 2800] (outpainting_updown_sampling_redux.py:72-82, 104-105; UODD is up-scaled to
    2048): the same Fill pipeline at 2048 x 2048 (17 625 joint tokens: attention is > half of a block's FLOPs), batch 2, 3 denoise steps —
    whole-pipeline time, per-kernel GEMM TFLOP/s (events around every GEMM launch) and the attention kernel alone at that shape.  A side
    field, outside the timed region; oracle parity at these sizes: tests/test_gpu_stage3_sizes.py."""
    import math
    from domain_rag_amd import ops
    from domain_rag_amd.fill_pipeline import SyntheticFillJob
    job = SyntheticFillJob(batch=batch, res=res, denoise_steps=steps, device=dev, seed=77)
    job.run_batch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    job.run_batch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = ops.GemmRecorder(every=1)
    job.run_batch(recorder=rec)
    flops, ms, launches = rec.totals()
    bk = rec.by_kernel()
    del job
    torch.cuda.empty_cache()
    S, H = (res // 16) ** 2 + 512 + 729, 24
    D = H * 128
    g = torch.Generator(device=dev).manual_seed(3)
    qkv = torch.randn(batch, S, 3 * D, generator=g, device=dev).bfloat16()
    s_pad = (S + 63) // 64 * 64
    vt = torch.empty(batch, H, 128, s_pad, dtype=torch.bfloat16, device=dev)
    ops.qk_norm_rope_vt(qkv, vt, None, None, None, None, None, None, batch, S, H, 3 * D, 0)
    o = torch.empty(batch, S, D, dtype=torch.bfloat16, device=dev)
    run = lambda: ops.attention(qkv, qkv.view(-1)[D:], vt, o, batch, S, H, 3 * D, S * 3 * D, D, S * D, 1 / math.sqrt(128))   # noqa: E731
    run(); run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    att_ms = e0.elapsed_time(e1) / 5
    att_tf = 4.0 * S * S * 128 * H * batch / (att_ms * 1e-3) / 1e12
    return {"workload": f"Fill pipeline {res}x{res} (S = {S} joint tokens), batch {batch}, {steps} denoise steps — the reference's stage-3 sizes",
            "seconds_per_batch": dt, "images_per_s_at_30_steps_extrapolated": None,
            "gemm": {"launches": launches, "achieved_tflops": flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                     "by_kernel": {k: {"launches": v[0], "achieved_tflops": v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0} for k, v in bk.items()}},
            "attention": {"shape": f"B={batch}, S={S}, 24 heads x 128", "ms_per_launch": att_ms, "achieved_tflops": att_tf, "mfma_frac": att_tf / MFMA_BF16_PEAK_TF}}


def side_retrieval(dev, k: int = 100, cpu_budget_s: float = 10.0) -> dict:
    """the retrieval side of the path as a side field of the DEFAULT line (N = 1 only, about 15 s; SURVEY 8(d)'s retrieval rows):
    the HBM-bound scan and the whole exact top-k call at N = 118 287 (BASELINE configs[4]'s corpus: 242 MB, Infinity-Cache
    resident across repeats) AND at N = 1 000 000 (2.05 GB: HBM, not cache), Q = 1 / 16 / 64 queries per call; the float32 CLIP
    tower on 4096 resident crops; BASELINE configs[0] on the host cores (a 10 s sample of it).  `--workload retrieval` is the full second line."""
    from domain_rag_amd import ops
    from domain_rag_amd.retrieval import embed_images, load_clip
    ev = lambda: torch.cuda.Event(enable_timing=True)            # noqa: E731

    def timed(fn, reps=20, warm=3):
        for _ in range(warm):
            fn()
        a, b = ev(), ev()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    out = {"k": k, "d": 512, "peak_GBs": HBM_PEAK_GBS, "measured_achievable_GBs": HBM_MEASURED_GBS, "shapes": {}}
    g = torch.Generator(device=dev).manual_seed(11)
    for N in (118287, 1000000):
        corpus = torch.randn(N, 512, device=dev, generator=g)
        corpus /= corpus.norm(dim=-1, keepdim=True)              # test-data preparation, outside every timed region
        planted = torch.randint(0, N, (64,), generator=g, device=dev)
        qs = (corpus[planted] + 0.02 * torch.randn(64, 512, generator=g, device=dev)).contiguous()
        row = {}
        for Q in (1, 16, 64):
            q = qs[:Q].contiguous()
            sc = ops.cosine_scores(corpus, q)
            scan_ms = timed(lambda: ops.cosine_scores(corpus, q, out=sc))
            call_ms = timed(lambda: ops.cosine_topk(corpus, q, k))
            D, I = ops.cosine_topk(corpus, q, k)
            scan_bytes = N * 512 * 4 + Q * 512 * 4 + Q * sc.shape[1] * 4
            call_bytes = N * 512 * 4 + Q * 512 * 4 + Q * k * 12
            mfma_ms = 2.0 * N * 512 * Q / (MFMA_F32_PEAK_TF * 1e12) * 1e3
            row[f"Q{Q}"] = {"scan_us": scan_ms * 1e3, "scan_GBs": scan_bytes / scan_ms / 1e6, "scan_frac_hbm": scan_bytes / scan_ms / 1e6 / HBM_PEAK_GBS,
                            "scan_tflops_f32": 2.0 * N * 512 * Q / scan_ms / 1e9,
                            "topk_call_us": call_ms * 1e3, "topk_GBs": call_bytes / call_ms / 1e6,
                            "topk_scan_plus_select_frac": call_bytes / call_ms / 1e6 / HBM_PEAK_GBS,
                            "f32_mfma_floor_us": mfma_ms * 1e3, "bound": "mfma_f32" if mfma_ms > call_bytes / (HBM_PEAK_GBS * 1e6) else "hbm",
                            "planted_neighbour_first": f"{int((I[:, 0] == planted[:Q]).sum().item())}/{Q}"}
        out["shapes"][f"N{N}"] = row
        del corpus
    torch.cuda.empty_cache()
    out["formula"] = ("topk bytes = N*512*4 + Q*512*4 + Q*k*12 (SURVEY 8d), ONE corpus pass for Q <= 64; scan bytes add the Q*ceil64(N)*4 "
                      "score rows the scan-only entry writes.  At Q = 64 the exact-f32 matrix core, not HBM, is the floor: 2*N*512*Q flop at "
                      f"{MFMA_F32_PEAK_TF} TFLOP/s (f32_mfma_floor_us)")
    # the tower: 4096 resident 224^2 crops through the float32 CLIP ViT-B/32
    model, _ = load_clip("ViT-B/32", dev, weights=None, seed=0, precision="fp32")
    crops = synthetic_crops(0, 4096, dev)
    embed_images(model, crops, 1024)
    rec = ops.GemmRecorder(f32=True)
    ops.set_recorder(rec)
    try:
        embed_images(model, crops, 1024)
    finally:
        ops.set_recorder(None)
    e_ms = timed(lambda: embed_images(model, crops, 1024), reps=3, warm=0)
    flops, ms, launches = rec.totals()
    out["embed"] = {"images_per_s": 4096 / e_ms * 1e3, "kernel": "conv2d_f32_kernel", "achieved_tflops": flops / (ms * 1e-3) / 1e12 if ms > 0 else None,
                    "peak_tflops": MFMA_F32_PEAK_TF, "launches_timed": launches, "batch": 1024}
    del model, crops
    torch.cuda.empty_cache()
    if cpu_budget_s > 0:
        out["cpu_config0"] = cpu_baseline_retrieval(k, budget_s=cpu_budget_s)
    return out


def synthetic_crops(lo: int, hi: int, device, chunk: int = 2048) -> torch.Tensor:
    """uint8 [hi-lo, 224, 224, 3]: rows lo..hi of a corpus that is a function of the GLOBAL row index only (each chunk of
    `chunk` rows has its own seed), so any sharding embeds the same images as the single-GPU run"""
    parts = []
    c = lo // chunk
    while c * chunk < hi:
        g = torch.Generator(device=device).manual_seed(100003 * c + 17)
        blk = torch.randint(0, 256, (chunk, 224, 224, 3), generator=g, device=device, dtype=torch.uint8)
        a, b = max(lo, c * chunk), min(hi, (c + 1) * chunk)
        parts.append(blk[a - c * chunk: b - c * chunk])
        c += 1
    return torch.cat(parts, 0) if len(parts) > 1 else parts[0].clone()


def run_retrieval(args, d: Dist):
    from domain_rag_amd import ops
    from domain_rag_amd.retrieval import allgather_rows, embed_images, load_clip, shard_bounds
    N, Q, k = args.corpus, args.queries, args.topk
    dev = d.dev
    model, _ = load_clip("ViT-B/32", dev, weights=None, seed=0, precision=args.clip_precision)
    s, e = shard_bounds(N, d.world, d.rank)
    crops = synthetic_crops(s, e, dev)
    qs, qe = shard_bounds(Q, d.world, d.rank)
    gq = torch.Generator(device=dev).manual_seed(7)
    planted = torch.randint(0, N, (Q,), generator=gq, device=dev)                 # each query sits next to one corpus row
    qnoise = 0.02 * torch.randn(Q, 512, generator=gq, device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)                             # noqa: E731

    def step(timers=None, rec=None):
        t = [ev() for _ in range(4)]
        t[0].record()
        ops.set_recorder(rec)
        try:
            local = embed_images(model, crops, args.embed_batch)
        finally:
            ops.set_recorder(None)
        t[1].record()
        corpus = allgather_rows(local, N)
        t[2].record()
        queries = (corpus[planted[qs:qe]] + qnoise[qs:qe]).contiguous()
        D, I = ops.cosine_topk(corpus, queries, k) if qe > qs else (None, None)
        t[3].record()
        if timers is not None:
            timers.append(t)
        return corpus, queries, D, I

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    d.barrier()
    timers: list = []
    rec = ops.GemmRecorder(f32=True) if args.clip_precision == "fp32" else ops.GemmRecorder()
    t0 = time.perf_counter()
    for i in range(args.steps):
        corpus, queries, D, I = step(timers, rec if i == 0 else None)
    torch.cuda.synchronize()
    d.barrier()
    dt = d.max_over_ranks(time.perf_counter() - t0)
    # checks (outside the timed region): planted neighbours come back first on this rank's queries
    hits = int((I[:, 0] == planted[qs:qe]).sum().item()) if qe > qs else 0
    # the HBM-bound scan kernel alone, events on the launch stream
    q16 = queries[: min(16, queries.shape[0])]                 # the roofline figure is quoted at 16 queries per pass, as in rounds 1-2
    sc = ops.cosine_scores(corpus, q16)
    reps = 20
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(reps):
        ops.cosine_scores(corpus, q16, out=sc)
    e1.record()
    f0, f1 = ev(), ev()
    f0.record()
    for _ in range(reps):
        ops.cosine_topk(corpus, q16, k)
    f1.record()
    torch.cuda.synchronize()
    scan_ms, topk_ms = e0.elapsed_time(e1) / reps, f0.elapsed_time(f1) / reps
    if d.rank != 0:
        return None
    qn = q16.shape[0]
    scan_bytes = N * 512 * 4 + qn * 512 * 4 + qn * ((N + 63) // 64 * 64) * 4        # corpus + queries read, score rows written
    topk_bytes = N * 512 * 4 + qn * 512 * 4 + qn * k * 12                             # SURVEY §8(d) formula
    emb_ms = statistics.mean(t[0].elapsed_time(t[1]) for t in timers)
    ag_ms = statistics.mean(t[1].elapsed_time(t[2]) for t in timers)
    tk_ms = statistics.mean(t[2].elapsed_time(t[3]) for t in timers)
    cap = shard_bounds(N, d.world, 0)[1]
    ag_bytes = (d.world - 1) * cap * 512 * 4                      # what one GPU receives: the other ranks' padded shards
    flops, ms, launches = rec.totals()
    out = {
        "metric": f"corpus images/sec through embed + all-gather + top-{k} (CLIP ViT-B/32, N={N})", "value": args.steps * N / dt,
        "unit": "images/s", "n_gpus": d.world, "rccl_ranks": d.rccl_ranks if d.rccl_ranks is not None else 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32" if args.clip_precision == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": f"retrieval: {N} synthetic 224^2 uint8 crops resident in HBM -> CLIP ViT-B/32 ({args.clip_precision}) -> "
                               f"unit-norm f32 [N,512], rank-sharded; one all-gather; exact top-{k} of {Q} queries (BASELINE configs[4] corpus)",
                   "corpus": N, "queries": Q, "k": k, "embed_batch": args.embed_batch, "parallelism": f"corpus-shard{d.world}",
                   "weights": "seeded random init of the CLIP ViT-B/32 architecture"},
        "stage_ms": {"embed_shard": emb_ms, "all_gather": ag_ms, "topk_shard_of_queries": tk_ms},
        "planted_neighbour_first": f"{hits}/{qe - qs} on rank 0's query shard",
        "roofline": {"bound": "hbm", "kernel": "ip_scan_kernel", "achieved": scan_bytes / (scan_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": scan_bytes / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "frac_of_measured_achievable": scan_bytes / (scan_ms * 1e-3) / 1e9 / HBM_MEASURED_GBS,
                     "traffic": _pmc_traffic("ip_scan_kernel")[0],
                     "traffic_note": "fabric-side bytes per launch of the scan at this N, FETCH_SIZE scaled by the factor measured on the same "
                                     "kernel at N = 1 000 000 (past the Infinity Cache), + WRITE_SIZE; Infinity-Cache hits are counted: a 242 MB "
                                     "corpus stays cache-resident across repeated launches",
                     "avg_launch_ms": scan_ms, "launches_timed": reps,
                     "algorithmic_bytes_per_launch": scan_bytes, "queries_per_pass": qn,
                     "topk_scan_plus_select": {"avg_call_ms": topk_ms, "achieved": topk_bytes / (topk_ms * 1e-3) / 1e9,
                                               "frac": topk_bytes / (topk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "bytes": topk_bytes, "formula": "N*512*4 + Q*512*4 + Q*k*12 (SURVEY 8d)"}},
        "embed": {"kernel": "conv2d_f32_kernel" if args.clip_precision == "fp32" else "gemm_bf16", "bound": "mfma",
                  "achieved": flops / (ms * 1e-3) / 1e12 if ms > 0 else None, "unit": "TFLOP/s",
                  "peak": MFMA_F32_PEAK_TF if args.clip_precision == "fp32" else MFMA_BF16_PEAK_TF,
                  "launches_timed": launches, "images_per_s_per_gpu": (e - s) / (emb_ms * 1e-3)},
        "allgather": {"bytes_received_per_gpu": ag_bytes, "ms": ag_ms,
                      "achieved": ag_bytes / (ag_ms * 1e-3) / 1e9 if d.world > 1 and ag_ms > 0 else None, "unit": "GB/s",
                      "peak": XGMI_LINKS * XGMI_LINK_GBS, "note": "one all_gather_into_tensor of zero-padded [ceil(N/W), 512] f32 shards "
                                                                   "(RCCL over xGMI); single rank: no collective"},
    }
    if not args.no_cpu_baseline and d.world == 1:
        out["cpu_baseline"] = cpu_baseline_retrieval(k)
    return out


def main():
    args = parse()
    self_launch_if_needed(args)
    d = Dist(args)
    try:
        if args.selftest_launcher:
            launcher_selftest(args, d)
            return
        import __graft_entry__ as ge
        if d.rank == 0:
            ge.build()
        d.barrier()
        out = run_generate(args, d) if args.workload == "generate" else run_retrieval(args, d)
        if d.rank == 0:
            if d.shared:
                out["debug_share_gpu"] = "all ranks on GPU 0 over gloo: control-flow check only, NOT a measurement"
            print(json.dumps(out), flush=True)
    finally:
        d.close()


if __name__ == "__main__":
    main()
