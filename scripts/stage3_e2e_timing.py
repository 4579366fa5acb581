"""Wall-clock of stage 3's per-sample loop on one GPU, full-size synthetic weights: 500x375 originals (-> 1365x1024 frames),
5 backgrounds per sample, strength 0.6 x 50 = 30 steps (Camouflage settings), PNGs encoded inline (the reference's pattern),
in background worker processes, and on the device (--png gpu, the CLI's default).  Prints seconds per sample and per composite."""
import argparse
import json
import os
import random
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=3)
    ap.add_argument("--dataset", default="Camouflage")
    a = ap.parse_args()
    ge.build()
    from PIL import Image
    from domain_rag_amd.cli import stage3_outpaint as s3
    from domain_rag_amd.engine import Engine
    from domain_rag_amd.io_pool import ImageWriter
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:375, 0:500]

    def picture(h, w):
        base = (np.sin(xx[:h, :w] / 37.0 + rng.random() * 6) + np.cos(yy[:h, :w] / 53.0)) * 60 + 128
        return np.clip(base[..., None] + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)

    with tempfile.TemporaryDirectory() as root:
        os.chdir(root)
        ds = a.dataset
        os.makedirs(f"datasets/{ds}/annotations"); os.makedirs(f"datasets/{ds}/train")
        images, anns, sdirs = [], [], {}
        for i in range(a.samples):
            name = f"s{i:02d}"
            Image.fromarray(picture(375, 500)).save(f"datasets/{ds}/train/{name}.jpg")
            images.append({"id": i + 1, "file_name": f"{name}.jpg", "width": 500, "height": 375})
            anns.append({"id": i + 1, "image_id": i + 1, "bbox": [150, 120, 180, 140], "category_id": 1})
            sd = f"result/{ds}_1shot_retrieval/results_x/{name}"
            os.makedirs(sd)
            big = np.kron(picture(256, 256), np.ones((4, 4, 1), np.uint8))
            for r in range(1, 6):
                Image.fromarray(np.roll(big, r * 37, axis=1)).save(f"{sd}/generated_image_rank{r}.png")
            sdirs[name] = sd
        json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "animal"}]}, open(f"datasets/{ds}/annotations/1_shot.json", "w"))
        args = s3.build_parser().parse_args(["--dataset", ds, "--shot", "1", "--synthetic-weights", "--seed", "1"])
        t = time.time()
        engine = Engine("fill", args.model_root, synthetic=True, device=torch.device("cuda", 0))
        print(f"engine ready in {time.time() - t:.1f} s", flush=True)
        s3.process_sample(engine, args, ds, "s00", sdirs["s00"], 1, "warm", random.Random(0), None)       # graph capture, caches
        # where the time goes: GPU pipeline (enqueue + wait) vs Redux prior vs everything else on the host
        acc = {"pipe": 0.0, "prior": 0.0}
        pipe0, prior0 = engine.pipe, engine.prior_embeds

        class TimedPipe:
            def __call__(self, *a_, **k_):
                torch.cuda.synchronize(); t_ = time.time()
                out = pipe0(*a_, **k_)
                torch.cuda.synchronize(); acc["pipe"] += time.time() - t_
                return out

        def timed_prior(*a_, **k_):
            torch.cuda.synchronize(); t_ = time.time()
            out = prior0(*a_, **k_)
            torch.cuda.synchronize(); acc["prior"] += time.time() - t_
            return out

        engine.pipe, engine.prior_embeds = TimedPipe(), timed_prior
        for label, workers, png_mode in (("inline Image.save", 0, "host"), ("background encoders", 4, "host"), ("device PNG encoder", 4, "gpu")):
            args.png = png_mode
            w = ImageWriter(workers)
            acc["pipe"] = acc["prior"] = 0.0
            torch.cuda.synchronize()
            t = time.time()
            logs = [s3.process_sample(engine, args, ds, n, sdirs[n], 1, label.split()[0], random.Random(0), w) for n in sorted(sdirs)]
            errs = w.flush()
            dt = time.time() - t
            w.close()
            ok = sum(lg["status"] == "completed" for lg in logs)
            n_img = sum(len(lg["outpainted_images"]) for lg in logs)
            print(f"{label:22s}: {dt / len(logs):6.2f} s/sample  {dt / max(n_img, 1):5.2f} s/composite  ({ok}/{len(logs)} samples ok, {n_img} composites, "
                  f"frame {logs[0]['upscaled_resolution']}, write errors {len(errs)}); per sample: Fill pipeline {acc['pipe'] / len(logs):.2f} s, "
                  f"Redux prior {acc['prior'] / len(logs):.2f} s, other host work {(dt - acc['pipe'] - acc['prior']) / len(logs):.2f} s", flush=True)
        os.chdir("/")


if __name__ == "__main__":
    main()
