"""Within-process, interleaved A/B of the fused head's FORWARD across workgroup shapes: the shipped 8-wave dual-plane core
(256 x 256, two waves per SIMD) against the one-wave-per-SIMD core (256 x 384 / 256 x 320, 512 registers per wave).
One micro-batch per shape (7B: 8192 x 3584 x 152 064, 32B: 8192 x 5120 x 152 064), fp32 weight (two planes) and bf16
weight (one plane); every variant's outputs are compared with the default's before it is timed.

    python scripts/lmhead_fwd_tile_ab.py [--rounds 4] [--iters 5] [--shapes 7b,32b] [--keep]
"""
import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pipelinerl_amd.fused_head import FusedLmHead  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--shapes", default="7b,32b")
ap.add_argument("--tiles", default="default,256x384,256x384:1,256x320,256x320:1")
ap.add_argument("--keep", action="store_true", help="also time the forward that keeps its logits")
ap.add_argument("--rows", type=int, default=8192)
args = ap.parse_args()
dev = torch.device("cuda", 0)
SHAPES = {"7b": (3584, 152064), "32b": (5120, 152064), "0p5b": (896, 151936)}
tiles = args.tiles.split(",")


def set_tile(t):
    """`default` | `<tile>` | `<tile>:<PRL_LMHEAD_EXP value>`"""
    tile, _, exp = t.partition(":")
    if tile == "default":
        os.environ.pop("PRL_LMHEAD_TILE", None)
    else:
        os.environ["PRL_LMHEAD_TILE"] = tile
    if exp:
        os.environ["PRL_LMHEAD_EXP"] = exp
    else:
        os.environ.pop("PRL_LMHEAD_EXP", None)


for shape in args.shapes.split(","):
    H, V = SHAPES[shape]
    T = args.rows
    g = torch.Generator(device=dev).manual_seed(7)
    hidden = torch.empty(1, T, H, device=dev).normal_(generator=g).to(torch.bfloat16)
    ids = torch.randint(3, V, (1, T), device=dev, generator=g)
    for wname, wdt in (("fp32_two_planes", torch.float32), ("bf16_one_plane", torch.bfloat16)):
        W = (torch.empty(V, H, device=dev).normal_(0.0, 0.02, generator=g)).to(wdt)
        head = FusedLmHead(W, backward=False, keep_logits=False)
        head.refresh()
        planes = 2 if wdt == torch.float32 else 1
        flop = planes * 2.0 * T * V * H
        set_tile("default")
        ref = head.logprob_entropy(hidden, ids, 1.0)
        torch.cuda.synchronize()
        times = {t: [] for t in tiles}
        err = {}
        for t in tiles:
            set_tile(t)
            out = head.logprob_entropy(hidden, ids, 1.0)
            torch.cuda.synchronize()
            err[t] = [float((a - b).abs().max()) for a, b in zip(out[:3], ref[:3])]
        for _ in range(args.rounds):
            for t in tiles:
                set_tile(t)
                head.logprob_entropy(hidden, ids, 1.0)
                torch.cuda.synchronize()
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
                for a, b in ev:
                    a.record()
                    head.logprob_entropy(hidden, ids, 1.0, keep=args.keep)
                    b.record()
                torch.cuda.synchronize()
                times[t].append(float(np.median([a.elapsed_time(b) for a, b in ev])))
        set_tile("default")
        rec = {"shape": shape, "rows": T, "hidden": H, "vocab": V, "weight": wname, "keep_logits": args.keep,
               "ms": {t: round(float(np.median(v)), 3) for t, v in times.items()},
               "executed_tflops": {t: round(flop / (float(np.median(v)) * 1e-3) / 1e12, 1) for t, v in times.items()},
               "max_abs_diff_vs_default(nlp,ent,lse2)": err}
        print(json.dumps(rec), flush=True)
        del W, head
        torch.cuda.empty_cache()
