#!/usr/bin/env python3
"""Headline benchmark: RVQ frames/s (all 8 levels) + real-time factor, bf16, batch 16 per GPU.

One "step" = one pass of the hot path over one synthetic batch: AR level-0 generation
(prefill + KV-cache decode, R frames) followed by the 7 NAR levels -- BASELINE.json configs[1]:
16 utterances, 80 phonemes, 225 prompt frames (3 s), R = 450 output frames (6 s), random-init
weights of the repo-default architecture, stop token suppressed through the AR head's bias so
every implementation generates exactly R frames.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: utterances shard by rank (one independent batch of 16 per GPU, weak scaling); the only
collective is an RCCL all-gather of the generated codes and per-rank timings.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from vall_e_amd import config as vcfg            # noqa: E402
from vall_e_amd.models import get_model          # noqa: E402
from vall_e_amd.synth import synth_batch         # noqa: E402
from vall_e_amd.weights import synth_state_dict  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak
FRAME_RATE = 75.0


def decode_step_bytes(cfg, B, ctx):
    """Algorithmic HBM bytes of one AR decode step, bf16, KV cache."""
    d, L, ff = cfg.dim, cfg.layers, cfg.dim * cfg.ffn
    weights = L * (4 * d * d + 3 * d * ff) * 2
    head = (cfg.audio_tokens + 1) * d * 2 + (cfg.audio_tokens + 1) * 4
    norms = (2 * L + 1) * d * 2
    kv_row = L * 2 * d * 2
    return weights + head + norms + sum(c * kv_row for c in ctx) + B * kv_row + B * d * 2


def dense_pass_flops(cfg, lens, out_rows, vocab):
    """Algorithmic FLOPs of one dense pass over packed utterances: linears 2*params per
    token, causal attention 2*T^2*d per layer and utterance, head only on the rows that are used."""
    d, L, ff = cfg.dim, cfg.layers, cfg.dim * cfg.ffn
    per_tok = 2 * L * (4 * d * d + 3 * d * ff)
    return sum(lens) * per_tok + sum(2 * t * t * d * L for t in lens) + 2 * out_rows * d * vocab


def cpu_baseline(cfg, sd, phns, proms, langs, R):
    """The oracle (CPU restatement of the reference, fp32, no KV cache like the reference) timed
    on this box's host cores on a bounded sample; extrapolated linearly in context."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import valle_oracle as vo
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    o = vo.Oracle(cfg, {k: v for k, v in sd.items()}, dtype=torch.float32)
    B = len(phns)
    empty = [torch.zeros(0, dtype=torch.int64) for _ in range(B)]
    plans = o._plans(phns, langs, proms, empty, 0)
    ctx0 = plans[0].pos.shape[0]
    def time_ar(plans_, n):
        t0 = time.perf_counter()
        with torch.inference_mode():
            for _ in range(n):
                o.forward(plans_, rows=[1] * B)
        return (time.perf_counter() - t0) / n

    g = torch.Generator().manual_seed(3)
    n_ar0, n_ar1 = 3, 2                                        # steps timed at the first and at the middle context
    t_ar0 = time_ar(plans, n_ar0)
    half = [torch.randint(0, 1024, (R // 2,), generator=g) for _ in range(B)]
    t_ar1 = time_ar(o._plans(phns, langs, proms, half, 0), n_ar1)
    Bn = 8                                                     # one NAR level on half the batch
    resp = [torch.randint(0, 1024, (R, 1), generator=g) for _ in range(Bn)]
    nplans = o._plans(phns[:Bn], langs[:Bn], proms[:Bn], resp, 1)
    t0 = time.perf_counter()
    with torch.inference_mode():
        o.forward(nplans, rows=[R] * Bn)
    t_nar = (time.perf_counter() - t0) * (B / Bn)
    # recompute AR: step n costs t(ctx0 + n), a line through the two measured contexts
    slope = (t_ar1 - t_ar0) / (R // 2)
    ar_total = sum(t_ar0 + slope * n for n in range(R))
    total = ar_total + 7 * t_nar
    return {
        "value": round(B * R / total, 3), "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": (f"oracle (fp32 restatement, full recompute per AR step as the reference does): {n_ar0} AR steps "
                   f"at ctx {ctx0} B={B} = {t_ar0:.2f} s/step and {n_ar1} at ctx {ctx0 + R // 2} = {t_ar1:.2f} s/step, one NAR "
                   f"level at T={ctx0 + R} on {Bn}/{B} of the batch = {t_nar:.2f} s/level scaled; whole job = the line through "
                   f"the two AR points summed over {R} steps + 7 NAR levels = {total:.0f} s"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=450)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; VALLE_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs
        # than ranks (ranks then share devices: local_rank % device_count)
        dist.init_process_group(os.environ.get("VALLE_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = vcfg.repo_default_model()
    B, R = args.batch, args.frames
    sd = synth_state_dict(cfg, 1234, stop_bias=-30000.0)
    model = get_model(cfg).load_state_dict(sd).to(dev, torch.bfloat16)
    phns, proms, langs = synth_batch(1234 + rank, B, n_phn=80, n_prom=225)
    tasks = ["tts"] * B
    prof, nar_prof = [], []

    def step(profile=None):
        codes = model(task_list=tasks, phns_list=phns, proms_list=proms, lang_list=langs, max_duration=R,
                      temperature=0.0, step_chunk=R, profile=profile)
        if profile is not None:
            n0, n1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n0.record()
        full = model(task_list=tasks, phns_list=phns, proms_list=proms, lang_list=langs,
                     resps_list=codes, temperature=0.0)
        if profile is not None:
            n1.record()
            nar_prof.append((n0, n1))
        out = torch.stack([f.to(torch.int16) for f in full])            # (B, R, 8)
        if dist is not None:
            raw = out.view(torch.uint8)                                 # RCCL has no int16: ship the bytes
            if dist.get_backend() == "gloo":                            # the CPU rehearsal backend gathers host tensors
                raw = raw.cpu()
            gathered = [torch.empty_like(raw) for _ in range(world)]
            dist.all_gather(gathered, raw)                              # RCCL: codes of every rank
            assert all(g.shape == raw.shape for g in gathered)
        return out

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    assert out.shape == (B, R, 8), out.shape
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(profile=prof)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    frames = world * B * R * args.steps
    ms_per_step = elapsed / args.steps * 1e3
    # roofline of the dominant kernel group: the AR decode step (weights + KV cache streamed once)
    dec_ms, dec_steps, dec_bytes = 0.0, 0, 0.0
    for e0, e1, n, ctx0 in prof:
        dec_ms += e0.elapsed_time(e1)
        dec_steps += n
        for j in range(n):
            dec_bytes += decode_step_bytes(cfg, B, [c + j for c in ctx0])
    roofline = None
    traffic, traffic_src = None, None
    pmc = ROOT / "profiles" / "r1_pmc_decode_layer.json"
    if dec_steps and pmc.exists():
        # HBM bytes per decode step from the PMC passes committed under profiles/ (FETCH_SIZE x2 on gfx950 +
        # WRITE_SIZE, collected on the same kernels at the bench's mean context): measured / algorithmic of one
        # layer, applied to this run's algorithmic bytes per step
        ratio = json.loads(pmc.read_text())["layer_traffic_over_algorithmic"]
        traffic = int(dec_bytes / dec_steps * ratio)
        traffic_src = f"profiles/{pmc.name}: measured/algorithmic = {ratio} per layer at ctx 537"
    if dec_steps:
        achieved = dec_bytes / (dec_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "AR decode step (valle_ar_decode: all kernels of one step)",
                    "us_per_decode_step": round(dec_ms * 1e3 / dec_steps, 2),
                    "algorithmic_bytes_per_step": int(dec_bytes / dec_steps)}
    dense = None
    if nar_prof:
        T = int(phns[0].shape[0]) + int(proms[0].shape[0]) + R + 6
        fl = 7 * dense_pass_flops(cfg, [T] * B, B * R, cfg.audio_tokens)
        nar_ms = sum(a.elapsed_time(b) for a, b in nar_prof) / len(nar_prof)
        tf = fl / (nar_ms * 1e-3) / 1e12
        dense = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "what": f"7 NAR levels, {B} x {T} tokens each, host time included",
                 "ms": round(nar_ms, 2)}
        mf = ROOT / "profiles" / "r1_pmc_mfma_dense_gemms.json"
        if mf.exists():     # MFMA pipe utilisation of the four layer GEMMs from the PMC pass committed under profiles/
            dense["mfma_busy_frac_pmc"] = {k.split("(")[0].replace("void ", ""): round(v["mfma_busy_frac"], 3)
                                           for k, v in json.loads(mf.read_text()).items()}
            dense["mfma_busy_source"] = f"profiles/{mf.name} (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 256 CUs x 4 SIMDs), tools/micro_gemm)"
    if rank == 0:
        line = {
            "metric": "RVQ frames/sec (all levels)", "value": round(frames / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: AR KV-cache decode + 7 NAR levels, repo-default ar+nar llama (d=1024, L=12), "
                                   f"batch {B}/GPU, 80 phonemes, 225 prompt frames, {R} output frames, greedy",
                       "global_batch": world * B, "frames_per_utterance": R, "parallelism": f"utterance-sharded x{world}"},
            "rtf": round((elapsed / args.steps) / (R / FRAME_RATE), 5),
            "roofline": roofline,
            "dense": dense,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, phns, proms, langs, R)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
