#!/usr/bin/env python3
"""Headline benchmark: image-text pairs/s of the Mammo-CLIP contrastive pre-training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Either the driver launches the ranks (``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N``: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or a
plain ``python bench.py --gpus N`` re-executes ITSELF under torch.distributed.run on 127.0.0.1 with a free port; rank 0
prints the one JSON line either way.

One "step" = one full training step of the hot path on one batch of synthetic data already resident in HBM:
2 image views + 2 reports per pair through EfficientNet + BioClinicalBERT, projection, fused RCCL all-gather,
symmetric InfoNCE (breast_clip loss), backward, gradient all-reduce, AdamW update, LR-schedule step.
Default workload = the configuration BASELINE.json's metric is quoted on (configs[3], "cfg4"): EfficientNet-B5 +
BioClinicalBERT, GLOBAL batch 1024, 1520x912 images, 256-token reports, bf16 compute.  Each GPU takes 1024 / N pairs
per step (strong scaling) in micro-batches of 32 pairs: the contrastive loss runs over all 1024 pairs of the step, the
micro-batching costs k - keep extra forwards per step of k micro-batches (engine.Trainer.step(batch, micro_batches=k):
the last `keep` micro-batches are forwarded once, graph kept; 2 full graphs = 232 GB, or all 4 graphs of the N = 8 load
with the MBConv recompute mode 3).  At N = 1 a step is 32 micro-batches (about 12.4 s).  "--workload cfg3" is BASELINE configs[2], 32 pairs per GPU in one pass (weak scaling).

The JSON line carries, besides the contract fields:
  roofline     -- the dominant kernel of the step (largest share of GPU time in the rocprofv3 kernel stats under
                  profiles/).  The candidates at the top of that table -- bnact_bwd_k<true>, the HBM-bound
                  BatchNorm(+SiLU) backward apply pass, and the two plain NT MFMA tile kernels (128x128 and 256x256) --
                  are each bracketed by HIP events on the launch stream at every launch inside the timed steps; the one
                  with the most GPU time in the run is "roofline", then "roofline_runner_up", "roofline_third";
                  achieved = algorithmic bytes (flops) per launch / average launch duration
  cpu_baseline -- the CPU oracle (oracle/, torch-fp32 restatement of the reference) timed on this box's host cores
                  on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time
import types

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (encoder name, arch, per-GPU pairs, H, W, T)
    "cfg1": ("tf_efficientnetv2-detect", "efficientnet-b2", 4, 224, 224, 64),
    "cfg2": ("tf_efficientnetv2-detect", "efficientnet-b2", 64, 912, 912, 256),
    "cfg3": ("tf_efficientnet_b5_ns-detect", "efficientnet-b5", 32, 1520, 912, 256),
    # BASELINE config #4: GLOBAL batch 1024 (strong scaling: 1024 / N pairs per GPU, in micro-batches of 32)
    "cfg4": ("tf_efficientnet_b5_ns-detect", "efficientnet-b5", 1024, 1520, 912, 256),
    # BASELINE config #5: cfg4's model with fp8 (OCP e4m3) operands for the late-stage 1x1 convolutions, GLOBAL batch 2048
    "cfg5": ("tf_efficientnet_b5_ns-detect", "efficientnet-b5", 2048, 1520, 912, 256),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_PEAK_TFS = 2500.0         # dense bf16 MFMA peak

# The top of the rocprofv3 kernel statistics of the default command (profiles/r02_cfg4_kernel_stats.csv):
#  * bnact_bwd_k<true>, the BatchNorm(+SiLU) backward "apply" pass  dx = A*dz + B*x + C  (reads the saved conv output x
#    and the upstream gradient g, writes dx): HBM-bound, 3 x rows x channels x 2 B algorithmic bytes per launch;
#  * the plain NT direct-to-LDS MFMA tile kernels (forward / data-gradient 1x1 convolutions of the late stages and the BERT
#    linears): gemm_kernel<128,128,64,2,2,0,0,false,true> and, where its 256 x 256 tiles fill the 256 CUs well,
#    g256::gemm256_kernel (mc_gemm_tile_config): MFMA-bound, 2 M N K flop per launch.
# All three are timed live and ranked by GPU time in the run: "roofline", "roofline_runner_up", "roofline_third".
GEMM_KERNEL = ("gemm_kernel<128,128,64,2,2,0,0,false,true> (plain NT direct-to-LDS MFMA tiles 128x128x64: late-stage 1x1 convs "
               "fwd/dgrad, BERT linears -- the problems whose 256x256 tiling would not fill the 256 CUs)")
GEMM256_KERNEL = ("g256::gemm256_kernel (plain NT direct-to-LDS MFMA tiles 256x256x64, 8 waves, 160 KB LDS: late-stage expand / "
                  "projection convs, BERT FFN1, where mc_gemm_tile_config picks it)")
ROOFLINE_OP = "mc_bnact_bwd_apply"
ROOFLINE_KERNEL = "bnact_bwd_k<true> (BatchNorm+SiLU backward apply pass)"
STREAM_OPS = (ROOFLINE_OP, "mc_gemm_bf16")


def model_cfg(enc_name, fp8=False, recompute=0):
    return {"name": "clip_custom", "temperature": 0.07,
            "image_encoder": {"source": "cnn", "name": enc_name, "pretrained": True, "model_type": "cnn", "fp8": fp8,
                              "recompute": recompute},
            "text_encoder": {"source": "huggingface", "name": "emilyalsentzer/Bio_ClinicalBERT", "pretrained": False,
                             "gradient_checkpointing": False, "pooling": "eos", "cache_dir": "", "trust_remote_code": True},
            "projection_head": {"name": "linear", "dropout": 0.1, "proj_dim": 512}}


LOSS_CFG = {"breast_clip": dict(label_smoothing=0.0, i2i_weight=1.0, t2t_weight=0.5, loss_ratio=1.0)}


def synth_batch_gpu(b, H, W, T, device, seed):
    """Synthetic batch resident in HBM: images ~ N(0,1) in the trainer's [b,3,H,W] permuted-NHWC view
    (trainer_ddp.py:288-291), full-length token rows ([CLS] ... [SEP]) as in SURVEY.md section 8d throughput runs."""
    g = torch.Generator(device=device).manual_seed(seed)
    batch = {}
    for k in ("images", "image_views"):
        batch[k] = torch.randn((b, H, W, 3), generator=g, device=device).permute(0, 3, 1, 2)
    for k in ("text_tokens", "text_tokens2"):
        ids = torch.randint(1000, 28996, (b, T), generator=g, device=device)
        ids[:, 0], ids[:, -1] = 101, 102
        batch[k] = {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": torch.ones_like(ids)}
    return batch


def cpu_baseline(arch_name, H, W, T, budget_s=30.0):
    """CPU oracle (port of the reference path), train-mode fwd + bwd on ONE pair-group sample, host cores of this box."""
    from oracle import arch as oarch, bert as obert, clip as oclip, loss as oloss, weights as ow
    cores = os.cpu_count() or 1
    threads = min(cores, 64)     # measured on the MI355X host: 64 threads 46 s vs 128 threads 79 s for the same sample
    torch.set_num_threads(threads)
    arch = oarch.build_arch(arch_name)
    cfg = obert.BertShape()
    b = 1                      # one pair = 2 images + 2 reports (bounded sample: ~20-30 s of CPU work)
    sd = ow.synth_state_dict(ow.clip_shapes(arch, cfg), seed=10)
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
    batch = ow.synth_batch(b, H, W, T, seed=10, full_length=True)
    t0 = time.time()
    out = oclip.forward(sdg, batch, arch, cfg, train=True, new_buffers={})
    loss = oloss.breast_clip_rank(out["image_embeddings"], out["text_embeddings"], out["text_embeddings2"],
                                  out["image_view_embeddings"], out["logit_scale"], 0, b)["loss"]
    loss.backward()
    dt = time.time() - t0
    return {"value": round(b / dt, 4), "unit": "image-text pairs/s", "cores": threads, "kind": "port",
            "sample": f"{b} pairs ({2*b} images {H}x{W} + {2*b} reports T={T}), one train-mode fwd+bwd of the oracle "
                      f"(no optimizer step), {dt:.1f} s on {threads} threads of {cores} host cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg4", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU (0 = workload default)")
    ap.add_argument("--micro-batches", type=int, default=1,
                    help="cut the per-GPU batch into k micro-batches (one extra forward per step; for batches beyond one pass)")
    ap.add_argument("--keep-graphs", type=int, default=0,
                    help="micro-batched step: micro-batches forwarded once with their graph kept (activation memory x this); "
                         "0 = 2 for the global-batch workloads at <= 512 pairs per GPU (2 x 32 pairs = 232 GB of the 288 GB), 1 otherwise")
    ap.add_argument("--recompute", type=int, default=-1, choices=(-1, 0, 1, 2, 3),
                    help="MBConv activation recompute mode (EfficientNet.set_recompute); -1 = chosen with --keep-graphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="fp8 (e4m3) operands for the late-stage 1x1 convolutions (config #5 arithmetic on any workload)")
    ap.add_argument("--op-profile", action="store_true", help="print a per-entry-point HIP-event breakdown (rank 0)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves (one process per GPU, RCCL) and relay rank 0's line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import mammo_clip_amd  # noqa: F401
    from mammo_clip_amd import lib as L
    from mammo_clip_amd import engine
    from mammo_clip_amd.breastclip import util
    from mammo_clip_amd.breastclip.loss import build_loss
    from mammo_clip_amd.breastclip.model import build_model
    from mammo_clip_amd.breastclip.optimizer import build_optimizer
    from mammo_clip_amd.breastclip.scheduler import LinearWarmupCosineAnnealingLR

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the HIP path is the only path (no CPU fallback)")
    L.load()
    rank, local, world, device = engine.init_distributed()
    assert world == args.gpus or (world == 1 and args.gpus == 1), f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    enc_name, arch_name, b, H, W, T = WORKLOADS[args.workload]
    if args.batch:
        b = args.batch
    strong = args.workload in ("cfg4", "cfg5") and not args.batch
    if strong:
        assert b % world == 0
        b = b // world
        args.micro_batches = max(1, b // 32)
    fp8 = args.workload == "cfg5" or args.fp8
    # Kept graphs against activation memory (measured on one MI355X at the per-GPU load, scripts/recompute_sweep.sh):
    #   128 pairs/GPU (N = 8): recompute mode 3 makes a 32-pair graph small enough to keep all four (216 GB peak, no
    #     re-forward at all): 1293-1305 ms/step against 1384 ms with two full graphs kept (227 GB) -- mode 1 with four kept
    #     is 1271 ms but peaks at 295 GB, too close to the 309 GB of the device;
    #   256 pairs/GPU (N = 4): mode 2 (28 GB per graph), all eight kept: 2758 ms/step, 219 GB, against 2894 ms with two
    #     full graphs (231 GB);
    #   512 pairs/GPU (N = 2): two full graphs (mode 2 with five kept graphs: 6394 ms -- every backward pays the recompute,
    #     eleven micro-batches are still forwarded twice);
    #   1024 pairs/GPU (N = 1): the fp32 input batch itself occupies 34 GB: one kept graph.
    if args.keep_graphs <= 0 and args.recompute < 0 and strong and b <= 256:
        args.keep_graphs, args.recompute = args.micro_batches, (3 if b <= 128 else 2)
    if args.keep_graphs <= 0:
        args.keep_graphs = 2 if (strong and b <= 512) else 1
    util.GlobalEnv.reset()
    torch.manual_seed(10)
    if args.recompute < 0:
        args.recompute = 0
    model = build_model(model_cfg(enc_name, fp8, args.recompute), LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(device)
    loss_func = build_loss(LOSS_CFG)
    opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 5e-5, "weight_decay": 1e-4}})
    sched = LinearWarmupCosineAnnealingLR(opt, total_steps=10000, warmup_steps=100)
    trainer = engine.Trainer(model, loss_func, opt, sched, device, keep_graphs=args.keep_graphs)
    batch = synth_batch_gpu(b, H, W, T, device, seed=10 + rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        ld = trainer.step(batch, args.micro_batches)
    sync()
    timer = L.OpTimer(only=None if args.op_profile else STREAM_OPS, kind_contains={"mc_gemm_bf16": "|glnt"})
    L.TIMER = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ld = trainer.step(batch, args.micro_batches)
    sync()
    dt = time.perf_counter() - t0
    L.TIMER = None
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    loss_val = float(ld["total"])
    summ = timer.summary()
    peak_gb = torch.cuda.max_memory_allocated() / 1e9

    if rank == 0:
        ms = dt / args.steps * 1e3
        pairs = b * world * args.steps / dt
        sc, st, sb, _ = summ.get(ROOFLINE_OP, (0, 0.0, 0, 0))
        ach = sb / (st * 1e-3) / 1e9 if st > 0 else 0.0
        traffic = gtraffic = g256traffic = None
        tpath = os.path.join(ROOT, "profiles", "r02_roofline_traffic.json")
        if not os.path.exists(tpath):
            tpath = os.path.join(ROOT, "profiles", "r01_roofline_traffic.json")
        if os.path.exists(tpath) and args.workload in ("cfg3", "cfg4") and not args.batch:   # same 32-pair kernel launches
            tj = json.load(open(tpath))                                     # PMC passes (rocprofv3 --pmc), same workload
            traffic, gtraffic = tj.get("hbm_bytes_per_launch"), tj.get("gemm_nt_hbm_bytes_per_launch")
            g256traffic = tj.get("gemm256_hbm_bytes_per_launch")
        # the two kernels that share the top of the rocprofv3 kernel statistics (profiles/r01_cfg4_kernel_stats.csv):
        # the HBM-bound BN-backward apply pass and the MFMA NT GEMM instance; the one with more GPU time in THIS run
        # is reported as "roofline", the other as "roofline_runner_up"
        timing = "HIP events on the launch stream around every launch inside the timed steps"
        r_hbm = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": ROOFLINE_KERNEL, "launches": sc,
                 "avg_launch_us": round(st / max(sc, 1) * 1e3, 1), "algorithmic_bytes_per_launch": int(sb / max(sc, 1)),
                 "gpu_ms_in_timed_steps": round(st, 1), "timing": timing}

        def mfma_entry(sel, name, traffic_bytes):
            gc = sum(v[0] for k, v in summ.items() if sel(k))
            gt = sum(v[1] for k, v in summ.items() if sel(k))
            gb = sum(v[2] for k, v in summ.items() if sel(k))
            gf = sum(v[3] for k, v in summ.items() if sel(k))
            gach = gf / (gt * 1e-3) / 1e12 if gt > 0 else 0.0
            return {"bound": "mfma", "achieved": round(gach, 1), "peak": MFMA_PEAK_TFS, "unit": "TFLOP/s",
                    "frac": round(gach / MFMA_PEAK_TFS, 4), "traffic": traffic_bytes, "kernel": name, "launches": gc,
                    "avg_launch_us": round(gt / max(gc, 1) * 1e3, 1), "algorithmic_flops_per_launch": int(gf / max(gc, 1)),
                    "algorithmic_bytes_per_launch": int(gb / max(gc, 1)),
                    "gpu_ms_in_timed_steps": round(gt, 1), "timing": timing}
        # the plain NT MFMA tile kernels are two different kernels (rocprofv3 lists them separately): timed separately
        r_g128 = mfma_entry(lambda k: k.startswith("mc_gemm_bf16") and k.endswith("|glnt"), GEMM_KERNEL, gtraffic)
        r_g256 = mfma_entry(lambda k: k.startswith("mc_gemm_bf16") and k.endswith("|glnt256"), GEMM256_KERNEL, g256traffic)
        ranked = sorted([r_hbm, r_g128, r_g256], key=lambda r: -r["gpu_ms_in_timed_steps"])
        first, second, third = ranked
        res = {
            "metric": "image-text pairs/s (whole node), EN-B5+BioClinicalBERT contrastive pre-training step",
            "value": round(pairs, 3), "unit": "image-text pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "bf16" + (" + fp8 e4m3 pointwise-conv operands" if fp8 else ""), "data": "synthetic",
            "config": {"workload": f"{args.workload}: {arch_name} + BioClinicalBERT(BERT-base), {b} pairs/GPU "
                                   f"(2 views + 2 reports each), {H}x{W} images, {T}-token reports, breast_clip loss, "
                                   f"AdamW; fwd+loss+bwd+optimizer; dropout/drop-connect on",
                       "global_batch": b * world, "parallelism": f"dp{world}" + (f" x {args.micro_batches} micro-batches" if args.micro_batches > 1 else ""),
                       "loss": round(loss_val, 5), "peak_hbm_gb": round(peak_gb, 1),
                       "keep_graphs": args.keep_graphs, "recompute": args.recompute},
            "roofline": first, "roofline_runner_up": second, "roofline_third": third,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(arch_name, H, W, T)
            except Exception as e:                # pragma: no cover
                res["cpu_baseline"] = {"value": None, "error": repr(e)}
        if args.op_profile:
            rows = sorted(summ.items(), key=lambda kv: -kv[1][1])
            tot = sum(v[1] for _, v in rows)
            print(f"# per-entry-point HIP-event time over {args.steps} steps (sum {tot:.1f} ms, wall {dt*1e3:.1f} ms)", file=sys.stderr)
            for k, (cnt, t_ms, by, fl) in rows:
                print(f"# {k:24s} n={cnt:6d} {t_ms:10.2f} ms {100*t_ms/tot:5.1f}%  {by/(t_ms*1e6+1e-9):8.1f} GB/s "
                      f"{fl/(t_ms*1e9+1e-9):8.1f} TFLOP/s", file=sys.stderr)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
