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
with the MBConv recompute mode 3).  At N = 1 a step is 32 micro-batches (about 8.5 s in round 5).  "--workload cfg3" is BASELINE configs[2], 32 pairs per GPU in one pass (weak scaling).

The JSON line carries, besides the contract fields:
  roofline     -- the dominant kernel class of THIS run.  The last warm-up step is run with HIP events around every C-ABI
                  launch (all ~70 entry points, keyed by entry point + shape class, e.g. "mc_dwconv_fwd:k5s1"); the three
                  classes with the most GPU time are then bracketed by HIP events on the launch stream at every launch
                  inside the timed steps and reported as "roofline", "roofline_runner_up", "roofline_third"
                  (achieved = algorithmic bytes or flops per launch / average launch duration; bound = hbm or mfma by
                  the class's flop / byte ratio against the 2.5 PFLOP/s : 8 TB/s ridge; traffic = PMC bytes per launch
                  from profiles/rNN_roofline_traffic.json (cfg3 launch mix; `traffic_source` says so) where that class was measured, else null)
  n8_load      -- (default workload, N = 1 only) ms/step of the step ONE GPU runs at N = 8: 128 pairs as 4 micro-batches,
                  recompute mode 3, all four graphs kept, no communication -- the like-for-like single-GPU time a
                  1 -> 8 scaling curve should be read against (the N = 1 point itself carries 31 re-forwards)
  cpu_baseline -- the CPU oracle (oracle/, torch-fp32 restatement of the reference) timed on this box's host cores
                  on a bounded sample of the same workload (rank 0, N = 1 only)
  parity       -- (N = 1) |loss - fp32 oracle| of THIS run's kernel-library build at the workload's per-sample shape, eval and
                  train mode, 2 pairs, identical weights / inputs (oracle/probe.py: the oracle is the checker, run on the GPU
                  after the timed region)
  parity_build -- (N = 1, bf16 runs) the same workload on the f16 storage build (the reference's AMP dtype; the build that
                  meets north_star's 1e-3 in train mode), timed for one step in a child process of this
                  run, with ITS parity object: throughput and tolerance are evidenced by the same command
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

RIDGE = MFMA_PEAK_TFS * 1e12 / (HBM_PEAK_GBS * 1e9)      # flop per byte above which a kernel class is MFMA-bound

# entry point (+ shape class) -> the kernel behind it, for the report
KERNEL_NAMES = {
    "mc_bnact_bwd_apply": "bnact_bwd_k<true, ACT, MA> (BatchNorm(+SiLU) backward apply pass: dx = A*dz + B*x + C; all three instantiations)",
    "mc_bnact_bwd_reduce": "bnact_bwd_k<false, ACT, MA> (BatchNorm backward reduce pass)",
    "mc_bnact_se_sums": "bnact_se_sums_k (SE-gate gradient + BatchNorm1 backward sums, one pass over (d, dA1))",
    "mc_bnact_pool": "bnact_img_reduce_k (BN+SiLU + squeeze-excite average pool)",
    "mc_bnact_apply": "bnact_apply_k (BatchNorm2 + drop-connect + residual)",
    "mc_dwconv_fwd": "dwconv_march_fwd_kernel + lane::dwconv_lane_fwd_kernel<K,S,NCOL,0|1> (depthwise conv forward / stride-1 data gradient: marching LDS kernels for 3x3, lane = column kernels with the taps in SGPRs for 5x5)",
    "mc_dwconv_bwd_weight": "dwconv_march_bww_kernel + lane::dwconv_lane_fwd_kernel<K,S,NCOL,2> (depthwise conv weight gradient)",
    "mc_dwconv_bwd_data": "dwconv_march_bwd_s2_kernel (depthwise conv stride-2 data gradient, with the BatchNorm0+SiLU backward epilogue)",
    "mc_dwconv_bwd_fused": "lane::dwconv_lane_fwd_kernel<3,1,1,3,G> (round 5: whole stride-1 3x3 depthwise backward in one launch -- data gradient + BatchNorm0 epilogue + weight gradient from one staging of (dd, e))",
    "mc_mbconv_xdw_fwd": "lane::dwconv_lane_fwd_kernel<K,S,NCOL,4,G,KC> (round 6: expand 1x1 conv (MFMA, weight slice in LDS) + BatchNorm0 + swish inside the staging of the depthwise forward -- the expanded tensor never reaches HBM; algorithmic bytes = block input + depthwise output)",
    "mc_xbwd_rows_bf16": "xbwd_rows_kernel (round 5: expand-conv backward, weight + data gradient from one pass over the upstream gradient, LDS transpose-reads + register-resident weight slice)",
    "mc_gemm_rows_bf16": "gemm_rows_kernel (row-streaming 1x1 conv forward / data gradient, weights resident in LDS)",
    "mc_wgrad_rows_bf16": "wgrad_rows_kernel (row-streaming 1x1 conv weight gradient, LDS transpose-reads)",
    "mc_gemm_bf16|glnt256": "g8::gemm8p_kernel (plain NT 256x256x64 MFMA tiles, 8 waves, 4 phases per K tile, LDS-direct DMA)",
    "mc_gemm_bf16|glnt": "gemm_kernel<128,128,64,2,2,0,0,false,true> (plain NT 128x128x64 MFMA tiles)",
    "mc_gemm_bf16|tn256": "g8t::gemm256_tn_kernel (TN weight-gradient 256x256x64 MFMA tiles, LDS transpose-reads, split-K)",
    "mc_adamw_step": "adamw_multi_k (multi-tensor AdamW)",
}


def kernel_name(key):
    """entry point ':' class tags -> kernel description; for the plain-operand GEMM entry point the class tag names the
    kernel, and forward / data-gradient calls of one kernel are one class (class_key)"""
    side = {"mfma": " -- launches whose shape is MFMA-bound (flop/byte >= 312)", "hbm": " -- launches whose shape is HBM-bound (flop/byte < 312)"}
    for k, v in KERNEL_NAMES.items():
        ep, _, tag = k.partition("|")
        base = key[:-5] if key.endswith("|mfma") else (key[:-4] if key.endswith("|hbm") else key)
        if key.split(":")[0] == ep and (not tag or base.endswith("|" + tag)):
            return v + side.get(key.rsplit("|", 1)[-1], "") + (" [" + key + "]" if ":" in key else "")
    return key


def class_key(key):
    """merge the timer's record keys into kernel classes = one class per kernel TEMPLATE (rocprof kernel-name family):
    'mc_gemm_bf16:fwd|glnt256' and ':dgrad|glnt256' -> one class per tile kernel and roofline side; every depthwise
    forward / stride-1 data-gradient launch (all kernel sizes and strides) -> 'mc_dwconv_fwd' (dwconv_march_fwd_kernel);
    every row-streaming 1x1 convolution launch (forward, data gradient, the epilogue forms) -> 'mc_gemm_rows_bf16'"""
    ep, _, kind = key.partition(":")
    for tag in ("glnt256", "glnt", "tn256"):
        for side in ("mfma", "hbm"):
            if ep == "mc_gemm_bf16" and kind.endswith(f"|{tag}|{side}"):
                return f"{ep}:|{tag}|{side}"
    if ep in ("mc_dwconv_fwd", "mc_gemm_rows_bf16", "mc_dwconv_bwd_weight", "mc_wgrad_rows_bf16", "mc_dwconv_bwd_fused", "mc_xbwd_rows_bf16",
              "mc_mbconv_xdw_fwd"):
        return ep
    return key


def merged(summ):
    out = {}
    for k, (cnt, t_ms, by, fl) in summ.items():
        c = out.get(class_key(k), (0, 0.0, 0, 0))
        out[class_key(k)] = (c[0] + cnt, c[1] + t_ms, c[2] + by, c[3] + fl)
    return out


def model_cfg(enc_name, fp8=False, recompute=0):
    return {"name": "clip_custom", "temperature": 0.07,
            "image_encoder": {"source": "cnn", "name": enc_name, "pretrained": True, "model_type": "cnn", "fp8": fp8,
                              "recompute": recompute},
            "text_encoder": {"source": "huggingface", "name": "emilyalsentzer/Bio_ClinicalBERT", "pretrained": False,
                             "gradient_checkpointing": False, "pooling": "eos", "cache_dir": "", "trust_remote_code": True},
            "projection_head": {"name": "linear", "dropout": 0.1, "proj_dim": 512}}


LOSS_CFG = {"breast_clip": dict(label_smoothing=0.0, i2i_weight=1.0, t2t_weight=0.5, loss_ratio=1.0)}
# SURVEY.md section 8d's secondary metric: the single-view loss (1 image + 1 report per pair through the encoders, half the
# encoder work per pair) [ref: loss/breast_clip_contrastive.py:28-59]; `--loss breast_clip_contrastive`
LOSS_CFG_SINGLE = {"breast_clip_contrastive": dict(label_smoothing=0.0, loss_ratio=1.0)}


def synth_batch_gpu(b, H, W, T, device, seed, views=2):
    """Synthetic batch resident in HBM: images ~ N(0,1) in the trainer's [b,3,H,W] permuted-NHWC view
    (trainer_ddp.py:288-291), full-length token rows ([CLS] ... [SEP]) as in SURVEY.md section 8d throughput runs."""
    g = torch.Generator(device=device).manual_seed(seed)
    batch = {}
    for k in ("images", "image_views")[:views]:
        batch[k] = torch.randn((b, H, W, 3), generator=g, device=device).permute(0, 3, 1, 2)
    for k in ("text_tokens", "text_tokens2")[:views]:
        ids = torch.randint(1000, 28996, (b, T), generator=g, device=device)
        ids[:, 0], ids[:, -1] = 101, 102
        batch[k] = {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": torch.ones_like(ids)}
    return batch


def cpu_baseline(arch_name, H, W, T, budget_s=80.0):
    """CPU oracle (port of the reference path) on the host cores of this box: ONE full training step (train-mode forward
    + loss + backward + AdamW) of a bounded sample -- 1 pair = 2 images + 2 reports at the workload's size.  Protocol
    (bounded): one untimed warm-up step, then the median of up to 3 timed steps within ``budget_s``
    (at least one timed step)."""
    from oracle import arch as oarch, bert as obert, clip as oclip, loss as oloss, weights as ow
    cores = os.cpu_count() or 1
    threads = min(cores, 64)     # measured on the MI355X host: 64 threads 46 s vs 128 threads 79 s for the same sample
    torch.set_num_threads(threads)
    arch = oarch.build_arch(arch_name)
    cfg = obert.BertShape()
    b = 1                      # one pair = 2 images + 2 reports (bounded sample: ~15-20 s of CPU work per step)
    sd = ow.synth_state_dict(ow.clip_shapes(arch, cfg), seed=10)
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
    params = [v for v in sdg.values() if torch.is_tensor(v) and v.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4)
    batch = ow.synth_batch(b, H, W, T, seed=10, full_length=True)

    def step():
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        out = oclip.forward(sdg, batch, arch, cfg, train=True, new_buffers={})
        loss = oloss.breast_clip_rank(out["image_embeddings"], out["text_embeddings"], out["text_embeddings2"],
                                      out["image_view_embeddings"], out["logit_scale"], 0, b)["loss"]
        loss.backward()
        opt.step()
        return time.time() - t0
    t_start = time.time()
    warm = step()                                                      # untimed: thread pool, primitive caches, allocator
    times = [step()]
    while len(times) < 3 and (time.time() - t_start) + 1.1 * times[-1] <= budget_s:
        times.append(step())
    dt = sorted(times)[len(times) // 2]
    return {"value": round(b / dt, 4), "unit": "image-text pairs/s", "cores": threads, "kind": "port",
            "sample": f"{b} pair ({2*b} images {H}x{W} + {2*b} reports T={T}) per step, full train step of the oracle "
                      f"(fwd + loss + bwd + AdamW); 1 untimed warm-up ({warm:.1f} s), median of {len(times)} timed steps "
                      f"({', '.join('%.1f' % t for t in times)} s) on {threads} threads of {cores} host cores"}


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
    ap.add_argument("--recompute", type=int, default=-1, choices=(-1, 0, 1, 2, 3, 4),
                    help="MBConv activation recompute mode (EfficientNet.set_recompute); -1 = chosen with --keep-graphs")
    ap.add_argument("--keep-kept", type=int, default=7, help="kept mode-2 graphs of the N = 1 / 2 global-batch runs")
    ap.add_argument("--keep-mode", type=int, default=2, choices=(1, 2, 3, 4), help="recompute mode of those kept graphs")
    ap.add_argument("--as-gpus", type=int, default=0, help="with --gpus 1: run ONE rank's share of the N-GPU strong-scaling run "
                    "(1024 / N pairs, that run's micro-batch / kept-graph policy, no collectives): the per-GPU load of the N-GPU point")
    ap.add_argument("--loss", default="breast_clip", choices=("breast_clip", "breast_clip_contrastive"),
                    help="breast_clip = the configured multi-view loss (2 views + 2 reports per pair: the headline metric); "
                         "breast_clip_contrastive = the single-view loss (1 image + 1 report per pair: SURVEY 8d's secondary metric)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity / parity_build objects (loss deviation from the fp32 oracle)")
    ap.add_argument("--parity-child", action="store_true", help=argparse.SUPPRESS)     # the f16-build leg spawned by the default run
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-n8-load", action="store_true", help="skip the n8_load block of the default N = 1 run")
    ap.add_argument("--fp8", action="store_true", help="fp8 (e4m3) operands for the late-stage 1x1 convolutions (config #5 arithmetic on any workload)")
    ap.add_argument("--op-profile", action="store_true", help="print a per-entry-point HIP-event breakdown (rank 0)")
    ap.add_argument("--streams", type=int, default=None, choices=(0, 1, 2, 3, 7),
                    help="encoder chains of one forward on separate HIP streams (model/clip.py MC_STREAMS: bit 0 the text encoder, "
                         "bit 1 the second image view; default: MC_STREAMS or 3)")
    ap.add_argument("--roofline-in-timed-region", action="store_true",
                    help="with streams: no extra one-stream steps -- the roofline objects quote the (shared) launch durations of the timed steps")
    ap.add_argument("--storage", default=None, choices=("bf16", "f16"),
                    help="16-bit storage / MFMA operand build of the kernel library (default: MC_STORAGE or bf16 -- BASELINE's dtype; "
                         "f16 = the reference's AMP dtype with a dynamic loss scale, the parity configuration of DESIGN.md (c))")
    args = ap.parse_args()
    if args.storage:
        os.environ["MC_STORAGE"] = args.storage        # read when the package is imported (below; the spawned ranks inherit it)

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
    # evidence that the collectives ran on RCCL with N ranks: RCCL's own INIT log goes to a per-process file
    # (NCCL_DEBUG_FILE), rank 0 reads its communicator line ("... nranks N ...") back after the first collective
    rccl_log = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("MC_DIST_BACKEND", "nccl") == "nccl":
        import tempfile
        rccl_log = os.path.join(tempfile.gettempdir(), f"mc_rccl_{os.getpid()}.log")
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT")
        os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log)
        rccl_log = os.environ["NCCL_DEBUG_FILE"]
    rank, local, world, device = engine.init_distributed()
    assert world == args.gpus or (world == 1 and args.gpus == 1), f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    enc_name, arch_name, b, H, W, T = WORKLOADS[args.workload]
    if args.batch:
        b = args.batch
    strong = args.workload in ("cfg4", "cfg5") and not args.batch
    if strong:
        share = args.as_gpus if (args.as_gpus and world == 1) else world
        assert b % share == 0
        b = b // share
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
    keep_recompute = None
    if args.keep_graphs <= 0 and args.recompute < 0 and strong and b <= 256:
        args.keep_graphs, args.recompute = args.micro_batches, (3 if b <= 128 else 2)
    elif args.keep_graphs <= 0 and args.recompute < 0 and strong:
        # N = 1 / 2 (round 3): SEVEN kept graphs in recompute mode 2 (25 GB each; their backward pays the rebuild of e and d,
        # ~32 ms, and saves a ~90 ms forward), the re-forwarded micro-batches stay in mode 0 (Trainer.keep_recompute);
        # 225 GB allocated / 239 GB reserved at N = 1 (6 kept: 199 GB, 0.9 % slower; 8 kept: 0.1 % faster at 252 / 263 GB;
        # 9 kept: 278 GB -- too close to the device's 288 GiB)
        args.keep_graphs, keep_recompute = args.keep_kept, args.keep_mode
    if args.keep_graphs <= 0:
        args.keep_graphs = 2 if (strong and b <= 512) else 1
    util.GlobalEnv.reset()
    torch.manual_seed(10)
    if args.recompute < 0:
        args.recompute = 0
    single = args.loss == "breast_clip_contrastive"
    loss_cfg = LOSS_CFG_SINGLE if single else LOSS_CFG
    model = build_model(model_cfg(enc_name, fp8, args.recompute), loss_cfg, types.SimpleNamespace(vocab_size=28996)).to(device)
    loss_func = build_loss(loss_cfg)
    opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 5e-5, "weight_decay": 1e-4}})
    sched = LinearWarmupCosineAnnealingLR(opt, total_steps=10000, warmup_steps=100)
    trainer = engine.Trainer(model, loss_func, opt, sched, device, keep_graphs=args.keep_graphs, keep_recompute=keep_recompute)
    from mammo_clip_amd.breastclip.model import clip as clipmod
    if args.streams is not None:
        clipmod._STREAMS = args.streams
    streams = clipmod._STREAMS
    batch = synth_batch_gpu(b, H, W, T, device, seed=10 + rank, views=1 if single else 2)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    dist_info = None
    if world > 1:
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)                      # SUM of one per rank over the job's backend: = the ranks that took part
        dist_info = {"backend": dist.get_backend(), "allreduce_of_ones": int(ones.item()), "rccl_ranks": None}
        if rccl_log and rank == 0:
            import glob
            import re
            for fn in glob.glob(rccl_log.replace("%h", "*").replace("%p", "*")):
                try:
                    m = re.findall(r"nranks (\d+)", open(fn, errors="replace").read())
                except OSError:
                    m = []
                if m:
                    dist_info["rccl_ranks"] = max(int(v) for v in m)

    # warm-up; the LAST warm-up step (an extra step when --warmup 0) is the survey step: HIP events around every C-ABI
    # launch -> GPU time per kernel class -> the three classes the timed steps bracket
    for _ in range(max(args.warmup - 1, 0)):
        ld = trainer.step(batch, args.micro_batches)
    sync()
    survey = L.OpTimer()
    L.TIMER = survey
    ld = trainer.step(batch, args.micro_batches)
    sync()
    L.TIMER = None
    raw = survey.summary()
    ssum = merged(raw)
    del survey
    top = [k for k, v in sorted(ssum.items(), key=lambda kv: -kv[1][1]) if v[2] > 0 or v[3] > 0][:3]
    timer = L.OpTimer(keys=None if args.op_profile else [k for k in raw if class_key(k) in top])
    L.TIMER = timer
    torch.cuda.reset_peak_memory_stats()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step spread (no sync inside the region)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        ld = trainer.step(batch, args.micro_batches)
        marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    L.TIMER = None
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    loss_val = float(ld["total"])
    raw_timed = timer.summary()
    summ = merged(raw_timed)
    peak_gb = torch.cuda.max_memory_allocated() / 1e9
    peak_res_gb = torch.cuda.max_memory_reserved() / 1e9

    # Kernel rooflines need launch durations that belong to ONE kernel.  With the encoder chains on separate streams a
    # bracketed launch shares the GPU with launches of the other chains: the HIP-event time around it is no longer that
    # kernel's own time (two HBM-bound kernels side by side each see about half the bandwidth).  The roofline objects are
    # therefore taken from `xsteps` further steps of the same trainer on ONE stream, right after the timed region (one untimed
    # step first: the allocator pools of the side streams are released and the main pool regrows); the shared durations of the
    # timed region are reported beside them (`in_timed_region`).
    summ_x, xsteps = None, 0
    if streams and not args.roofline_in_timed_region:
        clipmod._STREAMS = 0
        trainer.optimizer.zero_grad(set_to_none=True)
        torch.cuda.empty_cache()
        ld = trainer.step(batch, args.micro_batches)
        sync()
        xsteps = 1 if args.micro_batches >= 8 else max(1, min(args.steps, 2))     # (a micro-batched step is thousands of launches per class)
        timer_x = L.OpTimer(keys=[k for k in raw if class_key(k) in top])
        L.TIMER = timer_x
        for _ in range(xsteps):
            ld = trainer.step(batch, args.micro_batches)
        sync()
        L.TIMER = None
        summ_x = merged(timer_x.summary())
        clipmod._STREAMS = streams

    n8 = None
    if world == 1 and args.workload == "cfg4" and strong and not args.no_n8_load and not single:
        # the step one GPU runs at N = 8 (128 pairs = 4 micro-batches, recompute mode 3, four kept graphs), no communication
        del batch, ld
        trainer.optimizer.zero_grad(set_to_none=True)
        torch.cuda.empty_cache()
        model.image_encoder.set_recompute(3)
        trainer.keep_graphs, trainer.keep_recompute = 4, None
        b8 = synth_batch_gpu(128, H, W, T, device, seed=99)
        for _ in range(2):                         # (two untimed steps: the allocator pools of all streams settle)
            trainer.step(b8, 4)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t8 = time.perf_counter()
        for _ in range(2):
            trainer.step(b8, 4)
        torch.cuda.synchronize()
        ms8 = (time.perf_counter() - t8) / 2 * 1e3
        n8 = {"ms_per_step": round(ms8, 1), "pairs_per_s_per_gpu": round(128 / ms8 * 1e3, 2), "pairs_per_gpu": 128,
              "micro_batches": 4, "keep_graphs": 4, "recompute": 3, "steps": 2, "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1),
              "note": "per-GPU load of the N = 8 run of this workload on ONE GPU, no collectives; 8 x this rate is the no-communication ceiling of the 8-GPU point"}
        del b8

    stat_tapes_on = bool(getattr(trainer, "stat_tapes", False)) and args.micro_batches > args.keep_graphs
    parity = parity_build = None
    if rank == 0 and world == 1 and not args.no_parity and args.workload in ("cfg2", "cfg3", "cfg4") and not single:
        # checker legs, after every timed region: free the benchmark's memory first
        try:
            del batch, ld
        except NameError:
            pass
        trainer.optimizer.zero_grad(set_to_none=True)
        del trainer, opt, sched
        torch.cuda.empty_cache()
        try:
            from oracle import probe
            model.image_encoder.set_recompute(0)
            r = probe.loss_deviation(model, loss_func, arch_name, H, W, T, pairs=2, device=str(device))
            parity = {"storage": os.environ.get("MC_STORAGE", "bf16").lower(), "tolerance": 1e-3,
                      "eval_dloss": round(r["eval_dloss"], 6), "train_dloss": round(r["train_dloss"], 6),
                      "eval_min_cos": round(r["eval_min_cos"], 6), "train_min_cos": round(r["train_min_cos"], 6),
                      "within_tolerance": {"eval": abs(r["eval_dloss"]) <= 1e-3, "train": abs(r["train_dloss"]) <= 1e-3},
                      "probe": f"{r['pairs']} pairs at {r['shape']}, identical synthetic weights and inputs, dropout / drop-connect off; "
                               "checker = oracle/ (torch fp32 on this GPU, pinned to the reference by tests/golden)"}
        except Exception as e:                    # pragma: no cover
            parity = {"error": repr(e)}
        if os.environ.get("MC_STORAGE", "bf16").lower() == "bf16" and not args.parity_child:
            # the f16 storage build on the same workload: one timed step in a child process (the storage type is fixed when the
            # kernel library is loaded), with its own parity object
            del model
            torch.cuda.empty_cache()
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--storage", "f16", "--parity-child", "--workload", args.workload,
                   "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-n8-load", "--roofline-in-timed-region"]
            if args.batch:
                cmd += ["--batch", str(args.batch)]
            try:
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, MC_STORAGE="f16"))
                cj = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])
                parity_build = {"storage": "f16", "pairs_per_s": cj["value"], "ms_per_step": cj["ms_per_step"], "steps": cj["steps"],
                                "peak_hbm_gb": cj["config"]["peak_hbm_gb"], "parity": cj.get("parity"),
                                "note": "same workload, same policies, libmammoclip_hip_f16.so (IEEE f16 storage / MFMA operands, dynamic loss scale = "
                                        "the reference's AMP configuration); child process of this run"}
            except Exception as e:                # pragma: no cover
                parity_build = {"error": repr(e)}

    if rank == 0:
        ms = dt / args.steps * 1e3
        pairs = b * world * args.steps / dt
        # per-workload PMC tables: rNN_roofline_traffic_cfg4.json (the default run's launch mix) when present, else the cfg3 table
        cands = ([os.path.join(ROOT, "profiles", f"r{r:02d}_roofline_traffic_{args.workload}.json") for r in (6, 5, 4)] +
                 [os.path.join(ROOT, "profiles", f"r{r:02d}_roofline_traffic.json") for r in (6, 5, 4, 3)])
        tpath = next((q for q in cands if os.path.exists(q)), "")
        own_mix = tpath.endswith(f"_{args.workload}.json") or args.workload == "cfg3"
        tj = json.load(open(tpath)) if tpath and args.workload in ("cfg3", "cfg4") and not args.batch else {}
        # the PMC bytes were collected on the cfg3 launch mix (32 pairs in one pass): the same 32-pair launches as cfg4's
        # micro-batches, but cfg4 adds the re-forward launches -- its per-class launch MIX differs, so the ratio of `traffic`
        # to `algorithmic_bytes_per_launch` is only meaningful for --workload cfg3
        tsrc = (f"{os.path.basename(tpath)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, per-launch average of the class's kernels "
                + (f"on this workload's own launch mix ({args.workload})" if own_mix else
                   "on the cfg3 launch mix; this run's launch mix differs (micro-batch re-forwards), compare traffic with "
                   "algorithmic bytes on --workload cfg3 only")) if tj else None
        timing = "HIP events on the launch stream around every launch of this class inside the timed steps"
        if summ_x is not None:
            timing = (f"HIP events on the launch stream around every launch of this class in {xsteps} step(s) of the same trainer on ONE "
                      f"stream (MC_STREAMS=0) run right after the timed region: the timed steps run the encoder chains on "
                      f"{1 + bin(streams).count('1')} concurrent streams, where a bracketed launch shares the GPU with other chains' "
                      "launches (those shared durations: in_timed_region)")

        def entry(key, src=None, nsteps=None):
            src = (summ_x if summ_x is not None else summ) if src is None else src
            nsteps = (xsteps if summ_x is not None else args.steps) if nsteps is None else nsteps
            cnt, t_ms, by, fl = src.get(key, (0, 0.0, 0, 0))
            # the tile-GEMM classes are split per launch by the roofline that bounds the launch's shape (ops.gemm tags them);
            # every other class by its aggregate intensity
            mfma = key.endswith("|mfma") or (not key.endswith("|hbm") and by > 0 and fl / by >= RIDGE)
            if mfma:
                ach, peak, unit = (fl / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0), MFMA_PEAK_TFS, "TFLOP/s"
            else:
                ach, peak, unit = (by / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0), HBM_PEAK_GBS, "GB/s"
            tkey = key[:-5] if key.endswith("|mfma") else (key[:-4] if key.endswith("|hbm") else key)
            return {"bound": "mfma" if mfma else "hbm", "achieved": round(ach, 1), "peak": peak, "unit": unit,
                    "frac": round(ach / peak, 4), "traffic": tj.get(key, tj.get(tkey)), "traffic_source": tsrc if tj.get(key, tj.get(tkey)) is not None else None,
                    "kernel": kernel_name(key), "class": tkey, "launches": cnt, "launches_per_step": cnt // max(nsteps, 1),
                    "avg_launch_us": round(t_ms / max(cnt, 1) * 1e3, 1), "algorithmic_bytes_per_launch": int(by / max(cnt, 1)),
                    "algorithmic_flops_per_launch": int(fl / max(cnt, 1)), "gpu_ms_in_timed_steps": round(summ.get(key, (0, 0.0, 0, 0))[1], 1),
                    "gpu_ms_per_step": round(t_ms / max(nsteps, 1), 2),
                    "share_of_gpu_time_in_survey_step": round(ssum[key][1] / max(sum(v[1] for v in ssum.values()), 1e-9), 4),
                    "timing": timing}

        def entry2(key):
            e = entry(key)
            if summ_x is not None:
                t = entry(key, summ, args.steps)
                e["in_timed_region"] = {"avg_launch_us": t["avg_launch_us"], "achieved": t["achieved"], "frac": t["frac"],
                                        "launches": t["launches"], "concurrent_streams": 1 + bin(streams).count("1")}
            return e
        ranked = sorted((entry2(k) for k in top), key=lambda r: -r["gpu_ms_per_step"])
        first, second, third = (ranked + [None, None, None])[:3]
        res = {
            "metric": "image-text pairs/s (whole node), EN-B5+BioClinicalBERT contrastive pre-training step",
            "value": round(pairs, 3), "unit": "image-text pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": os.environ.get("MC_STORAGE", "bf16").lower() + (" + fp8 e4m3 pointwise-conv operands" if fp8 else ""), "data": "synthetic",
            "config": {"workload": f"{args.workload}: {arch_name} + BioClinicalBERT(BERT-base), {b} pairs/GPU "
                                   f"({'1 image + 1 report' if single else '2 views + 2 reports'} each), {H}x{W} images, {T}-token reports, {args.loss} loss, "
                                   f"AdamW; fwd+loss+bwd+optimizer; dropout/drop-connect on",
                       "global_batch": b * world, "parallelism": f"dp{world}" + (f" x {args.micro_batches} micro-batches" if args.micro_batches > 1 else ""),
                       "loss": round(loss_val, 5), "peak_hbm_gb": round(peak_gb, 1), "peak_reserved_gb": round(peak_res_gb, 1),
                       "keep_graphs": args.keep_graphs, "recompute": args.recompute, "keep_recompute": keep_recompute,
                       "stat_tapes": stat_tapes_on,
                       "streams": streams,
                       "step_ms": {"first": round(step_ms[0], 1), "last": round(step_ms[-1], 1), "min": round(min(step_ms), 1),
                                   "max": round(max(step_ms), 1)}},     # rank 0's steps (HIP events on the main stream, which joins the chains)
            "roofline": first, "roofline_runner_up": second, "roofline_third": third,
        }
        # the whole step against both roofs: algorithmic bytes / flops of every launch of ONE step (the survey step's tags,
        # ops._note: each kernel reads its inputs and writes its outputs once) over the measured step time -- independent of
        # how the launches overlap
        sb, sf = sum(v[2] for v in ssum.values()), sum(v[3] for v in ssum.values())
        res["whole_step"] = {"algorithmic_gb_per_step": round(sb / 1e9, 1), "algorithmic_tflop_per_step": round(sf / 1e12, 2),
                             "hbm_gb_per_s": round(sb / 1e9 / (ms * 1e-3), 1), "hbm_frac": round(sb / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
                             "mfma_tflop_per_s": round(sf / 1e12 / (ms * 1e-3), 1), "mfma_frac": round(sf / (ms * 1e-3) / (MFMA_PEAK_TFS * 1e12), 4),
                             # 497 GB per 32-pair B5 step when every conv reads its input and writes its output once
                             "survey_8d_min_gb_per_step": (round(497.0 * b / 32, 1) if args.workload in ("cfg3", "cfg4", "cfg5") else None),
                             "hbm_frac_of_survey_min": (round(497e9 * b / 32 / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4) if args.workload in ("cfg3", "cfg4", "cfg5") else None),
                             "note": "per GPU; bytes / flops as tagged per launch (sum over the kernels' own inputs + outputs, i.e. the traffic of THIS pass structure, not SURVEY 8d's one-read-one-write minimum)"}
        if n8 is not None:
            res["n8_load"] = n8
        if parity is not None:
            res["parity"] = parity
        if parity_build is not None:
            res["parity_build"] = parity_build
        if dist_info is not None:
            res["rccl_ranks"] = dist_info["rccl_ranks"]
            res["dist"] = dist_info
        if world == 1 and not args.no_cpu_baseline and not single:
            try:
                res["cpu_baseline"] = cpu_baseline(arch_name, H, W, T)
            except Exception as e:                # pragma: no cover
                res["cpu_baseline"] = {"value": None, "error": repr(e)}
        if args.op_profile:
            rows = sorted(raw_timed.items(), key=lambda kv: -kv[1][1])
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
