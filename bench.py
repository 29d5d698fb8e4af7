#!/usr/bin/env python
"""Headline benchmark of the DreamLLM hot path on MI355X (contract: see the round brief; BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one stage-II interleaved training step on a synthetic batch already resident in HBM: DreamLLM-7B
(Vicuna-7B dims, 32008 vocab) forward over B=16 x S=2048 interleaved documents with 2 comprehension + 2 creation images per
document -> CLIP-ViT-L/14 encode -> multimodal splice -> 32 decoder layers -> dream-state gather -> SD-2.1 head (VAE encode,
add_noise, UNet forward) -> lm_head + CE -> loss = 10*vm + 1*lm -> backward (LLM dgrad+wgrad, UNet dgrad) -> [RCCL gradient
all-reduce through DDP when N > 1] -> global-norm clip + AdamW.  Nothing is skipped inside the timed region.

value = whole-job training samples (sequences)/s = N*16*K / max-over-ranks wall time.  Weak scaling (16 seq/GPU).
Also reported in the same JSON line (not part of `value`): SD-2.1 512 px denoise steps/s (50 deterministic DDIM steps,
CFG 7.5 => UNet batch 2*B_img per step), `roofline` for the dominant kernel (the bf16 MFMA GEMM family, timed per launch
with HIP events on the launch stream), `cpu_baseline` (oracle decoder layer fwd+bwd on the host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
# analytic algorithmic FLOPs (multiply-add = 2), SURVEY.md §8(d)
FLOPS_TRAIN_SAMPLE = 88.1e12
FLOPS_UNET_FWD = 0.803e12
FLOPS_LM_HEAD_TOKEN = 2 * 4096 * 32000  # lm_head forward per token (0.262 GFLOP); x3 with both gradients


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="sequences per GPU (projects/dreamllm/configs/stage2/base.py:78)")
    ap.add_argument("--seq-len", type=int, default=2048)
    ap.add_argument("--images-per-sample", type=int, default=2)
    ap.add_argument("--model", default="7b", choices=["7b", "tiny"])
    ap.add_argument("--denoise-batch", type=int, default=1)
    ap.add_argument("--denoise-steps", type=int, default=50)
    ap.add_argument("--no-denoise", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-pack", action="store_true", help="A/B knob: one GEMM per projection instead of packed q|k|v and gate|up")
    ap.add_argument("--unfused-ce", action="store_true", help="A/B knob: lm_head + CE through the full fp32 logits")
    ap.add_argument("--no-configs", action="store_true", help="skip the short config-2 / config-5 legs (N = 1 only)")
    ap.add_argument("--no-ragged", action="store_true", help="skip the secondary ragged (varlen) training leg (S ~ U{S/2..S}, seqlens passed)")
    ap.add_argument("--ragged-steps", type=int, default=3)
    ap.add_argument("--no-grad-ckpt-leg", action="store_true", help="skip the secondary gradient-checkpointing leg (2 steps)")
    ap.add_argument("--sharded-grad", action="store_true",
                    help="N>1: reduce-scatter + sharded AdamW + all-gather (distributed.ShardedGradAdamW) instead of DDP all-reduce")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; anything but nccl (= RCCL) is accepted only together with --launch-check")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher dry run: form the N-rank process group, all-reduce a one per rank, print one JSON line, exit "
                         "(no GPU work; with --backend gloo it runs on CPU: tests/test_distributed_cpu.py)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: re-exec this command line under
    torch.distributed.run (one rank per GPU, 127.0.0.1, a free port) -- the reference's launcher is torchrun with 8 ranks per
    node (scripts/train/dreamllm/run_stage2.sh:1).  Rank 0's JSON line is the only stdout of the ranks, so it passes through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


def launch_check(a, rank, world):
    """The part of main() that decides whether an N-rank run is what it claims to be, without any GPU work."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(a.backend, init_method="env://")
    ranks = 1
    if a.gpus > 1:
        if not dist.is_initialized() or dist.get_backend() != a.backend or dist.get_world_size() != a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: needs {a.gpus} ranks on backend {a.backend}")
        ones = torch.ones(1, device=torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))) if a.backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks = int(ones.item())
        if ranks != a.gpus:
            raise SystemExit(f"all-reduce saw {ranks} ranks, expected {a.gpus}")
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": a.gpus, "ranks": ranks, "backend": a.backend if world > 1 else None}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _median3(fn, reps=3):
    """median wall time of `reps` runs of fn() (median of >= 3 after one warm-up)."""
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def cpu_baseline(seq_len, budget_s=60.0):
    """Oracle (`oracle/*_ref.py`, kind "port") timed on the host cores (baseline only): (i) one Vicuna-7B decoder
    layer forward+backward at B=1, S=seq_len, fp32, x32 layers + lm_head/CE; (ii) the CLIP-ViT-L/14 encoder forward on one
    image (x K_c = 2, frozen: forward only); (iii) the restated SD-2.1 UNet on one CFG batch (2 x [4,64,64], 64 context tokens):
    its time is the CPU denoise step, and 2 x forward stands for the training pass (forward + input gradient) of each of the
    K_g = 2 dream images.  Every piece: one warm-up, then the MEDIAN of 3 runs (a piece whose warm-up alone shows that three
    more runs would not fit the budget is timed once and says so)."""
    from oracle import llm_ref, unet_ref
    torch.manual_seed(0)
    H, Fd, nh = 4096, 11008, 32
    S = seq_len
    t_start = time.perf_counter()
    reps_used = {}

    def piece(name, fn):
        t0 = time.perf_counter()
        fn()                                        # warm-up (thread pool, allocator, lazy inits)
        w = time.perf_counter() - t0
        left = budget_s - (time.perf_counter() - t_start)
        reps = 3 if 3.2 * w < left else 1
        reps_used[name] = reps
        return _median3(fn, reps)

    sd = {}
    for n, shp in (("self_attn.q_proj.weight", (H, H)), ("self_attn.k_proj.weight", (H, H)), ("self_attn.v_proj.weight", (H, H)),
                   ("self_attn.o_proj.weight", (H, H)), ("mlp.gate_proj.weight", (Fd, H)), ("mlp.up_proj.weight", (Fd, H)),
                   ("mlp.down_proj.weight", (H, Fd))):
        sd[n] = (torch.randn(shp) * 0.02).requires_grad_(True)
    sd["input_layernorm.weight"] = torch.ones(H, requires_grad=True)
    sd["post_attention_layernorm.weight"] = torch.ones(H, requires_grad=True)
    cfg = dict(num_attention_heads=nh, num_key_value_heads=nh, rms_norm_eps=1e-6)
    cos, sin = llm_ref.rope_tables(H // nh, S)
    mask = llm_ref.causal_mask_4d(None, 1, S, torch.float32)
    pos = torch.arange(S)[None]

    def layer():
        x = torch.randn(1, S, H, requires_grad=True)
        llm_ref.decoder_layer(x, sd, "", cfg, cos, sin, pos, mask).square().mean().backward()

    t_layer = piece("layer", layer)
    # the same layer in bf16 (SURVEY §8d asks for fp32 AND bf16): bf16 weights / activations, torch's CPU bf16 GEMMs
    t_layer_bf16 = None
    try:
        sdb = {k: v.detach().to(torch.bfloat16).requires_grad_(True) for k, v in sd.items()}
        maskb = mask.to(torch.bfloat16)

        def layer_bf16():
            x = torch.randn(1, S, H).to(torch.bfloat16).requires_grad_(True)
            llm_ref.decoder_layer(x, sdb, "", cfg, cos, sin, pos, maskb).float().square().mean().backward()

        t_layer_bf16 = piece("layer_bf16", layer_bf16)
        del sdb, maskb
    except Exception:
        t_layer_bf16 = None
    w = (torch.randn(32008, H) * 0.02).requires_grad_(True)
    h = torch.randn(S, H, requires_grad=True)
    lab = torch.randint(0, 32008, (S,))
    t_head = piece("head", lambda: torch.nn.functional.cross_entropy((h @ w.t()).float(), lab).backward())
    del sd, w, h, mask
    # (ii) CLIP-ViT-L/14 forward, one image
    t_clip = None
    try:  # the installed transformers CLIPVisionModel: the class the reference wraps (modeling_plugins.py:214-219)
        from transformers import CLIPVisionConfig, CLIPVisionModel
        clip = CLIPVisionModel(CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                                num_attention_heads=16, image_size=224, patch_size=14)).eval()
        px = torch.randn(1, 3, 224, 224)
        with torch.no_grad():
            t_clip = piece("clip", lambda: clip(px, output_hidden_states=True))
        del clip
    except Exception:  # the CLIP share is 0.4 % of the sample's FLOPs: report without it rather than fail
        t_clip = None
    # (iii) restated SD-2.1 UNet, one CFG step (batch 2) at 64x64 latents
    ucfg = dict(unet_ref.SD21_BASE)
    usd = unet_ref.random_state_dict(ucfg, seed=0)
    xin, tin, cin = torch.randn(2, 4, 64, 64), torch.tensor([500]), torch.randn(2, 64, 1024)
    with torch.no_grad():
        t_unet2 = piece("unet", lambda: unet_ref.unet_forward(xin, tin, cin, usd, ucfg))
    del usd
    K = 2
    t_sample = 32 * t_layer + t_head + K * (t_clip or 0.0) + K * t_unet2  # K_g images x (fwd + dgrad ~ 2 fwd) = K x one batch-2 fwd x 2 / 2
    timing = "one warm-up per piece, then " + ", ".join(
        f"{n}: {'median of 3 runs' if r == 3 else f'{r} run (three more would not fit the {budget_s:.0f} s budget)'}" for n, r in reps_used.items())
    return dict(value=1.0 / t_sample, unit="samples/s", cores=torch.get_num_threads(), kind="port",
                denoise_steps_per_s=round(1.0 / t_unet2, 4), timing=timing, runs_per_piece=reps_used,
                decoder_layer_fwd_bwd_s=dict(fp32=round(t_layer, 3), bf16=None if t_layer_bf16 is None else round(t_layer_bf16, 3)),
                value_bf16_layers=None if t_layer_bf16 is None else 1.0 / (t_sample - 32 * t_layer + 32 * t_layer_bf16),
                sample=f"oracle ports (CLIP: installed transformers class), fp32: decoder layer fwd+bwd B=1 S={S} ({t_layer:.2f} s, x32) + lm_head/CE "
                       f"({t_head:.2f} s) + CLIP-L/14 fwd ({'n/a' if t_clip is None else f'{t_clip:.2f} s'}, x{K}) + SD-2.1 UNet CFG "
                       f"step batch 2 ({t_unet2:.2f} s = the CPU denoise step; x{K} stands for fwd+dgrad of {K} dream images); "
                       f"VAE not included; host has {os.cpu_count()} logical cores")


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)   # does not return
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.launch_check:
        return launch_check(a, rank, world)
    if a.backend != "nccl":
        raise SystemExit("bench.py measures on RCCL only (--backend gloo is for --launch-check)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from dreamllm_amd import distributed as D, ops
    from dreamllm_amd.factory import TINY, TINY_CLIP, TINY_DIFFUSION, VICUNA_7B, build_dreamllm
    from dreamllm_amd.optim import HipAdamW
    from dreamllm_amd.schedulers import DDIMScheduler
    from dreamllm_amd.synthetic import make_interleaved_batch

    D.init_distributed("nccl")
    # N > 1 must be RCCL with exactly N ranks -- fail loudly, never measure a silent single-process or gloo run
    rccl_ranks = 1
    if a.gpus > 1:
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_backend() != "nccl" or dist.get_world_size() != a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: needs {a.gpus} ranks on backend nccl (= RCCL), found "
                             f"{'no process group' if not dist.is_initialized() else (dist.get_backend(), dist.get_world_size())}; "
                             "launch with python -m torch.distributed.run --nproc-per-node N ...")
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                      # an all-reduce that actually ran on RCCL: every rank contributes 1
        rccl_ranks = int(ones.item())
        if rccl_ranks != a.gpus:
            raise SystemExit(f"RCCL all-reduce saw {rccl_ranks} ranks, expected {a.gpus}")
    tiny = a.model == "tiny"
    if tiny:
        model = build_dreamllm(TINY, device=dev, clip=TINY_CLIP, diffusion=TINY_DIFFUSION, num_dream_queries=8)
    else:
        model = build_dreamllm(VICUNA_7B, device=dev)
    head = model.stable_diffusion_head
    if a.no_pack:
        model.config.pack_projection_weights = False
    if a.unfused_ce:
        model.config.fused_lm_head_ce = False
    result = {}

    # ------------------------------------------------------------------ M2: SD-2.1 512 px denoise steps/s (replicas)
    denoise = None
    if not a.no_denoise:
        nq = model.model.dream_embedding.embed_len
        legs = []
        for Bi in ([a.denoise_batch] if a.denoise_batch != 1 else ([1] if (tiny or world > 1) else [1, 8])):  # B_img in {1, 8}
            g = torch.Generator().manual_seed(42)
            pe = (torch.randn(Bi, nq, model.config.hidden_size, generator=g) * 0.02).to(dev, torch.bfloat16)
            ne = (torch.randn(Bi, nq, model.config.hidden_size, generator=g) * 0.02).to(dev, torch.bfloat16)
            sched = DDIMScheduler()
            kw = dict(num_inference_steps=a.denoise_steps, guidance_scale=7.5, prompt_embeds=pe, negative_prompt_embeds=ne,
                      output_type="latent", scheduler=sched)
            if tiny:
                kw.update(height=128, width=128)
            head.pipeline(generator=torch.Generator().manual_seed(42), **{**kw, "num_inference_steps": 2})  # warm-up + graph capture
            dts = []
            for _ in range(3):      # median of 3 full 50-step loops (each 0.3-1.5 s): one loop alone moves +-3 % with the clock
                D.synchronize()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                head.pipeline(generator=torch.Generator().manual_seed(42), **kw)
                torch.cuda.synchronize()
                dts.append(D.max_over_ranks(time.perf_counter() - t0))
            dt = sorted(dts)[1]
            sps = world * a.denoise_steps / dt
            tf = sps / world * 2 * Bi * FLOPS_UNET_FWD / 1e12
            legs.append(dict(
                metric="SD-2.1 512px denoise steps/s (50 DDIM eta=0, CFG 7.5, replicas)", value=round(sps, 3), unit="steps/s",
                batch_images=Bi, unet_batch=2 * Bi, ms_per_step=round(1e3 * dt / a.denoise_steps, 3),
                loops_timed=3, loop_s=[round(x, 4) for x in dts],
                roofline=None if tiny else dict(
                    bound="mfma", kernel="SD-2.1 UNet forward (conv + GEMM + attention launches of one step, hipGraph replay)",
                    achieved=round(tf, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(tf / PEAK_BF16_TFLOPS, 4), traffic=None,
                    note="achieved = algorithmic 2*B_img*0.803 TFLOP per step / measured step time (whole loop body: the launches "
                         "live inside one hipGraph replay, so per-launch HIP events do not apply); kernel-level shares: "
                         "profiles/r06_denoise_kernel_stats.csv (B_img 1), r06_denoise_b8_kernel_stats.csv (B_img 8)")))
        denoise = dict(legs[0], legs=legs) if legs else None
        if denoise is not None and not tiny:
            denoise["frac_mfma_peak"] = legs[0]["roofline"]["frac"]

    # ------------------------------------------------------------------ M1: interleaved train samples/s
    train = {}
    if not a.no_train:
        model.train()
        params = [p for p in model.parameters() if p.requires_grad]
        timeline = None
        if a.sharded_grad and world > 1:
            ddp = model
            from dreamllm_amd.modeling_dreamllm import packed_parameter_groups
            opt = D.ShardedGradAdamW(params, lr=2e-5, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, overlap=True,
                                     atomic_groups=packed_parameter_groups(model))
        else:
            timeline = D.BucketTimeline() if world > 1 else None   # per-bucket ready / all-reduce-done events -> comm_exposed_ms
            ddp = D.wrap_ddp(model, timeline=timeline)
            opt = HipAdamW(params, lr=2e-5, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0)
        if tiny:
            batch = make_interleaved_batch(a.batch, 512, 1, n_dream=8, n_patch=16, seed=1234 + rank, device=dev, image_size=56,
                                           dm_size=128)
        else:
            batch = make_interleaved_batch(a.batch, a.seq_len, a.images_per_sample, seed=1234 + rank, device=dev)

        def step(b=None):
            out = ddp(**(batch if b is None else b), return_dict=True)
            out.loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            return out

        # setup (not a benchmark step): one priming step creates the optimizer state, sizes the allocator's pools and loads
        # every kernel, so that even `--warmup 0/1` times steady-state steps (measured: the second step of a fresh process can
        # run 10 % slow)
        step()
        if world > 1 and a.warmup < 1:
            step()   # torch DDP (static_graph) uses ONE all-gradient bucket in its first two iterations and the rebuilt 512 MB buckets
                     # from the third (tools/ddp_dry_7b.py): the timed region must never contain one of those two
        # warm-up steps run with the per-launch HIP-event timing switched on as well, so that the timed region starts in
        # steady state; the events the timed region will need are created here, outside it
        ops.GEMM_PROFILE = []
        for _ in range(a.warmup):
            out = step()
        per_step = (len(ops.GEMM_PROFILE) // a.warmup) if a.warmup > 0 else 1400
        ops.GEMM_PROFILE = None
        ops.prealloc_gemm_events(2 * per_step * a.steps + 64)
        D.synchronize()
        torch.cuda.synchronize()
        ops.GEMM_PROFILE = []
        if timeline is not None:
            per_step_buckets = max(1, len(timeline.records) // max(1, a.warmup + 1))
            timeline.reset()
            timeline.prealloc(2 * per_step_buckets * a.steps + 64)   # event creation stays outside the timed region
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = step()
        torch.cuda.synchronize()
        D.synchronize()
        dt = D.max_over_ranks(time.perf_counter() - t0)
        comm = timeline.summary() if timeline is not None else None
        prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
        loss_val = float(out.loss.item())
        gsum = sum(f for _, _, f, _ in prof)
        tsum = sum(s.elapsed_time(e) for s, e, _, _ in prof) * 1e-3
        by_tag = {}
        for s, e, f, tag in prof:
            d = by_tag.setdefault(tag, [0.0, 0.0, 0])
            d[0] += f
            d[1] += s.elapsed_time(e) * 1e-3
            d[2] += 1
        samples = world * a.batch * a.steps
        value = samples / dt
        # share of the S rows of a sample that carry an LM label (after the shift): what the fused lm_head + CE unit runs on
        lab = batch.get("labels")
        lab_share = float((lab[:, 1:] != -100).sum().item()) / float(lab.shape[0] * lab.shape[1]) if torch.is_tensor(lab) else 1.0
        flops_required = FLOPS_TRAIN_SAMPLE - 3 * FLOPS_LM_HEAD_TOKEN * a.seq_len * (1.0 - lab_share)
        achieved = gsum / tsum / 1e12 if tsum > 0 else 0.0
        # HBM-side bytes of one launch of the dominant shape, from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        # (profiles/roofline_traffic.json; corrected as MI355X_MICROARCH.md prescribes).  A number, per launch, like `achieved`.
        traffic, traffic_detail = None, None
        tf = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                fwd = next(v for k, v in tj["per_launch"].items() if k.startswith("fwd"))
                traffic = int(tj.get("gemm_bf16_kernel_hbm_bytes_per_launch") or fwd["hbm_bytes_corrected"])
                traffic_detail = {"shape": tj.get("shape"), "algorithmic_bytes": fwd["algorithmic_bytes"],
                                  "source": "STATIC: read from profiles/roofline_traffic.json, not measured in this run (rocprofv3 "
                                            "--pmc FETCH_SIZE/WRITE_SIZE in separate passes, FETCH doubled per MI355X_MICROARCH.md; "
                                            "L2->fabric requests, mostly Infinity-Cache hits)"}
            except Exception:
                traffic, traffic_detail = None, None
        train = dict(
            value=value, ms_per_step=1e3 * dt / a.steps, loss=loss_val,
            roofline=dict(bound="mfma", kernel="bf16 MFMA GEMM family: gemm_w4m_kernel (the decoder's linears) / gemm_pipe_kernel / gemm_pipe_tail_kernel / gemm_ring_kernel / gemm_bf16_kernel (linear fwd/dgrad/wgrad + implicit-GEMM conv)",
                          achieved=round(achieved, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(achieved / PEAK_BF16_TFLOPS, 4),
                          traffic=traffic, traffic_measured=False, traffic_detail=traffic_detail, launches_per_step=len(prof) // max(a.steps, 1),
                          avg_launch_ms=round(1e3 * tsum / max(len(prof), 1), 4),
                          time_share_of_step=round(tsum / dt, 4),
                          by_kind={k: dict(tflops=round(v[0] / v[1] / 1e12, 1), launches=v[2] // max(a.steps, 1))
                                   for k, v in by_tag.items() if v[1] > 0}),
            # SURVEY §8(d)'s 88.1 TFLOP / sample counts lm_head fwd + bwd on all S rows; the step computes it on the labelled rows only
            # (the rest of an interleaved document carries no LM loss): the headline fraction uses the FLOPs the step REQUIRES
            e2e_frac_mfma_peak=None if tiny else round(value / world * flops_required / (PEAK_BF16_TFLOPS * 1e12), 4),
            e2e_frac_mfma_peak_full_lm_head=None if tiny else round(value / world * FLOPS_TRAIN_SAMPLE / (PEAK_BF16_TFLOPS * 1e12), 4),
            flops_per_sample=dict(required=flops_required, survey_8d=FLOPS_TRAIN_SAMPLE, labelled_row_share=round(lab_share, 4)),
            peak_hbm_gb=round(torch.cuda.max_memory_allocated() / 1e9, 1),
            comm=comm,
        )

    # ------------------------------------------------------------------ secondary leg: ragged documents (SURVEY §8d "Synthetic inputs")
    # S ~ U{S/2 .. S} per document, right-padded to S, `seqlens` passed: the varlen path of modeling_dreamllm.py:521-545 (per-row spans in
    # the attention kernels, loss on the labelled rows).  Round 6: the decoder runs on COMPACT rows (valid tokens back to back; attention
    # alone on the padded grid -- modeling_dreamllm._token_pack; DREAMLLM_PACK_RAGGED=0 restores the padded-grid GEMMs of the reference).
    # Same model / optimizer state; 1 warm-up + `--ragged-steps` timed steps.
    ragged = None
    if not a.no_train and not a.no_ragged and not tiny and world == 1:   # N = 1 only: a rank that fails alone inside this try
        try:                                                                  # would leave the others in a collective
            rb = make_interleaved_batch(a.batch, a.seq_len, a.images_per_sample, seed=4321 + rank, device=dev, ragged=True)
            lens = rb["attention_mask"].sum(1).tolist()
            labr = rb["labels"]
            n_lab = int((labr[:, 1:] != -100).sum().item())
            step(rb)                                     # warm-up on the new shapes (kernel choices, allocator)
            ops.GEMM_PROFILE = None
            nlaunch = 1400
            ops.prealloc_gemm_events(2 * nlaunch * a.ragged_steps + 64)
            D.synchronize()
            torch.cuda.synchronize()
            ops.GEMM_PROFILE = []
            t0 = time.perf_counter()
            for _ in range(a.ragged_steps):
                out_r = step(rb)
            torch.cuda.synchronize()
            D.synchronize()
            dtr = D.max_over_ranks(time.perf_counter() - t0)
            profr, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
            gs = sum(f for _, _, f, _ in profr)
            tsr = sum(s_.elapsed_time(e_) for s_, e_, _, _ in profr) * 1e-3
            # algorithmic FLOPs from the ACTUAL lengths: 3 x fwd; fwd per token = 32 layers x (404.8 MFLOP linear +
            # 2 * L * d causal attention) ; lm_head on the labelled rows; per image the CLIP / UNet / projector terms of the dense sample
            per_tok_lin = 32 * 404.8e6
            fl = sum(3 * L * (per_tok_lin + 32 * 2 * L * 4096) for L in lens) + 3 * FLOPS_LM_HEAD_TOKEN * n_lab \
                + a.batch * a.images_per_sample * (0.162e12 + 1.61e12 + 3 * (2.15e9 + 0.54e9))
            tokens = sum(lens)
            ragged = dict(metric="interleaved train samples/sec, ragged documents (S ~ U{S/2..S}, right-padded, seqlens passed)",
                          value=round(world * a.batch * a.ragged_steps / dtr, 4), unit="samples/s", steps=a.ragged_steps,
                          ms_per_step=round(1e3 * dtr / a.ragged_steps, 3), tokens_per_s=round(world * tokens * a.ragged_steps / dtr, 1),
                          mean_len=round(tokens / a.batch, 1), min_len=int(min(lens)), max_len=int(max(lens)), loss=float(out_r.loss.item()),
                          flops_per_step=fl,
                          e2e_frac_mfma_peak=round(fl * a.ragged_steps / dtr / (PEAK_BF16_TFLOPS * 1e12), 4),
                          roofline=dict(bound="mfma", kernel="bf16 MFMA GEMM family (same kernels as the dense leg; M = the padded token count)",
                                        achieved=round(gs / tsr / 1e12, 1) if tsr > 0 else None, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                                        frac=round(gs / tsr / 1e12 / PEAK_BF16_TFLOPS, 4) if tsr > 0 else None, traffic=None,
                                        launches_per_step=len(profr) // max(a.ragged_steps, 1), time_share_of_step=round(tsr / dtr, 4),
                                        note="the decoder's GEMMs run on the compact rows of the valid tokens (round 6; executed FLOPs of the "
                                             "launches); e2e_frac_mfma_peak beside it uses the algorithmic FLOPs of the actual lengths"))
        except Exception as ex:  # secondary leg: never lose the headline line
            ragged = {"error": repr(ex)}

    # ------------------------------------------------------------------ secondary leg: the reference recipe's gradient checkpointing
    # (stage2/base.py:99 `gradient_checkpointing=True`, modeling_dreamllm.py:994-1003): whole-layer recompute inside _DecoderLayerFn.
    # Not the headline (288 GB hold the 7B activations: recompute only costs time here); reported so that both modes have a measured
    # step time and peak.  Same model / optimizer / batch; 1 warm-up + 2 timed steps.
    grad_ckpt = None
    if not a.no_train and not a.no_grad_ckpt_leg and not tiny and world == 1:
        try:
            model.gradient_checkpointing_enable()
            step()
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            t0 = time.perf_counter()
            for _ in range(2):
                out_c = step()
            torch.cuda.synchronize()
            dtc = time.perf_counter() - t0
            grad_ckpt = dict(metric="interleaved train samples/sec with gradient_checkpointing_enable() (whole-layer recompute)",
                             value=round(a.batch * 2 / dtc, 4), unit="samples/s", steps=2, ms_per_step=round(1e3 * dtc / 2, 3),
                             peak_hbm_gb=round(torch.cuda.max_memory_allocated() / 1e9, 1), loss=float(out_c.loss.item()),
                             vs_keeping_path=round((a.batch * 2 / dtc) / train["value"], 4) if train.get("value") else None)
        except Exception as ex:
            grad_ckpt = {"error": repr(ex)}
        finally:
            model.gradient_checkpointing_disable()

    # ------------------------------------------------------------------ the other BASELINE.json configs (N = 1 only, short)
    configs = None
    if world == 1 and not tiny and not a.no_configs:
        configs = {}
        try:
            del batch, opt, ddp
        except NameError:
            pass
        try:
            from tools import bench_configs as BC
            res = []
            BC.config2(None, BC.default_args(), model=model, sink=res)   # image-comprehension prefill + greedy decode (config 2)
            configs["config2"] = res
        except Exception as ex:
            configs["config2"] = {"error": repr(ex)}
        try:
            del model, head
            torch.cuda.empty_cache()
            res = []
            BC.config5(None, BC.default_args(), sink=res)                # DreamLLM-SDXL stage-I step + SDXL denoise (config 5)
            configs["config5"] = res
        except Exception as ex:
            configs["config5"] = {"error": repr(ex)}

    if rank == 0:
        line = {
            "metric": "interleaved train samples/sec (DreamLLM-7B stage-II fwd+bwd+allreduce+AdamW)",
            "value": round(train.get("value", 0.0), 4), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(train.get("ms_per_step", 0.0), 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("tiny smoke config" if tiny else
                                    "DreamLLM-7B (Vicuna-7B + CLIP-ViT-L/14 + SD-2.1 UNet/VAE) stage-II interleaved-doc training step, "
                                    "random-init weights"),
                       "model": "dreamllm-7b" if not tiny else "tiny", "global_batch": world * a.batch, "per_gpu_batch": a.batch,
                       "seq_len": a.seq_len if not tiny else 512, "images_per_sample": a.images_per_sample,
                       "parallelism": f"dp{world}" + ("-shardedgrad" if (a.sharded_grad and world > 1) else ""), "optimizer": "AdamW bf16 states + global-norm clip 1.0",
                       "activation_recompute": "none in the headline leg (normed inputs and the SwiGLU product are kept: +40 GB, peak_hbm_gb); "
                                               "the reference recipe's gradient checkpointing is measured beside it: train_grad_ckpt"},
            "loss": train.get("loss"),
            "roofline": train.get("roofline"),
            "e2e_frac_mfma_peak": train.get("e2e_frac_mfma_peak"),
            "e2e_frac_mfma_peak_full_lm_head": train.get("e2e_frac_mfma_peak_full_lm_head"),
            "flops_per_sample": train.get("flops_per_sample"),
            "peak_hbm_gb": train.get("peak_hbm_gb"),
            # ranks an all-reduce on backend nccl (= RCCL) actually summed over; DDP's per-bucket timeline of rank 0 (N > 1):
            # comm_exposed_ms = last all-reduce done - last bucket ready = communication the backward did not hide
            "rccl_ranks": rccl_ranks,
            "comm_exposed_ms": (train.get("comm") or {}).get("comm_exposed_ms"),
            "comm": train.get("comm"),
            "train_ragged": ragged,
            "train_grad_ckpt": grad_ckpt,
            "denoise": denoise,
            "configs": configs,
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(a.seq_len)
            except Exception as ex:  # the baseline is informational; never lose the GPU line
                line["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
