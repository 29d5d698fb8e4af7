#!/usr/bin/env python
"""Secondary BASELINE.json configs on one MI355X, one JSON line each:

  config 2  DreamLLM-7B image-comprehension forward: prefill tokens/s (1 image -> 258 image tokens + text) and greedy
            decode tokens/s with the KV cache (batch 1: weight-bandwidth bound, 13.48 GB/token => <= ~590 tok/s at 8 TB/s)
  config 3  SD-2.1 512 px denoise steps/s at B_img = 1 and 8 (50 DDIM steps, CFG => UNet batch 2*B_img)
  config 5  DreamLLM-SDXL stage-I step (frozen LLM fwd + dgrad-only bwd, SDXL UNet fwd + dgrad at 128x128 latents,
            196 dream queries): samples/s; plus SDXL denoise steps/s (CFG, UNet batch 2*B_img)

    python tools/bench_configs.py [--only 2,3,5] [--out gpurun_out/configs.jsonl]
Synthetic inputs, random-init weights of the named architectures; inputs resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BF = torch.bfloat16
PEAK_TF, PEAK_GBS = 2500.0, 8000.0


def timed(fn, iters, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def emit(out, sink=None, **kw):
    if sink is not None:  # called from bench.py: collect instead of printing (bench.py prints ONE JSON line)
        sink.append(kw)
        return
    line = json.dumps(kw)
    print(line, flush=True)
    if out:
        with open(out, "a") as f:
            f.write(line + "\n")


def config2(out, a, model=None, sink=None):
    from dreamllm_amd.factory import build_dreamllm
    from dreamllm_amd.synthetic import make_interleaved_batch
    own = model is None
    model = (build_dreamllm(None, device="cuda", dtype=BF, with_sd=False) if own else model).eval()
    # one comprehension image per prompt: [bos] text <im_start> 256 patches <im_end> text
    S = a.prompt_len
    b = make_interleaved_batch(batch_size=a.prefill_batch, seq_len=S + 66, images_per_sample=1, device="cuda", dtype=BF)
    ids, img, mask = b["input_ids"], b["images"], b["attention_mask"]
    with torch.no_grad():
        t = timed(lambda: model(input_ids=ids, images=img, attention_mask=mask, image_index=b["image_index"], use_cache=True,
                                return_dict=True), a.iters, a.warmup)
    ntok = ids.numel()
    flops = 13.75e9 * ntok + 0.162e12 * img.shape[0]
    emit(out, sink, config=2, metric="image-comprehension prefill tokens/s", value=round(ntok / t, 1), unit="tokens/s",
         batch=a.prefill_batch, seq_len=ids.shape[1], images=img.shape[0], ms=round(t * 1e3, 2),
         frac_mfma_peak=round(flops / t / 1e12 / PEAK_TF, 4), dtype="bf16", data="synthetic")
    # greedy decode, batch 1, KV cache: token steps on the decode kernels, one hipGraph replay per token
    from dreamllm_amd.decode import GreedyDecodeSession
    ids1, img1 = ids[:1], img[:1]
    new = a.decode_tokens
    sess = GreedyDecodeSession(model, 1, ids1.shape[1] + 2 * new + 16)
    sess.prefill(ids1, images=img1)
    sess.generate(4)  # warm-up + graph capture
    sess.prefill(ids1, images=img1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sess.generate(new)
    torch.cuda.synchronize()
    per_tok = (time.perf_counter() - t0) / new
    wbytes = sum(p.numel() for n, p in model.named_parameters()
                 if n.startswith("model.layers") or n.startswith("lm_head") or n == "model.norm.weight") * 2
    emit(out, sink, config=2, metric="greedy decode tokens/s (batch 1, KV cache)", value=round(1.0 / per_tok, 1), unit="tokens/s",
         ms_per_token=round(per_tok * 1e3, 3), new_tokens=new, context=int(ids1.shape[1]),
         roofline=dict(bound="hbm", achieved=round(wbytes / per_tok / 1e9, 1), peak=PEAK_GBS, unit="GB/s",
                       frac=round(wbytes / per_tok / 1e9 / PEAK_GBS, 4), algorithmic_bytes_per_token=wbytes),
         dtype="bf16", data="synthetic")
    model._decode_session = None
    del sess
    if own:
        del model
    torch.cuda.empty_cache()


def config3(out, a):
    """SD-2.1 512 px denoise loop (BASELINE config 3 / metric M2) at B_img = 1 and 8: 50 deterministic DDIM steps, CFG 7.5
    => one UNet forward at batch 2*B_img per step.  Only the head is built (UNet + VAE + projector)."""
    from dreamllm_amd.modeling_plugins import StableDiffusionHead
    from dreamllm_amd.schedulers import DDIMScheduler
    torch.manual_seed(0)
    head = StableDiffusionHead("sd21-base", embed_hidden_size=4096).to("cuda", BF).eval()
    steps = 50
    for Bi in [int(x) for x in a.denoise_batches.split(',')]:
        g = torch.Generator().manual_seed(42)
        pe = (torch.randn(Bi, 64, 4096, generator=g) * 0.02).to("cuda", BF)
        ne = (torch.randn(Bi, 64, 4096, generator=g) * 0.02).to("cuda", BF)
        kw = dict(guidance_scale=7.5, prompt_embeds=pe, negative_prompt_embeds=ne, output_type="latent", scheduler=DDIMScheduler())
        head.pipeline(num_inference_steps=2, generator=torch.Generator().manual_seed(42), **kw)  # warm-up + graph capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        head.pipeline(num_inference_steps=steps, generator=torch.Generator().manual_seed(42), **kw)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        emit(out, config=3, metric="SD-2.1 512px denoise steps/s (50 DDIM eta=0, CFG 7.5)", value=round(steps / t, 2), unit="steps/s",
             batch_images=Bi, unet_batch=2 * Bi, ms_per_step=round(t / steps * 1e3, 3),
             frac_mfma_peak=round(steps / t * 2 * Bi * 0.803e12 / 1e12 / PEAK_TF, 4), dtype="bf16", data="synthetic")
    del head
    torch.cuda.empty_cache()


def config5(out, a, sink=None):
    from dreamllm_amd.factory import build_dreamllm_sdxl
    from dreamllm_amd.optim import HipAdamW
    from dreamllm_amd.schedulers import DDIMScheduler
    from dreamllm_amd.synthetic import make_creation_batch
    model = build_dreamllm_sdxl(None, device="cuda", dtype=BF, with_clip=False).train()
    B, S = a.sdxl_batch, a.sdxl_seq_len
    b = make_creation_batch(batch_size=B, seq_len=S, n_dream=196, device="cuda", dtype=BF, dm_size=a.sdxl_px)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = HipAdamW(params, lr=2e-3, weight_decay=0.0, max_grad_norm=1.0)

    def step():
        loss = model(**b).loss
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    t = timed(step, a.iters, a.warmup)
    lat = a.sdxl_px // 8
    unet_fwd = 6.89e12 * (lat / 128.0) ** 2  # SDXL UNet fwd @128x128, 196 ctx tokens
    flops = B * (2 * unet_fwd + 2 * 13.75e9 * S)
    emit(out, sink, config=5, metric="DreamLLM-SDXL stage-I train samples/s (frozen LLM fwd+dgrad, SDXL UNet fwd+dgrad)",
         value=round(B / t, 3), unit="samples/s", batch=B, seq_len=S, image_px=a.sdxl_px, ms_per_step=round(t * 1e3, 1),
         trainable_params=sum(p.numel() for p in params), frac_mfma_peak=round(flops / t / 1e12 / PEAK_TF, 4), dtype="bf16",
         data="synthetic", n_gpus=1)
    # SDXL denoise loop: steps/s at B_img = 1 (UNet batch 2)
    head = model.stable_diffusion_head
    model.eval()
    pe = torch.randn(1, 196, 4096, device="cuda", dtype=BF) * 0.02
    ne = torch.randn(1, 196, 4096, device="cuda", dtype=BF) * 0.02
    steps = a.sdxl_denoise_steps
    sched = DDIMScheduler()
    lat0 = torch.randn(1, 4, lat, lat, generator=torch.Generator().manual_seed(42)).cuda()

    def run():
        head.pipeline(height=a.sdxl_px, width=a.sdxl_px, num_inference_steps=steps, guidance_scale=7.5, latents=lat0.clone(),
                      prompt_embeds=pe, negative_prompt_embeds=ne, output_type="latent", scheduler=sched)

    t = timed(run, 2, 1)
    emit(out, sink, config=5, metric="SDXL denoise steps/s (DDIM eta=0, CFG 7.5, B_img=1)", value=round(steps / t, 2), unit="steps/s",
         image_px=a.sdxl_px, ms_per_step=round(t / steps * 1e3, 2), frac_mfma_peak=round(2 * unet_fwd * steps / t / 1e12 / PEAK_TF, 4),
         dtype="bf16", data="synthetic")


def _parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="2,3,5")
    ap.add_argument("--out", default="")
    ap.add_argument("--iters", type=int, default=5, help="timed iterations per leg (round 2: 3, after ONE warm-up: not reproducible)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prompt-len", type=int, default=448)
    ap.add_argument("--prefill-batch", type=int, default=8)
    ap.add_argument("--decode-tokens", type=int, default=64)
    ap.add_argument("--sdxl-batch", type=int, default=16)
    ap.add_argument("--sdxl-seq-len", type=int, default=256)
    ap.add_argument("--sdxl-px", type=int, default=1024)
    ap.add_argument("--sdxl-denoise-steps", type=int, default=20)
    ap.add_argument("--denoise-batches", default="1,8", help="config 3: images per denoise loop (UNet batch = 2x)")
    return ap


def default_args():
    return _parser().parse_args([])


def main():
    a = _parser().parse_args()
    if a.out and os.path.exists(a.out):
        os.remove(a.out)
    sel = set(a.only.split(","))
    if "2" in sel:
        config2(a.out, a)
    if "3" in sel:
        config3(a.out, a)
    if "5" in sel:
        config5(a.out, a)


if __name__ == "__main__":
    main()
