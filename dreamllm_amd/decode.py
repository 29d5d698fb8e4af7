"""KV-cache greedy decoding of DreamLLMForCausalMLM on the HIP decode kernels, one hipGraph replay per token.

Reference loop: omni/eval/language_eval/modeling_dreamllm.py:76-97 with temperature 0 (argmax, :92) over
DreamLLMForCausalMLM.forward (modeling_dreamllm.py:1353) with `use_cache=True`.  The reference re-enters the whole Python model
per token (~15 kernels x 32 layers + host logic); at batch 1 that is launch-bound by an order of magnitude over the weight
stream.  Here the token step is a fixed sequence of launches on static buffers:

    embed[tok] -> 32 x { [RMSNorm + q/k/v GEMV] -> [RoPE(pos on device) + K/V append to the cache at pos] -> decode attention
    over the cache (valid length on device) -> [o GEMV + residual] -> [RMSNorm + gate/up GEMV + SwiGLU] -> [down GEMV + residual] }
    -> [final RMSNorm + lm_head GEMV] (fp32 logits) -> argmax -> pos/len/step += 1      ([..] = one launch; 7 per layer;
    round 4: RoPE + append folded into the attention launch -> 6 per layer; the split-KV merge can be folded in
    too (`fuse_combine=True`: 5 per layer) but measured 1 % slower than the separate combine launch)

captured once per (batch, max_len) in a hipGraph (`torch.cuda.CUDAGraph`) and replayed per token; nothing in it depends on the
host.  Prefill runs through the normal model forward (MFMA GEMMs + flash attention) and its K/V are copied into the cache.
"""
from __future__ import annotations

import math

import torch

from . import ops


class GreedyDecodeSession:
    def __init__(self, model, batch_size: int, max_len: int, use_graph: bool = True, nsplit: int = 8, fused: bool = True,
                 vocab_limit: int | None = 32000, fuse_rope: bool = True, fuse_combine: bool = False, merge_in_oproj: bool = False):
        """`vocab_limit`: argmax runs over `logits[..., :vocab_limit]` -- the reference loop slices `[..., :32000]`
        (omni/eval/language_eval/modeling_dreamllm.py:79,86) so that none of the added special tokens (<dream_start>, <im_*>,
        [PAD] ...) can be emitted; None = whole vocabulary."""
        cfg = model.config
        self.model = model
        self.B, self.max_len, self.use_graph, self.nsplit, self.fused = batch_size, max_len, use_graph, nsplit, fused
        self.fuse_rope = fuse_rope
        # round 6 (opt-in, measured SLOWER: 298.6 against 308.3 tok/s in one process, profiles/r06_decode_merge_ab.log): the o projection's GEMV
        # merges the split-KV partials itself while staging x (5 launches per layer instead of 6) -- each of its ~512 blocks then re-reads the
        # 133 KB of partials from L2, which costs more than the 5.5 us combine launch it removes; LDS staging covers up to 4 sequences
        self.merge_oproj = bool(merge_in_oproj) and fuse_rope and fused and batch_size <= 4 and \
            batch_size * cfg.hidden_size * 2 <= 60 * 1024
        # arrival counters of the in-launch split-KV merge (zero at rest; one array per session = per stream in flight)
        self.attn_counters = (torch.zeros(batch_size * cfg.num_attention_heads, dtype=torch.int32, device=model.device)
                              if fuse_combine else None)
        if batch_size > 8:
            raise ValueError("the decode GEMV handles up to 8 sequences per step")
        dev, dt = model.device, model.dtype
        self.H, self.Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
        self.D = cfg.hidden_size // self.H
        L = cfg.num_hidden_layers
        self.kc = [torch.zeros(batch_size, max_len, self.Hkv, self.D, dtype=dt, device=dev) for _ in range(L)]
        self.vc = [torch.zeros(batch_size, max_len, self.Hkv, self.D, dtype=dt, device=dev) for _ in range(L)]
        self.tok = torch.zeros(batch_size, dtype=torch.long, device=dev)          # token fed to the next step
        self.pos = torch.zeros(batch_size, 1, dtype=torch.long, device=dev)       # its position id
        self.kv_len = torch.zeros(batch_size, dtype=torch.int32, device=dev)      # valid cache length INCLUDING that token
        self.slot = torch.zeros(batch_size, dtype=torch.long, device=dev)         # row of cache.view(B*max_len, -1) to write
        self.step_idx = torch.zeros(1, dtype=torch.long, device=dev)
        self.out_tokens = torch.zeros(max_len, batch_size, dtype=torch.long, device=dev)
        self.vocab_limit = cfg.vocab_size if vocab_limit is None else min(int(vocab_limit), cfg.vocab_size)
        self.kv_start = torch.zeros(batch_size, dtype=torch.int32, device=dev)    # first valid cache slot (left-padded prompts)
        # teacher forcing of rows that are still inside their prompt (ragged batches, reference loop :93-94): step i emits
        # force_tok[i] where force_mask[i] is set, the argmax otherwise
        self.force_tok = torch.zeros(max_len, batch_size, dtype=torch.long, device=dev)
        self.force_mask = torch.zeros(max_len, batch_size, dtype=torch.bool, device=dev)
        rope = model.model.layers[0].self_attn.rotary_emb
        self.cos, self.sin = rope.tables(max_len, dev)
        self.graph = None
        self.logits = None

    # ---- one token -------------------------------------------------------------------------------------------------
    def _token_step(self):
        m = self.model
        cfg = m.config
        B, H, Hkv, D = self.B, self.H, self.Hkv, self.D
        eps = cfg.rms_norm_eps
        x = m.model.embed_tokens.weight.index_select(0, self.tok)  # [B, d]
        pos = self.pos.view(-1)
        for li, layer in enumerate(m.model.layers):
            at, mlp = layer.self_attn, layer.mlp
            if self.fused:
                # 7 launches per layer: [RMSNorm + q/k/v GEMV] [RoPE + cache append] [attention x2] [o GEMV + residual]
                # [RMSNorm + gate/up GEMV + SwiGLU] [down GEMV + residual]
                q, k, v = ops.gemv_fused(x, (at.q_proj.weight, at.k_proj.weight, at.v_proj.weight),
                                         norm_w=layer.input_layernorm.weight, eps=eps)
                q = q.view(B, H, D)
                if self.merge_oproj:
                    # round 6: 5 launches per layer -- the split-KV partials go straight into the o projection, whose blocks merge them
                    # while staging x (no combine launch)
                    ws = ops.attn_decode_rope(q, k.view(B, Hkv, D), v.view(B, Hkv, D), self.kc[li], self.vc[li], self.kv_len, self.cos,
                                              self.sin, pos, 1.0 / math.sqrt(D), self.nsplit, kv_start=self.kv_start, partials_only=True)
                    x2 = ops.gemv_attn_combine(ws, at.o_proj.weight, B, H, D, self.nsplit, residual=x)
                    act = ops.gemv_fused(x2, (mlp.gate_proj.weight, mlp.up_proj.weight), norm_w=layer.post_attention_layernorm.weight,
                                         eps=eps, swiglu=True)
                    x = ops.gemv(act, mlp.down_proj.weight, residual=x2)
                    continue
                if self.fuse_rope:   # RoPE + cache append inside the attention launch (round 4)
                    o = ops.attn_decode_rope(q, k.view(B, Hkv, D), v.view(B, Hkv, D), self.kc[li], self.vc[li], self.kv_len, self.cos,
                                             self.sin, pos, 1.0 / math.sqrt(D), self.nsplit, kv_start=self.kv_start,
                                             counters=self.attn_counters)
                else:
                    ops.rope_append_(q, k.view(B, Hkv, D), v.view(B, Hkv, D), self.kc[li], self.vc[li], self.cos, self.sin, pos,
                                     kv_len=self.kv_len)
                    o = ops.attn_decode(q, self.kc[li], self.vc[li], self.kv_len, 1.0 / math.sqrt(D), self.nsplit,
                                        kv_start=self.kv_start)
                x2 = ops.gemv(o.view(B, H * D), at.o_proj.weight, residual=x)
                act = ops.gemv_fused(x2, (mlp.gate_proj.weight, mlp.up_proj.weight), norm_w=layer.post_attention_layernorm.weight,
                                     eps=eps, swiglu=True)
                x = ops.gemv(act, mlp.down_proj.weight, residual=x2)
                continue
            h, _, _ = ops.rmsnorm_fwd(x, layer.input_layernorm.weight, eps)
            q = ops.gemv(h, at.q_proj.weight).view(B, 1, H, D)
            k = ops.gemv(h, at.k_proj.weight).view(B, 1, Hkv, D)
            v = ops.gemv(h, at.v_proj.weight)
            ops.rope_(q, self.cos, self.sin, self.pos)
            ops.rope_(k, self.cos, self.sin, self.pos)
            self.kc[li].view(B * self.max_len, Hkv * D).index_copy_(0, self.slot, k.view(B, Hkv * D))
            self.vc[li].view(B * self.max_len, Hkv * D).index_copy_(0, self.slot, v)
            o = ops.attn_decode(q.view(B, H, D), self.kc[li], self.vc[li], self.kv_len, 1.0 / math.sqrt(D), self.nsplit,
                                kv_start=self.kv_start)
            x2 = ops.gemv(o.view(B, H * D), at.o_proj.weight, residual=x)
            h2, _, _ = ops.rmsnorm_fwd(x2, layer.post_attention_layernorm.weight, eps)
            g = ops.gemv(h2, mlp.gate_proj.weight)
            u = ops.gemv(h2, mlp.up_proj.weight)
            x = ops.gemv(ops.glu_fwd(g, u, 0), mlp.down_proj.weight, residual=x2)
        if self.fused:
            logits = ops.gemv_fused(x, (m.lm_head.weight,), norm_w=m.model.norm.weight, eps=eps, out_dtype=torch.float32)[0]
        else:
            hf, _, _ = ops.rmsnorm_fwd(x, m.model.norm.weight, eps)
            logits = ops.gemv(hf, m.lm_head.weight, out_dtype=torch.float32)
        nxt = logits[:, : self.vocab_limit].argmax(-1)
        nxt = torch.where(self.force_mask.index_select(0, self.step_idx)[0], self.force_tok.index_select(0, self.step_idx)[0], nxt)
        self.out_tokens.index_copy_(0, self.step_idx, nxt[None])
        self.tok.copy_(nxt)
        self.pos.add_(1)
        self.kv_len.add_(1)
        self.slot.add_(1)
        self.step_idx.add_(1)
        return logits

    # ---- public ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def prefill(self, input_ids, images=None, attention_mask=None, forced_tokens=None, forced_mask=None):
        """Run the prompt through the model (MFMA path), fill the cache, emit the first new token.

        `attention_mask` (LEFT padding, the layout of the reference's batched generate() callers,
        omni/eval/vqa/vqa_inference.py:276): position ids are mask-aware as HF generate computes them
        (`prepare_inputs_for_generation`, modeling_dreamllm.py:1527-1533); every row's next token goes to cache slot S, its
        position is the row's own length, and the pad slots in front are masked in the token steps (`kv_start`).  Right-padded
        or holey masks raise: a right-padded row would need a hole between its prompt and its new tokens.
        `forced_tokens` / `forced_mask` [B, n]: token i after the prompt is forced to `forced_tokens[:, i]` where
        `forced_mask[:, i]` is set (teacher forcing of rows still inside their prompt, reference loop :93-94)."""
        B, S = input_ids.shape
        if B != self.B or S >= self.max_len:
            raise ValueError(f"prompt [{B},{S}] does not fit the session (batch {self.B}, max_len {self.max_len})")
        dev = self.tok.device
        position_ids, lens = None, None
        if attention_mask is not None:
            am = attention_mask.to(dev) != 0
            lens = am.sum(-1)
            start = S - lens
            left = torch.arange(S, device=dev)[None] >= start[:, None]
            if not bool((left == am).all()):
                raise ValueError("GreedyDecodeSession.prefill: attention_mask must be LEFT-padded (one run of valid tokens ending "
                                 "at the last prompt position); use forced_tokens for right-padded ragged prompts")
            position_ids = (am.long().cumsum(-1) - 1).masked_fill(~am, 1)
            self.kv_start.copy_(start.to(torch.int32))
        else:
            self.kv_start.zero_()
        out = self.model(input_ids=input_ids, images=images, attention_mask=attention_mask, position_ids=position_ids,
                         use_cache=True, return_dict=True)
        for li, (k, v) in enumerate(out.past_key_values):  # reference layout [B, Hkv, S, D]
            self.kc[li][:, :S].copy_(k.transpose(1, 2))
            self.vc[li][:, :S].copy_(v.transpose(1, 2))
        first = out.logits[:, -1, : self.vocab_limit].argmax(-1)
        self.force_mask.zero_()
        if forced_tokens is not None:
            n = forced_tokens.shape[1]
            if n > 0:
                first = torch.where(forced_mask[:, 0].to(dev), forced_tokens[:, 0].to(dev), first)
                self.force_tok[: n - 1].copy_(forced_tokens[:, 1:].t())
                self.force_mask[: n - 1].copy_(forced_mask[:, 1:].t())
        self.tok.copy_(first)
        if lens is None:
            self.pos.fill_(S)
        else:
            self.pos.copy_(lens[:, None])
        self.kv_len.fill_(S + 1)
        self.slot.copy_(torch.arange(B, device=dev) * self.max_len + S)
        self.step_idx.zero_()
        self.prompt_len = S
        self.generated = 0  # tokens produced by generate() since this prefill (host-side count of graph replays)
        self.first = first
        return first

    @torch.no_grad()
    def generate(self, n_more: int):
        """n_more further tokens after those already produced since `prefill`; returns [B, n_more] (may be called repeatedly)."""
        if self.prompt_len + 1 + self.generated + n_more > self.max_len:
            raise ValueError("generation would overflow the KV cache")
        if n_more <= 0:
            return self.out_tokens[:0].t()
        if self.use_graph and self.graph is None:
            state = [t.clone() for t in (self.tok, self.pos, self.kv_len, self.slot, self.step_idx, self.out_tokens)]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up outside capture
                self._token_step()
            torch.cuda.current_stream().wait_stream(side)
            for t, s in zip((self.tok, self.pos, self.kv_len, self.slot, self.step_idx, self.out_tokens), state):
                t.copy_(s)  # the warm-up wrote cache row `slot`, which the first real step rewrites identically
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.logits = self._token_step()
            for t, s in zip((self.tok, self.pos, self.kv_len, self.slot, self.step_idx, self.out_tokens), state):
                t.copy_(s)  # capture does not execute, but keep the state explicit
        for _ in range(n_more):
            if self.use_graph:
                self.graph.replay()
            else:
                self.logits = self._token_step()
        start = self.generated
        self.generated += n_more
        return self.out_tokens[start:start + n_more].t().contiguous()
