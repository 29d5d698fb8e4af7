"""Data-parallel runtime: one process per GPU, `torch.distributed` with backend "nccl" (= RCCL over xGMI on ROCm).

The reference's only exchange step is the gradient all-reduce that accelerate's DDP wrap performs
(omni/train/trainer.py:577-601, backward at :1043).  Here: torch DDP over RCCL with
* `gradient_as_bucket_view=True` (gradients live in the communication buckets: no extra 13.5 GB copy),
* large buckets (default 512 MB): the 8-GPU xGMI mesh is point-to-point (7 links x ~153 GB/s), ring collectives are
  per-link bound, so fewer/larger messages amortise launch + protocol latency; buckets still overlap with the backward
  of earlier layers because the fused decoder-layer Function returns all of a layer's weight gradients at once,
* `static_graph=True`: every trainable parameter receives a gradient every step (the reference keeps that invariant with
  dummy forwards, modeling_dreamllm.py:1142-1144,1443-1445), so bucket order is fixed after the first step.
Helpers for rank bookkeeping mirror omni/utils/comm.py:10-58.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def get_rank() -> int:
    return int(os.environ.get("RANK", 0))


def get_local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", 0))


def get_world_size() -> int:
    return int(os.environ.get("WORLD_SIZE", 1))


def is_main_process() -> bool:
    return get_rank() == 0


def init_distributed(backend: str | None = None):
    """Initialise the default process group from torchrun's env (RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT)."""
    ws = get_world_size()
    if ws <= 1 or dist.is_initialized():
        return ws
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(get_local_rank())
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", get_local_rank()))
    else:
        dist.init_process_group(backend=backend)
    return ws


def synchronize():
    """omni/utils/comm.py:38-57."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


class BucketTimeline:
    """Per-bucket timeline of DDP's gradient all-reduce, so that the first scaling run on real xGMI is diagnosable: a comm hook
    (`DDP.register_comm_hook`) that performs the default all-reduce(mean) and records, per bucket, WHEN the bucket became ready
    (the backward has written its last gradient: an event on the compute stream) and WHEN its all-reduce finished (an event
    recorded from the future's completion callback, i.e. on the stream that is ordered after the collective).

    `summary()` -> per step (averaged over the recorded steps):
        comm_exposed_ms   last all-reduce done - last bucket ready: the communication the backward could NOT hide (at least the
                          last bucket's own all-reduce; everything beyond it is a backlog of earlier buckets)
        comm_busy_ms      sum over buckets of (done - max(ready, previous bucket done)): time the fabric was reducing
        backward_span_ms  first bucket ready -> last bucket ready
        buckets           count, bytes, and the bucket -> first-parameter order of one step (reverse layer order expected)
    CPU / gloo (the tests): time.perf_counter() instead of events."""

    def __init__(self, process_group=None):
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.records = []       # per bucket launch: dict(index, bytes, ready, done, is_last)
        self.enabled = True
        self._cuda = None
        self._pool = []         # pre-created events (hipEventCreate costs ~0.2 ms: keep it out of a timed region)

    def prealloc(self, n_events):
        """Create `n_events` timing events now (and force their lazy creation by recording them once)."""
        st = torch.cuda.current_stream()
        for _ in range(max(0, n_events - len(self._pool))):
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            self._pool.append(e)

    def _now(self, cuda):
        if cuda:
            e = self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            return e
        import time as _t
        return _t.perf_counter()

    def hook(self, _state, bucket):
        buf = bucket.buffer()
        cuda = buf.is_cuda
        self._cuda = cuda
        rec = None
        if self.enabled:
            rec = dict(index=bucket.index(), bytes=buf.numel() * buf.element_size(), ready=self._now(cuda), done=None,
                       is_last=bool(bucket.is_last()), param_ptrs=[int(q.data_ptr()) for q in bucket.parameters()])
            self.records.append(rec)
        fut = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True).get_future()
        world = self.world

        def _done(f):
            out = f.value()[0]
            out.div_(world)                       # mean, as DDP's built-in reducer
            if rec is not None:
                rec["done"] = self._now(cuda)
            return out

        return fut.then(_done)

    def reset(self):
        self.records = []

    def summary(self):
        recs = [r for r in self.records if r["done"] is not None]
        if not recs:
            return None
        if self._cuda:
            torch.cuda.synchronize()
            ms = lambda a, b: a.elapsed_time(b)
        else:
            ms = lambda a, b: (b - a) * 1e3
        steps, cur = [], []
        for r in recs:
            cur.append(r)
            if r["is_last"]:
                steps.append(cur)
                cur = []
        if not steps:
            steps = [recs]
        exposed, busy, span = [], [], []
        for st in steps:
            exposed.append(ms(st[-1]["ready"], st[-1]["done"]))
            span.append(ms(st[0]["ready"], st[-1]["ready"]))
            b, prev_done = 0.0, None
            for r in st:
                start_after_prev = prev_done is not None and ms(r["ready"], prev_done) > 0
                b += ms(prev_done, r["done"]) if start_after_prev else ms(r["ready"], r["done"])
                prev_done = r["done"]
            busy.append(b)
        avg = lambda v: round(sum(v) / len(v), 3)
        one = steps[-1]
        return dict(steps=len(steps), buckets_per_step=len(one), bucket_mb=[round(r["bytes"] / 2**20, 1) for r in one],
                    bucket_order=[r["index"] for r in one], bucket_param_ptrs=[r["param_ptrs"] for r in one],
                    comm_exposed_ms=avg(exposed), comm_busy_ms=avg(busy),
                    backward_span_ms=avg(span))


def wrap_ddp(model: torch.nn.Module, bucket_cap_mb: int = 512, timeline: BucketTimeline | None = None):
    """`timeline`: a `BucketTimeline` whose comm hook replaces DDP's built-in all-reduce by the same all-reduce(mean) plus
    per-bucket ready / done timestamps (bench.py exports its summary as `comm_exposed_ms`)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    kw = {}
    if next(model.parameters()).is_cuda:
        kw = dict(device_ids=[torch.cuda.current_device()], output_device=torch.cuda.current_device())
    ddp = DDP(model, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True, static_graph=True,
              find_unused_parameters=False, broadcast_buffers=False, **kw)
    if timeline is not None:
        ddp.register_comm_hook(None, timeline.hook)
    return ddp


def shard_for_rank(n_items: int, rank: int | None = None, world: int | None = None):
    """Contiguous disjoint slice of `n_items` work units for this rank (DistributedSampler-like partition; replicas of the
    denoising loop shard prompts the same way, omni/eval/text2img/ddp_sample_coco.py:167)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    per = (n_items + world - 1) // world
    return range(min(rank * per, n_items), min((rank + 1) * per, n_items))


def max_over_ranks(value: float) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_dict(d: dict, average: bool = True) -> dict:
    """omni/utils/comm.py:123-152 (all-reduce instead of reduce-to-0 so every rank can log)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return d
    keys = sorted(d)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.stack([torch.as_tensor(d[k], dtype=torch.float32, device=dev).reshape(()) for k in keys])
    dist.all_reduce(t)
    if average:
        t /= dist.get_world_size()
    return {k: v for k, v in zip(keys, t)}


class ShardedGradAdamW:
    """Sharded-gradient data parallelism: the `shard_grad_op` mode of the reference's stage-II / SFT recipes
    (omni/train/trainer.py:199-230, projects/dreamllm/configs/stage2/base.py:91-93, omni/utils/fsdp_utils.py:23-61) without FSDP.

    Parameters stay whole on every rank (forward and backward are the plain replica step: no parameter all-gather inside the
    model, no DDP wrapper); what is sharded is the gradient reduction, the AdamW moments and the update:

        backward  ->  per bucket: reduce_scatter(mean) of the flat gradient   (each rank keeps 1/N of it)
                  ->  global grad norm = sqrt(all_reduce(sum of the local shards' squares))  -> clip coefficient on device
                  ->  fused AdamW on the local 1/N slice of the flat parameter buffer (moments exist only for that slice)
                  ->  all_gather of the updated slices back into the flat parameter buffer

    i.e. ZeRO-2: per-rank optimizer memory drops from 2 x params to 2 x params / N and gradient traffic is one reduce-scatter
    + one all-gather of the parameter bytes (the same bytes on the wire as DDP's all-reduce), in buckets of `bucket_mb`.
    The trainable parameters are re-pointed at views of contiguous flat buffers (one per bucket, padded to a multiple of the
    world size); gradients are accumulated straight into flat gradient buffers (`p.grad` views), so no gather/scatter copies
    exist.  On the xGMI mesh both collectives are RCCL reduce-scatter / all-gather over the default process group.

    `update_fn(p, g, m, v, lr, b1, b2, eps, wd, step, clip_coef)` defaults to the HIP AdamW kernel (`ops.adamw_`); the CPU
    tests inject a torch restatement (the product path has no CPU fallback)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None, bucket_mb=512,
                 state_dtype=None, update_fn=None, sumsq_fn=None, process_group=None, overlap=False, atomic_groups=None):
        """`overlap=True`: a bucket's reduce-scatter is issued (async, on the communication stream) from a
        post-accumulate-grad hook the moment the LAST gradient of the bucket has been written by the backward -- buckets are
        filled in reverse layer order, so the collectives of the late layers run under the backward of the early ones, as DDP's
        bucket hooks do.  Requires one `step()` per backward (no gradient accumulation across backwards).

        `params`: an iterable of parameters, or torch-style param groups `[{"params": [...], "weight_decay": 0.0}, ...]`
        (the reference trainer excludes biases and `ALL_LAYERNORM_LAYERS` weights from decay, omni/train/trainer.py:388-411):
        a bucket never mixes decay values.  Invariant shared with DDP's `static_graph`: every trainable parameter receives
        a gradient every step (the gradients live in flat buffers, so "no gradient" cannot be told from a zero gradient).

        `atomic_groups`: parameter lists that must not be split by a bucket boundary -- the q|k|v and gate|up weights a decoder
        layer applies as ONE packed GEMM (`modeling_dreamllm.packed_parameter_groups(model)`): a cut between them would silently
        drop the layer back to one GEMM per projection."""
        self.pg = process_group
        self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
        self.rank = dist.get_rank(self.pg) if dist.is_initialized() else 0
        self.lr, self.betas, self.eps = lr, betas, eps
        self.max_grad_norm = max_grad_norm
        self._step = 0
        self.last_grad_norm = None
        # the collectives are chosen ONCE, from the backend: RCCL has reduce_scatter_tensor / all_gather_into_tensor; the gloo
        # transport of the CPU tests does not, and takes all_reduce + all_gather.  No fallback inside step(): a failing
        # collective must surface, not silently switch pattern on one rank.
        self.tensor_collectives = self.world > 1 and dist.get_backend(self.pg) == "nccl"
        if update_fn is None:
            from . import ops

            def update_fn(p, g, m, v, lr, b1, b2, eps, wd, step, coef):
                ops.adamw_(p, g, m, v, lr, b1, b2, eps, wd, step, 1.0, coef)

            def sumsq_fn(g):
                parts = torch.zeros(ops.SUMSQ_PARTS, dtype=torch.float32, device=g.device)
                ops.sumsq_partials_(g, parts)
                return ops.reduce_sum_f32(parts)
        self.update_fn, self.sumsq_fn = update_fn, sumsq_fn
        params = list(params)
        if params and isinstance(params[0], dict):
            groups = [([p for p in g["params"] if p.requires_grad], float(g.get("weight_decay", weight_decay))) for g in params]
        else:
            groups = [([p for p in params if p.requires_grad], float(weight_decay))]
        flat = [(p, wd) for ps, wd in groups for p in ps]
        if not flat:
            raise ValueError("no trainable parameters")
        # buckets in REVERSE registration order: the last layers' gradients are complete first
        cap = bucket_mb * 1024 * 1024
        gid, gbytes = {}, {}
        for gi, grp in enumerate(atomic_groups or ()):
            for q in grp:
                gid[id(q)] = gi
            gbytes[gi] = sum(q.numel() * q.element_size() for q in grp)
        self.buckets, self.bucket_wd, cur, cur_bytes, cur_wd = [], [], [], 0, None
        open_group = None   # atomic group whose first member (in reverse order) is already in `cur`
        for p, wd in reversed(flat):
            nb = p.numel() * p.element_size()
            g = gid.get(id(p))
            if g is not None and g == open_group:
                need = 0            # the whole group was accounted for when its first member arrived: never cut inside it
            else:
                need = gbytes[g] if g is not None else nb
                open_group = g
            if cur and ((need and cur_bytes + need > cap) or p.dtype != cur[0].dtype or p.device != cur[0].device or wd != cur_wd):
                self.buckets.append(cur)
                self.bucket_wd.append(cur_wd)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_wd = wd
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
            self.bucket_wd.append(cur_wd)
        self.flat_p, self.flat_g, self.shard_m, self.shard_v, self.layout = [], [], [], [], []
        # inside a bucket the parameters sit in REGISTRATION order: weights packed for single-GEMM use (q|k|v, gate|up:
        # modeling_dreamllm.pack_linear_weights) stay adjacent, in order, after they move into the flat buffer
        self.buckets = [b[::-1] for b in self.buckets]
        for bucket in self.buckets:
            n = sum(p.numel() for p in bucket)
            padded = (n + self.world - 1) // self.world * self.world
            fp = torch.zeros(padded, dtype=bucket[0].dtype, device=bucket[0].device)
            fg = torch.zeros(padded, dtype=bucket[0].dtype, device=bucket[0].device)
            off = 0
            for p in bucket:
                k = p.numel()
                fp[off:off + k].copy_(p.data.reshape(-1))
                p.data = fp[off:off + k].view(p.shape)   # the parameter now lives in the flat buffer
                p._dllm_flat_owned = True                # nobody else may re-point it (DreamLLMDecoderLayer.pack_weights)
                p.grad = fg[off:off + k].view(p.shape)   # autograd accumulates into the flat gradient buffer
                off += k
            shard = padded // self.world
            sd = state_dtype or bucket[0].dtype
            self.flat_p.append(fp)
            self.flat_g.append(fg)
            self.shard_m.append(torch.zeros(shard, dtype=sd, device=fp.device))
            self.shard_v.append(torch.zeros(shard, dtype=sd, device=fp.device))
            self.layout.append((n, padded, shard))

        self.overlap = bool(overlap) and self.world > 1
        self._pending = [None] * len(self.buckets)   # (work handle, output shard) of a reduce-scatter already in flight
        self._left = [len(b) for b in self.buckets]
        if self.overlap:
            for bi, bucket in enumerate(self.buckets):
                for p in bucket:
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))

    def _make_hook(self, bi):
        def hook(_param):
            self._left[bi] -= 1
            if self._left[bi] < 0 or (self._left[bi] == 0 and self._pending[bi] is not None):
                # a second backward before step() (gradient accumulation, a skipped step): the reduce-scatter already in flight
                # saw only the first backward's gradients.  Drop it; step() then reduces the accumulated buffer synchronously.
                pend, self._pending[bi] = self._pending[bi], None
                if pend is not None and pend[0] is not None:
                    pend[0].wait()
                    if not self.tensor_collectives:
                        raise RuntimeError("ShardedGradAdamW(overlap=True): a second backward before step() on a backend whose "
                                           "reduce is in place (gloo all_reduce) cannot be undone; use overlap=False for "
                                           "gradient accumulation")
                return
            if self._left[bi] == 0:
                self._pending[bi] = self._launch_reduce(bi, async_op=True)
        return hook

    def _launch_reduce(self, bi, async_op=False):
        """reduce-scatter(sum) of bucket bi's flat gradient -> (work or None, this rank's shard)."""
        fg, (n, padded, shard) = self.flat_g[bi], self.layout[bi]
        out = torch.empty(shard, dtype=fg.dtype, device=fg.device)
        if self.tensor_collectives:
            work = dist.reduce_scatter_tensor(out, fg, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
        else:
            work = dist.all_reduce(fg, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
        return (work if async_op else None), out

    # ---- optimizer-like surface ------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False):
        """Gradients live in the flat buffers: they are zeroed in place (set_to_none would detach the views)."""
        for fg in self.flat_g:
            fg.zero_()

    def state_bytes_per_rank(self) -> int:
        return sum(m.numel() * m.element_size() + v.numel() * v.element_size() for m, v in zip(self.shard_m, self.shard_v))

    def state_dict(self) -> dict:
        """This rank's optimizer shard (the moments exist only for the local 1/N slice of every bucket) plus the step and the
        bucket layout it belongs to.  Save one file per rank; `load_state_dict` refuses a shard of a different layout."""
        return dict(step=self._step, world=self.world, rank=self.rank, layout=[tuple(l) for l in self.layout],
                    weight_decay=list(self.bucket_wd), exp_avg=[m.clone() for m in self.shard_m],
                    exp_avg_sq=[v.clone() for v in self.shard_v])

    def load_state_dict(self, sd: dict):
        if sd["world"] != self.world or sd["rank"] != self.rank or [tuple(l) for l in sd["layout"]] != [tuple(l) for l in self.layout]:
            raise ValueError(f"optimizer shard of rank {sd['rank']}/{sd['world']} with layout {sd['layout']} does not match this "
                             f"run (rank {self.rank}/{self.world}, layout {self.layout}): same world size, parameters and "
                             "bucket_mb are required")
        self._step = int(sd["step"])
        for dst, src in zip(self.shard_m, sd["exp_avg"]):
            dst.copy_(src)
        for dst, src in zip(self.shard_v, sd["exp_avg_sq"]):
            dst.copy_(src)

    @torch.no_grad()
    def step(self):
        self._step += 1
        W, r = self.world, self.rank
        gshards = []
        for bi, (fg, (n, padded, shard)) in enumerate(zip(self.flat_g, self.layout)):
            if W > 1:
                pend, self._pending[bi] = self._pending[bi], None
                work, out = pend if pend is not None else self._launch_reduce(bi)
                if work is not None:
                    work.wait()  # orders the compute stream behind the collective; no host block on RCCL
                if not self.tensor_collectives:
                    out.copy_(fg[r * shard:(r + 1) * shard])
                out.div_(W)  # mean, as DDP (ReduceOp.AVG is not available on every backend)
            else:
                out = fg[:shard]
            gshards.append(out)
        self._left = [len(b) for b in self.buckets]
        coef = None
        if self.max_grad_norm is not None:
            sq = torch.stack([self.sumsq_fn(g).reshape(()) for g in gshards]).sum()
            if W > 1:
                dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.pg)
            norm = sq.sqrt()
            self.last_grad_norm = norm
            coef = torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0).reshape(1).to(torch.float32)
        b1, b2 = self.betas
        for fp, g, m, v, wd, (n, padded, shard) in zip(self.flat_p, gshards, self.shard_m, self.shard_v, self.bucket_wd, self.layout):
            pshard = fp[r * shard:(r + 1) * shard]
            self.update_fn(pshard, g, m, v, self.lr, b1, b2, self.eps, wd, self._step, coef)
            if W > 1:
                if self.tensor_collectives:
                    dist.all_gather_into_tensor(fp, pshard.clone(), group=self.pg)
                else:
                    parts = [torch.empty_like(pshard) for _ in range(W)]
                    dist.all_gather(parts, pshard.clone(), group=self.pg)
                    fp.copy_(torch.cat(parts))


def decay_param_groups(model: torch.nn.Module, weight_decay: float):
    """Param groups as the reference trainer builds them (omni/train/trainer.py:388-411, HF `get_parameter_names`): weights of
    `ALL_LAYERNORM_LAYERS` modules (DreamLLMRMSNorm registers itself there, modeling_dreamllm.py:94) and biases get no decay."""
    from transformers.pytorch_utils import ALL_LAYERNORM_LAYERS
    norm_types = tuple(ALL_LAYERNORM_LAYERS)
    no_decay = set()
    for mn, mod in model.named_modules():
        for pn, _ in mod.named_parameters(recurse=False):
            full = f"{mn}.{pn}" if mn else pn
            if isinstance(mod, norm_types) or pn.endswith("bias"):
                no_decay.add(full)
    dec = [p for n, p in model.named_parameters() if p.requires_grad and n not in no_decay]
    nod = [p for n, p in model.named_parameters() if p.requires_grad and n in no_decay]
    return [{"params": dec, "weight_decay": weight_decay}, {"params": nod, "weight_decay": 0.0}]


def save_dreamllm_full_state_dict(model, output_dir: str, rank: int | None = None):
    """Plugin-aware full-state-dict save: the counterpart of `save_dreamllm_fsdp_full_state_dict`
    (omni/utils/fsdp_utils.py:23-61, called by omni/train/dreamllm_trainer.py:54) for the sharded-gradient mode.  Parameters are
    whole on every rank here, so nothing has to be gathered: rank 0 writes `pytorch_model.bin` (the full state dict, plugin
    keys included, exactly like the reference) and one `{plugin.save_model_name}.bin` per plugin with the
    `model.{name}.` / `{name}.` prefix stripped -- the files `PluginBase.load_model` reads back."""
    import os as _os
    from collections import OrderedDict
    rank = get_rank() if rank is None else rank
    inner = model.module if hasattr(model, "module") else model
    state_dict = inner.state_dict()
    prefixes = {}
    for plugin_name, ptype in inner.config.plugins_type.items():
        if ptype == "embedding":
            plugin, prefix = getattr(inner.get_decoder(), plugin_name), f"model.{plugin_name}."
        elif ptype == "head":
            plugin, prefix = getattr(inner, plugin_name), f"{plugin_name}."
        else:
            continue
        prefixes[plugin.save_model_name] = prefix
    if rank == 0:
        _os.makedirs(output_dir, exist_ok=True)
        # `torch.save` serialises the WHOLE storage behind a view: parameters that live in ShardedGradAdamW's flat 512 MB buckets
        # (or in a packed q|k|v buffer) would each drag their bucket into the file (a 0.5 MB dream_embedding.bin of 512 MB).
        # Write compact CPU copies instead.
        state_dict = OrderedDict((k, v.detach().cpu() if v.is_cuda else v.detach().clone()) for k, v in state_dict.items())
        torch.save(state_dict, _os.path.join(output_dir, "pytorch_model.bin"))
        for name, prefix in prefixes.items():
            sd = OrderedDict((k[len(prefix):], v) for k, v in state_dict.items() if k.startswith(prefix))
            torch.save(sd, _os.path.join(output_dir, f"{name}.bin"))
    synchronize()
    return sorted(prefixes)
