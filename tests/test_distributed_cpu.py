"""CPU, world_size 2 over gloo: the data-parallel runtime logic of dreamllm_amd/distributed.py (rank bookkeeping, shard
partition, DDP wrap with static buckets -> averaged gradients identical on both ranks, max-over-ranks timing, dict reduce).
The HIP model itself cannot run on CPU (no fallback); the wrapper is exercised with a small torch module, which is what DDP
sees anyway: parameters that receive gradients."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dreamllm_amd import distributed as D
    assert D.init_distributed("gloo") == world and D.get_rank() == rank and D.is_main_process() == (rank == 0)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))
    ddp = D.wrap_ddp(model, bucket_cap_mb=1)
    shard = D.shard_for_rank(10)
    data = torch.arange(80, dtype=torch.float32).view(10, 8) / 80.0
    x = data[list(shard)]
    for _ in range(2):  # static_graph: second iteration reuses the bucket order
        ddp.zero_grad()
        ddp(x).sum().backward()
    g = torch.cat([p.grad.flatten() for p in model.parameters()])
    D.synchronize()
    t = D.max_over_ranks(1.0 + rank)
    red = D.reduce_dict({"a": torch.tensor(float(rank)), "b": torch.tensor(2.0)})
    q.put((rank, list(shard), g.tolist(), t, float(red["a"]), float(red["b"])))
    torch.distributed.destroy_process_group()


def test_ddp_gloo_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, s0, g0, t0, a0, b0), (r1, s1, g1, t1, a1, b1) = res
    g0, g1 = torch.tensor(g0), torch.tensor(g1)
    assert s0 == [0, 1, 2, 3, 4] and s1 == [5, 6, 7, 8, 9]
    assert torch.allclose(g0, g1)  # all-reduced (averaged) gradients agree
    # equals the mean of the two per-rank gradients computed serially
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))
    data = torch.arange(80, dtype=torch.float32).view(10, 8) / 80.0
    gs = []
    for sl in (slice(0, 5), slice(5, 10)):
        model.zero_grad()
        model(data[sl]).sum().backward()
        gs.append(torch.cat([p.grad.flatten() for p in model.parameters()]))
    assert torch.allclose(g0, (gs[0] + gs[1]) / 2, atol=1e-6)
    assert t0 == t1 == 2.0 and a0 == a1 == 0.5 and b0 == b1 == 2.0


def test_shard_partition_is_disjoint_and_complete():
    from dreamllm_amd.distributed import shard_for_rank
    for n, w in ((128, 8), (10, 4), (3, 8)):
        seen = [i for r in range(w) for i in shard_for_rank(n, r, w)]
        assert sorted(seen) == list(range(n))


# ------------------------------------------------------------------------------------------ sharded-gradient mode (ZeRO-2)
def _torch_adamw(p, g, m, v, lr, b1, b2, eps, wd, step, coef):
    """torch restatement of the fused AdamW kernel (decoupled weight decay, bias correction, optional clip coefficient)."""
    g = g.float() * (coef.float() if coef is not None else 1.0)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    mh = m / (1 - b1 ** step)
    vh = v / (1 - b2 ** step)
    p.mul_(1 - lr * wd).sub_(lr * mh / (vh.sqrt() + eps))


def _sumsq(g):
    return (g.float() ** 2).sum()


def _sharded_worker(rank, world, port, q, overlap=False):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dreamllm_amd import distributed as D
    D.init_distributed("gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 13), torch.nn.Tanh(), torch.nn.Linear(13, 3))  # odd sizes: padding path
    opt = D.ShardedGradAdamW(model.parameters(), lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1, max_grad_norm=0.5,
                             bucket_mb=1, update_fn=_torch_adamw, sumsq_fn=_sumsq, overlap=overlap)
    data = torch.arange(80, dtype=torch.float32).view(10, 8) / 80.0
    x = data[list(D.shard_for_rank(10))]
    norms = []
    for _ in range(3):
        opt.zero_grad()
        model(x).pow(2).mean().backward()
        opt.step()
        norms.append(float(opt.last_grad_norm))
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    q.put((rank, flat.tolist(), norms, opt.state_bytes_per_rank()))
    torch.distributed.destroy_process_group()


import pytest


@pytest.mark.parametrize("overlap", [False, True])
def test_sharded_grad_adamw_matches_single_process_adamw(overlap):
    """(overlap=True: the reduction of a bucket is launched from the post-accumulate-grad hook of its last gradient.)
    2 ranks, gloo: reduce-scatter(mean) -> global-norm clip -> AdamW on the local slice -> all-gather reproduces a
    single-process AdamW step on the mean gradient; both ranks end with identical parameters; moments are half-size."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, f0, n0, sb0), (_, f1, n1, sb1) = res
    assert f0 == f1 and n0 == n1  # replicas bit-identical after 3 steps
    # reference: one process, mean of the two shard losses, torch AdamW semantics with clip_grad_norm_
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 13), torch.nn.Tanh(), torch.nn.Linear(13, 3))
    params = list(model.parameters())
    ms = [torch.zeros_like(p) for p in params]
    vs = [torch.zeros_like(p) for p in params]
    data = torch.arange(80, dtype=torch.float32).view(10, 8) / 80.0
    norms = []
    for step in range(1, 4):
        model.zero_grad()
        (0.5 * (model(data[:5]).pow(2).mean() + model(data[5:]).pow(2).mean())).backward()
        norm = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in params))
        norms.append(float(norm))
        coef = torch.clamp(0.5 / (norm + 1e-6), max=1.0)
        with torch.no_grad():
            for p, m, v in zip(params, ms, vs):
                _torch_adamw(p, p.grad, m, v, 1e-2, 0.9, 0.95, 1e-8, 0.1, step, coef)
    ref = torch.cat([p.detach().flatten() for p in params])
    assert torch.allclose(torch.tensor(f0), ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(torch.tensor(n0), torch.tensor(norms), rtol=1e-5)
    n_params = sum(p.numel() for p in params)
    assert sb0 == sb1 and sb0 <= 2 * 4 * (n_params // 2 + 4)  # two fp32 moments over half the parameters (+ padding)


def _resume_worker(rank, world, port, q):
    """3 steps straight vs 2 steps -> state_dict -> fresh optimizer + load_state_dict -> 1 step (param groups: no decay on biases)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dreamllm_amd import distributed as D
    D.init_distributed("gloo")
    data = torch.arange(80, dtype=torch.float32).view(10, 8) / 80.0
    x = data[list(D.shard_for_rank(10))]

    def make():
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(8, 13), torch.nn.Tanh(), torch.nn.Linear(13, 3))
        groups = [{"params": [p for n, p in model.named_parameters() if n.endswith("weight")], "weight_decay": 0.1},
                  {"params": [p for n, p in model.named_parameters() if n.endswith("bias")], "weight_decay": 0.0}]
        opt = D.ShardedGradAdamW(groups, lr=1e-2, betas=(0.9, 0.95), max_grad_norm=0.5, bucket_mb=1, update_fn=_torch_adamw,
                                 sumsq_fn=_sumsq)
        return model, opt

    def run(model, opt, n):
        for _ in range(n):
            opt.zero_grad()
            model(x).pow(2).mean().backward()
            opt.step()

    m1, o1 = make()
    run(m1, o1, 3)
    m2, o2 = make()
    run(m2, o2, 2)
    sd, weights = o2.state_dict(), [p.detach().clone() for p in m2.parameters()]
    m3, o3 = make()
    with torch.no_grad():
        for p, w in zip(m3.parameters(), weights):
            p.copy_(w)  # parameters come from the model checkpoint; they stay views of the flat buffers
    o3.load_state_dict(sd)
    run(m3, o3, 1)
    a = torch.cat([p.detach().flatten() for p in m1.parameters()])
    b = torch.cat([p.detach().flatten() for p in m3.parameters()])
    bad = None
    try:
        o3.load_state_dict(dict(sd, world=4))
    except ValueError as e:
        bad = str(e)
    q.put((rank, a.tolist(), b.tolist(), sorted(set(o1.bucket_wd)), bad is not None, sd["step"]))
    torch.distributed.destroy_process_group()


def test_sharded_grad_adamw_checkpoint_resume_and_param_groups():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_resume_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for _, a, b, wds, refused, step in res:
        assert a == b, "resumed run must reproduce the uninterrupted one bit for bit (step + moments round-trip)"
        assert wds == [0.0, 0.1] and refused and step == 2


def test_decay_param_groups_and_plugin_aware_save(tmp_path):
    """`decay_param_groups`: RMSNorm weights / biases out of weight decay (omni/train/trainer.py:388-411);
    `save_dreamllm_full_state_dict`: pytorch_model.bin + one {save_model_name}.bin per plugin with the prefix stripped
    (omni/utils/fsdp_utils.py:23-61)."""
    from dreamllm_amd import distributed as D
    from dreamllm_amd.configuration_dreamllm import DreamLLMConfig
    from dreamllm_amd.modeling_dreamllm import DreamLLMForCausalMLM
    from dreamllm_amd.modeling_plugins import DreamEmbedding
    cfg = DreamLLMConfig(vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                         max_position_embeddings=32)
    lm = DreamLLMForCausalMLM(cfg)
    lm.model.dream_embedding = DreamEmbedding(num_dream_queries=4, embed_hidden_size=128)
    lm.config.plugins_type["dream_embedding"] = "embedding"
    groups = D.decay_param_groups(lm, 0.05)
    names = {id(p): n for n, p in lm.named_parameters()}
    nod = {names[id(p)] for p in groups[1]["params"]}
    assert groups[0]["weight_decay"] == 0.05 and groups[1]["weight_decay"] == 0.0
    assert nod == {n for n in names.values() if "layernorm" in n or n == "model.norm.weight"}
    saved = D.save_dreamllm_full_state_dict(lm, str(tmp_path), rank=0)
    assert saved == ["dream_embedding"]
    full = torch.load(tmp_path / "pytorch_model.bin")
    plug = torch.load(tmp_path / "dream_embedding.bin")
    assert "model.dream_embedding.dream_queries" in full and list(plug) == ["dream_queries"]
    assert torch.equal(plug["dream_queries"], lm.model.dream_embedding.dream_queries.data)


def test_full_state_save_is_compact_for_flat_owned_parameters(tmp_path):
    """plugin tensors that are views into ShardedGradAdamW's flat bucket buffers must be saved as compact copies --
    `torch.save` of a view serialises the whole underlying storage (a 2 KB dream_embedding.bin would carry the bucket)."""
    from dreamllm_amd import distributed as D
    from dreamllm_amd.configuration_dreamllm import DreamLLMConfig
    from dreamllm_amd.modeling_dreamllm import DreamLLMForCausalMLM
    from dreamllm_amd.modeling_plugins import DreamEmbedding
    cfg = DreamLLMConfig(vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                         max_position_embeddings=32)
    lm = DreamLLMForCausalMLM(cfg)
    lm.model.dream_embedding = DreamEmbedding(num_dream_queries=4, embed_hidden_size=128)
    lm.config.plugins_type["dream_embedding"] = "embedding"
    D.ShardedGradAdamW(lm.parameters(), update_fn=lambda *a: None, sumsq_fn=lambda g: g.float().pow(2).sum())
    dq = lm.model.dream_embedding.dream_queries
    assert dq.untyped_storage().nbytes() > 100 * dq.numel() * dq.element_size()   # it IS a view of a big flat buffer now
    D.save_dreamllm_full_state_dict(lm, str(tmp_path), rank=0)
    assert os.path.getsize(tmp_path / "dream_embedding.bin") < 4 * dq.numel() * dq.element_size() + 4096
    n_bytes = sum(p.numel() * p.element_size() for p in lm.state_dict().values())
    assert os.path.getsize(tmp_path / "pytorch_model.bin") < 1.1 * n_bytes + 65536
    assert torch.equal(torch.load(tmp_path / "dream_embedding.bin")["dream_queries"], dq.data)


def test_sharded_grad_buckets_keep_atomic_groups_together():
    """a bucket cut must not split q|k|v / gate|up (they are applied as ONE packed GEMM).  With a bucket size that
    would otherwise cut inside the groups, every group still lands in one bucket, adjacent and in registration order."""
    from dreamllm_amd import distributed as D
    from dreamllm_amd.configuration_dreamllm import DreamLLMConfig
    from dreamllm_amd.modeling_dreamllm import DreamLLMForCausalMLM, _packed_view, packed_parameter_groups
    cfg = DreamLLMConfig(vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                         max_position_embeddings=32)
    lm = DreamLLMForCausalMLM(cfg)
    groups = packed_parameter_groups(lm)
    assert len(groups) == 6 and [len(g) for g in groups] == [3, 2] * 3
    kw = dict(update_fn=lambda *a: None, sumsq_fn=lambda g: g.float().pow(2).sum())
    # 128x128 fp32 = 64 KiB per projection: 0.15 MiB buckets cut after every 2nd projection without the hint
    opt = D.ShardedGradAdamW(lm.parameters(), bucket_mb=0.15, atomic_groups=groups, **kw)
    where = {id(p): bi for bi, b in enumerate(opt.buckets) for p in b}
    for g in groups:
        assert len({where[id(p)] for p in g}) == 1
        assert _packed_view(*[p.data for p in g]) is not None      # adjacent row blocks of the flat buffer, in order
    assert len(opt.buckets) > 6
    lm2 = DreamLLMForCausalMLM(cfg)
    opt2 = D.ShardedGradAdamW(lm2.parameters(), bucket_mb=0.15, **kw)
    where2 = {id(p): bi for bi, b in enumerate(opt2.buckets) for p in b}
    assert any(len({where2[id(p)] for p in g}) > 1 for g in packed_parameter_groups(lm2))   # the hint is what keeps them whole


class _WholeLayerFn(torch.autograd.Function):
    """Stand-in with the autograd SHAPE of `modeling_dreamllm._DecoderLayerFn` (which needs the GPU): ONE Function per layer that
    takes every weight of the layer as an input and returns all of their gradients at once from a hand-written backward."""

    @staticmethod
    def forward(ctx, x, w1, w2, g):
        h = torch.tanh(x @ w1.t())
        ctx.save_for_backward(x, w1, w2, g, h)
        return x + (h @ w2.t()) * g

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, g, h = ctx.saved_tensors
        u = h @ w2.t()
        dg = (dy * u).sum(0)
        du = dy * g
        dw2 = du.t() @ h
        dh = (du @ w2) * (1 - h * h)
        return dy + dh @ w1, dh.t() @ x, dw2, dg


class _Layer(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.w1 = torch.nn.Parameter(torch.randn(2 * d, d) * 0.1)
        self.w2 = torch.nn.Parameter(torch.randn(d, 2 * d) * 0.1)
        self.g = torch.nn.Parameter(torch.ones(d))

    def forward(self, x):
        return _WholeLayerFn.apply(x, self.w1, self.w2, self.g)


def _timeline_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dreamllm_amd import distributed as D
    D.init_distributed("gloo")
    torch.manual_seed(0)
    d, L = 128, 4
    model = torch.nn.Sequential(*[_Layer(d) for _ in range(L)])   # 2 * 2d*d * 4 B = 256 KiB per layer
    ref = [p.detach().clone() for p in model.parameters()]
    tl = D.BucketTimeline()
    ddp = D.wrap_ddp(model, bucket_cap_mb=0.3, timeline=tl)         # ~one layer per bucket
    torch.manual_seed(10 + rank)
    x = torch.randn(8, d)
    for it in range(3):                                             # static_graph: buckets are rebuilt after iteration 1
        if it == 2:
            tl.reset()
        ddp.zero_grad()
        ddp(x).square().mean().backward()
    s = tl.summary()
    ptr2layer = {int(p.data_ptr()): int(n.split(".")[0]) for n, p in model.named_parameters()}
    layers_per_bucket = [sorted({ptr2layer[pp] for pp in ptrs}) for ptrs in s["bucket_param_ptrs"]]
    g = torch.cat([p.grad.flatten() for p in model.parameters()])
    # single-process reference of the averaged gradient
    xs = []
    for r in range(world):
        torch.manual_seed(10 + r)
        xs.append(torch.randn(8, d))
    m2 = torch.nn.Sequential(*[_Layer(d) for _ in range(L)])
    for p, v in zip(m2.parameters(), ref):
        p.data.copy_(v)
    for xr in xs:
        (m2(xr).square().mean() / world).backward()
    g2 = torch.cat([p.grad.flatten() for p in m2.parameters()])
    q.put((rank, s["bucket_order"], layers_per_bucket, s["comm_exposed_ms"], s["comm_busy_ms"], s["steps"],
           float((g - g2).abs().max()), float(g2.abs().max())))
    torch.distributed.destroy_process_group()


def test_ddp_bucket_timeline_and_reverse_layer_bucket_order():
    """(i) a whole-layer autograd Function (the shape of `_DecoderLayerFn`) fires DDP's bucket hooks under
    `static_graph` in REVERSE layer order -- the last layer's bucket is ready first, so its all-reduce runs under the backward of
    the earlier layers; (ii) `BucketTimeline` (the comm hook bench.py installs for N > 1) reports every bucket of a step with
    ready / done stamps, performs the same all-reduce(mean) as DDP's built-in reducer, and yields `comm_exposed_ms`."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, order, layers, exposed, busy, steps, err, scale in res:
        assert steps == 1 and order == list(range(len(order))) and len(order) >= 3
        firsts = [l[-1] for l in layers]
        assert firsts == sorted(firsts, reverse=True), layers          # bucket 0 holds the LAST layer, ... (reverse layer order)
        assert layers[0][-1] == 3 and layers[-1][0] == 0
        assert exposed >= 0.0 and busy >= exposed * 0.999
        assert err <= 1e-6 * max(scale, 1.0)                            # the hook's all-reduce(mean) == DDP's averaged gradient


def test_bench_self_launches_n_ranks_without_a_rendezvous_in_the_environment():
    """`python bench.py --gpus N` exactly as the driver types it (no WORLD_SIZE): bench.py must re-exec itself under
    torch.distributed.run with N ranks and rank 0 must print ONE JSON line.  Dry run of the launcher on gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "n_gpus": 2, "ranks": 2, "backend": "gloo"}
    # a measuring run refuses any backend but RCCL instead of silently timing gloo
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--backend", "gloo"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode != 0 and "RCCL" in (r.stderr + r.stdout)
