"""`-m gpu`: the real (tiny) DreamLLM model through `distributed.wrap_ddp` with 2 ranks sharing the one GPU of the test box
(gloo transports CUDA tensors; RCCL needs one device per rank, which the 8-GPU scaling run provides).  Checks what the N>1
bench depends on: DDP's bucket hooks fire for every trainable parameter of the custom autograd Functions under
static_graph, gradients are averaged and identical on both ranks, and two optimizer steps keep the replicas in sync."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, mode="ddp", backend="gloo"):
    local = rank if backend == "nccl" else 0  # RCCL: one device per rank; gloo: both ranks share the box's single GPU
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(local), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from dreamllm_amd import distributed as D
    from dreamllm_amd.factory import TINY, TINY_CLIP, TINY_DIFFUSION, build_dreamllm
    from dreamllm_amd.optim import HipAdamW
    from dreamllm_amd.synthetic import make_interleaved_batch
    torch.cuda.set_device(local)
    if backend == "nccl":
        assert D.init_distributed("nccl") == world and dist.get_backend() == "nccl"   # the path bench.py takes for N > 1
        assert D.max_over_ranks(1.0 + rank) == float(world)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    model = build_dreamllm(TINY, device=dev, seed=0, clip=dict(TINY_CLIP, num_hidden_layers=2), diffusion=TINY_DIFFUSION,
                           num_dream_queries=8).train()
    tl = None
    if mode == "ddp":
        tl = D.BucketTimeline()          # the comm hook bench.py installs for N > 1: same all-reduce(mean) + per-bucket ready / done stamps
        ddp = D.wrap_ddp(model, bucket_cap_mb=1, timeline=tl)
        assert ddp is not model
        opt = HipAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0)
    else:  # sharded-gradient mode: plain replica forward/backward, reduce-scatter + sharded AdamW + all-gather in the step
        ddp = model
        opt = D.ShardedGradAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0, bucket_mb=1,
                                 overlap=(mode == "sharded_overlap"))
        assert opt.tensor_collectives == (backend == "nccl")
    batch = make_interleaved_batch(2, 256, 1, n_dream=8, n_patch=16, seed=100 + rank, device=dev, image_size=56, dm_size=128)
    torch.manual_seed(5)  # same diffusion noise / timesteps on both ranks is not required; losses differ per rank by data
    losses, gsig = [], None
    for it in range(2):
        out = ddp(**batch, return_dict=True)
        out.loss.backward()
        if it == 0:
            missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
            gsig = torch.stack([p.grad.float().abs().sum() for p in model.parameters() if p.requires_grad]).cpu()
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(float(out.loss.detach()))
    if tl is not None:
        summ = tl.summary()
        assert summ is not None and summ["steps"] == 2 and summ["buckets_per_step"] >= 1, summ   # (the tiny model fits one bucket)
        assert summ["bucket_order"] == list(range(summ["buckets_per_step"])) and summ["comm_exposed_ms"] >= 0.0, summ
    psig = torch.stack([p.detach().float().sum() for p in model.parameters() if p.requires_grad]).cpu()
    if mode != "ddp":  # also ship the parameters themselves for the cross-mode comparison
        flat = torch.cat([p.detach().float().flatten() for p in model.parameters() if p.requires_grad]).cpu()
        q.put((rank, missing, gsig.tolist(), psig.tolist(), losses, flat.tolist(), opt.state_bytes_per_rank()))
    else:
        flat = torch.cat([p.detach().float().flatten() for p in model.parameters() if p.requires_grad]).cpu()
        q.put((rank, missing, gsig.tolist(), psig.tolist(), losses, flat.tolist(), 0))
    dist.destroy_process_group()


def _collect(q, procs, timeout=300):
    """One result per worker; a worker that dies (an assert inside it) fails the test at once instead of after the queue timeout."""
    import queue
    import time
    out, t0 = [], time.time()
    while len(out) < len(procs):
        try:
            out.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f"worker exited with {dead}"
            assert time.time() - t0 < timeout, "workers timed out"
    return out


def test_ddp_two_ranks_tiny_model():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(q, procs), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, miss0, g0, p0, l0, _, _), (_, miss1, g1, p1, l1, _, _) = res
    assert miss0 == [] and miss1 == []
    assert g0 == g1   # all-reduced gradients identical
    assert p0 == p1   # replicas stay bit-identical after 2 steps (deterministic clip norm)
    assert l0 != l1  # different data shards


def _run(mode, backend="gloo"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(q, procs), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_sharded_grad_mode_matches_ddp():
    """`distributed.ShardedGradAdamW` (the reference's FSDP `shard_grad_op` recipe) on the tiny DreamLLM
    model, 2 ranks: replicas end bit-identical, the AdamW moments take half the memory per rank, and two steps land on the
    same parameters as DDP + HipAdamW (different reduction order => bf16-ulp differences only)."""
    sh = _run("sharded")
    dd = _run("ddp")
    assert sh[0][1] == [] and sh[1][1] == []
    assert sh[0][5] == sh[1][5]                       # replicas identical
    a, b = torch.tensor(sh[0][5]), torch.tensor(dd[0][5])
    assert ((a - b).norm() / b.norm()) < 2e-3         # same trajectory as DDP
    n_train = a.numel()
    assert sh[0][6] <= 2 * 2 * (n_train // 2 + 64)    # two bf16 moments over half the parameters (+ padding per bucket)
    assert abs(sh[0][4][0] - dd[0][4][0]) < 1e-3 * abs(dd[0][4][0])  # first-step loss identical up to rounding


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: runs on multi-GPU nodes only")
def test_rccl_two_ranks_ddp_and_sharded_grad():
    """The multi-GPU path of bench.py executed on RCCL (backend "nccl") with 2 ranks on 2 devices: `init_distributed` ->
    `wrap_ddp` -> 2 steps, and `ShardedGradAdamW` (reduce_scatter_tensor / all_gather_into_tensor, with and without the
    backward overlap) -> 2 steps; replicas identical, the three modes agree with each other."""
    dd = _run("ddp", "nccl")
    sh = _run("sharded", "nccl")
    so = _run("sharded_overlap", "nccl")
    for res in (dd, sh, so):
        assert res[0][1] == [] and res[1][1] == []
        assert res[0][5] == res[1][5]                 # replicas identical after 2 steps
        assert res[0][4] != res[1][4]                 # different data shards
    a, b, c = (torch.tensor(r[0][5]) for r in (dd, sh, so))
    assert ((b - a).norm() / a.norm()) < 2e-3 and ((c - a).norm() / a.norm()) < 2e-3
    assert torch.equal(b, c)                          # overlap changes when the collectives run, not what they compute
