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
