import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dreamllm_amd.factory import VICUNA_7B, build_dreamllm
from dreamllm_amd.optim import HipAdamW
from dreamllm_amd.synthetic import make_interleaved_batch
dev = torch.device("cuda", 0)
model = build_dreamllm(VICUNA_7B, device=dev).train()
opt = HipAdamW([p for p in model.parameters() if p.requires_grad], lr=2e-5, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0)
batch = make_interleaved_batch(16, 2048, 2, seed=1234, device=dev)
for i in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model(**batch, return_dict=True); out.loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print(f"step {i}: {dt*1e3:8.1f} ms  reserved {torch.cuda.memory_reserved()/2**30:6.1f} GiB  alloc_retries {st.get('num_alloc_retries',0)}  segments {st.get('segment.all.current',0)}", flush=True)
