"""Conv launches of one SD-2.1 UNet forward at a given batch (shape, split-K, eligibility for the LDS-DMA kernels): python tools/conv_shapes.py [batch]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dreamllm_amd import ops, _lib
from dreamllm_amd.modeling_plugins import StableDiffusionHead
log = collections.Counter()
orig = ops.check
def check(name, *a):
    if name == "dllm_conv2d_nhwc_bf16_splitk":
        N, H, W, C, OH, OW, CO, KH, KW, stride, pad, up2, even = a[6:19]
        sk = a[21]
        log[(N, H, W, C, CO, KH, stride, up2, even, sk)] += 1
    return orig(name, *a)
ops.check = check
torch.manual_seed(0)
head = StableDiffusionHead("sd21-base", embed_hidden_size=4096).to("cuda", torch.bfloat16).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.randn(B, 64, 64, 8, device="cuda").to(torch.bfloat16)
ctx = torch.randn(B, 64, 1024, device="cuda").to(torch.bfloat16)
t = torch.full((B,), 500, device="cuda")
with torch.no_grad():
    head.unet(x, t, ctx, nhwc_io=True)
for k, v in sorted(log.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2] * kv[0][3] * kv[0][4] * kv[0][5] ** 2):
    N, H, W, C, CO, KH, stride, up2, even, sk = k
    M = N * (H * (2 if up2 else 1) // stride) * (W * (2 if up2 else 1) // stride)
    print(f"x{v:2d} N{N} {H}x{W} C{C}->CO{CO} k{KH} s{stride} up{up2} ev{even} sk{sk}  M={M} tiles128={-(-M//128)*-(-CO//128)} C%64={C%64}")
