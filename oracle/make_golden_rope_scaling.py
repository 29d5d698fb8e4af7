"""TEST INFRASTRUCTURE -- golden cos/sin tables of the reference's RoPE scaling variants, from EXECUTING
`LinearScalingRotaryEmbedding` / `DynamicNTKScalingRotaryEmbedding` / `RotaryEmbedding`
(omni/models/dreamllm/modeling_dreamllm.py:97-173) in the call order that exposes their caching behaviour (a table rebuilt for a
longer sequence stays in use for shorter ones).  python -m oracle.make_golden_rope_scaling -> tests/golden/rope_scaling.pt"""
import os

import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rope_scaling.pt")
DIM, MAXPOS, BASE = 16, 48, 10000.0
CALLS = [24, 48, 100, 40, 150, 100]     # below / at / above max_position_embeddings, then shorter again, longer, shorter


def main():
    from oracle import ref_loader
    m = ref_loader.load_modeling()
    x = torch.zeros(1, 1, 1, DIM)
    out = dict(dim=DIM, max_position_embeddings=MAXPOS, base=BASE, calls=CALLS, cases=[])
    for typ, cls, kw in (("none", m.RotaryEmbedding, {}), ("linear", m.LinearScalingRotaryEmbedding, dict(scaling_factor=2.0)),
                         ("linear", m.LinearScalingRotaryEmbedding, dict(scaling_factor=4.0)),
                         ("dynamic", m.DynamicNTKScalingRotaryEmbedding, dict(scaling_factor=2.0)),
                         ("dynamic", m.DynamicNTKScalingRotaryEmbedding, dict(scaling_factor=3.5))):
        rot = cls(DIM, max_position_embeddings=MAXPOS, base=BASE, **kw)
        steps = []
        for n in CALLS:
            cos, sin = rot(x, seq_len=n)
            steps.append(dict(seq_len=n, cos=cos.clone(), sin=sin.clone()))
        out["cases"].append(dict(type=typ, factor=kw.get("scaling_factor", 1.0), steps=steps))
    torch.save(out, OUT)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
