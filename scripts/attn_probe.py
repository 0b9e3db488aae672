"""attn_mha_flash alone: time per launch vs sequence length (fixed cost vs per-tile cost), back-to-back on one stream."""
import sys, os
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops
dev = 'cuda'
B, H = 16, 8
for T in (32, 64, 128, 192, 256, 344, 512, 688):
    qkv = torch.randn(B, T, 3 * H * 64, device=dev).half()
    hd = H * 64
    q, k, v = qkv[..., :hd], qkv[..., hd:2 * hd], qkv[..., 2 * hd:]
    for _ in range(20): o = ops.attn_mha(q, k, v, H, out_dtype=torch.float16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300
    e0.record()
    for _ in range(n): o = ops.attn_mha(q, k, v, H, out_dtype=torch.float16)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    fl = 4.0 * B * H * T * T * 64
    print(f'T={T:4d} blocks={((T + 127) // 128) * H * B:5d}: {us:7.2f} us per launch (incl. boundary)  {fl / us / 1e6:7.1f} TFLOP/s')
