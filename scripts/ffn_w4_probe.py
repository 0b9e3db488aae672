"""Per-launch time of the fused feed-forward launch in its two forms (eight waves, one workgroup per CU: ASTTS_TFM_FFN_W4=0;
four waves, two workgroups per CU: =1) at 16 / 32 / 64 / 128 sequences x 344 frames."""
import os, sys
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops
dev = 'cuda'
c, hidden, k0 = 256, 1024, 512
p1 = ops.PackedWeight(torch.randn(hidden, c) / 16, torch.randn(hidden) * 0.1)
p2 = ops.PackedWeight(torch.randn(c, hidden) / 32, torch.randn(c) * 0.1)
po = ops.PackedWeight(torch.randn(c, k0) / 24, torch.randn(c) * 0.1)
f1, f2, fo = ops.tfm_pack_frag(p1), ops.tfm_pack_frag(p2), ops.tfm_pack_frag(po)


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for seqs in (16, 32, 64, 128):
    m = seqs * 344
    x = torch.randn(m, c, device=dev); at = torch.randn(m, k0, device=dev).half()
    fl = 4.0 * m * c * hidden + 2.0 * m * c * k0
    res = []
    for form in ("0", "1"):
        os.environ["ASTTS_TFM_FFN_W4"] = form
        t = timed(lambda: ops.tfm_ffn_fused(x, p1, f1, p2, f2, attn=at, wo=po, wo_frag=fo))
        res.append(f"form {form}: {t:.1f} us = {fl / t * 1e-6:.0f} TFLOP/s ({fl / t * 1e-6 / 2500:.3f} of MFMA)")
    print(f"{seqs} sequences ({m} rows): " + "; ".join(res), flush=True)
