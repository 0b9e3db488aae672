"""Per-launch time of astts_op_tfm_ffn_fused at the benchmark's row counts, next to the three launches it replaces."""
import sys
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops
dev = 'cuda'
c, hidden = 256, 1024
p1 = ops.PackedWeight(torch.randn(hidden, c) / 16, torch.randn(hidden) * 0.1)
p2 = ops.PackedWeight(torch.randn(c, hidden) / 32, torch.randn(c) * 0.1)
f1, f2 = ops.tfm_pack_frag(p1), ops.tfm_pack_frag(p2)
ident = (torch.ones(c, device=dev), torch.zeros(c, device=dev))


def timed(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for m in (5504, 11008):
    x = torch.randn(m, c, device=dev)

    def unfused():
        n16 = ops.layernorm(x, *ident, 1e-5, out_dtype=torch.float16)
        f16 = ops.linear(n16, p1, act="gelu", out_dtype=torch.float16)
        return ops.linear(f16, p2, residual=x)
    tf, tu = timed(lambda: ops.tfm_ffn_fused(x, p1, f1, p2, f2)), timed(unfused)
    print(f'm={m}: fused {tf:.2f} us ({4 * m * c * hidden / tf * 1e-6:.0f} TFLOP/s), three launches {tu:.2f} us')

po = ops.PackedWeight(torch.randn(c, 512) / 24, torch.randn(c) * 0.1)
fo = ops.tfm_pack_frag(po)
for m in (5504, 11008, 44032):
    x = torch.randn(m, c, device=dev); at = torch.randn(m, 512, device=dev).half()
    t1 = timed(lambda: ops.tfm_ffn_fused(x, p1, f1, p2, f2, attn=at, wo=po, wo_frag=fo))
    t2 = timed(lambda: ops.tfm_ffn_fused(ops.linear(at, po, residual=x), p1, f1, p2, f2))
    print(f'm={m}: with the output projection inside {t1:.2f} us, as two launches {t2:.2f} us')
