import sys, math, ctypes
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops, _lib
dev = 'cuda'
b, t, c, heads = 16, 344, 256, 8
x = torch.randn(b, t, c, device=dev)
pw = ops.PackedWeight(torch.randn(1536, c) / 16, torch.randn(1536) * 0.1)
out = torch.empty(b, t, 512, dtype=torch.float16, device=dev)
L = _lib.load()
wfrag = ops.tfm_pack_frag(pw)
for sc, what in ((0.125, 'whole kernel'),):
    for _ in range(50):
        _lib.check(L.astts_op_tfm_attn_fused(x.data_ptr(), wfrag.data_ptr(), pw.bias.data_ptr(), None, out.data_ptr(), b, heads, t, c, 1e-5, sc, _lib.stream_ptr()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        L.astts_op_tfm_attn_fused(x.data_ptr(), wfrag.data_ptr(), pw.bias.data_ptr(), None, out.data_ptr(), b, heads, t, c, 1e-5, sc, _lib.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    print(f'{what}: {e0.elapsed_time(e1) * 1e3 / 300:.2f} us per launch (incl. boundary)')
