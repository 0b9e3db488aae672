"""Per-launch time of ops.conv1d_snake at the vocoder's stage shapes, both resblock forms, 3 / 7 / 11 taps."""
import sys, math
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops
dev = 'cuda'


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for c, l in ((128, 27520), (256, 3440)):
    b = 8
    x = torch.randn(b, l, c, device=dev)
    h = torch.randn(b, l, c, device=dev).half()
    al = torch.rand(c, device=dev) + 0.5
    y32, y16 = torch.empty_like(x), torch.empty_like(h)
    L = ops._L()
    from astts import _lib
    for taps, dil in ((3, 1), (3, 5), (7, 1), (7, 5), (11, 1), (11, 5)):
        pw = ops.PackedWeight.from_conv1d(torch.randn(c, c, taps) / math.sqrt(c * taps), torch.zeros(c))
        wf = ops.conv_pack_frag(pw)
        st = _lib.stream_ptr()
        f1 = lambda: L.astts_op_conv1d_snake(x.data_ptr(), 0, al.data_ptr(), wf.data_ptr(), pw.bias.data_ptr(), None, y16.data_ptr(), 1, None, 1.0, 0, b, l, c, taps, dil, st)
        f2 = lambda: L.astts_op_conv1d_snake(h.data_ptr(), 1, al.data_ptr(), wf.data_ptr(), pw.bias.data_ptr(), x.data_ptr(), y32.data_ptr(), 0, None, 1.0, 0, b, l, c, taps, dil, st)
        f3 = lambda: L.astts_op_conv1d_snake(h.data_ptr(), 1, None, wf.data_ptr(), pw.bias.data_ptr(), None, y16.data_ptr(), 1, None, 1.0, 0, b, l, c, taps, dil, st)
        t1, t2, t3 = timed(f1), timed(f2), timed(f3)
        gf = 2.0 * b * l * c * c * taps * 1e-9
        print(f'c={c} l={l} taps={taps} dil={dil}: snake fp32->fp16 {t1:6.1f} us ({gf / t1 * 1e3:5.0f} TFLOP/s) | snake fp16->fp32 + res {t2:6.1f} us | plain fp16->fp16 {t3:6.1f} us ({gf / t3 * 1e3:5.0f} TFLOP/s)')
