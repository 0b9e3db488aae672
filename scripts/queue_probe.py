"""Which torch streams really run launch chains side by side?  For stream pairs (0, j): two chains of short dependent kernels
enqueued from two host threads; wall time of the pair vs one chain alone."""
import sys, time, threading, os
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts import ops, _lib
lib = _lib.load()
dev = torch.device('cuda')
NS = int(os.environ.get('NS', '12'))
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
COUNT, US, BLK = 400, 3, 256


def run(idx):
    torch.cuda.synchronize()
    def one(s):
        _lib.check(lib.astts_stream_chain(COUNT, US, BLK, int(s.cuda_stream)))
        s.synchronize()
    th = [threading.Thread(target=one, args=(streams[i],)) for i in idx]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return (time.perf_counter() - t0) * 1e6 / COUNT


run([0]); run([0, 1])
print('env GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'))
print('alone:', ' '.join(f'{run([i]):.1f}' for i in range(NS)), 'us per launch')
for i in range(min(NS, 5)):
    print(f'pair with stream {i}:', ' '.join('  -- ' if j == i else f'{run([i, j]):5.1f}' for j in range(NS)))
for grp in ([0, 1, 2], [0, 1, 2, 3], [0, 1, 2, 3, 4], [0, 1, 2, 3, 4, 5], [0, 2, 4, 6], [0, 1, 4, 5], [1, 2, 3, 4], [4, 5, 6, 7]):
    if max(grp) < NS:
        print(f'group {grp}: {run(grp):.1f} us per launch of each chain')
