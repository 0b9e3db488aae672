"""The query embedder's pass alone (Llama-3.2-3B widths, 3 layers, EP_TEXTS texts of 60 tokens): for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path[:0] = ['.', 'autostyle-tts_amd']
import torch
from astts.llm.config import LlamaShape
from astts.llm.embedder import LlamaEmbedder
from astts.llm.weights import make_llama_weights
shape = LlamaShape.wide()
emb = LlamaEmbedder(make_llama_weights(shape, 0), shape, torch.device('cuda'))
b, t = int(os.environ.get('EP_TEXTS', 256)), 60
g = torch.Generator().manual_seed(5)
ids = torch.randint(3, shape.vocab, (b, t), generator=g)
lens = torch.full((b,), t, dtype=torch.int32)
for _ in range(2):
    emb.embed_ids(ids, lens)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    emb.embed_ids(ids, lens)
torch.cuda.synchronize()
print(f'{b} texts: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per pass')
