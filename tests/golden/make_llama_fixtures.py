"""Generates tests/golden/llama_{tiny,wide}.npz: inputs and expected outputs of the query embedder's LLM, produced by the
third-party implementation the reference calls -- transformers' LlamaForCausalLM (src/search_milvus.py:36-72,75-108;
milvus/search_json.py:154-198) -- on the seeded synthetic weights of astts.llm.weights.make_llama_weights.
Run in the BUILD container only (python tests/golden/make_llama_fixtures.py); the .npz files are data: token ids in,
hidden states / pooled embeddings / greedy tokens out.  Nothing of transformers travels."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "autostyle-tts_amd")]

from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402

from astts.llm.config import LlamaShape  # noqa: E402
from astts.llm.weights import make_llama_weights  # noqa: E402


def run(name, cfg, seed, lens, gen_len, per_layer=False):
    torch.manual_seed(0)
    sd = make_llama_weights(cfg, seed)
    model = LlamaForCausalLM(LlamaConfig(**cfg.hf_kwargs())).eval().float()
    missing = model.load_state_dict(sd, strict=False, assign=True)      # (assign: the 13 GB of the full-depth model are held once)
    model.tie_weights()
    assert set(missing.missing_keys) <= {"lm_head.weight"} and not missing.unexpected_keys, missing
    g = torch.Generator().manual_seed(seed + 1)
    out = {"seed": np.int64(seed), "lens": np.asarray(lens, np.int64)}
    tmax = max(lens)
    ids = torch.zeros((len(lens), tmax), dtype=torch.int64)
    emb, last = [], []
    with torch.no_grad():
        for i, n in enumerate(lens):                      # the reference embeds ONE text per call (no padding)
            row = torch.randint(3, cfg.vocab, (1, n), generator=g)
            row[0, 0] = cfg.bos_token_id
            ids[i, :n] = row[0]
            o = model.model(input_ids=row, attention_mask=torch.ones_like(row), output_hidden_states=True)
            hs = o.hidden_states[-1]
            emb.append(hs.mean(dim=1).squeeze().float().numpy())          # src/search_milvus.py:99-100
            last.append(hs[0, -1].numpy())
            if i == 0:
                out["hidden_layer1_row0"] = o.hidden_states[1][0].numpy()  # input of layer 1: localises a mismatch
                out["hidden_final_row0"] = hs[0].numpy()
                if per_layer:       # the last token's residual stream after every layer (before the final norm; the last entry after it):
                    out["hidden_by_layer_last_token_row0"] = np.stack([h[0, -1].numpy() for h in o.hidden_states])   # error growth by depth
        prompt = ids[:1, :lens[0]]
        gen = model.generate(prompt, attention_mask=torch.ones_like(prompt), max_new_tokens=gen_len, do_sample=False,
                             eos_token_id=cfg.eos_token_id, pad_token_id=cfg.eos_token_id)
        out["logits_last_row0"] = model(input_ids=prompt).logits[0, -1].numpy()
    out.update(ids=ids.numpy(), embedding=np.stack(emb), hidden_last_token=np.stack(last), greedy=gen[0].numpy())
    path = os.path.join(ROOT, "tests", "golden", f"llama_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, os.path.getsize(path) // 1024, "KB; greedy tail", gen[0, lens[0]:].tolist())


if __name__ == "__main__":
    if "--3b" in sys.argv:
        # the model the reference runs (src/search_milvus.py:75-108): all 28 layers of Llama-3.2-3B over its 128 256-entry vocabulary.
        # 3.2 B fp32 parameters = 13 GB on the build container's CPU; a few short prompts; ~10 minutes.
        run("3b", LlamaShape.llama32_3b(), seed=9, lens=[40, 17, 60], gen_len=4, per_layer=True)
    else:
        run("tiny", LlamaShape.tiny(), seed=7, lens=[23, 5, 64, 130], gen_len=10)
        run("wide", LlamaShape.wide(), seed=8, lens=[40, 17], gen_len=6)
