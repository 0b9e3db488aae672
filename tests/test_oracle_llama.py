"""oracle/llama.py (the CPU restatement of the embedder LLM) against the golden fixtures produced by transformers'
LlamaForCausalLM (tests/golden/make_llama_fixtures.py): hidden states, mean-pooled embeddings, last-token logits and the
greedy continuation.  fp32 vs fp32: 1e-4 of the tensor's scale (different matmul blocking)."""
import os

import numpy as np
import pytest
import torch

from oracle import llama as ol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    from astts.llm.config import LlamaShape
    from astts.llm.weights import make_llama_weights

    fx = np.load(os.path.join(GOLD, f"llama_{name}.npz"))
    cfg = getattr(LlamaShape, name)()
    return fx, cfg, make_llama_weights(cfg, int(fx["seed"]))


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())


@pytest.mark.parametrize("name", ["tiny", "wide"])
def test_restatement_matches_transformers_fixtures(name):
    fx, cfg, sd = _load(name)
    torch.set_num_threads(8)
    ids, lens = torch.from_numpy(fx["ids"]), fx["lens"]
    with torch.no_grad():
        row0 = ids[:1, :lens[0]]
        final, hs = ol.forward_hidden(sd, cfg, row0, all_layers=True)
        assert _rel(hs[1][0], fx["hidden_layer1_row0"]) < 1e-4
        assert _rel(final[0], fx["hidden_final_row0"]) < 1e-4
        assert _rel(ol.logits_last(sd, cfg, row0)[0], fx["logits_last_row0"]) < 1e-4
        for i, n in enumerate(lens):                          # one text at a time, as the reference embeds
            e = ol.get_embedding(sd, cfg, ids[i:i + 1, :n])[0]
            assert _rel(e, fx["embedding"][i]) < 1e-4, i
        # batched with right padding + lens (what the GPU path does) == one at a time
        eb = ol.get_embedding(sd, cfg, ids, torch.from_numpy(lens))
        assert _rel(eb, fx["embedding"]) < 1e-4
        gen = ol.generate_greedy(sd, cfg, row0, len(fx["greedy"]) - int(lens[0]))
        assert gen[0].tolist() == fx["greedy"].tolist()


def test_llama3_rope_frequencies_known_values():
    """_compute_llama3_parameters at the Llama-3.2 settings: high frequencies untouched, low ones divided by 32."""
    from astts.llm.config import LlamaShape

    c = LlamaShape.llama32_3b()
    inv = ol.llama3_inv_freq(c.head_dim, c.rope_theta, c.rope_factor, c.rope_low_freq_factor, c.rope_high_freq_factor, c.rope_original_max_pos)
    base = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2).float() / c.head_dim))
    assert inv.shape == (64,) and float(inv[0]) == 1.0
    assert torch.equal(inv[:20], base[:20])                                   # wavelength < 2048: unchanged
    assert torch.allclose(inv[-1], base[-1] / 32.0)                           # wavelength > 8192: / factor
    assert bool((inv[:-1] >= inv[1:]).all())


def test_combined_query_layout():
    """src/search_milvus.py:220-221: [emotion | biography], float32, un-normalised."""
    e, b = torch.arange(4.0), torch.arange(4.0) + 10
    q = ol.combined_query(e, b)
    assert q.dtype == torch.float32 and q.tolist() == [0, 1, 2, 3, 10, 11, 12, 13]
