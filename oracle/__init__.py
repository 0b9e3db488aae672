"""CPU oracle for the AutoStyle-TTS hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import anything from this package.  The product path (autostyle-tts_amd/astts)
never imports it and fails loudly when the HIP extension is missing.

Parity status
-------------
* kNN (oracle/knn.py; the bank itself is read by tests/golden/make_fixtures.py): PINNED by data artefacts of the
  reference -- the shipped style bank milvus/milvus_demo.db (130 x 6144) and the
  recorded retrieval run output_emb/search_results.json -- see tests/golden/.
  The arithmetic itself lives in un-vendored third-party code (pymilvus /
  milvus-lite, no version pinned anywhere in the reference tree), so the
  restatement follows Milvus' published COSINE semantics and is anchored on the
  reference's own call sites.
* synthesis (oracle/synth.py): PARITY UNPINNED.  The arithmetic lives in the
  authors' private CosyVoice fork (not public, not vendored, no weights); the
  oracle is this build's own fp32 PyTorch-CPU restatement of the published
  CosyVoice-300M architecture with seeded synthetic weights.
"""
