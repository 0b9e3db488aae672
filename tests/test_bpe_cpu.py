"""astts.bpe.TiktokenBPE (the reference frontend's multilingual BPE; upstream uses the tiktoken package, absent
here) pinned against an independent implementation: a byte-level BPE is trained with the `tokenizers` library (GPT-2 pre-tokeniser,
the pattern whisper's multilingual vocabulary uses), its vocabulary is written out in tiktoken's file format, and the ids TiktokenBPE
produces from that file must equal the library's on multilingual text, emoji, digits and whitespace runs."""
import json
import os

import pytest

from astts.bpe import TiktokenBPE

CORPUS = [
    "I did it, I asked her to marry me.", "He did. In Niagara Falls.", "你好，世界！今天天气不错。我们去公园散步吧。",
    "Yeah. 😊 it's 12345 dollars... isn't it?", "こんにちは、元気ですか？", "Don't you think they'll've   gone   by 10:30?\n\nMaybe.",
    "The quick brown fox jumps over the lazy dog. " * 3, "Ça va très bien, merci — et vous ?",
]
SAMPLES = CORPUS + ["I'll marry her in Niagara, 你好 12 34!", "  leading and trailing   ", "tabs\tand\nnewlines\r\n", "unseen: ζωή Ω≈ç√∫ 🤖🤖",
                    "", "a", "'s't're", "1234567890" * 3]


def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


@pytest.fixture(scope="module")
def trained(tmp_path_factory):
    from tokenizers import Tokenizer, models, pre_tokenizers, trainers

    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=True)
    tr = trainers.BpeTrainer(vocab_size=700, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), special_tokens=[])
    tok.train_from_iterator(CORPUS * 40, tr)
    u2b = {c: b for b, c in _bytes_to_unicode().items()}
    vocab = json.loads(tok.to_str())["model"]["vocab"]
    ranks = {bytes(u2b[ch] for ch in t): i for t, i in vocab.items()}
    assert len(ranks) == len(vocab) and all(bytes([b]) in ranks for b in range(256))
    path = os.path.join(tmp_path_factory.mktemp("bpe"), "test.tiktoken")
    TiktokenBPE(ranks).to_file(path)
    return tok, path


def test_ids_equal_the_tokenizers_library(trained):
    ref, path = trained
    bpe = TiktokenBPE.from_file(path)
    assert bpe.n_vocab == ref.get_vocab_size()
    for text in SAMPLES:
        assert bpe.encode(text) == ref.encode(text).ids, text
        assert bpe.decode(bpe.encode(text)) == text


def test_special_tokens_and_frontend_plug(trained):
    _, path = trained
    base = TiktokenBPE.from_file(path)
    n = base.n_vocab
    bpe = TiktokenBPE.from_file(path, special_tokens={"<|endoftext|>": n, "<|en|>": n + 1})
    ids = bpe.encode("<|en|>Hello<|endoftext|>", allowed_special="all")
    assert ids[0] == n + 1 and ids[-1] == n and bpe.decode(ids) == "<|en|>Hello<|endoftext|>"
    assert n not in bpe.encode("<|endoftext|>")                       # not allowed: ordinary text
    assert bpe.encode("<|en|>x", allowed_special={"<|endoftext|>"})[0] != n + 1
    with pytest.raises(ValueError):
        TiktokenBPE({b"a": 0}, special_tokens={"<|x|>": 0})
    from astts.frontend import Frontend
    from astts.synth.config import SynthConfig

    fe = Frontend(SynthConfig.tiny(), tokenizer=bpe)
    t = fe.text_ids("He did. In Niagara Falls.")
    assert t.shape[0] == 1 and t.tolist()[0] == bpe.encode("He did. In Niagara Falls.")
