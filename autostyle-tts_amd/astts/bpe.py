"""Byte-pair tokenizer over a tiktoken-format vocabulary file: the text tokenizer behind the reference's ``CosyVoice`` frontend
("tokenizer (multilingual BPE, vocab 51 866)"; upstream: whisper's ``multilingual.tiktoken`` loaded through the
``tiktoken`` package, which is not installed here).  Given the vocabulary FILE (one ``base64(token bytes) rank`` pair per line) this
class encodes exactly as tiktoken does -- split the text with the GPT-2 / Whisper pre-tokenisation pattern, then merge adjacent byte
pairs of each piece in order of increasing rank -- and plugs into ``Frontend(tokenizer=TiktokenBPE.from_file(path))``.

The vocabulary file itself does not exist offline (the frontend's default stays the labelled byte-level stand-in); the ALGORITHM is
pinned against the ``tokenizers`` library on a byte-level BPE trained in the test (tests/test_bpe_cpu.py): same ids on multilingual
samples, emoji, digits and whitespace runs."""
from __future__ import annotations

import base64
from typing import Dict, Iterable, List, Optional, Sequence

import regex

# the pre-tokenisation pattern of GPT-2, used unchanged by whisper's multilingual vocabulary
GPT2_PATTERN = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""


class TiktokenBPE:
    def __init__(self, ranks: Dict[bytes, int], special_tokens: Optional[Dict[str, int]] = None, pattern: str = GPT2_PATTERN):
        self.ranks = dict(ranks)
        self.special = dict(special_tokens or {})
        self.pattern = regex.compile(pattern)
        self._by_id = {v: k for k, v in self.ranks.items()}
        self._special_by_id = {v: k for k, v in self.special.items()}
        clash = set(self._by_id) & set(self._special_by_id)
        if clash:
            raise ValueError(f"special token ids collide with the vocabulary: {sorted(clash)[:5]}")
        self._special_re = regex.compile("|".join(regex.escape(t) for t in sorted(self.special, key=len, reverse=True))) if self.special else None

    @property
    def n_vocab(self) -> int:
        return max(list(self._by_id) + list(self._special_by_id)) + 1

    @classmethod
    def from_file(cls, path: str, special_tokens: Optional[Dict[str, int]] = None, pattern: str = GPT2_PATTERN) -> "TiktokenBPE":
        ranks = {}
        with open(path, "rb") as f:
            for line in f:
                if not line.strip():
                    continue
                tok, rank = line.split()
                ranks[base64.b64decode(tok)] = int(rank)
        return cls(ranks, special_tokens, pattern)

    def to_file(self, path: str) -> None:
        with open(path, "wb") as f:
            for tok, rank in sorted(self.ranks.items(), key=lambda kv: kv[1]):
                f.write(base64.b64encode(tok) + b" " + str(rank).encode() + b"\n")

    # ------------------------------------------------------------------ encoding
    def _merge(self, piece: bytes) -> List[int]:
        """tiktoken's byte_pair_merge: start from single bytes, repeatedly merge the adjacent pair whose concatenation has the
        lowest rank (leftmost on ties), until no adjacent pair is in the vocabulary."""
        if piece in self.ranks:
            return [self.ranks[piece]]
        parts = [piece[i:i + 1] for i in range(len(piece))]
        while len(parts) > 1:
            best, best_rank = -1, None
            for i in range(len(parts) - 1):
                r = self.ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = i, r
            if best < 0:
                break
            parts[best:best + 2] = [parts[best] + parts[best + 1]]
        try:
            return [self.ranks[p] for p in parts]
        except KeyError as e:                       # a vocabulary without all 256 single bytes
            raise ValueError(f"byte sequence {e.args[0]!r} is not in the vocabulary") from None

    def _encode_ordinary(self, text: str) -> List[int]:
        out: List[int] = []
        for m in self.pattern.finditer(text):
            out.extend(self._merge(m.group(0).encode("utf-8")))
        return out

    def encode(self, text: str, allowed_special: Iterable[str] | str = ()) -> List[int]:
        """``allowed_special``: the special tokens that may appear verbatim in ``text`` (``"all"`` = every one, as upstream's frontend
        passes); any other occurrence is encoded as ordinary text."""
        if not self._special_re or not allowed_special:
            return self._encode_ordinary(text)
        allowed = set(self.special) if allowed_special == "all" else set(allowed_special)
        out: List[int] = []
        pos = 0
        for m in self._special_re.finditer(text):
            if m.group(0) not in allowed:
                continue
            out.extend(self._encode_ordinary(text[pos:m.start()]))
            out.append(self.special[m.group(0)])
            pos = m.end()
        out.extend(self._encode_ordinary(text[pos:]))
        return out

    def decode(self, ids: Sequence[int]) -> str:
        buf = bytearray()
        for i in ids:
            i = int(i)
            if i in self._by_id:
                buf += self._by_id[i]
            elif i in self._special_by_id:
                buf += self._special_by_id[i].encode("utf-8")
            else:
                raise ValueError(f"token id {i} is not in the vocabulary")
        return buf.decode("utf-8", errors="replace")
