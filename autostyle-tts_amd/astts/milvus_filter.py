"""Milvus boolean ``filter`` expressions -> a row mask (uint8 ``[N]``) for astts_knn_search_masked.

The reference passes ``filter=None`` everywhere (milvus/search_json.py:246-252, milvus/RAG.py:381-387), so this is the long
tail of the ``MilvusClient.search`` surface: scalar filtering on the primary key and on the
dynamic ``$meta`` fields the bank builders store (``file_id``, ``text``; milvus/RAG.py:541-544).  The expression is evaluated
on the HOST over the collection's scalar fields (strings and small integers -- not arithmetic of the search) and handed to
the kNN kernels as one byte per row; masked rows are never candidates, so the hits are the exact top-k of the allowed rows.

Grammar (the subset of Milvus' expression language that scalar / JSON fields of such a bank need):
    expr    := or
    or      := and ( ("or" | "||") and )*
    and     := unary ( ("and" | "&&") unary )*
    unary   := ("not" | "!") unary | "(" expr ")" | cmp
    cmp     := operand ( "==" | "!=" | "<" | "<=" | ">" | ">=" ) operand
             | operand ["not"] "in" "[" literal ("," literal)* "]"
             | operand ["not"] "like" string            ('%' matches any run of characters, '_' one character)
    operand := literal | field | field "[" string "]"     ($meta["file_id"], or any dynamic key by its bare name)
    literal := string | number | true | false
A comparison whose field is missing in a row, or whose operand types do not compare, is false for that row (Milvus'
behaviour for dynamic fields).
"""
from __future__ import annotations

import re
from typing import Any, Callable, Dict, List, Sequence

import numpy as np

_TOKEN = re.compile(r"""\s*(?:
    (?P<str>"(?:[^"\\]|\\.)*"|'(?:[^'\\]|\\.)*') |
    (?P<num>[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?) |
    (?P<op>==|!=|<=|>=|&&|\|\||[<>()\[\],!]) |
    (?P<name>\$?[A-Za-z_][A-Za-z_0-9]*)
)""", re.VERBOSE)


class FilterSyntaxError(ValueError):
    pass


def _tokens(text: str) -> List[tuple]:
    out, pos = [], 0
    text = text.rstrip()
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m or m.end() == pos:
            raise FilterSyntaxError(f"cannot parse filter expression at {text[pos:pos + 20]!r}")
        pos = m.end()
        kind = m.lastgroup
        val = m.group(kind)
        if kind == "str":
            val = re.sub(r"\\(.)", r"\1", val[1:-1])          # \" \' \\ -> the character itself
        elif kind == "num":
            val = float(val) if re.search(r"[.eE]", val) else int(val)
        out.append((kind, val))
    return out


Row = Dict[str, Any]
Pred = Callable[[Row], Any]
_MISSING = object()


class _Parser:
    def __init__(self, text: str):
        self.toks = _tokens(text)
        self.i = 0

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else (None, None)

    def take(self, kind=None, val=None):
        k, v = self.peek()
        if k is None or (kind is not None and k != kind) or (val is not None and (v.lower() if isinstance(v, str) and k == "name" else v) != val):
            raise FilterSyntaxError(f"filter expression: expected {val or kind}, found {v!r}")
        self.i += 1
        return v

    def is_kw(self, *words) -> bool:
        k, v = self.peek()
        return (k == "name" and v.lower() in words) or (k == "op" and v in words)

    def parse(self) -> Pred:
        e = self.p_or()
        if self.i != len(self.toks):
            raise FilterSyntaxError(f"filter expression: unexpected {self.peek()[1]!r}")
        return e

    def p_or(self) -> Pred:
        left = self.p_and()
        while self.is_kw("or", "||"):
            self.i += 1
            right = self.p_and()
            left = (lambda a, b: lambda r: bool(a(r)) or bool(b(r)))(left, right)
        return left

    def p_and(self) -> Pred:
        left = self.p_unary()
        while self.is_kw("and", "&&"):
            self.i += 1
            right = self.p_unary()
            left = (lambda a, b: lambda r: bool(a(r)) and bool(b(r)))(left, right)
        return left

    def p_unary(self) -> Pred:
        if self.is_kw("not", "!"):
            self.i += 1
            inner = self.p_unary()
            return lambda r: not bool(inner(r))
        if self.peek() == ("op", "("):
            self.i += 1
            e = self.p_or()
            self.take("op", ")")
            return e
        return self.p_cmp()

    def p_operand(self) -> Pred:
        k, v = self.peek()
        if k in ("str", "num"):
            self.i += 1
            return lambda r, v=v: v
        if k == "name":
            self.i += 1
            if v.lower() in ("true", "false"):
                b = v.lower() == "true"
                return lambda r: b
            if self.peek() == ("op", "["):            # $meta["key"] / json_field["key"]
                self.i += 1
                key = self.take("str")
                self.take("op", "]")
                if v == "$meta":
                    return lambda r, key=key: r.get(key, _MISSING)
                return lambda r, v=v, key=key: (r.get(v) or {}).get(key, _MISSING) if isinstance(r.get(v), dict) else _MISSING
            return lambda r, v=v: r.get(v, _MISSING)
        raise FilterSyntaxError(f"filter expression: expected a field or a literal, found {v!r}")

    def p_cmp(self) -> Pred:
        left = self.p_operand()
        neg = False
        if self.is_kw("not"):
            self.i += 1
            neg = True
            if not self.is_kw("in", "like"):
                raise FilterSyntaxError("filter expression: 'not' must be followed by 'in' or 'like' here")
        if self.is_kw("in"):
            self.i += 1
            self.take("op", "[")
            items = []
            while self.peek() != ("op", "]"):
                k, v = self.peek()
                if k not in ("str", "num"):
                    raise FilterSyntaxError(f"filter expression: list items must be literals, found {v!r}")
                items.append(v)
                self.i += 1
                if self.peek() == ("op", ","):
                    self.i += 1
            self.take("op", "]")

            def pred_in(r, left=left, items=items, neg=neg):
                x = left(r)
                if x is _MISSING:
                    return False
                hit = any(_same_kind(x, y) and x == y for y in items)
                return (not hit) if neg else hit
            return pred_in
        if self.is_kw("like"):
            self.i += 1
            pat = self.take("str")
            rx = re.compile("".join(".*" if ch == "%" else "." if ch == "_" else re.escape(ch) for ch in pat) + r"\Z", re.DOTALL)

            def pred_like(r, left=left, rx=rx, neg=neg):
                x = left(r)
                if not isinstance(x, str):
                    return False
                hit = rx.match(x) is not None
                return (not hit) if neg else hit
            return pred_like
        k, op = self.peek()
        if k != "op" or op not in ("==", "!=", "<", "<=", ">", ">="):
            raise FilterSyntaxError(f"filter expression: expected a comparison, found {op!r}")
        self.i += 1
        right = self.p_operand()

        def pred(r, left=left, right=right, op=op):
            a, b = left(r), right(r)
            if a is _MISSING or b is _MISSING or not _same_kind(a, b):
                return False
            if op == "==":
                return a == b
            if op == "!=":
                return a != b
            if isinstance(a, bool):
                return False
            return a < b if op == "<" else a <= b if op == "<=" else a > b if op == ">" else a >= b
        return pred


def _same_kind(a, b) -> bool:
    num = lambda x: isinstance(x, (int, float)) and not isinstance(x, bool)
    return (num(a) and num(b)) or (isinstance(a, str) and isinstance(b, str)) or (isinstance(a, bool) and isinstance(b, bool))


def compile_filter(expr: str) -> Pred:
    """-> predicate over one row's scalar fields (a dict: primary key under its field name + the dynamic / scalar fields)."""
    return _Parser(expr).parse()


def row_mask(expr: str, pk_field: str, pks: Sequence[int], metas: Sequence[Dict[str, Any]]) -> np.ndarray:
    """uint8 ``[N]``: 1 where the row satisfies ``expr``."""
    pred = compile_filter(expr)
    out = np.zeros(len(pks), np.uint8)
    for i, (pk, meta) in enumerate(zip(pks, metas)):
        row = dict(meta)
        row[pk_field] = pk
        out[i] = 1 if pred(row) else 0
    return out
