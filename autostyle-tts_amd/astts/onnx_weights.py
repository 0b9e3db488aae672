"""Weights out of an ONNX file without onnxruntime: the *initializers* of ``speech_tokenizer_v1.onnx`` / ``campplus.onnx``.

The reference loads those two files inside ``CosyVoice(model_dir)`` (/root/reference/tts_with_rag.py:159) and runs them through
onnxruntime [EXT]; neither onnxruntime nor onnx is installed here, and the networks themselves run as HIP kernels
(astts/frontend_nets.py).  All that is needed from the file is its trained tensors, and those are plain protobuf:

    ModelProto   { 7: GraphProto graph }
    GraphProto   { 1: repeated NodeProto node, 5: repeated TensorProto initializer }
    NodeProto    { 1: repeated string input, 2: repeated string output, 4: string op_type, 5: repeated AttributeProto attribute }
    AttributeProto { 1: name, 5: TensorProto t }                      (Constant nodes carry their value as attribute "value")
    TensorProto  { 1: repeated int64 dims, 2: int32 data_type, 4: float_data (packed), 5: int32_data, 7: int64_data, 8: name,
                   9: bytes raw_data (little endian), 10: double_data, 13: external_data, 14: data_location }

read with the protobuf wire decoder of astts.milvus_lite.  A writer for the same subset exists for the tests (round trip) and
for exporting this build's seeded synthetic networks in the layout the reader expects.
"""
from __future__ import annotations

from typing import Dict, Iterable, Tuple

import numpy as np

from .milvus_lite import _packed_varints, iter_fields

# TensorProto.DataType
FLOAT, UINT8, INT8, INT32, INT64, BOOL, FLOAT16, DOUBLE, BFLOAT16 = 1, 2, 3, 6, 7, 9, 10, 11, 16
_NP = {FLOAT: "<f4", UINT8: "u1", INT8: "i1", INT32: "<i4", INT64: "<i8", BOOL: "u1", FLOAT16: "<f2", DOUBLE: "<f8"}


class OnnxFormatError(ValueError):
    pass


def _tensor(buf: bytes) -> Tuple[str, np.ndarray]:
    dims, dtype, name, raw = [], None, "", None
    floats, int32s, int64s, doubles = [], [], [], []
    external = False
    for fno, wt, v in iter_fields(buf):
        if fno == 1:
            dims.extend(_packed_varints(v) if wt == 2 else [v])
        elif fno == 2:
            dtype = v
        elif fno == 4:
            floats.append(np.frombuffer(v, "<f4") if wt == 2 else np.frombuffer(v, "<f4", count=1))
        elif fno == 5:
            int32s.extend(_packed_varints(v) if wt == 2 else [v])
        elif fno == 7:
            int64s.extend(_packed_varints(v) if wt == 2 else [v])
        elif fno == 8:
            name = v.decode("utf-8")
        elif fno == 9:
            raw = v
        elif fno == 10:
            doubles.append(np.frombuffer(v, "<f8") if wt == 2 else np.frombuffer(v, "<f8", count=1))
        elif fno == 14 and v == 1:
            external = True
    if external:
        raise OnnxFormatError(f"tensor {name!r} keeps its data in an external file: not supported (re-export with the data embedded)")
    if dtype is None:
        raise OnnxFormatError(f"tensor {name!r} has no data_type")
    shape = tuple(int(d) for d in dims)
    n = int(np.prod(shape)) if shape else 1
    if raw is not None:
        if dtype == BFLOAT16:
            a = (np.frombuffer(raw, "<u2").astype(np.uint32) << 16).view("<f4")
        elif dtype in _NP:
            a = np.frombuffer(raw, _NP[dtype])
        else:
            raise OnnxFormatError(f"tensor {name!r}: data_type {dtype} is not supported")
    elif dtype == FLOAT:
        a = np.concatenate(floats) if floats else np.zeros(0, "<f4")
    elif dtype == DOUBLE:
        a = np.concatenate(doubles) if doubles else np.zeros(0, "<f8")
    elif dtype == INT64:
        a = np.asarray(int64s, "<i8")
    elif dtype in (INT32, INT8, UINT8, BOOL):
        a = np.asarray(int32s, "<i4").astype(_NP[dtype])
    elif dtype == FLOAT16:                       # stored as the bit patterns in int32_data
        a = np.asarray(int32s, "<u2").view("<f2")
    else:
        raise OnnxFormatError(f"tensor {name!r}: data_type {dtype} is not supported")
    if a.size != n:
        raise OnnxFormatError(f"tensor {name!r}: {a.size} values for shape {shape}")
    return name, a.reshape(shape).copy()


def read_initializers(path: str, constants: bool = True) -> Dict[str, np.ndarray]:
    """name -> array for every initializer of the file's graph (and, with ``constants``, for every Constant node's tensor
    value under the node's output name: some exporters keep weights there)."""
    with open(path, "rb") as f:
        buf = f.read()
    graph = None
    for fno, wt, v in iter_fields(buf):
        if fno == 7 and wt == 2:
            graph = v
    if graph is None:
        raise OnnxFormatError(f"{path}: no graph in the file (not an ONNX ModelProto?)")
    out: Dict[str, np.ndarray] = {}
    for fno, wt, v in iter_fields(graph):
        if fno == 5 and wt == 2:
            name, a = _tensor(v)
            out[name] = a
        elif fno == 1 and wt == 2 and constants:
            op, outputs, value = None, [], None
            for nf, nwt, nv in iter_fields(v):
                if nf == 4:
                    op = nv.decode("utf-8")
                elif nf == 2:
                    outputs.append(nv.decode("utf-8"))
                elif nf == 5 and nwt == 2:
                    an, at = None, None
                    for af, awt, av in iter_fields(nv):
                        if af == 1:
                            an = av.decode("utf-8")
                        elif af == 5 and awt == 2:
                            at = av
                    if an == "value" and at is not None:
                        value = at
            if op == "Constant" and value is not None and outputs:
                _, a = _tensor(value)
                out.setdefault(outputs[0], a)
    return out


# ------------------------------------------------------------------------------------------ writer (tests / export)
def _enc_varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(fno: int, wt: int, payload) -> bytes:
    key = _enc_varint((fno << 3) | wt)
    if wt == 0:
        return key + _enc_varint(int(payload))
    return key + _enc_varint(len(payload)) + payload


def encode_tensor(name: str, a: np.ndarray, raw: bool = True) -> bytes:
    a = np.asarray(a)
    code = {np.dtype("float32"): FLOAT, np.dtype("float16"): FLOAT16, np.dtype("float64"): DOUBLE, np.dtype("int64"): INT64,
            np.dtype("int32"): INT32}[a.dtype]
    msg = b"".join(_field(1, 0, d) for d in a.shape) + _field(2, 0, code)
    if raw or code == FLOAT16:
        msg += _field(8, 2, name.encode()) + _field(9, 2, np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<")).tobytes())
    elif code == FLOAT:
        msg += _field(4, 2, a.astype("<f4").tobytes()) + _field(8, 2, name.encode())
    elif code == DOUBLE:
        msg += _field(8, 2, name.encode()) + _field(10, 2, a.astype("<f8").tobytes())
    else:
        msg += _field(7 if code == INT64 else 5, 2, b"".join(_enc_varint(int(x)) for x in a.reshape(-1))) + _field(8, 2, name.encode())
    return msg


def write_initializers(path: str, tensors: Iterable[Tuple[str, np.ndarray]], raw: bool = True, producer: str = "astts") -> None:
    """A minimal ModelProto holding ``tensors`` as graph initializers (ir_version 8, no nodes)."""
    graph = _field(2, 2, b"astts_weights") + b"".join(_field(5, 2, encode_tensor(n, a, raw)) for n, a in tensors)
    model = _field(1, 0, 8) + _field(2, 2, producer.encode()) + _field(7, 2, graph)
    with open(path, "wb") as f:
        f.write(model)


__all__ = ["read_initializers", "write_initializers", "encode_tensor", "OnnxFormatError"]
