"""The oracle of the learned frontend (oracle/frontend_nets.py) against third-party code, and the host-side pieces around it
(CPU).  Fixture: tests/golden/frontend_blocks.npz, produced by transformers' WhisperEncoder / WhisperEncoderLayer on the seeded
weights of astts.frontend_weights (tests/golden/make_frontend_fixtures.py)."""
import os

import numpy as np
import pytest
import torch

from astts.frontend_weights import (CamPlusShape, SpeechTokenizerShape, check_against_manifest, load_frontend_weights, make_campplus_weights,
                                    make_speech_tokenizer_weights, manifest, sinusoids)
from oracle import frontend_nets as ofn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_tokenizer_encoder_matches_transformers_whisper_encoder():
    fx = np.load(os.path.join(GOLD, "frontend_blocks.npz"))
    cfg = SpeechTokenizerShape.tiny()
    sd = make_speech_tokenizer_weights(cfg, int(fx["seed"]))
    mel = torch.from_numpy(fx["mel"])
    pos = sinusoids(cfg.n_ctx, cfg.d)
    assert torch.equal(pos[:fx["positions"].shape[0]], torch.from_numpy(fx["positions"]))       # transformers' own table
    stem = ofn.tokenizer_stem(sd, mel, pos)
    assert float((stem - torch.from_numpy(fx["stem"])).abs().max()) < 1e-5
    x = stem
    for i in range(cfg.layers):
        x = ofn.whisper_block(sd, f"encoder.blocks.{i}.", x, cfg.heads)
        assert float((x - torch.from_numpy(fx[f"layer{i}"])).abs().max()) < 2e-5
    lens = torch.from_numpy(fx["masked_lens"])
    mask = torch.arange(stem.shape[1])[None, :] < lens[:, None]
    y = ofn.whisper_block(sd, "encoder.blocks.0.", stem, cfg.heads, mask)
    ref = torch.from_numpy(fx["layer0_masked"])
    for b in range(2):       # rows behind a sequence's length are padding (their values are not defined)
        assert float((y[b, :lens[b]] - ref[b, :lens[b]]).abs().max()) < 2e-5
    full, out_lens = ofn.tokenizer_encode(sd, cfg, mel)
    assert torch.equal(full, x) and out_lens.tolist() == [mel.shape[2] // 2] * 2


def test_quantiser_is_the_nearest_code_and_padding_does_not_leak():
    cfg = SpeechTokenizerShape.tiny()
    sd = make_speech_tokenizer_weights(cfg, 4)
    g = torch.Generator().manual_seed(0)
    frames = torch.randn(50, cfg.d, generator=g)
    codes = ofn.vq_encode(sd, cfg, frames)
    e = sd["quantizer._codebook.embed"].double()
    x = frames.double() / frames.double().norm(dim=1, keepdim=True)
    brute = torch.stack([((e - x[i]) ** 2).sum(1) for i in range(50)]).argmin(1)
    assert torch.equal(codes, brute)
    assert torch.equal(ofn.vq_encode(sd, cfg, e[:7].float() * 3.0), torch.arange(7))            # a (scaled) code word maps to itself
    # a shorter row inside a padded batch gets the tokens it gets alone
    mel = torch.randn(2, cfg.n_mels, 120, generator=g)
    lens = torch.tensor([120, 80])
    both, out_lens = ofn.speech_tokens(sd, cfg, mel, lens)
    alone, _ = ofn.speech_tokens(sd, cfg, mel[1:, :, :80])
    assert out_lens.tolist() == [60, 40]
    # (the convolution stem sees the padded frames right of the cut: the last frame of the short row may differ)
    assert torch.equal(both[1, :39], alone[0, :39])


def test_campplus_oracle_shapes_and_invariances():
    cfg = CamPlusShape.tiny()
    sd = make_campplus_weights(cfg, 3)
    g = torch.Generator().manual_seed(1)
    fb = torch.randn(2, 130, cfg.feat_dim, generator=g)
    e = ofn.speaker_embedding(sd, cfg, fb)
    assert e.shape == (2, cfg.emb) and bool(torch.isfinite(e).all())
    assert torch.allclose(ofn.speaker_embedding(sd, cfg, fb[1:]), e[1:], atol=1e-5)              # rows are independent
    head = ofn.campplus_head(sd, cfg, fb)
    assert head.shape == (2, cfg.head_out, 130)
    fr = ofn.campplus_xvector(sd, cfg, head, return_frames=True)
    assert fr.shape[2] == 65                                                                     # the TDNN halves the frame rate
    # the segment pooling: the last, shorter segment is averaged over the frames that exist
    x = torch.arange(50.0).view(1, 1, 50)
    seg = ofn._seg_pool(x, 20)
    assert torch.allclose(seg[0, 0, :20], torch.full((20,), 9.5)) and torch.allclose(seg[0, 0, 40:], torch.full((10,), 44.5))
    full = CamPlusShape()
    n_par = sum(v.numel() for k, v in make_campplus_weights(full, 0).items() if "running" not in k)
    assert 6.5e6 < n_par < 7.5e6                                                                  # CAM++ is published as a 7.2 M-parameter network


def test_manifest_check_and_weight_files(tmp_path):
    cfg = CamPlusShape.tiny()
    sd = make_campplus_weights(cfg, 5)
    want = manifest(sd)
    assert load_frontend_weights(str(tmp_path), "campplus", want) is None
    torch.save(sd, tmp_path / "campplus.pt")
    got = load_frontend_weights(str(tmp_path), "campplus", want)
    assert set(got) == set(want) and all(torch.equal(got[k], sd[k]) for k in want)
    bad = dict(sd)
    bad.pop("head.conv1.weight")
    bad["xvector.tdnn.linear.weight"] = bad["xvector.tdnn.linear.weight"][:, :, :3]
    with pytest.raises(ValueError) as ei:
        check_against_manifest(bad, want, "campplus.pt")
    assert "missing head.conv1.weight" in str(ei.value) and "xvector.tdnn.linear.weight: shape" in str(ei.value)
    # the same tensors as ONNX initializers
    from astts.onnx_weights import write_initializers
    os.remove(tmp_path / "campplus.pt")
    write_initializers(str(tmp_path / "campplus.onnx"), [(k, v.numpy()) for k, v in sd.items()])
    got = load_frontend_weights(str(tmp_path), "campplus", want)
    assert all(torch.equal(got[k], sd[k]) for k in want)


def test_onnx_initializer_reader_against_googles_protobuf_runtime(tmp_path):
    """The reader decodes what the official protobuf runtime ENCODES for the onnx.proto messages it needs (descriptors built here
    from the field numbers of onnx.proto3: ModelProto.graph = 7, GraphProto.node = 1 / initializer = 5, TensorProto dims = 1,
    data_type = 2, float_data = 4, int64_data = 7, name = 8, raw_data = 9; NodeProto output = 2 / op_type = 4 / attribute = 5;
    AttributeProto name = 1 / t = 5) -- an encoder independent of astts.onnx_weights' own writer."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    from astts.onnx_weights import OnnxFormatError, read_initializers, write_initializers

    fd = descriptor_pb2.FileDescriptorProto(name="onnx_subset.proto", package="onnxs", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = ".onnxs." + tname
    R, O = T.LABEL_REPEATED, T.LABEL_OPTIONAL
    msg("TensorProto", [("dims", 1, T.TYPE_INT64, R, None), ("data_type", 2, T.TYPE_INT32, O, None), ("float_data", 4, T.TYPE_FLOAT, R, None),
                        ("int64_data", 7, T.TYPE_INT64, R, None), ("name", 8, T.TYPE_STRING, O, None), ("raw_data", 9, T.TYPE_BYTES, O, None)])
    msg("AttributeProto", [("name", 1, T.TYPE_STRING, O, None), ("t", 5, T.TYPE_MESSAGE, O, "TensorProto")])
    msg("NodeProto", [("input", 1, T.TYPE_STRING, R, None), ("output", 2, T.TYPE_STRING, R, None), ("op_type", 4, T.TYPE_STRING, O, None),
                      ("attribute", 5, T.TYPE_MESSAGE, R, "AttributeProto")])
    msg("GraphProto", [("node", 1, T.TYPE_MESSAGE, R, "NodeProto"), ("name", 2, T.TYPE_STRING, O, None),
                       ("initializer", 5, T.TYPE_MESSAGE, R, "TensorProto")])
    msg("ModelProto", [("ir_version", 1, T.TYPE_INT64, O, None), ("producer_name", 2, T.TYPE_STRING, O, None),
                       ("graph", 7, T.TYPE_MESSAGE, O, "GraphProto")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Model = message_factory.GetMessageClass(pool.FindMessageTypeByName("onnxs.ModelProto"))
    rng = np.random.default_rng(0)
    a = rng.standard_normal((4, 3, 5)).astype(np.float32)
    b = rng.standard_normal((6,)).astype(np.float32)
    c = np.arange(-4, 8, dtype=np.int64).reshape(3, 4)
    h = rng.standard_normal((2, 9)).astype(np.float16)
    k = rng.standard_normal((3, 3)).astype(np.float32)
    m = Model(ir_version=8, producer_name="pytorch")
    t = m.graph.initializer.add(name="encoder.conv1.weight", data_type=1, dims=list(a.shape), raw_data=a.tobytes())
    t = m.graph.initializer.add(name="b", data_type=1, dims=list(b.shape))
    t.float_data.extend(b.tolist())                              # the typed (packed) field instead of raw_data
    t = m.graph.initializer.add(name="c", data_type=7, dims=list(c.shape))
    t.int64_data.extend(c.reshape(-1).tolist())
    m.graph.initializer.add(name="h", data_type=10, dims=list(h.shape), raw_data=h.tobytes())
    node = m.graph.node.add(op_type="Constant", output=["onnx::MatMul_77"])
    node.attribute.add(name="value").t.CopyFrom(type(t)(name="", data_type=1, dims=list(k.shape), raw_data=k.tobytes()))
    m.graph.node.add(op_type="Relu", input=["x"], output=["y"])
    path = tmp_path / "m.onnx"
    path.write_bytes(m.SerializeToString())
    got = read_initializers(str(path))
    assert set(got) == {"encoder.conv1.weight", "b", "c", "h", "onnx::MatMul_77"}
    for name, want in (("encoder.conv1.weight", a), ("b", b), ("c", c), ("h", h), ("onnx::MatMul_77", k)):
        assert got[name].dtype == want.dtype and np.array_equal(got[name], want), name
    assert "onnx::MatMul_77" not in read_initializers(str(path), constants=False)
    # and the other direction: what the package's writer emits parses with the official runtime
    write_initializers(str(tmp_path / "w.onnx"), [("a", a), ("c", c)], raw=False)
    m2 = Model()
    m2.ParseFromString((tmp_path / "w.onnx").read_bytes())
    assert [t.name for t in m2.graph.initializer] == ["a", "c"] and list(m2.graph.initializer[0].dims) == [4, 3, 5]
    assert np.array_equal(np.asarray(m2.graph.initializer[0].float_data, np.float32).reshape(a.shape), a)
    assert list(m2.graph.initializer[1].int64_data) == c.reshape(-1).tolist()
    (tmp_path / "junk.onnx").write_bytes(b"\x08\x08")
    with pytest.raises(OnnxFormatError):
        read_initializers(str(tmp_path / "junk.onnx"))
