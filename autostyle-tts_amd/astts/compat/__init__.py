"""Drop-in shims for the two third-party object APIs the reference's hot path sits behind:
``pymilvus.MilvusClient`` and ``cosyvoice.cli.cosyvoice.CosyVoice`` (+ ``load_wav``).

``install()`` registers alias modules so that the reference's own import lines
(``from pymilvus import MilvusClient`` -- milvus/search_embeddings.py:3;
``from cosyvoice.cli.cosyvoice import CosyVoice`` -- tts_with_rag.py:1-2) resolve to this build.
"""
import sys
import types


def install(pymilvus: bool = True, cosyvoice: bool = True) -> None:
    if pymilvus and "pymilvus" not in sys.modules:
        from . import pymilvus as _pm

        sys.modules["pymilvus"] = _pm
    if cosyvoice and "cosyvoice" not in sys.modules:
        from . import cosyvoice as _cv

        root = types.ModuleType("cosyvoice")
        cli = types.ModuleType("cosyvoice.cli")
        utils = types.ModuleType("cosyvoice.utils")
        cli_cv = types.ModuleType("cosyvoice.cli.cosyvoice")
        cli_cv.CosyVoice = _cv.CosyVoice
        fu = types.ModuleType("cosyvoice.utils.file_utils")
        fu.load_wav = _cv.load_wav
        root.cli, root.utils, cli.cosyvoice, utils.file_utils = cli, utils, cli_cv, fu
        sys.modules.update({"cosyvoice": root, "cosyvoice.cli": cli, "cosyvoice.utils": utils,
                            "cosyvoice.cli.cosyvoice": cli_cv, "cosyvoice.utils.file_utils": fu})
