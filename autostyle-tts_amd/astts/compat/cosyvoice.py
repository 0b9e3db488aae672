"""``CosyVoice`` / ``load_wav`` call surface (tts_with_rag.py:1-2,159,195).  Placeholder until the
synthesis kernels land: constructing it raises, nothing is faked."""


class CosyVoice:
    def __init__(self, model_dir, **kw):
        raise NotImplementedError("astts synthesis path is not built yet in this revision")


def load_wav(path, target_sr):
    raise NotImplementedError("astts synthesis path is not built yet in this revision")
