"""Dialogue driver: restatement of /root/reference/tts_for_dialog.py (same flags, same data formats, same file names).

    python -m astts.cli.tts_for_dialog --corresponding_json map.json --dialogue_json dialogue.jsonl \\
        --style_wav_json styles.jsonl --style_wav_dir wavs --result_dir out --timbre_map speakers.json

Reference behaviour (file:line in tts_for_dialog.py):
  * ``--dialogue_json`` / ``--style_wav_json`` are JSON-lines files (:27-35); lookups are 1-based: entry ``index`` is
    ``data[index - 1]`` (``zh_text`` :38-42, ``file_id`` :54-55).
  * ``--corresponding_json`` maps a dialogue line number (string key) to ``{"value": style line number, "speaker": name,
    "emotion": ...}`` or the string ``"null"`` (skipped, :175-176 -- the counter only advances for synthesised lines).
  * per entry: ``inference_tts_with_st(zh_text, style_wav_text, style_wav@16k, timbre_wav@16k)`` (:188), segment ``i``
    saved as ``{result_dir}_{MMDDHHMM}/{cnt}_{style_file_id}_to_{speaker}_{i}.wav`` at 22 050 Hz (:160-163,189-190).
  * ``--is_exp`` (:74-143): zero-shot with the style prompt, saved as ``{file_id}_prompt_{cnt}_{i}.wav``, then the saved audio
    is resampled to 16 kHz and voice-converted to the speaker's timbre (``{file_id}_{cnt}_to_{speaker}_exp_{i}.wav``).
DIVERGENCES, documented: the reference hard-codes one timbre wav per speaker name (:44-52, two names, absolute paths of
the authors' cluster) and the model directory (:152); here they are ``--timbre_map`` (JSON: speaker -> wav path) and
``--model_dir``.  In ``--is_exp`` mode the reference re-loads ``result_wav_path`` WITHOUT the ``_{i}.wav`` suffix it
saved under (:118-121) -- a path that does not exist; this driver converts the audio it just synthesised (all segments
concatenated) and keeps ``cnt`` at 0 as the reference does (it never increments it in that mode).
"""
import argparse
import json
import os
from datetime import datetime

REF_MODEL_DIR = "/apdcephfs_cq10/share_1615176/cq2/rodenluo/CosyVoice/pretrained_models/CosyVoice-300M"


class JsonDataReader:
    """JSON-lines table with the reference's 1-based lookups."""

    def __init__(self, file_path):
        self.file_path = file_path
        with open(file_path, "r", encoding="utf-8") as f:
            self.data = [json.loads(ln) for ln in (x.strip() for x in f) if ln]

    def get_zh_text_by_index(self, index):
        if 0 <= index <= len(self.data):
            return self.data[index - 1]["zh_text"]
        return "索引超出范围"          # the reference's out-of-range sentinel (:42)

    def get_fileid(self, value):
        return self.data[value - 1]["file_id"]


def _entries(args):
    with open(args.corresponding_json, "r", encoding="utf-8") as f:
        mapping = json.load(f)
    with open(args.timbre_map, "r", encoding="utf-8") as f:
        timbre_map = json.load(f)
    dialogue = JsonDataReader(args.dialogue_json)
    styles = JsonDataReader(args.style_wav_json)
    for key, value in mapping.items():
        if value == "null":
            continue
        speaker = value["speaker"]
        if speaker not in timbre_map:
            raise KeyError(f"speaker {speaker!r} has no entry in --timbre_map")
        file_id = styles.get_fileid(int(value["value"]))
        yield (dialogue.get_zh_text_by_index(int(key)), styles.get_zh_text_by_index(int(value["value"])), file_id,
               os.path.join(args.style_wav_dir, file_id + ".wav"), speaker, timbre_map[speaker])


def tts_for_infer(args, cosyvoice=None):
    from astts import audio
    from astts.compat.cosyvoice import CosyVoice, load_wav

    cosyvoice = cosyvoice or CosyVoice(args.model_dir, allow_random_init=True if getattr(args, "allow_random_init", False) else None)
    result_dir = args.result_dir + "_" + (args.time_tag or datetime.now().strftime("%m%d%H%M"))
    os.makedirs(result_dir, exist_ok=True)
    written = []
    bs = max(1, int(getattr(args, "batch_size", 1)))
    entries = list(_entries(args))
    wav_cache = {}

    def wav16(path):            # one tensor per file: the batched surface featurises each distinct prompt tensor once
        if path not in wav_cache:
            wav_cache[path] = load_wav(path, 16000)
        return wav_cache[path]

    if bs == 1:     # the reference's schedule (tts_for_dialog.py:172-190)
        for cnt, (zh_text, style_text, file_id, style_path, speaker, timbre_path) in enumerate(entries, start=1):
            print(zh_text, style_text, speaker)
            for i, j in enumerate(cosyvoice.inference_tts_with_st(zh_text, style_text, wav16(style_path), wav16(timbre_path), stream=False)):
                path = os.path.join(result_dir, f"{cnt}_{file_id}_to_{speaker}_{i}.wav")
                audio.write_wav(path, j["tts_speech"], 22050)
                written.append(path)
        return written
    for c0 in range(0, len(entries), bs):       # turns are independent: `batch_size` of them share ragged GPU batches
        chunk = entries[c0:c0 + bs]
        for zh_text, style_text, _, _, speaker, _ in chunk:
            print(zh_text, style_text, speaker)
        out = cosyvoice.inference_tts_with_st_batch([(e[0], e[1], wav16(e[3]), wav16(e[5])) for e in chunk], max_batch=bs)
        for k, segs in enumerate(out):
            for i, j in enumerate(segs):
                path = os.path.join(result_dir, f"{c0 + k + 1}_{chunk[k][2]}_to_{chunk[k][4]}_{i}.wav")
                audio.write_wav(path, j["tts_speech"], 22050)
                written.append(path)
    return written


def tts_for_exp(args, cosyvoice=None):
    import torch

    from astts import audio
    from astts.compat.cosyvoice import CosyVoice, load_wav

    cosyvoice = cosyvoice or CosyVoice(args.model_dir, allow_random_init=True if getattr(args, "allow_random_init", False) else None)
    os.makedirs(args.result_dir, exist_ok=True)
    written = []
    cnt = 0
    for zh_text, style_text, file_id, style_path, speaker, timbre_path in _entries(args):
        style_wav = load_wav(style_path, 16000)
        timbre_wav = load_wav(timbre_path, 16000)
        print(zh_text, style_text, speaker)
        segs = []
        for i, j in enumerate(cosyvoice.inference_zero_shot(zh_text, style_text, style_wav, stream=False)):
            path = os.path.join(args.result_dir, f"{file_id}_prompt_{cnt}_{i}.wav")
            audio.write_wav(path, j["tts_speech"], 22050)
            written.append(path)
            segs.append(j["tts_speech"])
        if not segs:
            continue
        src16 = audio.resample(torch.cat(segs, dim=1), 22050, 16000)
        for i, j in enumerate(cosyvoice.inference_vc(src16, timbre_wav, stream=False)):
            path = os.path.join(args.result_dir, f"{file_id}_{cnt}_to_{speaker}_exp_{i}.wav")
            audio.write_wav(path, j["tts_speech"], 22050)
            written.append(path)
    return written


def build_parser():
    p = argparse.ArgumentParser(description="Generate vc result from style_dir to timbre_dir.")
    p.add_argument("--corresponding_json", required=True, help="include text style timbre information")
    p.add_argument("--dialogue_json", required=True, help="provide text")
    p.add_argument("--style_wav_json", required=True, help="provide style_wav")
    p.add_argument("--style_wav_dir", required=True, help="style wav dir")
    p.add_argument("--result_dir", required=True, help="path to save results")
    p.add_argument("--is_exp", type=bool, default=False, help="two-stage zero-shot -> vc experiment")
    p.add_argument("--timbre_map", required=True, help="JSON {speaker: 16 kHz-loadable wav path} (the reference hard-codes two)")
    p.add_argument("--model_dir", default=REF_MODEL_DIR, help="CosyVoice model directory (the reference hard-codes it)")
    p.add_argument("--allow_random_init", action="store_true",
                   help="run on seeded random weights when model_dir holds no llm.pt / flow.pt / hift.pt (otherwise that is an error)")
    p.add_argument("--time_tag", default=None, help="suffix of the result directory (default: now as MMDDHHMM, as the reference)")
    p.add_argument("--batch_size", type=int, default=1, help="dialogue turns per ragged GPU batch (1 = the reference's schedule)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    print("flag:", args.is_exp)
    if args.is_exp:
        print("---exp---")
        return tts_for_exp(args)
    print("---not exp---")
    return tts_for_infer(args)


if __name__ == "__main__":
    main()
