"""Style x timbre x text sweep: restatement of /root/reference/vc_from_dir.py and vc_from_dir_seed.py (same flags, same
file names, same ``meta.lst``).

    python -m astts.cli.vc_from_dir --txt_path lines.txt --style_dir styles --timbre_dir timbres --result_dir out \\
        --style_num 2 --timbre_num 3 --style_json styles.json

Reference behaviour (file:line in vc_from_dir.py):
  * ``--style_num`` / ``--timbre_num`` files are drawn from the two directories with ``random.sample`` (:18-28); more than
    available is an error.
  * the style wav's transcript is the ``zh_text`` of the entry of a style JSON (a list) whose ``file_id`` is
    ``"denoise_" + <wav stem>`` (:35-47).
  * for every (style, timbre, text line): ``inference_tts_with_st(line, style_text, style@16k, timbre@16k)`` (:198),
    saved as ``{style}_to_{timbre}_{cnt}_new.wav`` at 22 050 Hz (:199-201), and one ``meta.lst`` row
    ``{name}|{style_text}|{timbre_wav_path}|{line}`` (:203-217).
DIVERGENCES, documented: the style JSON (:186) and the model directory (:91) are hard-coded absolute paths in the
reference; here ``--style_json`` / ``--model_dir``.  ``--seed`` makes the draw reproducible (the reference uses the global
RNG unseeded).  As in the other drivers the reference's file name has no ``{}`` placeholder, so every segment of a
multi-segment line overwrites the previous one; this driver concatenates the segments (``--keep_last_segment_only``
restores the on-disk result of the reference).  ``--style_meta_lst`` is the vc_from_dir_seed.py variant of the style
source (:60-77 there): rows ``name|text|wav|...`` of a seed-tts ``meta.lst``; ``--style_num`` of them are sampled and each
row's text is the style transcript.
"""
import argparse
import json
import os
import random

REF_MODEL_DIR = "/apdcephfs_cq10/share_1615176/cq2/rodenluo/CosyVoice/pretrained_models/CosyVoice-300M"


def get_path(directory, num, rng=random):
    if not os.path.isdir(directory):
        raise ValueError(f"'{directory}' is not a directory")
    files = sorted(os.path.join(directory, f) for f in os.listdir(directory) if os.path.isfile(os.path.join(directory, f)))
    if num > len(files):
        raise ValueError(f"{num} files requested, {len(files)} available in {directory}")
    return rng.sample(files, num)


def get_text(txt_path):
    with open(txt_path, "r", encoding="utf-8") as f:
        return [line.strip() for line in f.readlines()]


def get_style_wav_text(json_path, file_id):
    with open(json_path, "r", encoding="utf-8") as f:
        for entry in json.load(f):
            if entry["file_id"] == "denoise_" + file_id:
                return entry["zh_text"]
    raise KeyError(f"no entry with file_id 'denoise_{file_id}' in {json_path}")


def get_style_and_text(lst_path, num, rng=random):
    """vc_from_dir_seed.py: (wav path, transcript) pairs from a ``name|text|wav|...`` list."""
    rows = []
    with open(lst_path, "r", encoding="utf-8") as f:
        for line in f:
            parts = line.strip().split("|")
            if len(parts) >= 4:
                rows.append((parts[2], parts[1]))
    return rng.sample(rows, num) if len(rows) >= num else rows


def main(argv=None, cosyvoice=None):
    import torch

    from astts import audio
    from astts.compat.cosyvoice import CosyVoice, load_wav

    args = build_parser().parse_args(argv)
    rng = random.Random(args.seed) if args.seed is not None else random
    cosyvoice = cosyvoice or CosyVoice(args.model_dir, allow_random_init=True if getattr(args, "allow_random_init", False) else None)
    lines = get_text(args.txt_path)
    os.makedirs(args.result_dir, exist_ok=True)
    if args.style_meta_lst:
        base = os.path.dirname(os.path.abspath(args.style_meta_lst))
        styles = [(p if os.path.isabs(p) else os.path.join(base, p), t) for p, t in get_style_and_text(args.style_meta_lst, args.style_num, rng)]
    else:
        if not args.style_json:
            raise SystemExit("--style_json (or --style_meta_lst) is required")
        styles = [(p, get_style_wav_text(args.style_json, os.path.basename(p)[:-4])) for p in get_path(args.style_dir, args.style_num, rng)]
    timbres = get_path(args.timbre_dir, args.timbre_num, rng)
    rows = []
    bs = max(1, int(args.batch_size))
    for style_path, style_text in styles:
        style_wav = load_wav(style_path, 16000)
        style = os.path.basename(style_path)[:-4]
        for timbre_path in timbres:
            timbre_wav = load_wav(timbre_path, 16000)
            timbre = os.path.basename(timbre_path)[:-4]
            for c0 in range(0, len(lines), bs):
                chunk = lines[c0:c0 + bs]
                if bs == 1:     # the reference's schedule (vc_from_dir.py:196-201)
                    outs = [list(cosyvoice.inference_tts_with_st(chunk[0], style_text, style_wav, timbre_wav, stream=False))]
                else:           # the lines of one (style, timbre) pair share ragged GPU batches
                    outs = cosyvoice.inference_tts_with_st_batch([(line, style_text, style_wav, timbre_wav) for line in chunk], max_batch=64)
                for k, (line, out) in enumerate(zip(chunk, outs)):
                    segs = [j["tts_speech"] for j in out]
                    name = f"{style}_to_{timbre}_{c0 + k + 1}_new"
                    if segs:
                        wav = segs[-1] if args.keep_last_segment_only else torch.cat(segs, dim=1)
                        audio.write_wav(os.path.join(args.result_dir, name + ".wav"), wav, 22050)
                    rows.append([name, style_text, timbre_path, line])
    meta = os.path.join(args.result_dir, "meta.lst")
    with open(meta, "w", encoding="utf-8") as f:
        for row in rows:
            f.write("|".join(row) + "\n")
    print(f"wrote {meta}")
    return rows


def build_parser():
    p = argparse.ArgumentParser(description="Generate vc result from style_dir to timbre_dir.")
    p.add_argument("--txt_path", required=True, help="tts text")
    p.add_argument("--style_dir", required=True, help="style dir")
    p.add_argument("--timbre_dir", required=True, help="timbre dir")
    p.add_argument("--result_dir", required=True, help="path to save results")
    p.add_argument("--style_num", type=int, required=True, help="number of style wavs drawn")
    p.add_argument("--timbre_num", type=int, required=True, help="number of timbre wavs drawn")
    p.add_argument("--style_json", default=None, help="JSON list of {file_id, zh_text} (hard-coded path in the reference)")
    p.add_argument("--style_meta_lst", default=None, help="vc_from_dir_seed.py variant: name|text|wav|... list of style prompts")
    p.add_argument("--model_dir", default=REF_MODEL_DIR, help="CosyVoice model directory (hard-coded in the reference)")
    p.add_argument("--allow_random_init", action="store_true",
                   help="run on seeded random weights when model_dir holds no llm.pt / flow.pt / hift.pt (otherwise that is an error)")
    p.add_argument("--seed", type=int, default=None, help="seed of the file draw (default: unseeded, as the reference)")
    p.add_argument("--keep_last_segment_only", action="store_true", help="reproduce the reference's overwrite of multi-segment lines")
    p.add_argument("--batch_size", type=int, default=1, help="lines of one (style, timbre) pair synthesised together (1 = the reference's schedule)")
    return p


if __name__ == "__main__":
    main()
