"""Fixed style + timbre driver: restatement of /root/reference/tts_with_style_and_timbre.py (same flags).

    python -m astts.cli.tts_with_style_and_timbre --style_wav_path s.wav --timbre_wav_path t.wav \\
        --style_wav_text "..." --txt_path lines.txt --result_dir out

Reference behaviour (file:line in tts_with_style_and_timbre.py): one synthesis per text line (:91-93), output
``{style}_{cnt}_to_{timbre}.wav`` at 22 050 Hz (:94-95).  DIVERGENCE, documented: the reference's file name has no
``{}`` placeholder, so ``.format(i)`` is a no-op and every segment of a multi-segment line overwrites the previous
one -- only the last segment survives on disk.  This driver concatenates the segments of a line into that one file
(nothing is lost); ``--keep_last_segment_only`` restores the reference's on-disk result exactly.
``--is_exp`` (two-stage zero-shot -> VC, :23-63) is reproduced: zero-shot with the style prompt, then
inference_vc of the 16 kHz resampled result against the timbre wav.
"""
import argparse
import os

REF_MODEL_DIR = "/apdcephfs_cq10/share_1615176/cq2/rodenluo/CosyVoice/pretrained_models/CosyVoice-300M"


def get_text(txt_path):
    with open(txt_path, "r", encoding="utf-8") as file:
        return [line.strip() for line in file.readlines()]


def tts_for_infer(args, cosyvoice=None):
    import torch

    from astts import audio
    from astts.compat.cosyvoice import CosyVoice, load_wav

    cosyvoice = cosyvoice or CosyVoice(args.model_dir, allow_random_init=True if getattr(args, "allow_random_init", False) else None)
    style = os.path.basename(args.style_wav_path)[:-4]
    timbre = os.path.basename(args.timbre_wav_path)[:-4]
    lines = get_text(args.txt_path)
    style_wav = load_wav(args.style_wav_path, 16000)
    timbre_wav = load_wav(args.timbre_wav_path, 16000)
    os.makedirs(args.result_dir, exist_ok=True)
    written = []
    for cnt, line in enumerate(lines, start=1):
        segs = [j["tts_speech"] for j in cosyvoice.inference_tts_with_st(line, args.style_wav_text, style_wav, timbre_wav, stream=False)]
        if not segs:
            continue
        wav = segs[-1] if args.keep_last_segment_only else torch.cat(segs, dim=1)
        path = os.path.join(args.result_dir, f"{style}_{cnt}_to_{timbre}.wav")
        audio.write_wav(path, wav, 22050)
        written.append(path)
    return written


def tts_for_exp(args, cosyvoice=None):
    import torch

    from astts import audio
    from astts.compat.cosyvoice import CosyVoice, load_wav

    cosyvoice = cosyvoice or CosyVoice(args.model_dir, allow_random_init=True if getattr(args, "allow_random_init", False) else None)
    style = os.path.basename(args.style_wav_path)[:-4]
    timbre = os.path.basename(args.timbre_wav_path)[:-4]
    style_wav = load_wav(args.style_wav_path, 16000)
    timbre_wav = load_wav(args.timbre_wav_path, 16000)
    os.makedirs(args.result_dir, exist_ok=True)
    written = []
    for cnt, line in enumerate(get_text(args.txt_path), start=1):
        segs = [j["tts_speech"] for j in cosyvoice.inference_zero_shot(line, args.style_wav_text, style_wav, stream=False)]
        styled = torch.cat(segs, dim=1)
        p1 = os.path.join(args.result_dir, f"{style}_prompt_{cnt}.wav")
        audio.write_wav(p1, styled, 22050)
        styled16 = audio.resample(styled, 22050, 16000)
        for i, j in enumerate(cosyvoice.inference_vc(styled16, timbre_wav, stream=False)):
            p2 = os.path.join(args.result_dir, f"{style}_{cnt}_to_{timbre}_exp_{i}.wav")
            audio.write_wav(p2, j["tts_speech"], 22050)
            written.append(p2)
    return written


def build_parser():
    parser = argparse.ArgumentParser(description="Generate vc result from style_dir to timbre_dir.")
    parser.add_argument("--style_wav_path", required=True, help="style wav (exp:pdd or jj_ljq)")
    parser.add_argument("--timbre_wav_path", required=True, help="timbre wav (exp:test80)")
    parser.add_argument("--style_wav_text", required=True, help="style text ")
    parser.add_argument("--txt_path", required=True, help="text for tts")
    parser.add_argument("--result_dir", required=True, help="path to save results")
    parser.add_argument("--is_exp", type=bool, default=False, help="path to save results")
    parser.add_argument("--model_dir", default=REF_MODEL_DIR)
    parser.add_argument("--allow_random_init", action="store_true",
                        help="run on seeded random weights when model_dir holds no llm.pt / flow.pt / hift.pt (otherwise that is an error)")
    parser.add_argument("--keep_last_segment_only", action="store_true")
    return parser


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
