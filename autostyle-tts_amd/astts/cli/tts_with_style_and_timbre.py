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

    from astts import parallel

    dist, rank, world, local = parallel.init_from_env()
    if cosyvoice is None:
        kw = {}
        if dist is not None and dist.get_backend() == "nccl":
            torch.cuda.set_device(local)
            kw["device"] = torch.device("cuda", local)
        cosyvoice = CosyVoice(args.model_dir, allow_random_init=True if getattr(args, "allow_random_init", False) else None, **kw)
    style = os.path.basename(args.style_wav_path)[:-4]
    timbre = os.path.basename(args.timbre_wav_path)[:-4]
    all_lines = get_text(args.txt_path)
    first, last, _ = parallel.shard_bounds(len(all_lines), world, rank)      # data-parallel run: this rank's lines (global numbering kept)
    lines = all_lines[first:last]
    style_wav = load_wav(args.style_wav_path, 16000)
    timbre_wav = load_wav(args.timbre_wav_path, 16000)
    os.makedirs(args.result_dir, exist_ok=True)
    seed = getattr(args, "seed", None)
    if dist is not None and seed is None:
        seed = 0
    written = []

    def save(cnt, segs):
        if not segs:
            return
        wav = segs[-1] if args.keep_last_segment_only else torch.cat(segs, dim=1)
        path = os.path.join(args.result_dir, f"{style}_{cnt}_to_{timbre}.wav")
        audio.write_wav(path, wav, 22050)
        written.append(path)

    bs = max(1, int(getattr(args, "batch_size", 1)))

    def lines_of_this_rank():
        if bs == 1:     # the reference's schedule: one line at a time (tts_with_style_and_timbre.py:91-95)
            for cnt, line in enumerate(lines, start=first + 1):
                kw = {} if seed is None else {"seed": int(seed) * 1000003 + cnt}
                save(cnt, [j["tts_speech"] for j in cosyvoice.inference_tts_with_st(line, args.style_wav_text, style_wav, timbre_wav, stream=False, **kw)])
        else:           # lines are independent: the text segments of `batch_size` lines share ragged GPU batches (BASELINE config 3)
            step = max(bs, 256)     # the surface schedules up to 256 lines' segments itself (LM jobs on two streams, render overlapped)
            for c0 in range(0, len(lines), step):
                chunk = lines[c0:c0 + step]
                kw = {} if seed is None else {"seeds": [int(seed) * 1000003 + first + c0 + k + 1 for k in range(len(chunk))]}
                if getattr(args, "fixed_tokens", None):
                    kw["fixed_tokens"] = int(args.fixed_tokens)
                out = cosyvoice.inference_tts_with_st_batch([(line, args.style_wav_text, style_wav, timbre_wav) for line in chunk],
                                                            max_batch=getattr(args, "max_segments", None) or 64, **kw)
                for k, segs in enumerate(out):
                    save(first + c0 + k + 1, [j["tts_speech"] for j in segs])

    with parallel.rank_work(dist, "tts_with_style_and_timbre"):      # one agreed flag in place of a barrier: no rank waits for a failed peer
        lines_of_this_rank()
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
    parser.add_argument("--batch_size", type=int, default=1,
                        help="lines synthesised together: their text segments share ragged GPU batches (1 = the reference's schedule)")
    parser.add_argument("--max_segments", type=int, default=64, help="text segments per ragged GPU batch in the batched schedule")
    parser.add_argument("--fixed_tokens", type=int, default=None,
                        help="batched schedule only: every text segment decodes exactly this many speech tokens (EOS ignored) -- "
                             "fixed-length throughput runs, e.g. 250 for BASELINE config 3")
    parser.add_argument("--seed", type=int, default=None,
                        help="per-line random streams (line cnt draws from (seed, cnt)); always on in a data-parallel run (default 0 there)")
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
    from astts import parallel
    parallel.shutdown()
