"""RAG-TTS driver: the build's restatement of /root/reference/tts_with_rag.py (same flags, same JSONL input,
same output naming and sample rate), running on the MI355X engine through the CosyVoice call surface.

    python -m astts.cli.tts_with_rag --corresponding_json search_results.json --result_dir out

Reference behaviour kept (file:line in /root/reference/tts_with_rag.py):
  * input = JSONL of retrieval records {zh_text, speaker, retrieved_file_id, retrieved_text, distance, whisper} (:85-94)
  * speaker in {w1, w2, m1, m2} picks the timbre wav (:66-75); any other speaker is an error (the reference
    dies with UnboundLocalError, here a KeyError naming the speaker)
  * whisper rows use one fixed timbre wav (:179-182)
  * result dir gets a ``_%m%d%H%M`` suffix (:165-168); files ``{cnt}_{style_id}_to_{speaker}_{i}.wav`` at 22 050 Hz (:196-197)
  * ``--is_exp`` is ``type=bool``: any non-empty string is True (:230); the exp branch needs arguments its parser
    never defines (dead code in the reference, :98-148) and is not reproduced.
Additions (the reference hard-codes /apdcephfs_cq10 paths): --model_dir, --timbre_dir, --whisper_timbre_wav, and
--batch_size N: rows are independent, so N of them share one ragged GPU batch (left-padded LM prefixes, per-row EOS
windows, masked flow matching); file names and contents layout are unchanged; --seed S: row cnt draws its randomness from its
own stream (S, cnt) instead of the process-wide generator, so a row's audio does not depend on the rows before it.

Data-parallel form (BASELINE config 4; the reference's loop :172-197 has no cross-row state):

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m astts.cli.tts_with_rag --corresponding_json ... (same flags)

one process per GPU, model replicated; rank r synthesises rows [r ceil(Q/W), (r+1) ceil(Q/W)) and writes ITS OWN wavs under the
global row number ``cnt`` into the one result directory (rank 0's time stamp, broadcast); no audio crosses GPUs.  Rows are always
seeded per row here (--seed, default 0): the files equal those of the one-process run with the same --seed and --batch_size.
"""
import argparse
import json
import os
from datetime import datetime

REF_TIMBRE_DIR = "/apdcephfs_cq10/share_1615176/cq2/rodenluo/tts_vc/test_data/rag_test/youtube4/timbre"
REF_MODEL_DIR = "/apdcephfs_cq10/share_1615176/cq2/rodenluo/CosyVoice/pretrained_models/CosyVoice-300M"
TIMBRE_FILES = {
    "w1": "engagement_0h0m4dot0s_0h0m9dot51s_w1.wav",
    "w2": "engagement_0h0m10dot51s_0h0m13dot0s_w2.wav",
    "m1": "engagement_0h0m16dot73s_0h0m21dot34s_m1.wav",
    "m2": "engagement_0h0m13dot18s_0h0m16dot73s_m2.wav",
}
WHISPER_TIMBRE_FILE = "engagement_0h0m46dot41s_0h0m48dot99s.wav"


def get_timbre_wav_path(speaker, timbre_dir=REF_TIMBRE_DIR):
    if speaker not in TIMBRE_FILES:
        raise KeyError(f"speaker {speaker!r} has no timbre wav (known: {sorted(TIMBRE_FILES)})")
    return os.path.join(timbre_dir, TIMBRE_FILES[speaker])


def get_text_and_wav(corr_json_path, timbre_dir=REF_TIMBRE_DIR):
    """JSONL -> work items, field for field as the reference builds them (:77-96)."""
    data_list = []
    with open(corr_json_path, "r", encoding="utf-8") as file:
        for line in file:
            if not line.strip():
                continue
            data = json.loads(line)
            speaker = data.get("speaker")
            data_list.append({
                "is_whisper": data.get("whisper"),
                "tts_text": data.get("zh_text"),
                "speaker": speaker,
                "timbre_wav_path": get_timbre_wav_path(speaker, timbre_dir),
                "style_wav_path": data.get("retrieved_file_id"),
                "style_wav_text": data.get("retrieved_text"),
            })
    return data_list


def output_name(cnt, style_wav_path, speaker, i):
    style_wav_fileid = os.path.basename(style_wav_path)[:-4]     # the reference strips 4 chars, extension or not (:189)
    return f"{cnt}_{style_wav_fileid}_to_{speaker}_{i}.wav"


def tts_for_infer(args, cosyvoice=None, now=None):
    from astts import audio, parallel
    from astts.compat.cosyvoice import CosyVoice, load_wav

    dist, rank, world, local = parallel.init_from_env()
    if cosyvoice is None:
        kw = {}
        if dist is not None and dist.get_backend() == "nccl":
            import torch
            torch.cuda.set_device(local)
            kw["device"] = torch.device("cuda", local)
        cosyvoice = CosyVoice(args.model_dir, allow_random_init=True if getattr(args, "allow_random_init", False) else None, **kw)
    stamp = parallel.broadcast_object((now or datetime.now()).strftime("%m%d%H%M"), dist)      # one directory for all ranks
    result_dir = args.result_dir + "_" + stamp
    os.makedirs(result_dir, exist_ok=True)
    written = []
    all_items = get_text_and_wav(args.corresponding_json, args.timbre_dir)
    seed = getattr(args, "seed", None)
    if dist is not None and seed is None:
        seed = 0                     # ranks share nothing: without per-row streams every rank would replay the same draws
    first, last, _ = parallel.shard_bounds(len(all_items), world, rank)
    items = all_items[first:last]    # this rank's rows; global row number = first + position + 1

    def row_seed(cnt):
        return None if seed is None else int(seed) * 1000003 + cnt

    def wavs_of(item):
        style_wav = load_wav(item["style_wav_path"], 16000)
        timbre_path = args.whisper_timbre_wav if item["is_whisper"] else item["timbre_wav_path"]
        return style_wav, load_wav(timbre_path, 16000)

    bs = max(1, int(getattr(args, "batch_size", 1)))

    def rows_of_this_rank():
        if bs == 1:     # the reference's schedule: one utterance at a time (tts_with_rag.py:172-197)
            for cnt, item in enumerate(items, start=first + 1):
                print(item)
                style_wav, timbre_wav = wavs_of(item)
                kw = {} if seed is None else {"seed": row_seed(cnt)}
                for i, j in enumerate(cosyvoice.inference_tts_with_st(item["tts_text"], item["style_wav_text"], style_wav,
                                                                      timbre_wav, stream=False, **kw)):
                    path = os.path.join(result_dir, output_name(cnt, item["style_wav_path"], item["speaker"], i))
                    audio.write_wav(path, j["tts_speech"], 22050)
                    written.append(path)
            return
        # batched schedule: same files, same names; rows are independent so they share ragged GPU batches of `bs` rows.  The surface is
        # handed up to 256 rows at a time: it sorts them by length, decodes their 32-row LM jobs on two streams and renders one group
        # while the next ones decode (CosyVoice.synthesize_batch)
        step = max(bs, 256)
        for c0 in range(0, len(items), step):
            chunk = items[c0:c0 + step]
            for item in chunk:
                print(item)
            reqs = [(it["tts_text"], it["style_wav_text"], *wavs_of(it)) for it in chunk]
            kw = {} if seed is None else {"seeds": [row_seed(first + c0 + k + 1) for k in range(len(chunk))]}
            for k, segs in enumerate(cosyvoice.inference_tts_with_st_batch(reqs, max_batch=bs, **kw)):
                for i, j in enumerate(segs):
                    path = os.path.join(result_dir, output_name(first + c0 + k + 1, chunk[k]["style_wav_path"], chunk[k]["speaker"], i))
                    audio.write_wav(path, j["tts_speech"], 22050)
                    written.append(path)

    # a rank whose rows fail (a bad wav path, an allocation) must not leave its peers waiting: the block ends in one agreed flag
    with parallel.rank_work(dist, "tts_with_rag"):
        rows_of_this_rank()
    return written


def build_parser():
    parser = argparse.ArgumentParser(description="Generate vc result from style_dir to timbre_dir.")
    parser.add_argument("--corresponding_json", required=True, help="include text style timbre information")
    parser.add_argument("--result_dir", required=True, help="path to save results")
    parser.add_argument("--is_exp", type=bool, default=False, help="path to save results")
    parser.add_argument("--model_dir", default=REF_MODEL_DIR)
    parser.add_argument("--allow_random_init", action="store_true",
                        help="run on seeded random weights when model_dir holds no llm.pt / flow.pt / hift.pt (otherwise that is an error)")
    parser.add_argument("--timbre_dir", default=REF_TIMBRE_DIR)
    parser.add_argument("--whisper_timbre_wav", default=os.path.join(REF_TIMBRE_DIR, WHISPER_TIMBRE_FILE))
    parser.add_argument("--batch_size", type=int, default=1, help="rows synthesised per ragged GPU batch (1 = the reference's schedule)")
    parser.add_argument("--seed", type=int, default=None,
                        help="per-row random streams (row cnt draws from (seed, cnt)); always on in a data-parallel run (default 0 there)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    print("flag:", args.is_exp)
    if args.is_exp:
        raise SystemExit("--is_exp: the reference's experimental branch reads arguments its own parser never defines "
                         "(tts_with_rag.py:110-113 vs :224-230); it cannot run there either and is not reproduced")
    print("---not exp---")
    return tts_for_infer(args)


if __name__ == "__main__":
    main()
    from astts import parallel
    parallel.shutdown()
