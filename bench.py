#!/usr/bin/env python3
"""bench.py -- headline benchmark of the AutoStyle-TTS hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1 is launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``).  Rank 0 prints ONE
JSON line.

A step = one pass of the hot path over one batch of synthetic input, BASELINE.json configs[1]
(SURVEY.md 8d "Config 2"): style-kNN of Q=8 queries against the 1k-entry bank (k=3), then fixed-length
synthesis of the 8 utterances: Tt=32 text tokens, 3.0 s style/timbre prompts (Tp=150 tokens, Tm_p=258
mel frames), forced Ts=250 speech tokens (5.0 s -> Tm=430 frames -> L=110080 samples at 22 050 Hz),
LM decode -> flow matching (10 Euler steps x CFG) -> HiFT vocoder.  Weights are seeded random-init at
the CosyVoice-300M shapes (no checkpoints exist offline), inputs are resident in HBM when the timed
region starts, the waveform stays on the GPU.  value = synthesized audio seconds / wall seconds,
whole job.  The retrieval leg is additionally reported as queries/s (``knn_qps``).
"""
import argparse
import dataclasses
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "autostyle-tts_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# one hardware queue per command-processor pipe (astts/_lib.py sets the same default and says why; it must precede the first HIP call)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak


def make_config2_bank(n=1000, d=6144, seed=1234):
    """real 130 rows tiled + 0.05*N(0,1), re-rounded to fp16."""
    real = np.load(os.path.join(ROOT, "tests", "golden", "style_bank_130x6144.f16.npy"))
    rng = np.random.default_rng(seed)
    bank = real[np.arange(n) % real.shape[0]].astype(np.float32)
    bank = bank[:, :d] + 0.05 * rng.standard_normal((n, d)).astype(np.float32)
    return bank.astype(np.float16)


def make_queries(bank, nq, seed):
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, bank.shape[0], nq)
    return bank[rows].astype(np.float32) + 0.5 * rng.standard_normal((nq, bank.shape[1])).astype(np.float32)


class SynthInputs:
    """Synthetic fixed-length batch (SURVEY 8d config 2 shapes), all on the GPU."""

    def __init__(self, cfg, b, tt, tp, ts, dev, seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        self.b, self.ts = b, ts
        self.text = torch.randint(0, cfg.text_vocab, (b, tt), device=dev, generator=g)
        self.tlen = torch.full((b,), tt, dtype=torch.int32, device=dev)
        self.spk_style = torch.randn(b, cfg.spk_dim, device=dev, generator=g)
        self.spk_timbre = torch.randn(b, cfg.spk_dim, device=dev, generator=g)
        self.style_tok = torch.randint(0, cfg.speech_vocab, (b, tp), device=dev, generator=g)
        self.timbre_tok = torch.randint(0, cfg.speech_vocab, (b, tp), device=dev, generator=g)
        self.tmp = cfg.mel_frames_for_tokens(tp)
        self.tm = cfg.mel_frames_for_tokens(ts)
        self.timbre_mel = torch.randn(b, self.tmp, cfg.mel, device=dev, generator=g)
        self.u = torch.rand(ts, b, 2, device=dev, generator=g)
        self.z = torch.randn(b, self.tmp + self.tm, cfg.mel, device=dev, generator=g)
        nh = cfg.nb_harmonics + 1
        self.phase0 = (torch.rand(b, nh, device=dev, generator=g) * 2 - 1) * math.pi
        self.phase0[:, 0] = 0
        self.noise = torch.randn(b, self.tm * cfg.upsample_total, nh, device=dev, generator=g)
        self.audio_seconds = b * self.tm * cfg.upsample_total / cfg.sample_rate


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, weights, bank16, q_host, k, ts=250):
    """The oracle (kind "port": this build's fp32 PyTorch-CPU restatement, oracle/) timed on the host cores on a BOUNDED sample
    of the same workload: ONE full-length utterance of the config-2 batch (B=1 of the 8: Tt=32, 150-token prompt, Ts=250 speech
    tokens = 5 s of audio, the same shapes the GPU step runs for each of its 8 rows) plus the retrieval of the same 8 queries."""
    from oracle import knn as oknn
    from oracle import synth as osyn

    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    b, tt, tp = 1, 32, 150
    text = torch.randint(0, cfg.text_vocab, (b, tt), generator=g)
    tlen = torch.full((b,), tt)
    spk = torch.randn(b, cfg.spk_dim, generator=g)
    tok_p = torch.randint(0, cfg.speech_vocab, (b, tp), generator=g)
    tmp, tm = cfg.mel_frames_for_tokens(tp), cfg.mel_frames_for_tokens(ts)
    mel_p = torch.randn(b, tmp, cfg.mel, generator=g)
    u = torch.rand(ts, b, 2, generator=g)
    z = torch.randn(b, tmp + tm, cfg.mel, generator=g)
    nh = cfg.nb_harmonics + 1
    ph = torch.zeros(b, nh)
    noise = torch.randn(b, tm * cfg.upsample_total, nh, generator=g)
    with torch.no_grad():
        # The decode loop is 250 x 14 layers of one-row GEMVs: with all 64 threads of the GPU host it ran 5x SLOWER than with 8
        # (fork / join per small operator).  Untimed calibration: a few decode steps per thread count, the fastest one decodes.
        pre = osyn.lm_prefix(weights["llm"], cfg, text, tlen, spk, tok_p)
        lm_threads, best = threads, float("inf")
        for n in sorted({min(threads, c) for c in (4, 8, 16, 32, 64)}):
            torch.set_num_threads(n)
            osyn.lm_decode(weights["llm"], cfg, pre, 2, u[:2], True, None)          # warm: position tables, allocator
            tc = time.perf_counter()
            osyn.lm_decode(weights["llm"], cfg, pre, 4, u[:4], True, None)
            tc = time.perf_counter() - tc
            if tc < best:
                lm_threads, best = n, tc
        torch.set_num_threads(threads)
    t0 = time.perf_counter()
    with torch.no_grad():
        oknn.knn_search_fast_f32(bank16.astype(np.float32),
                                 (1.0 / np.linalg.norm(bank16.astype(np.float64), axis=1)).astype(np.float32), q_host, k)
        tk = time.perf_counter()
        torch.set_num_threads(lm_threads)
        pre = osyn.lm_prefix(weights["llm"], cfg, text, tlen, spk, tok_p)
        toks, _ = osyn.lm_decode(weights["llm"], cfg, pre, ts, u, True, None)
        torch.set_num_threads(threads)
        t1 = time.perf_counter()
        mel = osyn.flow_decode(weights["flow"], cfg, torch.cat([tok_p, toks.long()], 1), torch.full((b,), tp + ts), mel_p, spk, z, tmp + tm)
        t2 = time.perf_counter()
        wav = osyn.hift_forward(weights["hift"], cfg, mel, ph, noise)
    t3 = time.perf_counter()
    audio = wav.shape[1] / cfg.sample_rate
    return {"value": audio / (t3 - t0), "unit": "audio-s/wall-s, ONE utterance at a time (the reference's own loop: tts_with_rag.py:172-197)",
            "batch": 1, "gpu_leg_batch": 8,
            "comparison_note": "the GPU leg runs 8 utterances per step, this leg one (8 in a row would take 8x as long, not run 8x as fast: the "
                               "CPU decode is one-row GEMVs either way): a reported baseline beside `value`, not a like-for-like ratio",
            "cores": threads, "threads_by_stage": {"lm": lm_threads, "flow": threads, "vocoder": threads},
            "cpu": cpu_model_string(), "kind": "port",
            "sample": f"one full-length utterance of the batch (B=1, Tt={tt}, {tp}-token prompt, Ts={ts} tokens = {audio:.2f} s of audio) "
                      f"+ kNN of the 8 queries; oracle/ fp32 torch-CPU",
            "stage_seconds": {"knn": round(tk - t0, 4), "lm": round(t1 - tk, 2), "flow": round(t2 - t1, 2), "vocoder": round(t3 - t2, 2)}}


_JSON_OUT = None


def json_only_stdout():
    """Reserve the process's stdout for the ONE JSON line.  Libraries write there too -- RCCL prints its version banner
    ("RCCL version : ...", five lines) to C stdout, and with a pipe for stdout those lines are flushed at exit, i.e. AFTER the
    JSON line.  From here on file descriptor 1 is stderr; `emit_json` writes to the descriptor stdout had."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _JSON_OUT


def emit_json(obj):
    out = json_only_stdout()
    out.write(json.dumps(obj) + "\n")
    out.flush()


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, timeout=None):
    """Self-launcher for `python bench.py --gpus N` (no torch.distributed.run around it): one child process per GPU with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, same command line.  The parent never touches HIP (no
    torch.cuda call before or after: children are fresh interpreters started with subprocess, nothing is exec'ed from a
    process that has initialised the GPU).  Rank 0's stdout (the ONE JSON line) is relayed; every child's stderr passes
    through.  Returns 0 only if every rank exited 0; a failing rank takes the others down."""
    import subprocess

    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)   # keeps rank 0's pipe drained
    reader.start()
    t_end = None if timeout is None else time.time() + timeout
    rc = 0
    try:
        while rc == 0 and any(p.poll() is None for p in procs):
            time.sleep(0.05)
            rc = next((p.returncode for p in procs if p.poll() not in (None, 0)), 0)
            if rc == 0 and t_end is not None and time.time() > t_end:
                rc = 124
        rc = rc or next((p.returncode for p in procs if p.returncode not in (None, 0)), 0)
    finally:
        for p in procs:                 # exact PIDs this function started, never a pattern
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    out0 = b"".join(c for c in chunks if c)
    for line in out0.decode(errors="replace").splitlines():      # stdout carries the JSON line only; library chatter -> stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
    sys.stdout.flush()
    return rc


def stub_main(args, world, rank):
    """ASTTS_BENCH_STUB=1 (CPU tests of the launcher / timing protocol only, never a measurement): the same rendezvous,
    barrier, max-over-ranks timing and one-JSON-line protocol as main(), over gloo, with a sleep as the step."""
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group(backend="gloo")
    for _ in range(args.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, rank)
    else:
        gathered = [0]
    if os.environ.get("ASTTS_BENCH_STUB_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    if rank == 0:
        emit_json({"metric": "stub", "value": args.steps * world / dt, "unit": "steps/s", "n_gpus": world,
                   "rccl_world_size": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                   "ranks_seen": gathered, "data": "stub (launcher test)"})
    if world > 1:
        dist.destroy_process_group()
    return 0


def _profiled(ops, kind, fn, max_launches=40000):
    ops.prof_enable(kind, True, max_launches)
    fn()
    torch.cuda.synchronize()
    ms, n, work, dropped = ops.prof_read(kind)
    ops.prof_enable(kind, False)
    return {"ms": ms, "launches": n, "work": work, "dropped": dropped}


PROF_KINDS = {"gemm_tile": (0, "mfma"), "lm_gemv": (1, "hbm"), "attn_mha_flash": (2, "mfma"), "lm_attn": (3, "hbm")}


def kind_rooflines(ops, fn, dist=None, dev=None, traffic_table=None, max_launches=40000):
    """One pass of `fn` (ONE sequential unit of the workload: one batch on one stream) per profiled kernel kind, every launch of the kind
    timed on its own stream.  Returns (dominant single kernel's roofline, {kind: roofline}).  `gemm_tile` is a FAMILY (ring / tile GEMMs,
    the fused transformer-block, ResNet and vocoder convolution kernels: work = flops); the dominant KERNEL is the decode GEMV whenever
    it takes more than half of that family's time (the rule of the headline line)."""
    out = {}
    for name, (kind, bound) in PROF_KINDS.items():
        p = _profiled(ops, kind, fn, max_launches)
        ms = torch.tensor([p["ms"]], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(ms, op=dist.ReduceOp.MAX)
        n = max(p["launches"], 1)
        us = p["ms"] * 1e3 / n
        per = p["work"] / n
        if bound == "mfma":
            ach, peak, unit = (per / (us * 1e-6) / 1e12 if us > 0 else 0.0), MFMA_F16_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = (per / (us * 1e-6) / 1e9 if us > 0 else 0.0), HBM_PEAK_GBS, "GB/s"
        out[name] = {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "kernel": name, "avg_us": us,
                     "launches": p["launches"], "algorithmic_work_per_launch": per, "ms_total": p["ms"], "ms_total_max_over_ranks": float(ms.item()),
                     "dropped": p["dropped"],
                     "traffic": (traffic_table or {}).get(name, {}).get("hbm_bytes_per_launch")}
    tot = {k: v["ms_total_max_over_ranks"] for k, v in out.items()}
    dom = max(tot, key=tot.get)
    if dom == "gemm_tile" and tot["lm_gemv"] > 0.5 * tot["gemm_tile"]:
        dom = "lm_gemv"
    return dict(out[dom]), out


def load_traffic_table():
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f), name
        except OSError:
            pass
    return {}, None


_BANK100K = {}


def config5_bank(dim):
    """N = 100 000 rows ~ N(0, 1) rounded to fp16, seed 1234 (generated once per process: 2.4 GB of fp32 on the host)."""
    if dim not in _BANK100K:
        _BANK100K[dim] = np.random.default_rng(1234).standard_normal((100000, dim), dtype=np.float32).astype(np.float16)
    return _BANK100K[dim]


def knn_stress(args, dev, traffic_table):
    """BASELINE configs[4]'s retrieval leg on ONE GPU, each shape against ITS roofline: 100k x 6144 with Q = 8 (HBM: the bank
    is read once per search), Q = 256 (the GEMM scan: MFMA; the bank bytes per scan launch come from the committed FETCH_SIZE pass) and
    100k x 768 with Q = 256.  `scan_us` = HIP events around the scan launch(es) of a search (astts_knn_profile_*), `search_us` = wall time per
    whole search (prep + scan + select + fp64 re-score, back to back on one stream).  ids checked against oracle/knn.py on a sample."""
    from astts.knn import StyleBank
    from oracle import knn as oknn

    out = {}
    for dim, qs in ((6144, (8, 256)), (768, (256,))):
        bank = config5_bank(dim)
        sb = StyleBank(bank, device=dev)
        for nq in qs:
            q_host = make_queries(bank, nq, seed=0)
            q_dev = torch.from_numpy(q_host).to(dev)
            oi = torch.empty((nq, args.topk), dtype=torch.int64, device=dev)
            osc = torch.empty((nq, args.topk), dtype=torch.float32, device=dev)
            for _ in range(40):         # (a chip coming out of idle runs its first milliseconds at a lower clock: 5 warm-up searches read 12 % low)
                sb.search_device(q_dev, args.topk, out_idx=oi, out_score=osc)
            torch.cuda.synchronize()
            n_it = 100
            t0 = time.perf_counter()
            for _ in range(n_it):
                sb.search_device(q_dev, args.topk, out_idx=oi, out_score=osc)
            torch.cuda.synchronize()
            search_us = (time.perf_counter() - t0) / n_it * 1e6
            sb.profile_enable(True)
            for _ in range(n_it):
                sb.search_device(q_dev, args.topk, out_idx=oi, out_score=osc)
            torch.cuda.synchronize()
            ms, n_launch = sb.profile_read()
            sb.profile_enable(False)
            scan_us = ms * 1e3 / n_it                      # all scan launches of one search (two 128-query groups at Q = 256 before round 5)
            nbytes = bank.shape[0] * dim * 2 + nq * dim * 4 + nq * args.topk * 12      # bank read ONCE per search
            flops = 2.0 * nq * bank.shape[0] * dim
            sel = np.arange(0, nq, max(1, nq // 7))
            ok = bool(np.array_equal(oi.cpu().numpy()[sel], oknn.knn_search(bank, q_host[sel], args.topk)[0]))
            key = f"knn_scan_100000_x_{dim}_q{nq}"
            traffic = traffic_table.get(key, {}).get("hbm_bytes_per_launch")
            launches = max(n_launch // n_it, 1)
            hbm = {"bound": "hbm", "achieved": nbytes / (scan_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": nbytes / (scan_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
            mfma = {"bound": "mfma", "achieved": flops / (scan_us * 1e-6) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flops / (scan_us * 1e-6) / 1e12 / MFMA_F16_PEAK_TFLOPS}
            roof = dict(hbm if nq < 64 else mfma)
            roof.update({"kernel": "knn_scan" if nq < 64 else "gemm_ring (the scan as one GEMM)", "avg_us": scan_us / launches, "scan_launches_per_search": launches,
                         "algorithmic_bytes_per_search": nbytes, "algorithmic_flops_per_search": flops,
                         "traffic": traffic * launches if traffic else None, "traffic_per_launch": traffic,
                         "other_bound": mfma if nq < 64 else hbm})
            out[f"N100000_D{dim}_Q{nq}"] = {"qps": nq / (search_us * 1e-6), "search_us": search_us, "scan_us": scan_us, "k": args.topk,
                                           "ids_match_oracle_sample": ok, "roofline": roof}
        del sb
    return out


def run_side(args, which, dev, dist, rank, world, cfg, weights, eng=None, steps=1, warmup=1, traffic_table=None):
    """BASELINE.json's other configurations, measured with the default line's protocol (W untimed warm-up steps, K timed steps between
    barrier + synchronize, MAX over ranks).  Returns the record on rank 0 (None elsewhere).
      config3  64 long-form lines = 384 text segments (Tt = 64, Ts = 250: 30 s per line) per step, as twelve 32-row batches through
               the stream pipeline; weak scaling (every rank its own 64 lines).  tts_with_style_and_timbre.py:91-95.
      config4  the 1 623 IEMOCAP test sentences, Ts_i = clamp(round(20 words_i), 25, 1500) forced, STRONG scaling: rank r takes rows
               [r ceil(Q/W), ...) through CosyVoice.inference_tts_with_st_batch (prompt wavs -> frontend -> ragged LM / flow / vocoder ->
               CPU waveforms) after the query-sharded retrieval of all 1 623 queries + one all-gather of the ids.  tts_with_rag.py:172-197.
      config5  256 queries against the 100 000 x 6144 bank (bank-sharded over the ranks when there are several) + 256 utterances at
               the config-2 shapes, STRONG scaling over the ranks, 32-row batches through the stream pipeline.  milvus/search_json.py:382-411."""
    from astts import ops, parallel
    from astts.knn import StyleBank
    from astts.synth.model import PipelinedSynth, SynthEngine

    traffic_table = traffic_table or {}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, warm_fn=None):
        for _ in range(warmup):
            (warm_fn or step_fn)()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    extra = {}
    if which in ("config3", "config5"):
        eng = eng if eng is not None else SynthEngine(weights, cfg, dev)
        rows = 32
        if which == "config3":
            n_batches, tt, scaling = 12, 64, "weak"
            total_rows = n_batches * rows * world
        else:
            b0, b1, _ = parallel.shard_bounds(256, world, rank)
            n_batches, tt, scaling = (b1 - b0 + rows - 1) // rows, args.text_tokens, "strong"
            total_rows = 256
            rows = min(rows, max(b1 - b0, 1))
            bank = config5_bank(args.dim)
            r0, r1, _ = parallel.shard_bounds(bank.shape[0], world, rank)
            sb = StyleBank(bank[r0:r1] if dist is not None else bank, device=dev)
            q_host = make_queries(bank, 256, seed=0)
            q_dev = torch.from_numpy(q_host).to(dev)

            def search():
                if dist is None:
                    return sb.search_device(q_dev, args.topk)[0]
                return parallel.bank_sharded_search(lambda qq, kk: (lambda i, _s, s64: (i, s64))(*sb.search_device(qq, kk, return_f64=True)),
                                                    q_dev, args.topk, r0, dist)[0]
        inp = SynthInputs(cfg, rows, tt, args.prompt_tokens, args.speech_tokens, dev, seed=100 + rank)
        sample = (inp.text, inp.tlen, inp.spk_style, inp.style_tok, inp.ts, inp.u, inp.timbre_tok, inp.timbre_mel, inp.spk_timbre,
                  inp.z, inp.phase0, inp.noise)
        # Schedules tried by the calibration: 2 / 3 decode chains of one 32-row batch each (rounds 3-4), and -- round 5 -- the LM stages of 4
        # or 8 consecutive batches co-batched into ONE 128- / 256-row chain on the engine's wide path (plain GEMMs: the weights are read
        # once per token for all rows; AcousticLM.decode(wide=True)), every 32-row batch still rendered on its own.
        wide = os.environ.get("ASTTS_BENCH_WIDE", "1") != "0" and rows == 32
        cands = ((2, 1), (3, 1)) + (((3, 2), (2, 4), (3, 4), (2, 8)) if wide else ())      # (one wide chain alone -- (1, 4), (1, 8) -- measured 10 % behind two)
        pipe = PipelinedSynth.autotune(eng, sample, depths=cands, trials=1 if steps <= 2 else 2, steps=8 if wide else (4 if steps <= 2 else 6), dist=dist,
                                       wide_lm=wide)
        last = {}

        def step():
            if which == "config5":
                last["ids"] = search()
            for _ in range(n_batches):
                pipe.submit(*sample)
            for d in pipe.drain():
                last["wav"] = d[2]

        with torch.cuda.stream(pipe.front_stream):
            dt = timed(step)
        audio = inp.audio_seconds / rows * total_rows * steps
        ok = bool(torch.isfinite(last["wav"]).all())
        extra["decode_chains"] = pipe.depth
        extra["batches_per_decode_chain"] = pipe.cobatch
        extra["lm_rows_per_chain"] = pipe.cobatch * rows
        extra["calibration_ms_per_batch"] = getattr(pipe, "tuned_table_ms", None)
        del pipe
        # one sequential 32-row batch: stage times, then the kernel kinds against their rooflines
        e0 = ev()
        toks = eng.tts_tokens(*sample[:6])
        e1 = ev()
        all_tok = torch.cat([inp.timbre_tok.to(torch.int32), toks], 1)
        tl = torch.full((inp.b,), all_tok.shape[1], dtype=torch.int32, device=dev)
        mel = eng.flow.decode(all_tok, tl, inp.timbre_mel, inp.spk_timbre, inp.z, inp.tmp + inp.tm)
        e2 = ev()
        eng.hift.forward(mel, inp.phase0, inp.noise)
        e3 = ev()
        torch.cuda.synchronize()
        extra["stages_ms_one_sequential_batch"] = {"rows": rows, "lm_ms": round(e0.elapsed_time(e1), 3), "flow_ms": round(e1.elapsed_time(e2), 3),
                                                    "vocoder_ms": round(e2.elapsed_time(e3), 3)}
        roof, by_kind = kind_rooflines(ops, lambda: eng.tts(*sample), dist, dev, traffic_table)
        roof["mode"] = f"one sequential {rows}-row batch on one stream (kernel-level dispatch timestamps for the decode kernels, HIP events for the rest)"
        if which == "config5":
            from oracle import knn as oknn
            sel = np.arange(0, 256, 37)
            extra["ids_match_oracle_sample"] = bool(np.array_equal(last["ids"].cpu().numpy()[sel], oknn.knn_search(bank, q_host[sel], args.topk)[0]))
            del sb
        work = (f"BASELINE configs[2]: 64 long-form lines/GPU = 384 text segments (Tt={tt}, Tp={args.prompt_tokens}, Ts={args.speech_tokens}: 30 s per line) "
                f"per step as {n_batches} batches of {rows} rows" if which == "config3" else
                f"BASELINE configs[4]: kNN Q=256 x N=100000 x D={args.dim} k={args.topk} ({'bank-sharded over the ranks, one all-gather + merge' if dist is not None else 'one GPU holds the bank'}) "
                f"+ 256 utterances (Tt={tt}, Tp={args.prompt_tokens}, Ts={args.speech_tokens}) over {world} GPU(s) as batches of {rows} rows")
    else:       # config4
        from astts.compat.cosyvoice import CosyVoice

        with open(os.path.join(ROOT, "tests", "golden", "iemocap_test_sentences.json")) as f:
            sents = json.load(f)["all"]
        want = [int(min(max(round(20 * len(x.split())), 25), 1500)) for x in sents]
        bank = make_config2_bank(args.bank_rows, args.dim)
        sb = StyleBank(bank, device=dev)
        q_all = torch.from_numpy(make_queries(bank, len(sents), seed=4)).to(dev)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cv = CosyVoice("/nonexistent", config=cfg, seed=0, device=dev, allow_random_init=True, engine=eng)
        if os.environ.get("ASTTS_BENCH_WIDE", "1") != "0":      # throughput run: LM jobs of 48 rows on the engine's wide path (CosyVoice.wide_lm)
            # (rows per LM job on the wide engine's round-5 kernels: 32 (decode-step kernels) 581x; 40: 674; 48: 687; 56: 681; 64: 662; 96: 647;
            # 128: 624 -- fatter chains cost fewer LM stream-seconds and more flow seconds beside them, profiles/r05_config4_lm_rows.log)
            cv.wide_lm, cv.lm_rows = True, int(os.environ.get("ASTTS_BENCH_LM_ROWS", "48"))
        extra["lm_rows_per_job"] = cv.lm_rows
        g = torch.Generator().manual_seed(0)
        t16 = torch.arange(int(2.5 * 16000)) / 16000
        style = (0.3 * torch.sin(2 * math.pi * 220 * t16) + 0.01 * torch.randn(t16.shape, generator=g))[None]
        timbre = (0.3 * torch.sin(2 * math.pi * 330 * t16[:32000]) + 0.01 * torch.randn(32000, generator=g))[None]
        b0, b1, _ = parallel.shard_bounds(len(sents), world, rank)
        items = [(x, "He did. In Niagara Falls.", style, timbre) for x in sents[b0:b1]]
        last = {}

        def run_rows(sel_rows):
            t_a = time.perf_counter()
            outs = cv.inference_tts_with_st_batch([items[i] for i in sel_rows], max_batch=32, split=False,
                                                  fixed_tokens=[want[b0 + i] for i in sel_rows], seeds=[b0 + i for i in sel_rows])
            last["audio"] = sum(o[0]["tts_speech"].shape[1] for o in outs) / cfg.sample_rate
            last["outs"] = outs                 # every waveform is on the host here; the checker's pass over them runs after the timed region
            last["host_s"] = {"batch_surface": round(time.perf_counter() - t_a, 3)}

        def step():
            cv._stage_events, cv._host_times = [], []              # keep the LAST timed pass's stage events only
            idx, _ = parallel.sharded_search(lambda qq, kk: sb.search_device(qq, kk)[:2], q_all, args.topk, dist)
            last["ids"] = idx
            run_rows(range(len(items)))

        def warm():          # bounded warm-up: every 8th row of the shard (allocator, position tables, every kernel variant)
            parallel.sharded_search(lambda qq, kk: sb.search_device(qq, kk)[:2], q_all, args.topk, dist)
            run_rows(range(0, len(items), 8))

        cv.collect_stage_times = True
        dt = timed(step, warm if steps <= 1 else None)
        st_s = cv.stage_seconds()          # the last timed pass; booked per stream
        cv.collect_stage_times = False
        extra["stage_stream_seconds"] = {k: round(v, 3) for k, v in st_s.items()}
        extra["host_seconds_of_the_timed_pass"] = last.get("host_s")
        extra["stage_note"] = ("HIP events around every LM job's decode (two worker streams) and every render group's flow / vocoder pass (render stream), "
                               "summed per stage over the last timed pass: the streams overlap in wall time, the sums exceed it")
        a = torch.tensor([last["audio"]], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(a)
        audio = float(a.item()) * steps
        t_chk = time.perf_counter()
        ok = all(bool(torch.isfinite(o[0]["tts_speech"]).all()) and float(o[0]["tts_speech"].abs().max()) <= cfg.audio_limit + 1e-6 for o in last.pop("outs"))
        last["host_s"]["finite_and_clamp_check_of_every_waveform_untimed"] = round(time.perf_counter() - t_chk, 3)
        scaling = "strong"
        from oracle import knn as oknn
        sel = np.arange(0, len(sents), 101)
        extra["ids_match_oracle_sample"] = bool(np.array_equal(last["ids"].cpu().numpy()[sel], oknn.knn_search(bank, q_all.cpu().numpy()[sel], args.topk)[0]))
        extra["host_side_included"] = ("prompt wavs -> GPU resample / log-mel frontend, tokenisation, ragged LM / flow / vocoder, D2H of every waveform "
                                       "(asynchronous copies, all landed when the timed region ends); the checker's finite / clamp pass over the waveforms is untimed")
        # rooflines: one ragged render group (32 rows of the shard, every 8th row: lengths 25 ... several hundred tokens) through the same surface
        prof_rows = list(range(0, len(items), max(1, len(items) // 32)))[:32]
        roof, by_kind = kind_rooflines(ops, lambda: run_rows(prof_rows), dist, dev, traffic_table, max_launches=120000)
        roof["mode"] = "one ragged group of 32 sentences of the shard through CosyVoice.inference_tts_with_st_batch (LM jobs on two streams, render overlapped)"
        work = (f"BASELINE configs[3]: the 1 623 IEMOCAP test sentences (Ts_i = clamp(round(20 words_i), 25, 1500), {sum(want)} speech tokens) sharded over {world} GPU(s), "
                f"kNN Q=1623 x N={args.bank_rows} x D={args.dim} k={args.topk} query-sharded + all-gather of the ids, ragged synthesis in length-bucketed groups of 32 "
                f"through CosyVoice.inference_tts_with_st_batch")
        del cv, sb
    if rank != 0:
        return None
    res = {"value": audio / dt, "unit": "audio-s/wall-s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
           "scaling": scaling, "dtype": "f16", "data": "synthetic", "workload": work + "; CosyVoice-300M shapes, random-init weights",
           "audio_seconds_per_step": audio / steps, "waveform_finite": ok, "roofline": roof,
           "roofline_by_kind": {k: {kk: v[kk] for kk in ("bound", "achieved", "peak", "unit", "frac", "avg_us", "launches", "algorithmic_work_per_launch", "ms_total", "dropped", "traffic")}
                                for k, v in by_kind.items()}}
    res.update(extra)
    return res


def side_extras(args, dev, cfg, eng):
    """Measurements of the rows next to the path that ride in the default line: filled in by the functions below as
    those rows are measured (rank 0 only, bounded to a few seconds each)."""
    out = {}
    for name, fn in (("embedder", bench_embedder), ("streaming", bench_streaming)):
        try:
            out[name] = fn(args, dev, cfg, eng)
        except Exception as e:      # noqa: BLE001  a side probe must never take the headline line down; the failure is reported in its place
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def bench_host_io(args, dev, cfg, eng, sb, pipe, inp, q_host, out_idx, out_sc, dist, barrier, world):
    """The headline workload with its host I/O and its frontend inside the timed region (see the call site).  Returns the dict that
    rides in the line as `host_io` (`value` = `value_with_host_io`)."""
    import statistics

    from astts.frontend import Frontend
    from astts.parallel import gather_style_ids

    b = args.batch
    fe = Frontend.from_model_dir("/nonexistent", cfg, dev, weights_loaded=False)
    g = torch.Generator().manual_seed(11)
    n16 = int(args.prompt_tokens / cfg.token_rate * 16000)
    t16 = torch.arange(n16) / 16000.0
    wavs = [(0.3 * torch.sin(2 * math.pi * (180.0 + 17.0 * i) * t16) + 0.02 * torch.randn(n16, generator=g))[None].pin_memory() for i in range(2 * b)]
    q_pin = torch.from_numpy(q_host).pin_memory()
    text_host = inp.text.cpu()

    def featurise():
        feats = fe.prompts(wavs)                               # host waveforms -> (speech tokens, speaker vector, prompt mel): one GPU batch, one sync
        st, tb = feats[:b], feats[b:]
        n_tok = min(int(f.speech_tokens.shape[1]) for f in feats)
        n_mel = min(int(f.mel.shape[1]) for f in tb)
        to = lambda xs, dt=None: torch.cat(xs, 0).to(dev, dt, non_blocking=True) if dt else torch.cat(xs, 0).to(dev, non_blocking=True)
        return (to([f.spk_embedding for f in st]), to([f.speech_tokens[:, :n_tok] for f in st]), to([f.speech_tokens[:, :n_tok] for f in tb]),
                to([f.mel[:, :n_mel] for f in tb]), to([f.spk_embedding for f in tb]), n_mel)

    spk_s, tok_s, tok_t, mel_t, spk_t, n_mel = featurise()     # (builds the synthetic-weight networks: untimed)
    z = torch.randn(b, n_mel + inp.tm, cfg.mel, device=dev)
    host_out = [None]
    n_done = [0]

    def take(done):
        if done is None:
            return
        for d in (done if isinstance(done, list) else [done]):
            host_out[0] = d[2].to("cpu", non_blocking=False)   # the step's waveforms, device -> host
            n_done[0] += 1

    def step():
        qd = q_pin.to(dev, non_blocking=True)
        sb.search_device(qd, args.topk, out_idx=out_idx, out_score=out_sc)
        if dist is not None:
            gather_style_ids(out_idx, dist)
        spk_s, tok_s, tok_t, mel_t, spk_t, _ = featurise()
        text = text_host.to(dev, non_blocking=True)
        take(pipe.submit(text, inp.tlen, spk_s, tok_s, inp.ts, inp.u, tok_t, mel_t, spk_t, z, inp.phase0, inp.noise))

    k = max(2, min(args.steps, 8))
    with torch.cuda.stream(pipe.front_stream):
        for _ in range(max(1, min(args.warmup, 2))):
            step()
        take(pipe.drain())
        barrier()
        n_done[0] = 0
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        take(pipe.drain())
        barrier()
        dt = time.perf_counter() - t0
    assert n_done[0] == k
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the frontend alone: one prompt, host waveform in, host features out
    per = []
    for _ in range(7):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fe.prompt(wavs[0])
        per.append((time.perf_counter() - t1) * 1e3)
    per16 = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fe.prompts(wavs)
        per16.append((time.perf_counter() - t1) * 1e3)
    audio = b * inp.tm * cfg.upsample_total / cfg.sample_rate
    return {"value": audio * k * world / dt, "unit": "audio-s/wall-s", "steps": k, "ms_per_step": 1e3 * dt / k,
            "frontend_ms_per_prompt": statistics.median(per), "frontend_ms_per_prompt_max": max(per), "prompts_per_step": 2 * b,
            "frontend_ms_per_step_batched": statistics.median(per16),
            "frontend": fe.describe(),
            "h2d_bytes_per_step": int(q_pin.numel() * 4 + sum(w.numel() for w in wavs) * 4 + text_host.numel() * 8),
            "d2h_bytes_per_step": int(host_out[0].numel() * 4), "waveform_finite": bool(torch.isfinite(host_out[0]).all()),
            "includes": "H2D of the query vectors, the 16 prompt waveforms and the text ids; kNN; the GPU frontend of every prompt (resample, prompt "
                        "log-mel, Whisper log-mel -> speech tokenizer (6-block encoder + L2 codebook search), Kaldi fbank -> CAM++) as ONE batch of 16 "
                        "equal-length prompts (Frontend.prompts; features handed over on the host, as the call surface does); LM prefill + decode, flow, vocoder; "
                        "D2H of the waveforms"}


def bench_embedder(args, dev, cfg, eng):
    """SURVEY.md 8f rank 2, the step before the retrieval (milvus/search_json.py:154-229, src/search_milvus.py:75-108): the Llama query
    embedder AS THE REFERENCE RUNS IT -- Llama-3.2-3B: 28 layers, hidden 3072, 24 / 8 heads of 128, FFN 8192, vocabulary 128 256 (3.2 B
    seeded random parameters drawn on the GPU; parity at this depth: tests/test_llm_gpu.py::test_embedder_at_full_depth_*) --
    32 texts of 60 tokens per pass: mean-pooled embeddings per second, and the 10-token greedy label decode with the KV cache."""
    from astts.llm.config import LlamaShape
    from astts.llm.embedder import LlamaEmbedder
    from astts.llm.weights import make_llama_weights

    shape = LlamaShape.llama32_3b() if not args.embedder_layers else dataclasses.replace(LlamaShape.llama32_3b(), layers=args.embedder_layers)
    emb = LlamaEmbedder(make_llama_weights(shape, 0, device=dev), shape, dev)
    b, t, n_new = 32, 60, 10
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, shape.vocab, (b, t), generator=g)
    lens = torch.full((b,), t, dtype=torch.int32)
    for _ in range(3):
        emb.embed_ids(ids, lens)
    torch.cuda.synchronize()
    n_it = 20
    t0 = time.perf_counter()
    for _ in range(n_it):
        e = emb.embed_ids(ids, lens)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_it
    hq, hk = shape.heads * shape.head_dim, shape.kv_heads * shape.head_dim
    per_tok_layer = 2.0 * (shape.hidden * (hq + 2 * hk) + hq * shape.hidden + 3 * shape.hidden * shape.ffn)
    flops = b * t * shape.layers * per_tok_layer + b * shape.layers * 4.0 * shape.heads * shape.head_dim * t * (t + 1) / 2
    # the batch of BASELINE configs[4]'s 256 queries in one pass (15 360 rows per GEMM instead of 1 920)
    b2 = 256
    ids2 = torch.randint(3, shape.vocab, (b2, t), generator=g)
    lens2 = torch.full((b2,), t, dtype=torch.int32)
    emb.embed_ids(ids2, lens2)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    for _ in range(5):
        e2 = emb.embed_ids(ids2, lens2)
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t3) / 5
    flops2 = flops * b2 / b
    prompts = [ids[i].tolist() for i in range(b)]
    emb.generate_greedy_batch(prompts, n_new)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        emb.generate_greedy_batch(prompts, n_new)
    dg = (time.perf_counter() - t1) / 5
    t2 = time.perf_counter()
    emb.generate_greedy_recompute(prompts[0], n_new)
    d1 = time.perf_counter() - t2
    return {"texts_per_s": b / dt, "ms_per_pass": dt * 1e3, "texts": b, "tokens_per_text": t, "layers": shape.layers, "hidden": shape.hidden,
            "vocab": shape.vocab, "parameters": sum(int(w.data.numel()) for L in emb.L for w in (L["wqkv"], L["wo"], L["wgu"], L["wd"])) + int(emb.embed.numel()),
            "finite": bool(torch.isfinite(e).all()),
            "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / dt / 1e12 / MFMA_F16_PEAK_TFLOPS, "algorithmic_flops_per_pass": flops,
                         "note": f"whole embedding pass (embedding lookup, {shape.layers} x [RMSNorm, q|k|v GEMM, RoPE, MFMA causal GQA attention, out GEMM, RMSNorm, "
                                 "gate|up GEMM, SwiGLU, down GEMM], final norm, mean-pool) over the pass's wall time; 1 920 rows per GEMM: launch- and "
                                 "ingest-bound, not MFMA-bound"},
            "greedy_label": {"texts": b, "prompt_tokens": t, "new_tokens": n_new, "ms_per_batch": dg * 1e3, "labels_per_s": b / dg,
                             "ms_one_text_prompt_rerun_per_token": d1 * 1e3,
                             "note": "KV cache + argmax on the device, one host synchronisation per batch (rounds 3-4 re-ran the prompt per token "
                                     "with a host sync each: the last figure, ONE text)"},
            "batch_of_256_texts": {"texts_per_s": b2 / dt2, "ms_per_pass": dt2 * 1e3, "finite": bool(torch.isfinite(e2).all()),
                                   "mfma_frac": flops2 / dt2 / 1e12 / MFMA_F16_PEAK_TFLOPS,
                                   "note": "the same pass over 256 texts (configs[4]'s query batch): 15 360 rows per GEMM"}}


def bench_streaming(args, dev, cfg, eng):
    """stream=True (tts_for_dialog.py:188, vc_from_dir.py:18 pass the flag): time to the FIRST yielded chunk of a 250-token segment through
    CosyVoice.inference_tts_with_st(stream=True), with the LM decoding hop by hop on its own stream while the chunks render (round 5),
    against the one-pass form (decode everything, then chunk: rounds 3-4).  Upstream's schedule needs hop + look-ahead = 120 tokens
    before the first chunk, so 120 / 250 of the decode is the floor of the ratio for this segment length."""
    import warnings

    from astts.compat.cosyvoice import CosyVoice

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cv = CosyVoice("/nonexistent", config=cfg, seed=0, device=dev, allow_random_init=True, engine=eng)
    g = torch.Generator().manual_seed(0)
    t16 = torch.arange(int(3.0 * 16000)) / 16000
    style = (0.3 * torch.sin(2 * math.pi * 220 * t16) + 0.01 * torch.randn(t16.shape, generator=g))[None]
    timbre = (0.3 * torch.sin(2 * math.pi * 330 * t16) + 0.01 * torch.randn(t16.shape, generator=g))[None]
    text, style_text = "I did it, I asked her to marry me.", "He did. In Niagara Falls."

    def run(live, n_tok):
        cv.stream_lm_live = live
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first, n, samples = None, 0, 0
        for out in cv.inference_tts_with_st(text, style_text, style, timbre, stream=True, seed=1, fixed_tokens=n_tok):
            if first is None:
                first = time.perf_counter() - t0
            n += 1
            samples += out["tts_speech"].shape[1]
        return first * 1e3, (time.perf_counter() - t0) * 1e3, n, samples

    import statistics
    res = {}
    for n_tok in (250, 500):
        run(True, n_tok), run(False, n_tok)          # warm both paths (allocator, stream probe, frontend networks)
        lives = [run(True, n_tok) for _ in range(5)]
        onces = [run(False, n_tok) for _ in range(5)]
        live = sorted(lives, key=lambda r: r[0])[len(lives) // 2]
        once = sorted(onces, key=lambda r: r[0])[len(onces) // 2]
        res[f"tokens_{n_tok}"] = {"first_chunk_ms": live[0], "first_chunk_ms_max": max(r[0] for r in lives), "first_chunk_ms_min": min(r[0] for r in lives),
                                  "trials": len(lives), "segment_ms": live[1], "chunks": live[2], "audio_s": live[3] / cfg.sample_rate,
                                  "first_chunk_over_segment": live[0] / live[1],
                                  "one_pass_first_chunk_ms": once[0], "one_pass_first_chunk_ms_max": max(r[0] for r in onces), "one_pass_segment_ms": once[1],
                                  "live_first_chunk_below_0p9_of_one_pass": bool(statistics.median(r[0] for r in lives) < 0.9 * statistics.median(r[0] for r in onces)),
                                  "floor_note": f"120 of {n_tok} decode steps precede the first chunk by upstream's schedule"}
    res["first_chunk_ms"] = res["tokens_250"]["first_chunk_ms"]
    res["first_chunk_independent_of_segment_length_ms"] = abs(res["tokens_500"]["first_chunk_ms"] - res["tokens_250"]["first_chunk_ms"])
    res["statistic"] = "median of 5 trials per form (max / min beside it); the decode stream is probed against the caller's stream (ops.stream_beside)"
    res["includes"] = ("prompt featurisation on the GPU (resample, prompt log-mel, Whisper log-mel -> speech tokenizer, Kaldi fbank -> CAM++ "
                       f"speaker network; {cv.frontend.describe()}), LM prefill, decode, flow + vocoder of the chunk, D2H copy")
    return res


def side_workload(args, which, dev, dist, rank, world):
    """`--workload config3|config4|config5`: one side configuration as a line of its own, with the default line's JSON schema (the default
    `python bench.py` = BASELINE configs[1] carries bounded passes of all three under `side_workloads`)."""
    from astts.synth.config import SynthConfig
    from astts.synth.weights import make_all

    cfg = SynthConfig(sample_rate=args.sample_rate)
    weights = make_all(cfg, seed=0)
    table, _ = load_traffic_table()
    r = run_side(args, which, dev, dist, rank, world, cfg, weights, None, steps=args.steps, warmup=args.warmup, traffic_table=table)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        res = {"metric": "synthesized audio sec/wall-sec (RTF^-1) + style-kNN QPS, IEMOCAP test batch", "value": r["value"], "unit": r["unit"],
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
               "scaling": r["scaling"], "vs_baseline": None, "dtype": "f16", "data": "synthetic",
               "config": {"workload": r["workload"], "parallelism": f"dp{world}"},
               "side_measurement": f"--workload {which}: NOT the headline line (python bench.py = BASELINE configs[1])"}
        res.update({k: v for k, v in r.items() if k not in res and k != "workload"})
        if not args.no_cpu_baseline:
            bank16 = make_config2_bank(args.bank_rows, args.dim)
            res["cpu_baseline"] = cpu_baseline(cfg, weights, bank16, make_queries(bank16, 8, seed=0), args.topk)
        emit_json(res)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--bank-rows", type=int, default=1000)
    ap.add_argument("--dim", type=int, default=6144)
    ap.add_argument("--topk", type=int, default=3)
    ap.add_argument("--text-tokens", type=int, default=32)
    ap.add_argument("--prompt-tokens", type=int, default=150)
    ap.add_argument("--speech-tokens", type=int, default=250)
    ap.add_argument("--sample-rate", type=int, default=22050)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-io", action="store_true", help="skip the host-I/O-inclusive side pass (value_with_host_io, frontend_ms)")
    ap.add_argument("--embedder-layers", type=int, default=0, help="query-embedder side probe: layers of the Llama-3.2-3B shape (0 = all 28)")
    ap.add_argument("--no-24khz", action="store_true", help="skip the 24 kHz side measurement (a second engine at sample_rate 24000)")
    ap.add_argument("--no-cobatch", action="store_true", help="skip the co-batched side measurement (16 / 32-row decode chains): profiling "
                    "runs use it so that the kernel population is the timed region's")
    ap.add_argument("--no-side", action="store_true", help="skip the bounded passes of BASELINE configs[2..4] (side_workloads), the 100k-bank "
                    "retrieval stress (knn_stress), the query embedder and the streaming latency probe that the default line carries")
    ap.add_argument("--workload", default="config2", choices=("config2", "config3", "config4", "config5"),
                    help="config2 (default) = BASELINE configs[1], the headline; config3 / config4 / config5 = side lines for BASELINE configs[2..4] (side_workload)")
    ap.add_argument("--force-dist", action="store_true", default=bool(os.environ.get("ASTTS_BENCH_FORCE_DIST")),
                    help="initialise torch.distributed (backend nccl = RCCL) even for ONE rank, so that the id all-gather and the "
                         "bank-sharded merge run through librccl on a single GPU (also: ASTTS_BENCH_FORCE_DIST=1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it has made no HIP call and makes none)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         f"or plainly as `python bench.py --gpus {args.gpus}` (self-launching)")
    json_only_stdout()        # every rank: nothing but rank 0's JSON line reaches stdout (RCCL's banner, library chatter -> stderr)
    if os.environ.get("ASTTS_BENCH_STUB"):
        return stub_main(args, world, rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                      # --force-dist on one GPU: a one-rank RCCL communicator
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist_mod.init_process_group(backend="nccl", device_id=dev)
        dist = dist_mod

    if args.workload != "config2":
        return side_workload(args, args.workload, dev, dist, rank, world)

    from astts import ops
    from astts.knn import StyleBank
    from astts.parallel import gather_style_ids
    from astts.synth.config import SynthConfig
    from astts.synth.model import PipelinedSynth, SynthEngine
    from astts.synth.weights import make_all

    cfg = SynthConfig(sample_rate=args.sample_rate)
    weights = make_all(cfg, seed=0)                       # identical on every rank (replicated model)
    eng = SynthEngine(weights, cfg, dev)
    bank16 = make_config2_bank(args.bank_rows, args.dim)
    sb = StyleBank(bank16, device=dev)                    # replicated per GPU (12.3 MB)
    q_host = make_queries(bank16, args.batch, seed=rank)  # this rank's utterance batch (weak scaling)
    q_dev = torch.from_numpy(q_host).to(dev)
    inp = SynthInputs(cfg, args.batch, args.text_tokens, args.prompt_tokens, args.speech_tokens, dev, seed=100 + rank)
    out_idx = torch.empty((args.batch, args.topk), dtype=torch.int64, device=dev)
    out_sc = torch.empty((args.batch, args.topk), dtype=torch.float32, device=dev)
    result = {}

    # decode chains of `depth` consecutive batches overlap flow + vocoder of the batch before them.  Setup (untimed): the
    # pipeline picks its HIP streams by measurement -- which hardware queues the streams land on decides how well the chains
    # overlap (PipelinedSynth.autotune) -- on this rank's own inputs.
    sample = (inp.text, inp.tlen, inp.spk_style, inp.style_tok, inp.ts, inp.u, inp.timbre_tok, inp.timbre_mel, inp.spk_timbre,
              inp.z, inp.phase0, inp.noise)
    depths = tuple(int(x) for x in os.environ.get("ASTTS_BENCH_DEPTHS", "2,3").split(","))      # decode chains tried by the calibration
    pipe = PipelinedSynth.autotune(eng, sample, depths=depths, trials=2, steps=max(2, min(args.steps, 8)), verbose=rank == 0 and bool(os.environ.get("ASTTS_BENCH_VERBOSE")),
                                   front=lambda: sb.search_device(q_dev, args.topk, out_idx=out_idx, out_score=out_sc), dist=dist)
    n_done = [0]

    def take(done):
        if done is None:
            return
        for d in (done if isinstance(done, list) else [done]):
            result["wav"] = d[2]
            n_done[0] += 1

    def step():
        sb.search_device(q_dev, args.topk, out_idx=out_idx, out_score=out_sc)
        result["ids"] = gather_style_ids(out_idx, dist) if dist is not None else out_idx   # RCCL all-gather of the ids only
        take(pipe.submit(inp.text, inp.tlen, inp.spk_style, inp.style_tok, inp.ts, inp.u, inp.timbre_tok,
                         inp.timbre_mel, inp.spk_timbre, inp.z, inp.phase0, inp.noise))

    def step_sequential():
        sb.search_device(q_dev, args.topk, out_idx=out_idx, out_score=out_sc)
        return eng.tts(inp.text, inp.tlen, inp.spk_style, inp.style_tok, inp.ts, inp.u, inp.timbre_tok,
                       inp.timbre_mel, inp.spk_timbre, inp.z, inp.phase0, inp.noise)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the per-step work of this thread (retrieval, id all-gather, submit) is enqueued on the pipeline's front stream
    with torch.cuda.stream(pipe.front_stream):
        for _ in range(args.warmup):
            step()
        take(pipe.drain())
        barrier()
        n_done[0] = 0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        take(pipe.drain())      # the last batch's flow + vocoder: every one of the K batches completes inside the timed region
        barrier()
        dt = time.perf_counter() - t0
    assert n_done[0] == args.steps, (n_done[0], args.steps)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    wav_ok = bool(torch.isfinite(result["wav"]).all()) and float(result["wav"].abs().max()) <= cfg.audio_limit + 1e-6

    # ---- side measurement (NOT `value`): the headline's second number "including host I/O" (SURVEY.md 8d).  The same pipeline, the same
    # protocol (warm-up, barrier + synchronize on both sides, MAX over ranks), but every step STARTS from host memory and ENDS in it:
    # the 8 query vectors and the 16 prompt waveforms (3 s @ 16 kHz: a style and a timbre prompt per utterance, as tts_with_rag.py:179-186
    # loads them) go host -> device, the frontend runs on the GPU (resampler, prompt log-mel, Whisper log-mel + speech tokenizer, Kaldi
    # fbank + CAM++ speaker network: astts/frontend.py, frontend_nets.py -- synthetic weights at the published shapes), the retrieval and
    # the synthesis run as in the headline, and the step's waveforms come back device -> host (pinned).
    host_io = None
    if not args.no_host_io:
        host_io = bench_host_io(args, dev, cfg, eng, sb, pipe, inp, q_host, out_idx, out_sc, dist, barrier, world)

    # ---- side measurement (NOT `value`): the same K steps with the LM stages of consecutive batches co-batched into ONE decode chain
    # (PipelinedSynth(cobatch=c): rows are independent in every LM kernel; up to 32 rows a batch's output stays bit-identical --
    # tests/test_synth_gpu.py -- beyond that the chain runs on the engine's wide path: plain GEMMs, the weights read once per token for
    # all rows, logits equal to rounding).  Every candidate (decode chains, batches per chain) runs the benchmark's own protocol -- W
    # warm-up steps, K timed steps with fill and drain inside -- and the fastest is reported, with the table of all of them.
    cob = {}
    main_pipe = pipe
    if args.steps >= 2 and not args.no_cobatch:
        cob_cfgs = ((2, 2),) if args.steps < 8 else ((2, 2), (3, 2), (2, 4), (1, 4), (2, 8), (2, 16), (1, 16))
        classes = ops.stream_pipe_classes(device=dev)
        table = {}
        for cd, cc in cob_cfgs:
            pipe = PipelinedSynth(eng, lm_depth=cd, lm_priority=0, render_priority=0, cobatch=cc, pipe_classes=classes, wide_lm=True)
            with torch.cuda.stream(pipe.front_stream):
                for _ in range(max(args.warmup, 1)):
                    step()
                take(pipe.drain())
                barrier()
                n_done[0] = 0
                tc = time.perf_counter()
                for _ in range(args.steps):
                    step()
                take(pipe.drain())
                barrier()
                dtc = time.perf_counter() - tc
            assert n_done[0] == args.steps
            if dist is not None:
                t = torch.tensor([dtc], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dtc = float(t.item())
            table[f"{cd} chains x {cc} batches ({cc * args.batch} rows)"] = round(1e3 * dtc / args.steps, 2)
            if not cob or dtc < cob["dt"]:
                cob = {"ms_per_step": 1e3 * dtc / args.steps, "dt": dtc, "chains": cd, "batches_per_chain": cc}
            pipe = None
        cob["table_ms_per_step"] = table
        pipe = main_pipe

    # ---- stage breakdown (one more step with events on the current stream)
    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e
    e0 = ev()
    sb.search_device(q_dev, args.topk, out_idx=out_idx, out_score=out_sc)
    e1 = ev()
    pre = eng.lm.prefix(inp.text, inp.tlen, inp.spk_style, inp.style_tok)
    toks = eng.lm.decode(pre, inp.ts, inp.u, True)
    e2 = ev()
    all_tok = torch.cat([inp.timbre_tok.to(torch.int32), toks], 1)
    tl = torch.full((inp.b,), all_tok.shape[1], dtype=torch.int32, device=dev)
    mel = eng.flow.decode(all_tok, tl, inp.timbre_mel, inp.spk_timbre, inp.z, inp.tmp + inp.tm)
    e3 = ev()
    eng.hift.forward(mel, inp.phase0, inp.noise)
    e4 = ev()
    torch.cuda.synchronize()
    stages = {"knn_ms": e0.elapsed_time(e1), "lm_ms": e1.elapsed_time(e2), "flow_ms": e2.elapsed_time(e3), "vocoder_ms": e3.elapsed_time(e4)}

    # ---- retrieval leg alone: queries/s
    for _ in range(50):
        sb.search_device(q_dev, args.topk, out_idx=out_idx, out_score=out_sc)
    torch.cuda.synchronize()
    tq = time.perf_counter()
    nsearch = 500
    for _ in range(nsearch):
        sb.search_device(q_dev, args.topk, out_idx=out_idx, out_score=out_sc)
    torch.cuda.synchronize()
    knn_qps = nsearch * args.batch / (time.perf_counter() - tq)
    traffic_table, traffic_file = load_traffic_table()      # HBM bytes per launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, own
                                                            # pass; cannot be collected inside this run)
    # retrieval kernel against ITS roofline (HBM: the bank is read once per search): HIP events around the scan launch
    sb.profile_enable(True)
    for _ in range(100):
        sb.search_device(q_dev, args.topk, out_idx=out_idx, out_score=out_sc)
    torch.cuda.synchronize()
    scan_ms, scan_n = sb.profile_read()
    sb.profile_enable(False)
    scan_us = scan_ms * 1e3 / max(scan_n, 1)
    knn_bytes = args.bank_rows * args.dim * 2 + args.batch * args.dim * 4 + args.batch * args.topk * 12
    knn_roof = {"bound": "hbm", "kernel": "knn_scan", "achieved": knn_bytes / (scan_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": knn_bytes / (scan_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_us": scan_us, "algorithmic_bytes_per_launch": knn_bytes,
                "traffic": traffic_table.get("knn_scan_1000_x_6144", {}).get("hbm_bytes_per_launch") if args.bank_rows == 1000 and args.dim == 6144 else None}
    from oracle import knn as oknn

    eidx, _ = oknn.knn_search(bank16, q_host, args.topk)
    ids_ok = bool(np.array_equal(out_idx.cpu().numpy(), eidx))
    # ---- the collectives of the path, checked (untimed): the ids every rank holds after the all-gather are the oracle's for the
    # rank that produced them, and the bank-sharded stress mode (each rank holds N/W rows, one all-gather of [Q, k] (row, fp64
    # score) pairs + local merge) returns the unsharded ids
    gathered_ok = bank_sharded_ok = None
    if dist is not None:
        from astts.parallel import bank_sharded_search, shard_bounds

        allids = gather_style_ids(out_idx, dist).cpu().numpy().reshape(world, args.batch, args.topk)
        gathered_ok = all(bool(np.array_equal(allids[r], oknn.knn_search(bank16, make_queries(bank16, args.batch, seed=r), args.topk)[0]))
                          for r in range(world))
        rb, re, _ = shard_bounds(args.bank_rows, world, rank)
        sb_shard = StyleBank(bank16[rb:re], device=dev)

        def local(qq, kk):
            i, _, s64 = sb_shard.search_device(qq, min(kk, re - rb), return_f64=True)
            return i, s64

        # bank-sharded mode: EVERY rank searches the same queries (rank 0's) against its own rows
        q0_host = make_queries(bank16, args.batch, seed=0)
        bidx, _ = bank_sharded_search(local, torch.from_numpy(q0_host).to(dev), args.topk, rb, dist)
        bank_sharded_ok = bool(np.array_equal(bidx.cpu().numpy(), oknn.knn_search(bank16, q0_host, args.topk)[0]))
        del sb_shard

    # ---- roofline: every launch of the profiled kernel kinds timed on its own stream (dispatch timestamps for the decode kernels, HIP
    # events for the rest), one SEQUENTIAL step per kind (one batch at a time on one stream: the mode `sequential_ms_per_step` /
    # `stages_ms` are measured in) -- that is `roofline.achieved / frac / avg_us`; the dominant kernel is then timed again inside the
    # PIPELINED schedule of the timed region (`roofline.pipelined`: kernels of different batches overlap there, so its launches x
    # avg_us is not a per-step attribution).  One kind per pass: event records perturb neighbouring launches.
    kinds = {"gemm_tile": ops.PROF_GEMM_TILE, "lm_gemv": ops.PROF_GEMM_SKINNY, "attn_mha_flash": ops.PROF_ATTN_FLASH,
             "lm_attn": ops.PROF_ATTN_DECODE}

    def profiled(kind, fn):
        ops.prof_enable(kind, True, 40000)
        fn()
        torch.cuda.synchronize()
        ms, n, work, dropped = ops.prof_read(kind)
        ops.prof_enable(kind, False)
        return {"ms_per_step": ms, "launches": n, "work": work, "dropped": dropped}

    prof = {name: profiled(kind, step_sequential) for name, kind in kinds.items()}
    # which kind is "dominant" must be the same on every rank (the passes below differ per kind): decide on the MAX over ranks
    kind_ms = torch.tensor([prof[k]["ms_per_step"] for k in kinds], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(kind_ms, op=dist.ReduceOp.MAX)
    kind_ms = dict(zip(kinds, kind_ms.tolist()))
    dom = max(kind_ms, key=kind_ms.get)
    if dom == "gemm_tile":      # a family of tiles with different shapes: the single dominant KERNEL is the decode GEMV
        dom = "lm_gemv" if kind_ms["lm_gemv"] > 0.5 * kind_ms["gemm_tile"] else dom
    seq = prof[dom]
    # the dominant kernel again, inside the PIPELINED schedule of the timed region (same pipeline object, same inputs, a few more
    # steps): the decode kernels are timed by their own dispatch timestamps (hipExtLaunchKernelGGL events -- no extra packet in
    # the chain), which works whatever else runs beside them; this is what `rocprofv3 --kernel-trace --stats` of this command sees.
    pip = None
    if dom in ("lm_gemv", "lm_attn"):
        k_p = 2          # 2 x 14 445 launches: inside the profiler's 40 000-launch event budget

        def pipelined_steps():      # collective-free (retrieval + submit only): a profiling pass must not add all-gathers
            with torch.cuda.stream(pipe.front_stream):
                for _ in range(k_p):
                    sb.search_device(q_dev, args.topk, out_idx=out_idx, out_score=out_sc)
                    take(pipe.submit(inp.text, inp.tlen, inp.spk_style, inp.style_tok, inp.ts, inp.u, inp.timbre_tok,
                                     inp.timbre_mel, inp.spk_timbre, inp.z, inp.phase0, inp.noise))
                take(pipe.drain())

        pip = profiled(kinds[dom], pipelined_steps)
        pip = {"ms_per_step": pip["ms_per_step"] / k_p, "launches": pip["launches"] // k_p, "work": pip["work"] / k_p, "dropped": pip["dropped"]}
    pipe_depth_now = pipe.depth
    p = seq          # `roofline` = the sequential figure (attributable per launch); the pipelined one rides beside it
    if dom in ("gemm_tile", "attn_mha_flash"):
        achieved = p["work"] / (p["ms_per_step"] / 1e3) / 1e12
        roof = {"bound": "mfma", "achieved": achieved, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F16_PEAK_TFLOPS}
    else:
        achieved = p["work"] / (p["ms_per_step"] / 1e3) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
    traffic = traffic_table.get(dom, {}).get("hbm_bytes_per_launch")
    pip_us = pip["ms_per_step"] * 1e3 / max(pip["launches"], 1) if pip else None
    roof.update({"traffic": traffic,
                 "traffic_source": f"profiles/{traffic_file} (separate rocprofv3 --pmc FETCH_SIZE pass, x2 gfx950 correction)" if traffic else None,
                 "kernel": dom, "avg_us": p["ms_per_step"] * 1e3 / max(p["launches"], 1),
                 "launches_per_step": p["launches"],
                 "algorithmic_work_per_launch": p["work"] / max(p["launches"], 1),
                 "mode": "one sequential step (one batch at a time on one stream: what sequential_ms_per_step / stages_ms are measured in; "
                         "kernel-level dispatch timestamps)",
                 "pipelined": ({"decode_chains": pipe_depth_now, "avg_us": pip_us, "achieved": pip["work"] / max(pip["launches"], 1) / (pip_us * 1e-6) / 1e9 if dom.startswith("lm_") else None,
                                "frac": pip["work"] / max(pip["launches"], 1) / (pip_us * 1e-6) / 1e9 / HBM_PEAK_GBS if dom.startswith("lm_") else None,
                                "launches_per_step": pip["launches"],
                                "note": "the same kernel inside the schedule of the timed region (the decode chains and a render stage share the GPU: "
                                        "kernels of different batches overlap, so launches x avg_us may exceed ms_per_step -- not a per-step attribution)"}
                               if pip else None),
                 "all_kinds_ms_per_sequential_step": {k: round(v["ms_per_step"], 3) for k, v in prof.items()}})

    # ---- per-stage rooflines (sequential mode): algorithmic work of the whole stage / the stage's wall time
    lm_w = profiled(ops.PROF_GEMM_SKINNY, lambda: eng.lm.decode(pre, inp.ts, inp.u, True))
    lm_a = profiled(ops.PROF_ATTN_DECODE, lambda: eng.lm.decode(pre, inp.ts, inp.u, True))
    fl_g = profiled(ops.PROF_GEMM_TILE, lambda: eng.flow.decode(all_tok, tl, inp.timbre_mel, inp.spk_timbre, inp.z, inp.tmp + inp.tm))
    fl_a = profiled(ops.PROF_ATTN_FLASH, lambda: eng.flow.decode(all_tok, tl, inp.timbre_mel, inp.spk_timbre, inp.z, inp.tmp + inp.tm))
    vo_g = profiled(ops.PROF_GEMM_TILE, lambda: eng.hift.forward(mel, inp.phase0, inp.noise))
    lm_bytes = lm_w["work"] + lm_a["work"]
    by_stage = {
        "lm_decode": {"bound": "hbm", "achieved": lm_bytes / (stages["lm_ms"] / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": lm_bytes / (stages["lm_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS,
                      "algorithmic_bytes": lm_bytes, "launches": lm_w["launches"] + lm_a["launches"],
                      "note": "weights of every decode GEMV + KV / position rows of every decode attention, each read once per step; "
                              "the stage is a chain of dependent launches (1.45 us boundary + one memory round trip each): latency-bound"},
        "flow": {"bound": "mfma", "achieved": (fl_g["work"] + fl_a["work"]) / (stages["flow_ms"] / 1e3) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS,
                 "unit": "TFLOP/s", "frac": (fl_g["work"] + fl_a["work"]) / (stages["flow_ms"] / 1e3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                 "algorithmic_flops": fl_g["work"] + fl_a["work"], "launches_gemm_attn": fl_g["launches"] + fl_a["launches"]},
        "vocoder": {"bound": "mfma", "achieved": vo_g["work"] / (stages["vocoder_ms"] / 1e3) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": vo_g["work"] / (stages["vocoder_ms"] / 1e3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                    "algorithmic_flops": vo_g["work"],
                    "note": "conv stack as implicit GEMMs; long thin convolutions, launch- and HBM-bound rather than MFMA-bound"},
    }

    # ---- 24 kHz side measurement (the north star's rate; the reference scripts save 22 050 Hz): same pipeline, second engine
    v24 = None
    pipe_depth, pipe_tuned_ms = pipe.depth, pipe.tuned_ms_per_batch
    if not args.no_24khz and args.sample_rate != 24000 and args.steps >= 2:
        pipe = None
        cfg24 = SynthConfig(sample_rate=24000)
        eng24 = SynthEngine(weights, cfg24, dev)
        inp24 = SynthInputs(cfg24, args.batch, args.text_tokens, args.prompt_tokens, args.speech_tokens, dev, seed=100 + rank)
        s24 = (inp24.text, inp24.tlen, inp24.spk_style, inp24.style_tok, inp24.ts, inp24.u, inp24.timbre_tok, inp24.timbre_mel,
               inp24.spk_timbre, inp24.z, inp24.phase0, inp24.noise)
        front24 = lambda: sb.search_device(q_dev, args.topk, out_idx=out_idx, out_score=out_sc)
        pipe24 = PipelinedSynth.autotune(eng24, s24, depths=(main_pipe.depth,), trials=2, steps=4, front=front24)
        k24 = max(4, args.steps)          # the same K as `value` (pipeline fill and drain are inside both timed regions: equal shares)
        with torch.cuda.stream(pipe24.front_stream):
            for _ in range(2):
                front24()
                pipe24.submit(*s24)
            pipe24.drain()
            barrier()
            t24 = time.perf_counter()
            for _ in range(k24):
                front24()
                pipe24.submit(*s24)
            last = pipe24.drain()
            barrier()
            dt24 = time.perf_counter() - t24
        if dist is not None:
            t = torch.tensor([dt24], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt24 = float(t.item())
        v24 = {"value": inp24.audio_seconds * k24 * world / dt24, "unit": "audio-s/wall-s", "steps": k24, "ms_per_step": 1e3 * dt24 / k24,
               "samples_per_utterance": inp24.tm * cfg24.upsample_total, "sample_rate": 24000,
               "note": "same workload with SynthConfig(sample_rate=24000): 468 mel frames per 250 tokens instead of 430"}
        del pipe24, eng24

    # ---- BASELINE's other configurations, bounded, with the same protocol (every rank takes part: configs[3] / [4] are sharded jobs)
    side = {}
    if not args.no_side:
        pipe = main_pipe = None
        torch.cuda.empty_cache()
        for which, (k_s, w_s) in (("config3", (1, 1)), ("config5", (2, 1)), ("config4", (1, 1))):
            r = run_side(args, which, dev, dist, rank, world, cfg, weights, eng, steps=k_s, warmup=w_s, traffic_table=traffic_table)
            if rank == 0:
                side[which] = r
            torch.cuda.empty_cache()

    total_audio = inp.audio_seconds * args.steps * world
    res = None
    if rank == 0:
        res = {
            "metric": "synthesized audio sec/wall-sec (RTF^-1) + style-kNN QPS, IEMOCAP test batch",
            "value": total_audio / dt,
            "unit": "audio-s/wall-s",
            "n_gpus": world,
            "rccl_world_size": dist.get_world_size() if dist is not None else None,
            "rccl_backend": dist.get_backend() if dist is not None else None,
            "gathered_ids_match_oracle": gathered_ok,
            "bank_sharded_ids_match_oracle": bank_sharded_ok,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "dtype_note": "fp16 weights / MFMA operands / KV cache, fp32 accumulation, residual stream, norms and softmax (kNN: fp16 scan + "
                          "fp64 re-score).  Deviation: the reference's CosyVoice(model_dir) default runs fp32 (SURVEY.md 8d sanctions "
                          "fp16 MFMA operands); parity tolerances vs the fp32 oracle are stated in tests/",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: batch={args.batch} utterances/GPU, kNN Q={args.batch} x N={args.bank_rows} x D={args.dim} k={args.topk}, "
                                   f"synthesis Tt={args.text_tokens} Tp={args.prompt_tokens} Ts={args.speech_tokens} (fixed-length decode, EOS ignored) "
                                   f"-> {inp.tm} mel frames -> {inp.tm * cfg.upsample_total} samples @ {cfg.sample_rate} Hz; CosyVoice-300M shapes, random-init weights",
                       "parallelism": f"dp{world} (model + bank replicated, utterances sharded, all-gather of style ids only)"},
            "knn_qps": knn_qps,
            "knn_roofline": knn_roof,
            "ids_match_oracle": ids_ok,
            "waveform_finite_and_clamped": wav_ok,
            "pipelining": f"{pipe_depth + 2} HIP streams (front: retrieval + LM prefix / prefill + submit, {pipe_depth} decode chains, render): the LM decode chains of {pipe_depth} consecutive batches overlap flow+vocoder of the batch before them (streams chosen by PipelinedSynth.autotune during setup: {pipe_tuned_ms:.1f} ms/batch in calibration; with several ranks all keep the configuration whose slowest rank is fastest); every one of the K batches completes inside the timed region",
            "cobatched_lm_side_measurement": ({"value": total_audio / cob["dt"], "ms_per_step": cob["ms_per_step"],
                                               "decode_chains": cob["chains"], "batches_per_chain": cob["batches_per_chain"],
                                               "rows_per_chain": cob["batches_per_chain"] * args.batch, "table_ms_per_step": cob["table_ms_per_step"],
                                               "note": "same K steps (fill and drain inside), LM stages of consecutive batches co-batched into ONE decode chain: "
                                                       "<= 32 rows on the decode-step kernels (a row's tokens do not depend on the chain's width: outputs "
                                                       "bit-identical per batch), 64 / 128 rows on the engine's wide path (plain GEMMs, the weights read once "
                                                       "per token for all rows; logits equal to rounding); the fastest candidate; reported beside `value`, not as it"} if cob else None),
            "stages_ms": {k: round(v, 3) for k, v in stages.items()},
            "sequential_ms_per_step": round(sum(stages.values()), 3),
            "roofline": roof,
            "roofline_by_stage": by_stage,
            "value_24khz": v24,
            "value_with_host_io": host_io["value"] if host_io else None,
            "host_io": host_io,
            "frontend_ms": host_io["frontend_ms_per_prompt"] if host_io else None,
            "side_workloads": side or None,
        }
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # rank 0 alone, after the process group is gone (no collective, the other ranks have left): the one-GPU retrieval stress, the
        # rows next to the path (query embedder, streaming latency) and the CPU baseline -- at every N
        if not args.no_side:
            res["knn_stress"] = knn_stress(args, dev, traffic_table)
            res.update(side_extras(args, dev, cfg, eng))
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, weights, bank16, q_host, args.topk)
        emit_json(res)


if __name__ == "__main__":
    main()
