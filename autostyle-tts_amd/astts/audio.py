"""Wave-file I/O, resampling and the mel-spectrogram front end (host-side plumbing around the GPU path).

Replaces, for the reference's call surface:
  cosyvoice.utils.file_utils.load_wav(path, sr)   /root/reference/tts_with_rag.py:2,180-186
  torchaudio.save(path, tensor[1, L], 22050)      /root/reference/tts_with_rag.py:197,
                                                  /root/reference/tts_with_style_and_timbre.py:95
  torchaudio.transforms.Resample(22050 -> 16000)  /root/reference/tts_with_rag.py:137
torchaudio / soundfile / librosa are not available here, so RIFF parsing, a windowed-sinc polyphase
resampler and the Slaney mel filter bank are written out.  These stages sit OUTSIDE the measured
GPU path (frontend features are inputs of the timed region).
"""
from __future__ import annotations

import functools
import math
import struct
from typing import Tuple

import numpy as np
import torch


# ------------------------------------------------------------------------------------------ RIFF/WAVE
def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """-> (float32 [channels, n] in [-1, 1], sample_rate).  PCM 8/16/24/32-bit and IEEE float 32/64."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    pcm = None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _br, _ba, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: sub-format tag
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, bits = fmt
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v & 0x800000, v - (1 << 24), v)
            x = v.astype(np.float32) / 8388608.0
        elif bits == 32:
            x = np.frombuffer(pcm, dtype="<i4").astype(np.float32) / 2147483648.0
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(pcm, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    n = x.size // ch
    return x[: n * ch].reshape(n, ch).T.copy(), sr


def write_wav(path: str, wav, sample_rate: int) -> None:
    """``wav``: tensor/array [channels, n] (or [n]) float32 -> 32-bit IEEE-float WAVE, the encoding
    torchaudio.save picks for a float32 tensor."""
    x = wav.detach().cpu().numpy() if torch.is_tensor(wav) else np.asarray(wav)
    x = np.atleast_2d(x.astype(np.float32))
    ch, n = x.shape
    payload = x.T.astype("<f4").tobytes()
    fmt = struct.pack("<HHIIHH", 3, ch, sample_rate, sample_rate * ch * 4, ch * 4, 32)
    fact = struct.pack("<I", n)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact + \
        b"data" + struct.pack("<I", len(payload)) + payload
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


# ------------------------------------------------------------------------------------------ resampling
@functools.lru_cache(maxsize=16)
def _resample_kernel(sr_in: int, sr_out: int, zeros: int = 6, rolloff: float = 0.99):
    """Hann-windowed sinc table of the polyphase resampler (the torchaudio ``sinc_interp_hann`` recipe)
    -> (kern fp32 [up, 2 width + down], up, down, width)."""
    g = math.gcd(sr_in, sr_out)
    up, down = sr_out // g, sr_in // g
    base = min(up, down) * rolloff
    width = int(math.ceil(zeros * down / base))
    idx = torch.arange(-width, width + down, dtype=torch.float64)[None, :] / down
    t = (torch.arange(0, -up, -1, dtype=torch.float64)[:, None] / up + idx) * base
    t = t.clamp(-zeros, zeros)
    window = torch.cos(t * math.pi / zeros / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / down)
    return kern.to(torch.float32), up, down, width


def resample(x: torch.Tensor, sr_in: int, sr_out: int, zeros: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """Band-limited polyphase resampling.  x: [..., n] float32.  CUDA tensors run astts_op_resample_poly (HIP); host
    tensors run the same table through a strided convolution (file loading, tests)."""
    if sr_in == sr_out:
        return x
    kern, up, down, width = _resample_kernel(sr_in, sr_out, zeros, rolloff)
    shape = x.shape
    n_out = int(math.ceil(up * shape[-1] / down))
    if x.is_cuda:
        from . import _lib
        key = (sr_in, sr_out, zeros, rolloff, x.device.index)
        kd = _KERNEL_CACHE.get(key)
        if kd is None:
            kd = _KERNEL_CACHE[key] = kern.contiguous().to(x.device)
        xx = x.reshape(-1, shape[-1]).to(torch.float32).contiguous()
        y = torch.empty((xx.shape[0], n_out), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().astts_op_resample_poly(xx.data_ptr(), kd.data_ptr(), y.data_ptr(), xx.shape[0], xx.shape[1], n_out, up, down,
                                                      width, _lib.stream_ptr()))
        return y.reshape(*shape[:-1], n_out)
    kern = kern[:, None, :]                                         # [up, 1, k]
    xx = x.reshape(-1, 1, shape[-1]).to(torch.float32)
    xx = torch.nn.functional.pad(xx, (width, width + down))
    y = torch.nn.functional.conv1d(xx, kern, stride=down)            # [N, up, frames]
    y = y.transpose(1, 2).reshape(xx.shape[0], -1)
    return y[:, :n_out].reshape(*shape[:-1], n_out)


_KERNEL_CACHE: dict = {}


def load_wav(path: str, target_sr: int) -> torch.Tensor:
    """cosyvoice.utils.file_utils.load_wav: read -> mono (mean over channels) -> resample -> FloatTensor [1, n]."""
    x, sr = read_wav(path)
    t = torch.from_numpy(x).mean(dim=0, keepdim=True)
    if sr != target_sr:
        t = resample(t, sr, target_sr)
    return t


# ------------------------------------------------------------------------------------------ mel spectrogram
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    mel = f / (200.0 / 3)
    log_t = f >= 1000.0
    return np.where(log_t, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), mel)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f = m * (200.0 / 3)
    log_t = m >= 15.0
    return np.where(log_t, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), f)


def mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """Slaney-style (area-normalised) triangular filters, the librosa.filters.mel default -> [n_mels, n_fft//2+1]."""
    freqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(pts)
    ramps = pts[:, None] - freqs[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    w *= (2.0 / (pts[2:n_mels + 2] - pts[:n_mels]))[:, None]
    return w.astype(np.float32)


def mel_spectrogram(wav: torch.Tensor, sr: int = 22050, n_fft: int = 1024, hop: int = 256, win: int = 1024,
                    n_mels: int = 80, fmin: float = 0.0, fmax: float = 8000.0) -> torch.Tensor:
    """matcha-style log-mel used for the timbre prompt: reflect pad (n_fft-hop)/2, magnitude STFT, Slaney mel,
    log(clamp(., 1e-5)).  wav [B, n] -> [B, frames, n_mels].  CUDA input: HIP kernel; host input: torch.stft (tests, stand-ins)."""
    pad = (n_fft - hop) // 2
    if wav.is_cuda:            # astts_op_mel_spectrogram (HIP): same definition, one block per frame
        from . import _lib
        assert win == n_fft
        key = ("mel", sr, n_fft, n_mels, fmin, fmax, wav.device.index)
        tabs = _KERNEL_CACHE.get(key)
        if tabs is None:
            tabs = _KERNEL_CACHE[key] = (torch.hann_window(win).to(wav.device),
                                         torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax)).contiguous().to(wav.device))
        w = wav.to(torch.float32).contiguous()
        b, n = w.shape
        frames = (n + 2 * pad - n_fft) // hop + 1
        out = torch.empty((b, frames, n_mels), dtype=torch.float32, device=wav.device)
        _lib.check(_lib.load().astts_op_mel_spectrogram(w.data_ptr(), tabs[0].data_ptr(), tabs[1].data_ptr(), out.data_ptr(), b, n, n_fft, hop,
                                                        n_mels, 1e-5, _lib.stream_ptr()))
        return out
    y = torch.nn.functional.pad(wav[:, None, :], (pad, pad), mode="reflect")[:, 0]
    spec = torch.stft(y, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=False,
                      return_complex=True)
    mag = torch.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-9)
    fb = torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax))
    mel = torch.matmul(fb, mag)
    return torch.log(torch.clamp(mel, min=1e-5)).transpose(1, 2).contiguous()


def whisper_log_mel(wav16k: torch.Tensor, n_mels: int = 128) -> torch.Tensor:
    """The input features of the reference's speech tokenizer ("whisper-style 128-bin log-mel of 16 k
    audio"; upstream frontend._extract_speech_token calls whisper.log_mel_spectrogram(speech, n_mels=128)): periodic Hann
    window of 400, hop 160, centred (reflect) STFT with the last frame dropped, POWER spectrum, Slaney mel (0 .. 8 kHz),
    log10(clamp(., 1e-10)), floor at max - 8, (x + 4) / 4.  wav [B, n] or [n] at 16 kHz -> [B, n_mels, n // 160] (any device).
    Pinned against transformers' WhisperFeatureExtractor (tests/test_oracle_synth_blocks.py); a real speech-tokenizer model plugs
    in behind ``Frontend(speech_tokenizer=...)`` and takes exactly this tensor."""
    w = wav16k if wav16k.dim() == 2 else wav16k[None]
    w = w.to(torch.float32)
    if w.is_cuda:              # astts_op_whisper_log_mel (HIP): the same definition as the host path below (tests hold both to the fixture)
        from . import _lib
        lib = _lib.load()
        key = ("whisper", n_mels, w.device.index)
        tabs = _KERNEL_CACHE.get(key)
        if tabs is None:
            tabs = _KERNEL_CACHE[key] = (torch.hann_window(400, periodic=True).to(w.device),
                                         torch.from_numpy(mel_filterbank(16000, 400, n_mels, 0.0, 8000.0)).contiguous().to(w.device))
        w = w.contiguous()
        b, n = w.shape
        out = torch.empty((b, n_mels, n // 160), dtype=torch.float32, device=w.device)
        ws = torch.empty((max(int(lib.astts_op_whisper_log_mel_workspace_bytes(b)), 4),), dtype=torch.uint8, device=w.device)
        _lib.check(lib.astts_op_whisper_log_mel(w.data_ptr(), tabs[0].data_ptr(), tabs[1].data_ptr(), out.data_ptr(), b, n, 400, 160, n_mels,
                                                ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        return out
    window = torch.hann_window(400, periodic=True, device=w.device)
    spec = torch.stft(w, 400, hop_length=160, win_length=400, window=window, center=True, pad_mode="reflect", return_complex=True)
    power = (spec.real ** 2 + spec.imag ** 2)[..., :-1]
    fb = torch.from_numpy(mel_filterbank(16000, 400, n_mels, 0.0, 8000.0)).to(w.device)
    log_spec = torch.log10(torch.clamp(torch.matmul(fb, power), min=1e-10))
    log_spec = torch.maximum(log_spec, log_spec.amax(dim=(1, 2), keepdim=True) - 8.0)
    return (log_spec + 4.0) / 4.0


# ------------------------------------------------------------------------------------------ Kaldi fbank (speaker-net input)
def kaldi_mel_filterbank(sr: int = 16000, n_fft: int = 512, n_mels: int = 80, fmin: float = 20.0, fmax: float = 0.0) -> np.ndarray:
    """Kaldi's mel bank (compute-fbank-feats / torchaudio.compliance.kaldi.get_mel_banks): mel(f) = 1127 ln(1 + f / 700), n_mels + 2
    points equally spaced in mel between fmin and fmax (<= 0: Nyquist + fmax), triangles drawn IN MEL SPACE over the FFT bin centres,
    no area normalisation -> [n_mels, n_fft // 2 + 1] float32."""
    hi = fmax if fmax > 0 else sr / 2 + fmax
    mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    pts = np.linspace(mel(fmin), mel(hi), n_mels + 2)
    bins = mel(np.arange(n_fft // 2 + 1) * (sr / n_fft))
    slopes = pts[None, :] - bins[:, None]                       # [bins, n_mels + 2]
    d = np.diff(pts)
    w = np.maximum(0.0, np.minimum(-slopes[:, :-2] / d[:-1], slopes[:, 2:] / d[1:]))
    return w.T.astype(np.float32)


def kaldi_fbank(wav16k: torch.Tensor, n_mels: int = 80, scale: float = 1.0, subtract_mean: bool = False, sr: int = 16000) -> torch.Tensor:
    """The input features of the reference's speaker-embedding network ("spk-emb ... ONNX on 80-bin Kaldi fbank";
    upstream frontend._extract_spk_embedding calls torchaudio.compliance.kaldi.fbank(speech, num_mel_bins=80, dither=0,
    sample_frequency=16000) and subtracts the mean over time): 25 ms frames every 10 ms without padding (snip_edges), per frame: remove the
    DC offset, pre-emphasis 0.97 (first sample x0 (1 - 0.97)), Povey window (hann^0.85, symmetric), zero-pad to 512, POWER spectrum,
    Kaldi mel bank (20 Hz .. Nyquist), log(max(., float32 eps)).  ``scale``: what the samples are multiplied by first (Kaldi's own
    tools read 16-bit integers: 32768; torchaudio and upstream take the [-1, 1] floats as they are: 1).  wav [B, n] or [n] ->
    [B, 1 + (n - 400) // 160, n_mels] (any device; CUDA input: the HIP kernel astts_op_kaldi_fbank).  Pinned against transformers'
    SeamlessM4TFeatureExtractor (its numpy "mimic Kaldi" path; tests/test_oracle_synth_blocks.py)."""
    w = wav16k if wav16k.dim() == 2 else wav16k[None]
    w = w.to(torch.float32)
    flen, hop, n_fft = int(0.025 * sr), int(0.010 * sr), 512
    b, n = w.shape
    if n < flen:
        raise ValueError(f"kaldi_fbank: {n} samples are shorter than one {flen}-sample frame")
    frames = 1 + (n - flen) // hop
    eps = 1.1920928955078125e-07
    if w.is_cuda:
        from . import _lib
        key = ("kaldi", sr, n_mels, w.device.index)
        tabs = _KERNEL_CACHE.get(key)
        if tabs is None:
            win = torch.from_numpy(np.power(np.hanning(flen), 0.85).astype(np.float32))
            tabs = _KERNEL_CACHE[key] = (win.to(w.device), torch.from_numpy(kaldi_mel_filterbank(sr, n_fft, n_mels)).contiguous().to(w.device))
        w = w.contiguous()
        out = torch.empty((b, frames, n_mels), dtype=torch.float32, device=w.device)
        _lib.check(_lib.load().astts_op_kaldi_fbank(w.data_ptr(), tabs[0].data_ptr(), tabs[1].data_ptr(), out.data_ptr(), b, n, flen, hop, n_fft,
                                                    n_mels, float(scale), 0.97, eps, _lib.stream_ptr()))
    else:
        idx = torch.arange(flen)[None, :] + hop * torch.arange(frames)[:, None]
        fr = w[:, idx].double() * scale                                             # [B, frames, flen]
        fr = fr - fr.mean(dim=-1, keepdim=True)
        fr = torch.cat([fr[..., :1] * (1.0 - 0.97), fr[..., 1:] - 0.97 * fr[..., :-1]], dim=-1)
        fr = fr * torch.from_numpy(np.power(np.hanning(flen), 0.85))
        spec = torch.fft.rfft(fr, n=n_fft)
        power = spec.real ** 2 + spec.imag ** 2
        fb = torch.from_numpy(kaldi_mel_filterbank(sr, n_fft, n_mels)).double()
        out = torch.log(torch.clamp(power @ fb.T, min=eps)).float()
    if subtract_mean:
        out = out - out.mean(dim=1, keepdim=True)
    return out
