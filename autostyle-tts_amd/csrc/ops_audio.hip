// ops_audio.hip -- HiFT-vocoder signal operators and the LM's token sampler (gfx950).
//
//   nsf_source   f0 -> harmonic sine source + noise -> linear(9->1) -> tanh   (SineGen/SourceModuleHnNSF)
//   stft16       torch.stft(n_fft=16, hop=4, hann, center/reflect) -> [B, F, 18] (re[0..8], im[0..8])
//   istft16      magnitude=exp(.), phase=sin(.) -> torch.istft(16, 4, hann) -> clamp       (HiFT head)
//   ras_sample   repetition-aware top-p/top-k sampling of one speech token per batch row
//
// Randomness is INJECTED (initial phases, Gaussian noise, uniforms) so that the CPU oracle and this
// path can be driven by identical draws.  Replaces third-party cosyvoice.hifigan.generator /
// cosyvoice.utils.common.ras_sampling arithmetic behind tts_with_rag.py:195.
#include "common.h"
#include "toplist.h"
#include "xlane.h"

namespace astts {

static constexpr double kTwoPi = 6.283185307179586476925286766559;

// exclusive prefix sum of f0 over frames, fp64, one block per batch row (frames <= a few thousand)
__global__ __launch_bounds__(256) void f0_prefix(const float* __restrict__ f0, double* __restrict__ pre, int tm) {
    __shared__ double carry;
    __shared__ double buf[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) carry = 0.0;
    __syncthreads();
    for (int base = 0; base < tm; base += 256) {
        const int i = base + tid;
        const double v = i < tm ? (double)f0[(int64_t)b * tm + i] : 0.0;
        buf[tid] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
            const double add = tid >= off ? buf[tid - off] : 0.0;
            __syncthreads();
            buf[tid] += add;
            __syncthreads();
        }
        if (i < tm) pre[(int64_t)b * tm + i] = carry + buf[tid] - v;
        __syncthreads();
        if (tid == 255) carry += buf[255];
        __syncthreads();
    }
}

struct NsfArgs {
    const float* f0;      // [B, Tm]
    const double* pre;    // [B, Tm] exclusive prefix of f0
    const float* phase0;  // [B, H+1] initial phase per harmonic (index 0 forced to 0 by the caller)
    const float* noise;   // [B, L, H+1] standard normal
    const float* lin_w;   // [H+1]
    const float* lin_b;   // [1]
    float* out;           // [B, L]
    int b, tm, up, nh;    // up = samples per frame (256), nh = H+1 (9)
    float sr, sine_amp, noise_std, voiced_thr;
};

__global__ __launch_bounds__(256) void nsf_source(NsfArgs a) {
    const int64_t L = (int64_t)a.tm * a.up;
    const int64_t total = (int64_t)a.b * L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int bb = (int)(i / L);
        const int64_t s = i - (int64_t)bb * L;
        const int f = (int)(s / a.up);
        const int r = (int)(s - (int64_t)f * a.up);
        const float f0 = a.f0[(int64_t)bb * a.tm + f];
        // cumsum of the nearest-upsampled f0 up to and including sample s (fp64)
        const double cum = (a.pre[(int64_t)bb * a.tm + f] * (double)a.up + (double)(r + 1) * (double)f0) / (double)a.sr;
        const float uv = f0 > a.voiced_thr ? 1.0f : 0.0f;
        const float namp = uv * a.noise_std + (1.0f - uv) * a.sine_amp / 3.0f;
        float acc = a.lin_b[0];
        for (int hI = 0; hI < a.nh; ++hI) {
            double ph = cum * (double)(hI + 1);
            ph -= floor(ph);
            const float theta = (float)(kTwoPi * ph);
            const float sine = a.sine_amp * sinf(theta + a.phase0[bb * a.nh + hI]);
            const float v = sine * uv + namp * a.noise[i * a.nh + hI];
            acc += a.lin_w[hI] * v;
        }
        a.out[i] = tanhf(acc);
    }
}

__device__ __forceinline__ float hann16(int n) {  // periodic Hann, N = 16
    return 0.5f - 0.5f * cosf((float)(kTwoPi / 16.0) * (float)n);
}

// x [B, L] -> y [B, F, 18], F = L/4 + 1, frame f covers reflect-padded samples 4f-8 .. 4f+7
// lens (ragged batches): samples of each row (a multiple of 4, >= 16; null: L).  A row is transformed as a signal of ITS length -- the
// reflection at its end mirrors its own last samples -- into its first lens[b] / 4 + 1 frames; the frames behind them are zero.
__global__ __launch_bounds__(256) void stft16(const float* __restrict__ x, float* __restrict__ y, int b, int64_t Lmax, int64_t F,
                                              const int* __restrict__ lens) {
    const int64_t total = (int64_t)b * F;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int bb = (int)(i / F);
        const int64_t f = i - (int64_t)bb * F;
        const float* xb = x + (int64_t)bb * Lmax;
        const int64_t L = lens ? min((int64_t)lens[bb], Lmax) : Lmax;
        float* o = y + i * 18;
        if (f > L / 4) {
#pragma unroll
            for (int k = 0; k < 18; ++k) o[k] = 0.0f;
            continue;
        }
        float w[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            int64_t s = 4 * f - 8 + n;
            if (s < 0) s = -s;
            if (s >= L) s = 2 * (L - 1) - s;
            w[n] = hann16(n) * xb[s];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float re = 0.0f, im = 0.0f;
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int ph = (k * n) & 15;
                const float cs = cosf((float)(kTwoPi / 16.0) * (float)ph);
                const float sn = sinf((float)(kTwoPi / 16.0) * (float)ph);
                re += w[n] * cs;
                im -= w[n] * sn;
            }
            o[k] = re;
            o[9 + k] = im;
        }
    }
}

// y [B, F, 18] (log-magnitude, phase pre-activation) -> wav [B, 4(F-1)], clamped
// frame_lens (ragged batches): frames of each row (null: F).  A row is synthesised from ITS frames only -- the overlap-add at its end sees
// no frame of the padding -- into its first 4 (frame_lens[b] - 1) samples; the samples behind them are zero.
__global__ __launch_bounds__(256) void istft16(const float* __restrict__ y, float* __restrict__ wav, int b, int64_t Fmax,
                                               float mag_clip, float audio_limit, const int* __restrict__ frame_lens) {
    const int64_t L = 4 * (Fmax - 1);
    const int64_t total = (int64_t)b * L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int bb = (int)(i / L);
        const int64_t t = i - (int64_t)bb * L;
        const int64_t F = frame_lens ? min((int64_t)frame_lens[bb], Fmax) : Fmax;
        if (t >= 4 * (F - 1)) {
            wav[i] = 0.0f;
            continue;
        }
        const int64_t tp = t + 8;  // position in the centre-padded signal
        float num = 0.0f, den = 0.0f;
        const int64_t f_hi = tp / 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t f = f_hi - q;
            const int n = (int)(tp - 4 * f);  // 0..15
            if (f < 0 || f >= F) continue;
            const float* fr = y + ((int64_t)bb * Fmax + f) * 18;
            float xs = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float mag = fminf(__expf(fr[k]), mag_clip);
                const float ph = sinf(fr[9 + k]);
                const float re = mag * cosf(ph), im = mag * sinf(ph);
                const int idx = (k * n) & 15;
                const float cs = cosf((float)(kTwoPi / 16.0) * (float)idx);
                const float sn = sinf((float)(kTwoPi / 16.0) * (float)idx);
                const float ck = (k == 0 || k == 8) ? 1.0f : 2.0f;
                xs += ck * (re * cs - ((k == 0 || k == 8) ? 0.0f : im * sn));
            }
            const float wn = hann16(n);
            num += wn * xs * (1.0f / 16.0f);
            den += wn * wn;
        }
        const float v = den > 1e-11f ? num / den : 0.0f;
        wav[i] = fminf(fmaxf(v, -audio_limit), audio_limit);
    }
}

// ------------------------------------------------------------------ repetition-aware sampling
// One block per batch row.  Definition (mirrored by oracle/synth.py::ras_sample):
//   p = softmax(logits) (eos masked when ignore_eos asks for the "mask" policy); sort descending (ties: lower id first);
//   nucleus = shortest prefix with cumulative p >= top_p, at most top_k entries;
//   token = inverse-CDF pick from the renormalised nucleus with uniform u1;
//   if token occurs >= win*tau_r times among the last `win` decoded tokens:
//       token = inverse-CDF pick from the full p (id order) with uniform u2.
//   "reject" policy (ignore_eos bit 1): EOS keeps its logit; the draw is upstream's re-draw loop in closed form (see the kernel).
struct SampleArgs {
    const float* logits;  // [B, V]
    const int* history;   // [B, hist_ld] decoded tokens so far
    const float* u;       // [B, 2]
    int* out;             // [B]
    int* hist_w;          // when set: history[b, hist_len] = token (the engine's on-device token log)
    const int* forced;    // when set: token = forced[b, hist_len] (teacher forcing), same layout as history
    int clamp_out;        // out[b] = min(token, clamp_out) (embedding row for the next step); < 0: no clamp
    const int* eos_min_rows;  // when set: EOS is masked for row b while hist_len < eos_min_rows[b] (ragged batches)
    int b, v, hist_len, hist_ld, top_k, win, eos, ignore_eos;
    float top_p, tau_r;
};

// 1024 threads per row: every pass over the 4097 logits is 4-5 elements per thread (the kernel is a chain of short
// dependent passes; at 256 threads it took 26 us of the 575 us decode step).  A thread touches the SAME elements
// (i = tid + 1024 n) in every pass up to the gather, so those passes need no barrier between them; the three radix-select
// histograms and every hand-over variable have their own LDS (zeroed once, up front): 11 workgroup barriers instead of 22.
static constexpr int RS_NT = 1024;
// Leading parameters = what the first loads (the row's logits, its EOS window) need: preloaded into SGPRs by the command
// processor (-amdgpu-kernarg-preload-count, csrc/Makefile; a by-value struct is not), the struct carries the rest.
__global__ __launch_bounds__(RS_NT) void ras_sample(const float* p_logits, const int* p_eos_min_rows, int p_v, int p_hist_len, int p_eos, int p_ignore_eos,
                                                    SampleArgs a_in) {
    __builtin_amdgcn_s_setprio(3);     // a decode-chain kernel: wins the VALU arbitration against co-resident render waves (lm_step.hip)
    SampleArgs a = a_in;
    a.logits = p_logits; a.eos_min_rows = p_eos_min_rows; a.v = p_v; a.hist_len = p_hist_len; a.eos = p_eos; a.ignore_eos = p_ignore_eos;
    extern __shared__ float prob[];  // [V rounded up to a multiple of 16]
    __shared__ float red_max[RS_NT / 64], red_sum[RS_NT / 64];
    __shared__ float sh_s[RS_NT];
    __shared__ int sh_i[RS_NT];
    __shared__ float s_bcast;
    __shared__ int s_tok;
    __shared__ unsigned hist[3][2048];
    __shared__ int s_sel_bin[3], s_sel_rem[3], s_cnt;
    const int bb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* lg = a.logits + (int64_t)bb * a.v;
    // ignore_eos: bit 0 = EOS may not be produced at this step (scalar form; with eos_min_rows the window is per row), bit 1 = the
    // policy inside that window: 0 "mask" (the EOS logit is removed before the softmax), 1 "reject" (upstream's sampling_ids: draw
    // again until the token is not EOS -- EOS keeps its probability, its place in the nucleus and its share of top_p / top_k)
    const bool eos_window = a.eos_min_rows ? (a.hist_len < a.eos_min_rows[bb]) : ((a.ignore_eos & 1) != 0);
    const bool reject = eos_window && (a.ignore_eos & 2) != 0;
    const bool mask_eos = eos_window && !reject;
    // operands of the LAST phases, requested now: the two uniforms and the repetition window of the token log were dependent
    // global round trips at the very end of the kernel (~1 us each on a kernel of 11)
    const int h0 = a.hist_len > a.win ? a.hist_len - a.win : 0;
    const bool win_in_wave = a.hist_len - h0 <= 64;
    float u_first = 0.0f, u_second = 0.0f;
    int hwin = -1;
    if (wid == 0) {
        u_first = a.u[bb * 2];
        u_second = a.u[bb * 2 + 1];
        if (win_in_wave && h0 + lane < a.hist_len) hwin = a.history[(int64_t)bb * a.hist_ld + h0 + lane];
    }
    for (int i = tid; i < 3 * 2048; i += RS_NT) (&hist[0][0])[i] = 0u;
    if (tid == 0) s_cnt = 0;
    if (tid < 16 && ((a.v + 15) & ~15) - 16 + tid >= a.v) prob[((a.v + 15) & ~15) - 16 + tid] = 0.0f;   // tail of the last 16-chunk
    float mx = -INFINITY;
    // eight logits per thread and round trip, unconditional (clamped index): a load inside the bounds check is waited for in its own
    // block, which made the 4 097-entry row five dependent round trips
    for (int base = 0; base < a.v; base += 8 * RS_NT) {
        float xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = lg[min(base + u * RS_NT + tid, a.v - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * RS_NT + tid;
            if (i < a.v) {
                float x = xv[u];
                if (mask_eos && i == a.eos) x = -INFINITY;
                prob[i] = x;
                mx = fmaxf(mx, x);
            }
        }
    }
    mx = xmax<1>(xmax<2>(xmax<4>(xmax<8>(xmax<16>(xmax<32>(mx))))));      // xlane.h: VALU lane exchanges instead of six ds_bpermute round trips
    if (lane == 0) red_max[wid] = mx;
    __syncthreads();
    mx = red_max[0];
#pragma unroll
    for (int w = 1; w < RS_NT / 64; ++w) mx = fmaxf(mx, red_max[w]);
    float sum = 0.0f;
    for (int i = tid; i < a.v; i += RS_NT) {
        const float e = __expf(prob[i] - mx);
        prob[i] = e;
        sum += e;
    }
    sum = wave_sum_desc(sum);                                            // same butterfly order (32, 16, ..., 1), same bits
    if (lane == 0) red_sum[wid] = sum;
    __syncthreads();
    float tot = 0.0f;      // fixed summation order (wave partials 0..15): same bits on every run
#pragma unroll
    for (int w = 0; w < RS_NT / 64; ++w) tot += red_sum[w];
    const float inv = 1.0f / tot;
    // top_k by (p desc, id asc).  Fast path: the exact kk-th largest probability by a 3-pass radix select on the float
    // bits (LDS histograms), then the <= 64 entries >= it are gathered and sorted by one wave.  If ties push the
    // gather past 64 entries (e.g. fewer than kk non-zero probabilities) the general chunked path below runs instead.
    const int kk = a.top_k < 64 ? a.top_k : 64;
    TopList<float> tl;
    tl.init();
    unsigned prefix = 0u, known = 0u;
    {
        int remaining = kk < a.v ? kk : a.v;
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
            const int bins = pass == 2 ? 1024 : 2048;
            unsigned* h = hist[pass];
            for (int i = tid; i < a.v; i += RS_NT) {
                float pv = prob[i];
                if (pass == 0) {                      // the normalisation rides on the first histogram pass
                    pv *= inv;
                    prob[i] = pv;
                }
                const unsigned key = __float_as_uint(pv);
                if ((key & known) == prefix) atomicAdd(&h[(key >> shift) & (bins - 1)], 1u);
            }
            __syncthreads();
            if (wid == 0) {
                const int per = bins >> 6;                   // 32 or 16 bins per lane, kept in registers for both walks
                unsigned hv[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) hv[j] = j < per ? h[lane * per + j] : 0u;
                unsigned local = 0u;
#pragma unroll
                for (int j = 0; j < 32; ++j) local += hv[j];
                unsigned incl = local;                       // sum over lanes >= lane
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned tv = __shfl_down(incl, off, 64);
                    if (lane + off < 64) incl += tv;
                }
                const unsigned above = incl - local;
                if (above < (unsigned)remaining && (unsigned)remaining <= incl) {
                    unsigned acc = above;
                    bool done = false;
#pragma unroll
                    for (int j = 31; j >= 0; --j) {
                        if (j < per && !done) {
                            if (acc + hv[j] >= (unsigned)remaining) {
                                s_sel_bin[pass] = lane * per + j;
                                s_sel_rem[pass] = remaining - (int)acc;
                                done = true;
                            } else {
                                acc += hv[j];
                            }
                        }
                    }
                }
            }
            __syncthreads();
            prefix |= (unsigned)s_sel_bin[pass] << shift;
            known |= (unsigned)(bins - 1) << shift;
            remaining = s_sel_rem[pass];
        }
        for (int i = tid; i < a.v; i += RS_NT) {
            if (__float_as_uint(prob[i]) >= prefix) {
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < 64) {
                    sh_s[pos] = prob[i];
                    sh_i[pos] = i;
                }
            }
        }
        __syncthreads();
    }
    const bool fast = s_cnt <= 64;
    if (fast) {
        if (wid == 0) tl.seed(sh_s[lane], sh_i[lane], lane < s_cnt, lane);   // sort fixes the order whatever the gather order was
    } else {
        bool seeded = false;
        for (int base = wid * 64; base < a.v; base += RS_NT) {
            const int i = base + lane;
            const bool valid = i < a.v;
            const float pv = valid ? prob[i] : -INFINITY;
            if (!seeded) {
                tl.seed(pv, i, valid, lane);
                seeded = true;
            } else {
                tl.offer(pv, i, valid, lane, kk);
            }
        }
        __syncthreads();
        merge_lists<float>(tl, sh_s, sh_i, kk);
    }
    if (wid == 0) {
        // nucleus: sequential float accumulation in rank order (matches the oracle's definition bit for bit).  The sorted list
        // goes through LDS so that the <= 64 loads are independent of the running sum (as __shfl of rank r inside a loop with
        // `break` every rank cost a dependent ds_bpermute round trip: ~50 of them, 2.5 us of this kernel)
        sh_s[lane] = tl.s;
        sh_i[lane] = tl.idx;
        __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0), then the wave re-converges: the list is visible to every lane
        __builtin_amdgcn_wave_barrier();
        float cum = 0.0f;
        int cnt = 0;
        bool open = true;
#pragma unroll 8
        for (int r = 0; r < kk; ++r) {
            const float pr = sh_s[r];
            const int ir = sh_i[r];
            open = open && ir != kNoIdx && cum < a.top_p && cnt < a.top_k;
            if (open) {
                cum += pr;
                ++cnt;
            }
        }
        // repetitions of a token among the last `win` decoded ones: lane j compares the j-th token of the window (one round trip,
        // not `win` dependent ones); `t` is wave-uniform
        auto repeated = [&](int t) -> bool {
            int rep = 0;
            if (win_in_wave) {
                rep = __popcll(__ballot(hwin == t));         // lanes outside the window hold -1
            } else {
                for (int i0 = h0; i0 < a.hist_len; i0 += 64) {
                    const int i = i0 + lane;
                    const bool hit = i < a.hist_len && a.history[(int64_t)bb * a.hist_ld + i] == t;
                    rep += __popcll(__ballot(hit));
                }
            }
            return (float)rep >= (float)a.win * a.tau_r;
        };
        int tok;
        bool fallback;
        if (!reject) {
            const float target = u_first * cum;
            float run = 0.0f;
            tok = sh_i[cnt > 0 ? cnt - 1 : 0];
            bool found = false;
#pragma unroll 8
            for (int r = 0; r < kk; ++r) {
                const float pr = sh_s[r];
                const int ir = sh_i[r];
                if (r < cnt) {
                    run += pr;
                    if (!found && run > target) {
                        tok = ir;
                        found = true;
                    }
                }
            }
            fallback = repeated(tok);
        } else {
            // "reject": the closed form of upstream's re-draw loop (one pass = nucleus draw, repetition check, full-distribution
            // draw when repeated; start over while the result is EOS).  One pass ends on a nucleus entry t that is neither EOS nor
            // repeated with probability p_t / cum, in the fallback with rho = sum over the repeated entries of p_t / cum, and the
            // fallback ends off EOS with probability 1 - p_eos.  Conditioned on "not EOS": weights a_t = p_t for the direct entries and
            // F = (sum of the repeated p_t) (1 - p_eos) for the fallback -- drawn with u1 in rank order, the fallback last.
            // Lane r holds rank r (probability, id, "repeated"): the walks below then read lanes (v_readlane with a uniform index), not LDS
            // words behind ballots -- as dependent LDS round trips per rank this branch cost ~4 us of a 14 us kernel.
            const float p_eos = prob[a.eos];
            const float pr_l = lane < cnt ? sh_s[lane] : 0.0f;
            const int ir_l = lane < cnt ? sh_i[lane] : -1;
            int repc = 0;
            if (win_in_wave) {
                const int wl = a.hist_len - h0;             // window tokens sit on lanes 0 .. wl - 1 of hwin
                for (int j = 0; j < wl; ++j) repc += __builtin_amdgcn_readlane(hwin, j) == ir_l ? 1 : 0;
            } else {
                for (int i = h0; i < a.hist_len; ++i) repc += a.history[(int64_t)bb * a.hist_ld + i] == ir_l ? 1 : 0;
            }
            const bool rep_l = (float)repc >= (float)a.win * a.tau_r;
            const unsigned long long direct = __ballot(lane < cnt && ir_l != a.eos && !rep_l);      // neither EOS nor repeated
            const unsigned long long again = __ballot(lane < cnt && ir_l != a.eos && rep_l);        // repeated: their mass goes to the fallback
            float asum = 0.0f, prep = 0.0f;                  // fp32 sums in rank order: the oracle's order
            for (int r = 0; r < cnt; ++r) {
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr_l), r));
                if (direct >> r & 1ull) asum += v;
                else if (again >> r & 1ull) prep += v;
            }
            const float target = u_first * (asum + prep * (1.0f - p_eos));
            float run = 0.0f;
            tok = -1;
            for (int r = 0; r < cnt && tok < 0; ++r) {
                if (direct >> r & 1ull) {
                    run += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr_l), r));
                    if (run > target) tok = __builtin_amdgcn_readlane(ir_l, r);
                }
            }
            fallback = tok < 0;                             // also: a nucleus that holds nothing but EOS
        }
        if (lane == 0) {
            s_tok = tok;
            s_bcast = fallback ? 1.0f : 0.0f;
        }
    }
    __syncthreads();
    if (s_bcast > 0.5f && tid == 0) {
        // repetition detected: random sampling from the full distribution, id order, inverse CDF with u2.  The running sum is
        // the oracle's sequential fp32 accumulation (a parallel scan rounds differently), taken 16 elements per LDS round trip
        // with ONE comparison per chunk: ~20 us worst case (an LDS read + compare + branch per element took 270 us).
        // "reject": the same walk over the distribution without EOS (its term is skipped; the target scales by 1 - p_eos).
        const float p_skip = reject ? prob[a.eos] : 0.0f;
        if (reject) prob[a.eos] = 0.0f;                      // only this thread reads prob from here on
        const float target = reject ? u_second * (1.0f - p_skip) : u_second;
        float run = 0.0f;
        int tok = -1;
        const int vpad = (a.v + 15) & ~15;
        for (int base = 0; base < vpad && tok < 0; base += 16) {
            const float4 q0 = *reinterpret_cast<const float4*>(prob + base), q1 = *reinterpret_cast<const float4*>(prob + base + 4);
            const float4 q2 = *reinterpret_cast<const float4*>(prob + base + 8), q3 = *reinterpret_cast<const float4*>(prob + base + 12);
            const float pv[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            float r[16];
            r[0] = run + pv[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) r[k] = r[k - 1] + pv[k];
            if (r[15] > target) {                    // the sums are non-decreasing: the first crossing is in this chunk
#pragma unroll
                for (int k = 15; k >= 0; --k)
                    if (r[k] > target) tok = base + k;
            }
            run = r[15];
        }
        if (tok < 0 || tok >= a.v) {                 // the sum never exceeded the target: the last token with p > 0
            tok = 0;
            for (int i = a.v - 1; i > 0; --i)
                if (prob[i] > 0.0f) {
                    tok = i;
                    break;
                }
        }
        s_tok = tok;
    }
    __syncthreads();
    if (tid == 0) {
        int tok = s_tok;
        if (a.forced) tok = a.forced[(int64_t)bb * a.hist_ld + a.hist_len];
        if (a.hist_w) a.hist_w[(int64_t)bb * a.hist_ld + a.hist_len] = tok;
        a.out[bb] = (a.clamp_out >= 0 && tok > a.clamp_out) ? a.clamp_out : tok;
    }
}


// ------------------------------------------------------------------------------------------ frontend
// Polyphase resampler of cosyvoice.utils.file_utils.load_wav / the prompt path (tts_with_rag.py:180-186 -> frontend [EXT]):
// y[f * up + p] = sum_j kern[p][j] * xpad[f * down + j], xpad = x zero-padded by `width` on the left (the host builds the
// Hann-windowed sinc table exactly as astts.audio.resample does).  One thread per output sample.
__global__ __launch_bounds__(256) void resample_poly(const float* __restrict__ x, const float* __restrict__ kern, float* __restrict__ y,
                                                     int b, int64_t n_in, int64_t n_out, int up, int down, int width, int taps) {
    const int64_t total = (int64_t)b * n_out;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t bb = i / n_out, o = i - bb * n_out;
        const int64_t f = o / up;
        const int ph = (int)(o - f * up);
        const float* xr = x + bb * n_in;
        const float* kr = kern + (int64_t)ph * taps;
        const int64_t s0 = f * down - width;          // input index of tap 0
        float acc = 0.0f;
        for (int j = 0; j < taps; ++j) {
            const int64_t s = s0 + j;
            const float kv = kr[j];
            if (kv != 0.0f && s >= 0 && s < n_in) acc = fmaf(kv, xr[s], acc);
        }
        y[i] = acc;
    }
}

// Log-mel spectrogram (matcha / whisper style): reflect padding (n_fft - hop) / 2, Hann window, magnitude of the n_fft-point
// DFT, mel filterbank, log(clamp(., floor)).  One block per frame: the windowed frame and the twiddle table sit in LDS, every
// thread evaluates its bins by direct summation in fp32 (index k n mod n_fft into the table) -- n_fft <= 1024 and a few
// hundred frames per prompt: 0.3 GFLOP, far below anything worth an FFT.
// WHISPER = the log-mel of Whisper's feature extractor (the input of the reference's speech tokenizer): centred frames
// (reflect padding n_fft / 2), POWER spectrum, log10(max(., floor)), output [b][n_mels][frames], and the maximum over an utterance's
// (frame, bin) values gathered in gmax[b] (order-preserving integer image of the float) for the "max - 8" floor of whisper_floor.
// KALDI = the fbank of Kaldi's compute-fbank-feats (the input of the reference's speaker-embedding net): frames of
// frame_len samples every hop samples WITHOUT padding, each scaled, freed of its DC offset, pre-emphasised (x[i] -= c x[i-1]; x[0] *=
// 1 - c), multiplied by window[frame_len] and zero-padded to n_fft; power spectrum, natural log, output [b][frames][n_mels].
enum { MEL_MATCHA = 0, MEL_WHISPER = 1, MEL_KALDI = 2 };
template <int MODE>
__global__ __launch_bounds__(256) void mel_frames(const float* __restrict__ wav, const float* __restrict__ window,
                                                  const float* __restrict__ fb, float* __restrict__ out, int64_t n, int frames,
                                                  int n_fft, int hop, int n_mels, float floor_, int* __restrict__ gmax, int frame_len,
                                                  float scale, float preemph) {
    constexpr bool WHISPER = MODE == MEL_WHISPER;
    constexpr bool KALDI = MODE == MEL_KALDI;
    extern __shared__ float sm[];
    float* fr = sm;                       // [n_fft] windowed frame
    float* twc = sm + n_fft;              // [n_fft] cos(2 pi i / n_fft)
    float* tws = twc + n_fft;             // [n_fft] sin
    float* mag = tws + n_fft;             // [n_fft / 2 + 1]
    const int f = blockIdx.x, bb = blockIdx.y, tid = threadIdx.x;
    const int pad = WHISPER ? n_fft / 2 : (n_fft - hop) / 2;
    const float* w = wav + (int64_t)bb * n;
    int n_time = n_fft;                                        // samples of the frame that can be non-zero
    if constexpr (KALDI) {
        __shared__ float s_part[4];
        n_time = frame_len;
        const float* src = w + (int64_t)f * hop;               // frames never reach beyond the signal (snip_edges)
        float part = 0.0f;
        for (int i = tid; i < n_fft; i += 256) {
            const float v = i < frame_len ? src[i] * scale : 0.0f;
            fr[i] = v;
            part += v;
            float sv, cv;
            sincospif(2.0f * (float)i / (float)n_fft, &sv, &cv);
            twc[i] = cv;
            tws[i] = sv;
        }
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
        if ((tid & 63) == 0) s_part[tid >> 6] = part;
        __syncthreads();
        const float mean = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) / (float)frame_len;
        // pre-emphasis reads the neighbour's value BEFORE anybody overwrites it: values into registers, barrier, then write back
        float cur[2], prv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j;
            cur[j] = i < frame_len ? fr[i] - mean : 0.0f;
            prv[j] = (i >= 1 && i < frame_len) ? fr[i - 1] - mean : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j;
            if (i < frame_len) fr[i] = (i == 0 ? cur[j] * (1.0f - preemph) : cur[j] - preemph * prv[j]) * window[i];
        }
    } else {
        for (int i = tid; i < n_fft; i += 256) {
            int64_t s = (int64_t)f * hop + i - pad;
            if (s < 0) s = -s;                                     // reflect (no edge repeat)
            if (s >= n) s = 2 * (n - 1) - s;
            s = s < 0 ? 0 : (s >= n ? n - 1 : s);
            fr[i] = w[s] * window[i];
            float sv, cv;
            sincospif(2.0f * (float)i / (float)n_fft, &sv, &cv);
            twc[i] = cv;
            tws[i] = sv;
        }
    }
    __syncthreads();
    const int nb = n_fft / 2 + 1;
    for (int k = tid; k < nb; k += 256) {
        float re = 0.0f, im = 0.0f;
        int idx = 0;
        for (int i = 0; i < n_time; ++i) {
            re = fmaf(fr[i], twc[idx], re);
            im = fmaf(fr[i], tws[idx], im);
            idx += k;
            if (idx >= n_fft) idx -= n_fft;
        }
        mag[k] = (WHISPER || KALDI) ? re * re + im * im : sqrtf(re * re + im * im + 1e-9f);
    }
    __syncthreads();
    float vmax = -INFINITY;
    for (int m = tid; m < n_mels; m += 256) {
        const float* fr_ = fb + (int64_t)m * nb;
        float acc = 0.0f;
        for (int k = 0; k < nb; ++k) acc = fmaf(fr_[k], mag[k], acc);
        if constexpr (WHISPER) {
            const float v = log10f(fmaxf(acc, floor_));
            out[((int64_t)bb * n_mels + m) * frames + f] = v;
            vmax = fmaxf(vmax, v);
        } else {
            out[((int64_t)bb * frames + f) * n_mels + m] = logf(fmaxf(acc, floor_));
        }
    }
    if constexpr (WHISPER) {
        for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if ((tid & 63) == 0 && vmax > -INFINITY) {
            // float order as integer order: non-negative floats compare as their bits, negative ones in reverse
            const int bits = __float_as_int(vmax);
            atomicMax(gmax + bb, bits >= 0 ? bits : (int)(0x80000000u - (unsigned)bits));
        }
    }
}

// second pass of the Whisper features: x -> (max(x, utterance maximum - 8) + 4) / 4
__global__ __launch_bounds__(256) void whisper_floor(float* __restrict__ x, const int* __restrict__ gmax, int64_t per_item, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int key = gmax[i / per_item];
        const float mx = __int_as_float(key >= 0 ? key : (int)(0x80000000u - (unsigned)key));
        x[i] = (fmaxf(x[i], mx - 8.0f) + 4.0f) * 0.25f;
    }
}

}  // namespace astts

using namespace astts;

static inline int grid_for_a(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" {

size_t astts_op_nsf_source_workspace_bytes(int32_t b, int32_t tm) { return (size_t)b * tm * sizeof(double); }

int astts_op_nsf_source(const float* f0, const float* phase0, const float* noise, const float* lin_w, const float* lin_b,
                        float* out, int32_t b, int32_t tm, int32_t upsample, int32_t n_harm_plus1, float sample_rate,
                        float sine_amp, float noise_std, float voiced_threshold, void* workspace, size_t workspace_bytes,
                        astts_stream_t stream) {
    ASTTS_REQUIRE(f0 && phase0 && noise && lin_w && lin_b && out && workspace, ASTTS_ERR_INVALID, "astts_op_nsf_source: null pointer");
    ASTTS_REQUIRE(b >= 1 && tm >= 1 && upsample >= 1 && n_harm_plus1 >= 1 && n_harm_plus1 <= 32, ASTTS_ERR_INVALID,
                  "astts_op_nsf_source: bad shape");
    ASTTS_REQUIRE(workspace_bytes >= astts_op_nsf_source_workspace_bytes(b, tm), ASTTS_ERR_WORKSPACE,
                  "astts_op_nsf_source: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(f0_prefix, dim3(b), dim3(256), 0, st, f0, (double*)workspace, tm);
    ASTTS_CHECK_LAUNCH();
    NsfArgs a{f0, (const double*)workspace, phase0, noise, lin_w, lin_b, out, b, tm, upsample, n_harm_plus1,
              sample_rate, sine_amp, noise_std, voiced_threshold};
    hipLaunchKernelGGL(nsf_source, dim3(grid_for_a((int64_t)b * tm * upsample)), dim3(256), 0, st, a);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_stft16(const float* x, float* y, int32_t b, int64_t n_samples, astts_stream_t stream) {
    return astts_op_stft16_lens(x, y, b, n_samples, nullptr, stream);
}

int astts_op_stft16_lens(const float* x, float* y, int32_t b, int64_t n_samples, const int32_t* lens, astts_stream_t stream) {
    ASTTS_REQUIRE(x && y && b >= 1 && n_samples >= 16 && n_samples % 4 == 0, ASTTS_ERR_INVALID, "astts_op_stft16: bad argument");
    const int64_t F = n_samples / 4 + 1;
    hipLaunchKernelGGL(stft16, dim3(grid_for_a((int64_t)b * F)), dim3(256), 0, (hipStream_t)stream, x, y, b, n_samples, F, lens);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_istft16(const float* y, float* wav, int32_t b, int64_t frames, float mag_clip, float audio_limit,
                     astts_stream_t stream) {
    return astts_op_istft16_lens(y, wav, b, frames, mag_clip, audio_limit, nullptr, stream);
}

int astts_op_istft16_lens(const float* y, float* wav, int32_t b, int64_t frames, float mag_clip, float audio_limit, const int32_t* frame_lens,
                          astts_stream_t stream) {
    ASTTS_REQUIRE(y && wav && b >= 1 && frames >= 2, ASTTS_ERR_INVALID, "astts_op_istft16: bad argument");
    hipLaunchKernelGGL(istft16, dim3(grid_for_a((int64_t)b * 4 * (frames - 1))), dim3(256), 0, (hipStream_t)stream, y,
                       wav, b, frames, mag_clip, audio_limit, frame_lens);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_ras_sample(const float* logits, const int32_t* history, const float* uniforms, int32_t* out_tokens,
                        int32_t b, int32_t vocab, int32_t hist_len, int32_t hist_ld, int32_t top_k, float top_p,
                        int32_t win_size, float tau_r, int32_t eos_id, int32_t ignore_eos, astts_stream_t stream) {
    ASTTS_REQUIRE(logits && uniforms && out_tokens && (history || hist_len == 0), ASTTS_ERR_INVALID, "astts_op_ras_sample: null pointer");
    ASTTS_REQUIRE(b >= 1 && vocab >= 2 && vocab <= 15000 && top_k >= 1 && top_k <= 64 && hist_len >= 0, ASTTS_ERR_INVALID,
                  "astts_op_ras_sample: bad shape b=%d vocab=%d top_k=%d", b, vocab, top_k);
    // the reject policy (bit 1) reads and clears prob[eos_id] in LDS: the id must be a vocabulary entry (the mask policy only compares with it)
    ASTTS_REQUIRE(!(ignore_eos & 2) || (eos_id >= 0 && eos_id < vocab), ASTTS_ERR_RANGE, "astts_op_ras_sample: eos_id=%d outside [0, %d) with the reject policy", eos_id, vocab);
    SampleArgs a{logits, history, uniforms, out_tokens, nullptr, nullptr, -1, nullptr, b, vocab, hist_len, hist_ld, top_k, win_size, eos_id, ignore_eos, top_p, tau_r};
    hipLaunchKernelGGL(ras_sample, dim3(b), dim3(RS_NT), (size_t)((vocab + 15) & ~15) * sizeof(float), (hipStream_t)stream, a.logits, a.eos_min_rows, a.v,
                       a.hist_len, a.eos, a.ignore_eos, a);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

/* Engine form: `history` is read (repetition window) and written at column hist_len; optional teacher forcing. */
int astts_op_ras_sample_ex(const float* logits, int32_t* history, const float* uniforms, int32_t* out_tokens, int32_t b,
                           int32_t vocab, int32_t hist_len, int32_t hist_ld, int32_t top_k, float top_p, int32_t win_size,
                           float tau_r, int32_t eos_id, int32_t ignore_eos, const int32_t* eos_min_rows, const int32_t* forced,
                           astts_stream_t stream) {
    ASTTS_REQUIRE(logits && uniforms && out_tokens && history, ASTTS_ERR_INVALID, "astts_op_ras_sample_ex: null pointer");
    ASTTS_REQUIRE(b >= 1 && vocab >= 2 && vocab <= 15000 && top_k >= 1 && top_k <= 64 && hist_len >= 0 && hist_len < hist_ld,
                  ASTTS_ERR_INVALID, "astts_op_ras_sample_ex: bad shape b=%d vocab=%d top_k=%d hist_len=%d", b, vocab, top_k, hist_len);
    ASTTS_REQUIRE(!(ignore_eos & 2) || (eos_id >= 0 && eos_id < vocab), ASTTS_ERR_RANGE, "astts_op_ras_sample_ex: eos_id=%d outside [0, %d) with the reject policy", eos_id, vocab);
    SampleArgs a{logits, history, uniforms, out_tokens, history, forced, eos_id - 1, eos_min_rows, b, vocab, hist_len, hist_ld, top_k, win_size,
                 eos_id, ignore_eos, top_p, tau_r};
    hipLaunchKernelGGL(ras_sample, dim3(b), dim3(RS_NT), (size_t)((vocab + 15) & ~15) * sizeof(float), (hipStream_t)stream, a.logits, a.eos_min_rows, a.v,
                       a.hist_len, a.eos, a.ignore_eos, a);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}


int astts_op_resample_poly(const float* x, const float* kern, float* y, int32_t b, int64_t n_in, int64_t n_out, int32_t up, int32_t down,
                           int32_t width, astts_stream_t stream) {
    ASTTS_REQUIRE(x && kern && y, ASTTS_ERR_INVALID, "astts_op_resample_poly: null pointer");
    ASTTS_REQUIRE(b >= 1 && n_in >= 1 && n_out >= 1 && up >= 1 && down >= 1 && width >= 0, ASTTS_ERR_INVALID,
                  "astts_op_resample_poly: bad shape");
    hipLaunchKernelGGL(resample_poly, dim3(grid_for_a((int64_t)b * n_out)), dim3(256), 0, (hipStream_t)stream, x, kern, y, b, n_in, n_out, up,
                       down, width, 2 * width + down);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_mel_spectrogram(const float* wav, const float* window, const float* mel_fb, float* out, int32_t b, int64_t n_samples,
                             int32_t n_fft, int32_t hop, int32_t n_mels, float log_floor, astts_stream_t stream) {
    ASTTS_REQUIRE(wav && window && mel_fb && out, ASTTS_ERR_INVALID, "astts_op_mel_spectrogram: null pointer");
    ASTTS_REQUIRE(b >= 1 && n_fft >= 16 && n_fft <= 2048 && hop >= 1 && hop <= n_fft && n_mels >= 1 && log_floor > 0.0f,
                  ASTTS_ERR_INVALID, "astts_op_mel_spectrogram: bad shape n_fft=%d hop=%d n_mels=%d", n_fft, hop, n_mels);
    const int pad = (n_fft - hop) / 2;
    ASTTS_REQUIRE(n_samples > pad, ASTTS_ERR_INVALID, "astts_op_mel_spectrogram: %lld samples are shorter than the reflect padding %d",
                  (long long)n_samples, pad);
    const int64_t frames = (n_samples + 2 * pad - n_fft) / hop + 1;
    ASTTS_REQUIRE(frames >= 1 && frames < (1 << 30), ASTTS_ERR_INVALID, "astts_op_mel_spectrogram: frames=%lld", (long long)frames);
    const size_t lds = sizeof(float) * ((size_t)3 * n_fft + n_fft / 2 + 1);
    hipLaunchKernelGGL(mel_frames<MEL_MATCHA>, dim3((unsigned)frames, (unsigned)b), dim3(256), lds, (hipStream_t)stream, wav, window, mel_fb, out,
                       n_samples, (int)frames, n_fft, hop, n_mels, log_floor, (int*)nullptr, n_fft, 1.0f, 0.0f);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

int astts_op_kaldi_fbank(const float* wav, const float* window, const float* mel_fb, float* out, int32_t b, int64_t n_samples, int32_t frame_len,
                         int32_t hop, int32_t n_fft, int32_t n_mels, float scale, float preemph, float log_floor, astts_stream_t stream) {
    ASTTS_REQUIRE(wav && window && mel_fb && out, ASTTS_ERR_INVALID, "astts_op_kaldi_fbank: null pointer");
    ASTTS_REQUIRE(b >= 1 && frame_len >= 16 && frame_len <= n_fft && n_fft <= 512 && hop >= 1 && n_mels >= 1 && log_floor > 0.0f, ASTTS_ERR_INVALID,
                  "astts_op_kaldi_fbank: bad shape frame_len=%d n_fft=%d hop=%d n_mels=%d (one thread holds two samples of a frame: n_fft <= 512)",
                  frame_len, n_fft, hop, n_mels);
    ASTTS_REQUIRE(n_samples >= frame_len, ASTTS_ERR_INVALID, "astts_op_kaldi_fbank: %lld samples are shorter than one frame of %d",
                  (long long)n_samples, frame_len);
    const int64_t frames = 1 + (n_samples - frame_len) / hop;
    ASTTS_REQUIRE(frames < (1 << 30), ASTTS_ERR_INVALID, "astts_op_kaldi_fbank: frames=%lld", (long long)frames);
    const size_t lds = sizeof(float) * ((size_t)3 * n_fft + n_fft / 2 + 1);
    hipLaunchKernelGGL(mel_frames<MEL_KALDI>, dim3((unsigned)frames, (unsigned)b), dim3(256), lds, (hipStream_t)stream, wav, window, mel_fb, out,
                       n_samples, (int)frames, n_fft, hop, n_mels, log_floor, (int*)nullptr, frame_len, scale, preemph);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

size_t astts_op_whisper_log_mel_workspace_bytes(int32_t b) { return (size_t)(b > 0 ? b : 0) * sizeof(int); }

int astts_op_whisper_log_mel(const float* wav, const float* window, const float* mel_fb, float* out, int32_t b, int64_t n_samples,
                             int32_t n_fft, int32_t hop, int32_t n_mels, void* workspace, size_t workspace_bytes, astts_stream_t stream) {
    ASTTS_REQUIRE(wav && window && mel_fb && out && workspace, ASTTS_ERR_INVALID, "astts_op_whisper_log_mel: null pointer");
    ASTTS_REQUIRE(b >= 1 && n_fft >= 16 && n_fft <= 2048 && hop >= 1 && hop <= n_fft && n_mels >= 1, ASTTS_ERR_INVALID,
                  "astts_op_whisper_log_mel: bad shape n_fft=%d hop=%d n_mels=%d", n_fft, hop, n_mels);
    ASTTS_REQUIRE(workspace_bytes >= astts_op_whisper_log_mel_workspace_bytes(b), ASTTS_ERR_WORKSPACE, "astts_op_whisper_log_mel: workspace too small");
    ASTTS_REQUIRE(n_samples > n_fft / 2, ASTTS_ERR_INVALID, "astts_op_whisper_log_mel: %lld samples are shorter than the reflect padding %d",
                  (long long)n_samples, n_fft / 2);
    const int64_t frames = n_samples / hop;      // centred frames 0 .. n / hop, the last one dropped (whisper.log_mel_spectrogram)
    ASTTS_REQUIRE(frames >= 1 && frames < (1 << 30), ASTTS_ERR_INVALID, "astts_op_whisper_log_mel: frames=%lld", (long long)frames);
    hipStream_t st = (hipStream_t)stream;
    // every key starts below any float's image (the image of -inf is 0x80000000 - 0xff800000 = 0x80800000 as int: still above INT_MIN)
    ASTTS_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)workspace, (int)0x80000000u, (size_t)b, st));
    const size_t lds = sizeof(float) * ((size_t)3 * n_fft + n_fft / 2 + 1);
    hipLaunchKernelGGL(mel_frames<MEL_WHISPER>, dim3((unsigned)frames, (unsigned)b), dim3(256), lds, st, wav, window, mel_fb, out, n_samples, (int)frames,
                       n_fft, hop, n_mels, 1e-10f, (int*)workspace, n_fft, 1.0f, 0.0f);
    ASTTS_CHECK_LAUNCH();
    const int64_t total = (int64_t)b * n_mels * frames;
    hipLaunchKernelGGL(whisper_floor, dim3(grid_for_a(total)), dim3(256), 0, st, out, (const int*)workspace, (int64_t)n_mels * frames, total);
    ASTTS_CHECK_LAUNCH();
    return ASTTS_OK;
}

}  // extern "C"
