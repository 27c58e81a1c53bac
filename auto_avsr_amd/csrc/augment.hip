// augment.hip -- the reference's per-sample augmentation / normalisation transforms and its padding collation as two
// batch-level kernels that run where the data is consumed (input pipeline on the device).
//
// Replaces, for a whole batch at once (the reference runs these per sample in DataLoader workers on the CPU):
//   VideoTransform           datamodule/transforms.py:89-110   x/255 -> Random/CenterCrop(88) -> Grayscale ->
//                                                               AdaptiveTimeMask(10, 25) -> Normalize(0.421, 0.165)
//   AudioTransform           datamodule/transforms.py:113-136  AdaptiveTimeMask(6400, 16000) -> AddNoise -> layer_norm
//   AdaptiveTimeMask.forward datamodule/transforms.py:50-64    (the intervals are drawn on the host with the reference's
//                                                               own RNG call sequence; the kernels apply them)
//   AddNoise.forward         datamodule/transforms.py:81-88    torchaudio.functional.add_noise (absent dependency,
//                                                               restated: scale noise to the requested SNR, add)
//   pad / collate_pad        datamodule/data_module.py:10-41   zero padding of every sample to the longest of the batch
//
// Both kernels are HBM-bound streaming passes: video reads 3 B and writes 4 B (or 2 B) per output pixel straight from
// the decoder's THWC uint8 frames into the padded [B, Tmax, 1, 88, 88] batch tensor the stem kernel consumes; nothing
// is materialised in between (the reference materialises five intermediate tensors per clip).  The f32 arithmetic of
// the video path is issued with explicitly rounded, non-contractable operations in the reference's order, so the result
// is bit-identical to torch's CPU ops.
#include <math.h>
#include "prims.h"
#include "avsr_hip.h"

// Every f32 operation in this file is individually rounded: the file is compiled with FMA contraction off
// AVSR_CXXFLAGS: -ffp-contract=off
// (build.py appends the flags of the line above to this file's compile command; the build's default is
// -ffp-contract=fast, under which the backend fuses mul + add pairs regardless of `#pragma clang fp contract(off)`, and
// HIP's __fmul_rn / __fadd_rn are plain operators that get fused too), and division is hipcc's default correctly rounded
// one.  That is what makes the video path bit-identical to torch's CPU ops.

namespace {

// ---- individually rounded f32 operations.  Device: plain operators under -ffp-contract=off (a `volatile` temporary
// would be a scratch-memory round trip per operation -- measured 8x slower).  Host emulator: volatile
// temporaries keep the host compiler from fusing them.
#ifdef AVSR_EMU
#define AVSR_RN_TMP volatile float
#else
#define AVSR_RN_TMP float
#endif
AVSR_DEV float mul_rn(float a, float b) {
    AVSR_RN_TMP r = a * b;
    return r;
}
AVSR_DEV float add_rn(float a, float b) {
    AVSR_RN_TMP r = a + b;
    return r;
}
AVSR_DEV float div_rn(float a, float b) {
    AVSR_RN_TMP r = a / b;
    return r;
}

// is frame / sample t inside one of the n masking intervals [iv[2i], iv[2i+1]) ?
AVSR_DEV bool masked_at(const int* __restrict__ iv, int n, long t) {
    bool m = false;
    for (int i = 0; i < n; i++) m = m || (t >= iv[2 * i] && t < iv[2 * i + 1]);
    return m;
}

// grid (ceil(crop*crop/8 / 256), Tmax, B); thread = 8 consecutive pixels of one output row (crop % 8 == 0)
template <class T>
__global__ __launch_bounds__(256) void video_transform_kernel(
    const int64_t* __restrict__ src_ptr, const int32_t* __restrict__ lens,
    const int32_t* __restrict__ crop_y, const int32_t* __restrict__ crop_x, const int32_t* __restrict__ iv,
    const int32_t* __restrict__ niv, int max_iv, T* __restrict__ out, int Tmax, int H, int W, int crop, float mean,
    float std) {
    const int b = blockIdx.z, t = blockIdx.y;
    const int per_row = crop >> 3;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= crop * per_row) return;
    const int y = id / per_row, x0 = (id - y * per_row) * 8;
    T* o = out + (((long)b * Tmax + t) * crop + y) * crop + x0;
    float v[8];
    if (t >= lens[b]) {  // collate_pad: frames past the clip are zeros
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = 0.f;
        store8(o, v);
        return;
    }
    const bool masked = iv != nullptr && masked_at(iv + (long)b * max_iv * 2, niv[b], t);
    if (masked) {  // AdaptiveTimeMask zeroes the grey frame BEFORE Normalize
        const float z = div_rn(add_rn(0.f, -mean), std);
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = z;
        store8(o, v);
        return;
    }
    const uint8_t* p = reinterpret_cast<const uint8_t*>(src_ptr[b]) + (((long)t * H + crop_y[b] + y) * (long)W + crop_x[b] + x0) * 3;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        // x / 255.0 per channel, then torchvision's rgb_to_grayscale: 0.2989 r + 0.587 g + 0.114 b, then (v - mean) / std
        const float r = div_rn((float)p[3 * k], 255.0f), g = div_rn((float)p[3 * k + 1], 255.0f),
                    bl = div_rn((float)p[3 * k + 2], 255.0f);
        const float grey = add_rn(add_rn(mul_rn(0.2989f, r), mul_rn(0.587f, g)), mul_rn(0.114f, bl));
        v[k] = div_rn(add_rn(grey, -mean), std);
    }
    store8(o, v);
}

constexpr int AUD_THREADS = 256;
constexpr int AUD_CHUNK = 2048;  // samples per block: 8 per thread, all loads of a thread in flight together
constexpr int AUD_PER = AUD_CHUNK / AUD_THREADS;

AVSR_DEV double block_sum_d(double v, double* red) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    __syncthreads();  // red may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < AUD_THREADS / 64; w++) s += red[w];
    return s;
}

// mask -> add noise at the requested SNR -> layer_norm over the whole utterance -> padded batch, as four grid-wide
// phases (an utterance is up to 256 000 samples: one block per utterance would leave 250 CUs idle for milliseconds).
// Block (chunk, utterance); the per-utterance statistics travel through per-chunk partial sums in `part`
// ([3 phases][B][nch][2] doubles, reduced in one fixed order by every block that needs them -- deterministic).
//   PHASE 0: E_speech, E_noise over the masked speech / the noise segment
//   PHASE 1: y = speech + scale * noise  -> written to `out` (un-normalised), sum(y)
//   PHASE 2: sum((y - mean)^2) from `out`
//   PHASE 3: out = (y - mean) * rstd, zero tail
template <int PHASE>
__global__ __launch_bounds__(AUD_THREADS) void audio_phase_kernel(
    const int64_t* __restrict__ wav_ptr, const int32_t* __restrict__ lens, const int32_t* __restrict__ iv,
    const int32_t* __restrict__ niv, int max_iv, const float* __restrict__ noise,
    const int64_t* __restrict__ noise_start, const float* __restrict__ snr_db, float eps, float* __restrict__ out,
    long Lmax, double* __restrict__ part, int nch) {
    __shared__ double red[AUD_THREADS / 64];
    const int b = blockIdx.y, ch = blockIdx.x, B = gridDim.y;
    const long n = lens[b];
    const long i0 = (long)ch * AUD_CHUNK + threadIdx.x;
    float* o = out + (long)b * Lmax;
    double* mine = part + (((long)PHASE * B + b) * nch + ch) * 2;
    // Sum of the per-chunk partials of an earlier phase.  The partials were written by blocks on other XCDs, so every
    // load is an HBM / Infinity-Cache round trip: the block fetches them in parallel (one chunk per thread) and reduces,
    // instead of one thread walking them (125 serialised round trips for a 256 000-sample utterance).  Every block of
    // an utterance reduces in the same order, so all of them see bit-identical statistics.
    auto total = [&](int phase, int k) {
        const double* p = part + (((long)phase * B + b) * nch) * 2 + k;
        double t = 0.0;
        for (int c = threadIdx.x; c < nch; c += AUD_THREADS) t += p[2 * c];
        return block_sum_d(t, red);
    };
    const bool noisy = noise != nullptr && noise_start[b] >= 0;
    if (PHASE <= 1) {
        // the masking runs of this utterance, staged in LDS once per block: every sample is tested against all of them
        constexpr int IV_LDS = 64;
        __shared__ int siv[2 * IV_LDS];
        const int nm = iv ? niv[b] : 0;
        const int* ivb = iv ? iv + (long)b * max_iv * 2 : nullptr;
        if (nm <= IV_LDS) {
            for (int i = threadIdx.x; i < 2 * nm; i += AUD_THREADS) siv[i] = ivb[i];
            __syncthreads();
            ivb = siv;
        }
        const float* s = reinterpret_cast<const float*>(wav_ptr[b]);
        const float* nz = noisy ? noise + noise_start[b] : nullptr;
        float sp[AUD_PER], nv[AUD_PER];
#pragma unroll
        for (int j = 0; j < AUD_PER; j++) {  // all loads first
            const long i = i0 + (long)j * AUD_THREADS;
            sp[j] = i < n ? s[i] : 0.f;
            nv[j] = (noisy && i < n) ? nz[i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < AUD_PER; j++)
            if (masked_at(ivb, nm, i0 + (long)j * AUD_THREADS)) sp[j] = 0.f;
        if (PHASE == 0) {
            double es = 0.0, en = 0.0;
#pragma unroll
            for (int j = 0; j < AUD_PER; j++) {
                es += (double)sp[j] * sp[j];
                en += (double)nv[j] * nv[j];
            }
            es = block_sum_d(es, red);
            en = block_sum_d(en, red);
            if (threadIdx.x == 0) mine[0] = es, mine[1] = en;
            return;
        }
        float scale = 0.f;
        if (noisy) {
            // torchaudio.functional.add_noise: scale = 10^((10 (log10 E_s - log10 E_n) - snr) / 20)
            const float orig = 10.f * (log10f((float)total(0, 0)) - log10f((float)total(0, 1)));
            scale = powf(10.f, (orig - snr_db[b]) / 20.f);
        }
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < AUD_PER; j++) {
            const long i = i0 + (long)j * AUD_THREADS;
            const float y = noisy ? sp[j] + scale * nv[j] : sp[j];
            if (i < n) {
                o[i] = y;
                sum += (double)y;
            }
        }
        sum = block_sum_d(sum, red);
        if (threadIdx.x == 0) mine[0] = sum, mine[1] = 0.0;
        return;
    }
    const double mean = n > 0 ? total(1, 0) / (double)n : 0.0;
    float y[AUD_PER];
#pragma unroll
    for (int j = 0; j < AUD_PER; j++) {
        const long i = i0 + (long)j * AUD_THREADS;
        y[j] = i < n ? o[i] : 0.f;
    }
    if (PHASE == 2) {
        double sq = 0.0;
#pragma unroll
        for (int j = 0; j < AUD_PER; j++) {
            const double d = (double)y[j] - mean;
            if (i0 + (long)j * AUD_THREADS < n) sq += d * d;
        }
        sq = block_sum_d(sq, red);
        if (threadIdx.x == 0) mine[0] = sq, mine[1] = 0.0;
        return;
    }
    const double var = n > 0 ? total(2, 0) / (double)n : 0.0;  // biased, as layer_norm
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), mu = (float)mean;
#pragma unroll
    for (int j = 0; j < AUD_PER; j++) {
        const long i = i0 + (long)j * AUD_THREADS;
        if (i < Lmax) o[i] = i < n ? (y[j] - mu) * rstd : 0.f;
    }
}

}  // namespace

// src_ptr[b]: device address of clip b, decoder-layout frames uint8 [lens[b]][H][W][3].
// out: [B][Tmax][1][crop][crop] (f32 or bf16), frames t >= lens[b] zero.  crop_y / crop_x[b]: crop origin (drawn on the
// host: RandomCrop for training, (H-crop)/2 for evaluation).  iv: int32 [B][max_iv][2] masking intervals in frames,
// niv[b] of them used (NULL: no time masking).  value = ((0.2989 r + 0.587 g + 0.114 b) / 255 - mean) / std, 0 -> masked.
extern "C" int avsr_video_transform(const int64_t* src_ptr, const int32_t* lens, const int32_t* crop_y,
                                    const int32_t* crop_x, const int32_t* iv, const int32_t* niv, int max_iv, void* out,
                                    int out_dtype, int B, int Tmax, int H, int W, int crop, float mean, float std,
                                    hipStream_t stream) {
    AVSR_REQUIRE(crop > 0 && crop % 8 == 0 && crop <= H && crop <= W, "video_transform: crop must be a multiple of 8 within the frame");
    AVSR_REQUIRE(std != 0.f, "video_transform: std must be non-zero");
    AVSR_REQUIRE(iv == nullptr || (niv != nullptr && max_iv > 0), "video_transform: interval list without counts");
    if (B <= 0 || Tmax <= 0) return 0;
    dim3 grid((crop * (crop >> 3) + 255) / 256, Tmax, B), block(256);
    if (out_dtype == 0)
        AVSR_LAUNCH((video_transform_kernel<float>), grid, block, 0, stream, src_ptr, lens, crop_y, crop_x, iv, niv, max_iv,
                    (float*)out, Tmax, H, W, crop, mean, std);
    else
        AVSR_LAUNCH((video_transform_kernel<bf16_t>), grid, block, 0, stream, src_ptr, lens, crop_y, crop_x, iv, niv, max_iv,
                    (bf16_t*)out, Tmax, H, W, crop, mean, std);
    AVSR_CHECK_LAUNCH("video_transform");
    return 0;
}

// wav_ptr[b]: device address of utterance b (f32 [lens[b]]); out: f32 [B][Lmax][1], samples >= lens[b] zero.
// iv / niv / max_iv: masking intervals in samples (NULL: none).  noise (NULL: none): f32 noise recording; utterance b
// adds noise[noise_start[b] + i] scaled to snr_db[b] (noise_start[b] < 0: clean).  eps: layer_norm epsilon (1e-8).
// workspace: avsr_audio_transform_workspace_bytes(B, Lmax) bytes of device scratch (per-chunk partial sums).
extern "C" int64_t avsr_audio_transform_workspace_bytes(int B, int64_t Lmax) {
    const int64_t nch = (Lmax + AUD_CHUNK - 1) / AUD_CHUNK;
    return (int64_t)3 * B * nch * 2 * (int64_t)sizeof(double);
}

extern "C" int avsr_audio_transform(const int64_t* wav_ptr, const int32_t* lens, const int32_t* iv,
                                    const int32_t* niv, int max_iv, const float* noise, const int64_t* noise_start,
                                    const float* snr_db, float eps, float* out, int B, int64_t Lmax, void* workspace,
                                    hipStream_t stream) {
    AVSR_REQUIRE(iv == nullptr || (niv != nullptr && max_iv > 0), "audio_transform: interval list without counts");
    AVSR_REQUIRE(noise == nullptr || (noise_start != nullptr && snr_db != nullptr), "audio_transform: noise without start / SNR");
    if (B <= 0 || Lmax <= 0) return 0;
    AVSR_REQUIRE(workspace != nullptr, "audio_transform: workspace of avsr_audio_transform_workspace_bytes(B, Lmax) needed");
    const int nch = (int)((Lmax + AUD_CHUNK - 1) / AUD_CHUNK);
    double* part = reinterpret_cast<double*>(workspace);
    dim3 grid(nch, B), block(AUD_THREADS);
#define AVSR_AUD_PHASE(P)                                                                                              \
    AVSR_LAUNCH((audio_phase_kernel<P>), grid, block, 0, stream, wav_ptr, lens, iv, niv, max_iv, noise, noise_start,    \
                snr_db, eps, out, (long)Lmax, part, nch)
    AVSR_AUD_PHASE(0);
    AVSR_AUD_PHASE(1);
    AVSR_AUD_PHASE(2);
    AVSR_AUD_PHASE(3);
#undef AVSR_AUD_PHASE
    AVSR_CHECK_LAUNCH("audio_transform");
    return 0;
}
