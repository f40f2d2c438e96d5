// On-device training mixtures (replaces the CPU / Lhotse pipeline of css/datasets/separation.py:163-267).
//
// A chunk of the batch is one window [w0, w0 + n) of a two-speaker mixture built exactly like the reference builds it:
//   source1 = utterance u1 padded on the right, source2 = utterance u2 delayed by `delay` samples and padded
//   (separation.py:196-206), mix = source1 + source2 (cut1.mix(cut2, offset_other_by), :211), cut into windows (:212);
//   the augmentations act on the MIXTURE only (:235-237): a sparse 3-tap impulse response (stand-in for
//   ReverbWithImpulseResponse, p = 0.5) and additive noise at a drawn SNR (stand-in for CutMix, p = 0.5, 5-20 dB);
//   the targets are the clean padded sources.
// The utterance / noise pools are device resident; the host draws the per-chunk parameters (a few integers) and uploads
// them; everything that touches samples runs here: two passes over the window (energies for the SNR gain, then the mix),
// every sample read once per pass, 16-byte stores.  HBM bound: 3 n reads (2 sources + noise) and 3 n writes per chunk.
#include "common.hpp"

namespace {

constexpr int MX_SLICES = 32;  // energy partials per chunk

struct MixParams {  // one per chunk, 16 x 4 bytes (mirrors css_amd.ops.MIX_FIELDS)
  int u1, u2, w0, delay, len1, len2, rir_on, d1, d2, noise_on, nu, n0;
  float h0, h1, h2, snr_db;
};

DEV float src_at(const float* __restrict__ pool, int64_t L, int u, int len, int64_t t) {
  return (t >= 0 && t < len) ? pool[(int64_t)u * L + t] : 0.f;
}
DEV float clean_mix_at(const float* __restrict__ pool, int64_t L, const MixParams& p, int64_t t) {  // t on the mixture's time line
  return src_at(pool, L, p.u1, p.len1, t) + src_at(pool, L, p.u2, p.len2, t - p.delay);
}
DEV float wet_mix_at(const float* __restrict__ pool, int64_t L, const MixParams& p, int64_t t) {
  float m = clean_mix_at(pool, L, p, t);
  if (p.rir_on) m = p.h0 * m + p.h1 * clean_mix_at(pool, L, p, t - p.d1) + p.h2 * clean_mix_at(pool, L, p, t - p.d2);
  return m;
}

// pass 1: per (chunk, slice) partial energies of the (reverberated) mixture and of the noise segment
__global__ __launch_bounds__(256) void mix_energy_kernel(const float* __restrict__ pool, int64_t L,
                                                         const float* __restrict__ noise, int64_t Ln,
                                                         const MixParams* __restrict__ params, int n,
                                                         float* __restrict__ part) {
  __shared__ float red[2][4];
  const int b = blockIdx.y, sl = blockIdx.x;
  const MixParams p = params[b];
  const int per = (n + MX_SLICES - 1) / MX_SLICES;
  const int t0 = sl * per, t1 = min(n, t0 + per);
  float em = 0.f, en = 0.f;
  if (p.noise_on)
    for (int t = t0 + (int)threadIdx.x; t < t1; t += 256) {
      const float m = wet_mix_at(pool, L, p, (int64_t)p.w0 + t);
      const float z = noise[(int64_t)p.nu * Ln + p.n0 + t];
      em += m * m;
      en += z * z;
    }
  em = wave_sum(em);
  en = wave_sum(en);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = em; red[1][threadIdx.x >> 6] = en; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((size_t)b * MX_SLICES + sl) * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[((size_t)b * MX_SLICES + sl) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// pass 2: mix, source1, source2 of the window; 4 consecutive samples per thread
__global__ __launch_bounds__(256) void mix_write_kernel(const float* __restrict__ pool, int64_t L,
                                                        const float* __restrict__ noise, int64_t Ln,
                                                        const MixParams* __restrict__ params, int n,
                                                        const float* __restrict__ part, float* __restrict__ mix,
                                                        float* __restrict__ s1, float* __restrict__ s2, int64_t ld) {
  const int b = blockIdx.y;
  const MixParams p = params[b];
  float gain = 0.f;
  if (p.noise_on) {  // the chunk's energies: fixed-order sum of its partials (every thread, 64 floats from L2)
    float em = 0.f, en = 0.f;
    for (int i = 0; i < MX_SLICES; ++i) {
      em += part[((size_t)b * MX_SLICES + i) * 2];
      en += part[((size_t)b * MX_SLICES + i) * 2 + 1];
    }
    gain = en > 0.f ? sqrtf(em / (en * __powf(10.f, p.snr_db * 0.1f))) : 0.f;
  }
  const int t4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t4 >= n) return;
  float m[4], a[4], c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = t4 + j;
    const int64_t tm = (int64_t)p.w0 + t;
    a[j] = src_at(pool, L, p.u1, p.len1, tm);
    c[j] = src_at(pool, L, p.u2, p.len2, tm - p.delay);
    float v = p.rir_on ? wet_mix_at(pool, L, p, tm) : a[j] + c[j];
    if (p.noise_on && t < n) v += gain * noise[(int64_t)p.nu * Ln + p.n0 + min(t, n - 1)];
    m[j] = v;
  }
  if (t4 + 3 < n) {
    *reinterpret_cast<float4*>(mix + (int64_t)b * ld + t4) = make_float4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<float4*>(s1 + (int64_t)b * ld + t4) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<float4*>(s2 + (int64_t)b * ld + t4) = make_float4(c[0], c[1], c[2], c[3]);
  } else {
    for (int j = 0; j < 4 && t4 + j < n; ++j) {
      mix[(int64_t)b * ld + t4 + j] = m[j];
      s1[(int64_t)b * ld + t4 + j] = a[j];
      s2[(int64_t)b * ld + t4 + j] = c[j];
    }
  }
}

}  // namespace

extern "C" size_t csship_mix_workspace(int32_t B) { return (size_t)B * MX_SLICES * 2 * sizeof(float); }

extern "C" int csship_mix_batch(const float* pool, int64_t pool_len, int32_t n_utt, const float* noise_pool,
                                int64_t noise_len, int32_t n_noise, const void* params, int32_t B, int32_t n,
                                float* mix, float* source1, float* source2, int64_t ld, void* workspace,
                                size_t workspace_bytes, csship_stream_t stream) {
  if (!pool || !params || !mix || !source1 || !source2 || !workspace) return CSSHIP_ENULL;
  if (B <= 0 || n <= 0 || n_utt <= 0 || pool_len <= 0 || ld < n || B > 65535) return CSSHIP_EBADSHAPE;
  if ((noise_pool == nullptr) != (n_noise == 0) || (noise_pool && noise_len < n)) return CSSHIP_EBADSHAPE;
  if (!aligned16(mix) || !aligned16(source1) || !aligned16(source2) || (ld % 4) != 0 || !aligned16(params)) return CSSHIP_EALIGN;
  if (workspace_bytes < csship_mix_workspace(B)) return CSSHIP_EWORKSPACE;
  hipStream_t s = as_stream(stream);
  const MixParams* p = reinterpret_cast<const MixParams*>(params);
  float* part = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(mix_energy_kernel, dim3(MX_SLICES, B), dim3(256), 0, s, pool, pool_len, noise_pool, noise_len, p, n, part);
  CSSHIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(mix_write_kernel, dim3((n + 1023) / 1024, B), dim3(256), 0, s, pool, pool_len, noise_pool, noise_len, p,
                     n, part, mix, source1, source2, ld);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
