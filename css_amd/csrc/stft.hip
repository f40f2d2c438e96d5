// STFT front-end (css/executor/feature.py:13-82): 512-point Hann-windowed real DFT, hop 256,
// no centre padding.  The reference evaluates it as a dense conv1d against a 514x512 DFT
// matrix (0.53 MFLOP/frame); here each frame is a real FFT computed by ONE 64-lane wave:
//
//   * a wave walks FW consecutive frames of one signal; consecutive frames overlap by 50 %, so the wave keeps the
//     second half of the previous frame in registers and loads only the 256 new samples (two 8-byte loads per
//     lane, issued one frame ahead): every sample is read from HBM once, with no LDS staging and no barrier;
//   * 512 real points are packed as 256 complex z[n] = x[2n] + i x[2n+1]; lane l holds
//     z[l], z[l+64], z[l+128], z[l+192]: four radix-4 stages in registers, the register <-> lane
//     regrouping between them by v_permlane{32,16}_swap and a 2 KB per-wave LDS scratch (fft.hpp);
//   * the digit-reversed result goes through the same scratch once more, for the
//     Z[k] / conj(Z[256-k]) pairing of the real-FFT post-pass and natural-order output;
//   * magnitude / re / im / phase are written frame-major ([S,T,257]), 64 consecutive bins
//     per store instruction.
// HBM-bound: 1024 B read + 1028 B written per frame for the magnitude path.
#include "fft.hpp"

namespace {

constexpr int FW = 8;        // consecutive frames per wave
constexpr int FB = 4 * FW;   // frames per workgroup (4 waves): 8.5 KB LDS (FFT scratch only)
constexpr int NBIN = 257;

template <bool MAG_ONLY>
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ audio, int64_t n1, int64_t stride0,
                                                   int64_t stride1, int64_t n_samples, int T,
                                                   float* __restrict__ mag, float* __restrict__ re,
                                                   float* __restrict__ im, float* __restrict__ pha) {
  __shared__ __attribute__((aligned(16))) cplx sz[4][FFT_SCRATCH];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int64_t s = blockIdx.y;
  const int tw0 = blockIdx.x * FB + w * FW;  // this wave's first frame
  if (tw0 >= T) return;                      // wave-uniform
  const int nf = min(FW, T - tw0);
  const float* sig = audio + (s / n1) * stride0 + (s % n1) * stride1 + (int64_t)tw0 * 256;

  // ---- per-lane constants (window, twiddles)
  float win[8];
  cplx tw1[4], tws[6], twp[4];
  fft_load_consts(0, l, win, tw1, tws, twp);
  // natural-order scratch image padded by one entry per 16 (slot(k) = k + k/16): the digit-reversed writes of a
  // 16-lane group then touch 16 distinct 8-byte slots of the 128-byte bank row instead of 4
  const int ko = fft_out_index(l), kos = ko + (ko >> 4);

  // raw[q] = samples 2(l+64q), 2(l+64q)+1 of the current frame; q = 2, 3 become q = 0, 1 of the next frame
  float2 raw[4], nxt[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) raw[q] = *reinterpret_cast<const float2*>(sig + 2 * (l + 64 * q));
  cplx* zs = sz[w];
  for (int fi = 0; fi < nf; ++fi) {
    const int t = tw0 + fi;
    if (fi + 1 < nf) {  // the 256 new samples of frame t + 1 (all inside the signal for a valid frame)
      nxt[0] = *reinterpret_cast<const float2*>(sig + (int64_t)(fi + 1) * 256 + 256 + 2 * l);
      nxt[1] = *reinterpret_cast<const float2*>(sig + (int64_t)(fi + 1) * 256 + 256 + 2 * (l + 64));
    }
    cplx z[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) z[q] = {raw[q].x * win[2 * q], raw[q].y * win[2 * q + 1]};
    cplx u[4];
    fft256_wave(z, u, tw1, tws, zs, l);
    // lane l, reg r holds Z[ko + 64 r]
#pragma unroll
    for (int r = 0; r < 4; ++r) zs[kos + 68 * r] = u[r];
    fft_wave_sync();
    // real-FFT post pass: X[k] = E + W512^k O, E = (Z[k]+conj Z[256-k])/2, O = -i (Z[k]-conj Z[256-k])/2
    const size_t orow = ((size_t)s * T + t) * NBIN;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = l + 64 * q;
      const int kc = (256 - k) & 255;
      const cplx A = zs[k + (k >> 4)], Bv = zs[kc + (kc >> 4)];
      const cplx E = {0.5f * (A.x + Bv.x), 0.5f * (A.y - Bv.y)};
      const cplx Dd = {A.x - Bv.x, A.y + Bv.y};
      const cplx O = {0.5f * Dd.y, -0.5f * Dd.x};
      const cplx WO = cmul(twp[q], O);
      const float xr = E.x + WO.x, xi = E.y + WO.y;
      if constexpr (MAG_ONLY) {
        mag[orow + k] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi);  // v_sqrt_f32, 1 ulp
      } else {
        if (mag) mag[orow + k] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi);
        if (re) re[orow + k] = xr;
        if (im) im[orow + k] = xi;
        if (pha) pha[orow + k] = atan2f(xi, xr);
      }
    }
    if (l == 0) {  // Nyquist bin k = 256: X = Re Z[0] - Im Z[0]
      const cplx A = zs[0];
      const float xr = A.x - A.y;
      if (MAG_ONLY || mag) mag[orow + 256] = fabsf(xr);
      if constexpr (!MAG_ONLY) {
        if (re) re[orow + 256] = xr;
        if (im) im[orow + 256] = 0.f;
        if (pha) pha[orow + 256] = atan2f(0.f, xr);
      }
    }
    fft_wave_sync();  // scratch is rewritten by the next frame
    raw[0] = raw[2];
    raw[1] = raw[3];
    raw[2] = nxt[0];
    raw[3] = nxt[1];
  }
}

}  // namespace

extern "C" int csship_stft(const float* audio, int64_t n0, int64_t stride0, int64_t n1, int64_t stride1,
                           int64_t n_samples, float* mag, float* re, float* im, float* pha,
                           csship_stream_t stream) {
  if (!audio || (!mag && !re && !im && !pha)) return CSSHIP_ENULL;
  if (n0 <= 0 || n1 <= 0 || n_samples < 512) return CSSHIP_EBADSHAPE;
  if (!aligned16(audio) || (stride0 % 4) != 0 || (stride1 % 4) != 0) return CSSHIP_EALIGN;
  const int64_t T = (n_samples - 512) / 256 + 1;
  const int64_t S = n0 * n1;
  if (S > 65535 || T > 0x7fffffff / 256) return CSSHIP_EBADSHAPE;
  dim3 grid((unsigned)((T + FB - 1) / FB), (unsigned)S);
  if (mag && !re && !im && !pha)
    hipLaunchKernelGGL(stft_kernel<true>, grid, dim3(256), 0, as_stream(stream), audio, n1, stride0, stride1, n_samples,
                       (int)T, mag, re, im, pha);
  else
    hipLaunchKernelGGL(stft_kernel<false>, grid, dim3(256), 0, as_stream(stream), audio, n1, stride0, stride1,
                       n_samples, (int)T, mag, re, im, pha);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
