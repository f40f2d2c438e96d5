// STFT front-end (css/executor/feature.py:13-82): 512-point Hann-windowed real DFT, hop 256,
// no centre padding.  The reference evaluates it as a dense conv1d against a 514x512 DFT
// matrix (0.53 MFLOP/frame); here each frame is a real FFT computed by ONE 64-lane wave:
//
//   * a wave walks fw consecutive frames of one signal; consecutive frames overlap by 50 %, so the wave keeps the
//     second half of the previous frame in registers and loads only the 256 new samples (two 8-byte loads per
//     lane, issued one frame ahead): every sample is read from HBM once, with no LDS staging and no barrier.
//     fw is picked per launch so that the whole grid is resident in ONE round (per-wave time stamps showed the
//     second round starting only when first-round waves retire, on a half-empty chip);
//   * the SIMD arbiter serves the oldest ready wave first, so with equal priorities the four waves of a SIMD finish
//     one after the other (11 .. 17.5 us for equal work); the priority is rotated frame by frame (s_setprio);
//   * complex values are float 2-vectors: butterflies / twiddles compile to v_pk_add_f32 / v_pk_fma_f32
//     (179 -> 125 VALU instructions per frame);
//   * 512 real points are packed as 256 complex z[n] = x[2n] + i x[2n+1]; lane l holds
//     z[l], z[l+64], z[l+128], z[l+192]: four radix-4 stages in registers, the register <-> lane
//     regrouping between them by v_permlane{32,16}_swap and a 2 KB per-wave LDS scratch (fft.hpp);
//   * the digit-reversed result goes through the same scratch once more, for the
//     Z[k] / conj(Z[256-k]) pairing of the real-FFT post-pass and natural-order output;
//   * magnitude / re / im / phase are written frame-major ([S,T,257]), 64 consecutive bins
//     per store instruction.
// HBM-bound: 1024 B read + 1028 B written per frame for the magnitude path.
#include <cstdlib>

#include "fft.hpp"

namespace {

#ifdef STFT_STAMPS
__device__ unsigned long long g_stft_stamps[3 * 8192];
__device__ unsigned g_stft_epoch;
#endif

#ifndef STFT_MINW
#define STFT_MINW 1
#endif
#ifndef STFT_RES
#define STFT_RES 4  // resident workgroups per CU at this kernel's register count
#endif
constexpr int FW_MAX = 32;   // a wave walks fw <= FW_MAX consecutive frames (runtime: chosen so the grid is one resident round)
constexpr int NBIN = 257;
#ifndef STFT_NC
#define STFT_NC 1
#endif
constexpr int NC = STFT_NC;  // frames in flight per wave (independent chains)

// LOGPOW (with MAG_ONLY): the `mag` output holds log(max(|X|^2, eps)) instead of |X| - the log-power feature variant
// (north_star "log-power / IPD features"; SURVEY 8c: the build-defined stand-in for Lhotse's unpinned training
// features), as an epilogue flag of the same kernel.
template <bool MAG_ONLY, bool LOGPOW = false>
__global__ __launch_bounds__(256, STFT_MINW) void stft_kernel(const float* __restrict__ audio, int64_t n1, int64_t stride0,
                                                   int64_t stride1, int64_t n_samples, int T, int fw,
                                                   float* __restrict__ mag, float* __restrict__ re,
                                                   float* __restrict__ im, float* __restrict__ pha, int64_t ld, float eps = 0.f) {
  __shared__ __attribute__((aligned(16))) cplx sz[4][NC * FFT_SCRATCH];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int64_t s = blockIdx.y;
  const int tw0 = (blockIdx.x * 4 + w) * fw;  // this wave's first frame
  if (tw0 >= T) return;                       // wave-uniform
#ifdef STFT_STAMPS
  const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int nf = min(fw, T - tw0);
  const float* sig = audio + (s / n1) * stride0 + (s % n1) * stride1 + (int64_t)tw0 * 256;

  // ---- per-lane constants (window, twiddles).  The window carries the 1/2 of the real-FFT post pass (exact: a power
  // of two), the post twiddle is pre-multiplied by -i: X[k] = S + (-i W512^k) D, S = Z[k] + conj Z[256-k],
  // D = Z[k] - conj Z[256-k]
  float win[8];
  cplx tw1[4], tws[6], twp[4], wh[4];
  fft_load_consts(0, l, win, tw1, tws, twp);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    wh[q] = {0.5f * win[2 * q], 0.5f * win[2 * q + 1]};
    twp[q] = {twp[q].y, -twp[q].x};
  }
  // natural-order scratch image padded by one entry per 16 (slot(k) = k + k/16): the digit-reversed writes of a
  // 16-lane group then touch 16 distinct 8-byte slots of the 128-byte bank row instead of 4
  const int ko = fft_out_index(l), kos = ko + (ko >> 4);

  // Two runs of frames advance together (chain 0: frames tw0 .., chain 1: the second half of the wave's frames):
  // raw[c][q] = samples 2(l+64q), 2(l+64q)+1 of chain c's current frame; q = 2, 3 become q = 0, 1 of its next frame
  const int nA = NC == 1 ? nf : (nf + 1) >> 1, nB = nf - nA;  // chain lengths (nB = nA or nA - 1)
  const float* sigc[2] = {sig, sig + (int64_t)nA * 256};
  cplx raw[NC][4], nxt[NC][2];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q)  // chain 1 of a one-frame wave re-reads chain 0's frame (never stored)
      raw[c][q] = *reinterpret_cast<const cplx*>((c == 1 && nB == 0 ? sig : sigc[c]) + 2 * (l + 64 * q));
  cplx* zs = sz[w];
#ifndef STFT_NO_PRIO
  // the SIMD arbiter serves the oldest ready wave first: with equal priorities the four waves of a SIMD finish one
  // after the other (11 .. 17.5 us) and the last one runs alone.  Rotating the priority frame by frame, offset by
  // the dispatch quartile the workgroup belongs to, shares the SIMD evenly.
  const int phase = (int)((4u * (blockIdx.y * gridDim.x + blockIdx.x)) / (gridDim.x * gridDim.y));
#endif
  for (int fi = 0; fi < nA; ++fi) {
#ifndef STFT_NO_PRIO
    switch ((fi + phase) & 3) {
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
#endif
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (fi + 1 < (c == 0 ? nA : nB)) {  // the 256 new samples of the chain's next frame (inside the signal)
        nxt[c][0] = *reinterpret_cast<const cplx*>(sigc[c] + (int64_t)(fi + 1) * 256 + 256 + 2 * l);
        nxt[c][1] = *reinterpret_cast<const cplx*>(sigc[c] + (int64_t)(fi + 1) * 256 + 256 + 2 * (l + 64));
      }
    cplx u[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) u[c][q] = raw[c][q] * wh[q];
    fft256_wave_n<NC>(u, tw1, tws, zs, l);
    // lane l, reg r holds Z[ko + 64 r]
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) zs[c * FFT_SCRATCH + kos + 68 * r] = u[c][r];
    fft_wave_sync();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c == 1 && fi >= nB) continue;  // wave-uniform
      const int t = tw0 + (c == 0 ? 0 : nA) + fi;
      const size_t orow = ((size_t)s * T + t) * (size_t)ld;
      const cplx* zc = zs + c * FFT_SCRATCH;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = l + 64 * q;
        const int kc = (256 - k) & 255;
        const cplx A = zc[k + (k >> 4)], Bv = zc[kc + (kc >> 4)];
        const cplx Bc = {Bv.x, -Bv.y};
        const cplx X = cmul(A - Bc, twp[q]) + (A + Bc);
        if constexpr (MAG_ONLY && LOGPOW) {
          mag[orow + k] = __logf(fmaxf(X.x * X.x + X.y * X.y, eps));
        } else if constexpr (MAG_ONLY) {
          mag[orow + k] = __builtin_amdgcn_sqrtf(X.x * X.x + X.y * X.y);  // v_sqrt_f32, 1 ulp
        } else {
          if (mag) mag[orow + k] = __builtin_amdgcn_sqrtf(X.x * X.x + X.y * X.y);
          if (re) re[orow + k] = X.x;
          if (im) im[orow + k] = X.y;
          if (pha) pha[orow + k] = atan2f(X.y, X.x);
        }
      }
      if (l == 0) {  // Nyquist bin k = 256: X = Re Z[0] - Im Z[0] of the unscaled Z (ours carries the 1/2)
        const cplx A = zc[0];
        const float xr = 2.f * (A.x - A.y);
        if constexpr (MAG_ONLY && LOGPOW) mag[orow + 256] = __logf(fmaxf(xr * xr, eps));
        else if (MAG_ONLY || mag) mag[orow + 256] = fabsf(xr);
        if constexpr (!MAG_ONLY) {
          if (re) re[orow + 256] = xr;
          if (im) im[orow + 256] = 0.f;
          if (pha) pha[orow + 256] = atan2f(0.f, xr);
        }
      }
    }
    fft_wave_sync();  // scratch is rewritten by the next frames
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      raw[c][0] = raw[c][2];
      raw[c][1] = raw[c][3];
      raw[c][2] = nxt[c][0];
      raw[c][3] = nxt[c][1];
    }
  }
#ifdef STFT_STAMPS
  if (l == 0) {
    const int wid = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + w;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_stft_stamps[3 * wid] = st0;
    g_stft_stamps[3 * wid + 1] = __builtin_amdgcn_s_memrealtime();
    g_stft_stamps[3 * wid + 2] = ((unsigned long long)g_stft_epoch << 48) | ((unsigned long long)(xcc & 0xff) << 32) | hw;
  }
#endif
}

}  // namespace

#ifdef STFT_STAMPS
extern "C" int csship_debug_stft_epoch(unsigned e) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stft_epoch), &e, sizeof(e));
}
extern "C" int csship_debug_stft_stamps(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stft_stamps), sizeof(unsigned long long) * 3 * 8192);
}
#endif

static int stft_launch(const float* audio, int64_t n0, int64_t stride0, int64_t n1, int64_t stride1, int64_t n_samples,
                       float* mag, float* re, float* im, float* pha, bool logpow, float eps, csship_stream_t stream,
                       int64_t ld = NBIN);

extern "C" int csship_stft(const float* audio, int64_t n0, int64_t stride0, int64_t n1, int64_t stride1,
                           int64_t n_samples, float* mag, float* re, float* im, float* pha,
                           csship_stream_t stream) {
  return stft_launch(audio, n0, stride0, n1, stride1, n_samples, mag, re, im, pha, false, 0.f, stream);
}

extern "C" int csship_stft_ld(const float* audio, int64_t n0, int64_t stride0, int64_t n1, int64_t stride1,
                              int64_t n_samples, float* mag, float* re, float* im, float* pha, int64_t ld_out,
                              csship_stream_t stream) {
  if (ld_out < NBIN) return CSSHIP_EBADSHAPE;
  return stft_launch(audio, n0, stride0, n1, stride1, n_samples, mag, re, im, pha, false, 0.f, stream, ld_out);
}

extern "C" int csship_stft_logpower(const float* audio, int64_t n0, int64_t stride0, int64_t n1, int64_t stride1,
                                    int64_t n_samples, float eps, float* logpow, csship_stream_t stream) {
  if (!(eps > 0.f)) return CSSHIP_EBADSHAPE;
  return stft_launch(audio, n0, stride0, n1, stride1, n_samples, logpow, nullptr, nullptr, nullptr, true, eps, stream);
}

static int stft_launch(const float* audio, int64_t n0, int64_t stride0, int64_t n1, int64_t stride1, int64_t n_samples,
                       float* mag, float* re, float* im, float* pha, bool logpow, float eps, csship_stream_t stream,
                       int64_t ld) {
  if (!audio || (!mag && !re && !im && !pha)) return CSSHIP_ENULL;
  if (n0 <= 0 || n1 <= 0 || n_samples < 512) return CSSHIP_EBADSHAPE;
  if (!aligned16(audio) || (stride0 % 4) != 0 || (stride1 % 4) != 0) return CSSHIP_EALIGN;
  const int64_t T = (n_samples - 512) / 256 + 1;
  const int64_t S = n0 * n1;
  if (S > 65535 || T > 0x7fffffff / 256) return CSSHIP_EBADSHAPE;
  // frames per wave: the smallest run length whose grid is resident in ONE round (4 workgroups per CU at this
  // kernel's register count) - a second round starts only when first-round waves retire and leaves the chip
  // half empty; 8 when even the longest run does not fit
  static int cus = 0, fw_env = -1;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    const char* e = getenv("CSSHIP_STFT_FW");  // tuning override
    fw_env = e ? atoi(e) : 0;
  }
  int fw = 8;
  if (fw_env >= 1 && fw_env <= FW_MAX) {
    fw = fw_env;
  } else {
    const int64_t cap = STFT_RES * (int64_t)cus;
    for (int f = 4; f <= FW_MAX; ++f)
      if (S * (((T + f - 1) / f + 3) / 4) <= cap) {
        fw = f;
        break;
      }
  }
  const int64_t runs = (T + fw - 1) / fw;
  dim3 grid((unsigned)((runs + 3) / 4), (unsigned)S);
  TimelineScope tl(CSSHIP_TL_STFT, as_stream(stream));
  if (logpow)
    hipLaunchKernelGGL((stft_kernel<true, true>), grid, dim3(256), 0, as_stream(stream), audio, n1, stride0, stride1,
                       n_samples, (int)T, fw, mag, re, im, pha, ld, eps);
  else if (mag && !re && !im && !pha)
    hipLaunchKernelGGL(stft_kernel<true>, grid, dim3(256), 0, as_stream(stream), audio, n1, stride0, stride1, n_samples,
                       (int)T, fw, mag, re, im, pha, ld);
  else
    hipLaunchKernelGGL(stft_kernel<false>, grid, dim3(256), 0, as_stream(stream), audio, n1, stride0, stride1,
                       n_samples, (int)T, fw, mag, re, im, pha, ld);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
