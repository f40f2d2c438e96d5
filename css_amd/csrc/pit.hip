// Fused mask application + permutation-invariant MSE (css/objectives/pit.py:5-62 looped over
// the batch as css/objectives/mse.py:31-37 does, with y_pred = mask_k * f_orig of
// css/models/conformer.py:103 formed on the fly).
//
// The reference evaluates K! permutations with K!*B tiny mse_loss launches.  Every candidate
// is a sum of entries of the K x K pairwise squared-error matrix
//     D[b][k][j] = sum_{t,f} (mask_k f_orig - target_j)^2 ,
// so one bandwidth-bound pass computes D (per-thread accumulators -> wave shuffle -> LDS ->
// one partial per workgroup, no atomics: bitwise reproducible), and a finalize kernel takes
// the minimum over permutations, the batch mean and the arg-min permutation.
#include "common.hpp"

namespace {

#ifndef PIT_RPW
#define PIT_RPW 2
#endif
#ifndef PIT_ROWS
#define PIT_ROWS 16
#endif
constexpr int RPW = PIT_RPW;       // rows (frames) a wave has in flight
constexpr int ROWS_WG = PIT_ROWS;  // rows per workgroup = one partial-sum chunk
constexpr int ROWS_WAVE = ROWS_WG / 4;
constexpr int MAXK = 4;

struct PitArgs {
  const float* masks;
  const float* f;
  const float* tgt[MAXK];
  float* y_pred;
  float* partial;  // [B][nchunk][K*K]
  int64_t ld_mask, ld_f;
  int T, F, nchunk;
};

template <int K>
DEV void pit_finalize_body(const float* __restrict__ partial, int B, int nchunk, double norm, float* __restrict__ loss,
                           int* __restrict__ perm_out, float* __restrict__ pair_sse, float* __restrict__ per_sample,
                           double* best_sum);

// 16-byte access that is only 4-byte aligned (rows are F = 257 floats).  The texture-address unit spends the same
// ~16-20 cycles on a wave's load whatever its width, so one dwordx4 per four elements instead of four dword loads
// is what takes this kernel off the instruction-rate limit of the memory pipeline.
struct __attribute__((packed, aligned(4))) f4u {
  float v[4];
};
DEV f4u load4(const float* p) { return *reinterpret_cast<const f4u*>(p); }

// One wave takes RPW rows (frames) of one sample at a time: lane l covers bins 4l .. 4l+3 of each row with one
// 16-byte load per operand (f_orig, K masks, K targets), the F mod 4 bins left at the end of the rows by one more
// instruction with ROWS_WAVE * (F mod 4) lanes active.  All loads of a pass are issued before any arithmetic.
template <int K>
__global__ __launch_bounds__(256) void pit_sse_kernel(PitArgs a) {
  __shared__ float red[4][K * K];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int F = a.F, F4 = F & ~3, T = a.T;
  const size_t TF = (size_t)T * F;
  const int tw = chunk * ROWS_WG + w * ROWS_WAVE;
  float acc[K][K];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < K; ++j) acc[k][j] = 0.f;
  for (int t0 = tw; t0 < tw + ROWS_WAVE; t0 += RPW)
  for (int f0 = 0; f0 < F4; f0 += 256) {
    const int f = f0 + 4 * lane;
    const bool fok = f < F4;
    const int fc = fok ? f : 0;
    f4u fo[RPW], mk[RPW][K], tg[RPW][K];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int tc = min(t0 + r, T - 1);
      const size_t row = (size_t)b * T + tc;
      fo[r] = load4(a.f + row * a.ld_f + fc);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        mk[r][k] = load4(a.masks + row * a.ld_mask + k * F + fc);
        tg[r][k] = load4(a.tgt[k] + (size_t)b * TF + (size_t)tc * F + fc);
      }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const bool ok = fok && t0 + r < T;
      f4u y[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int c = 0; c < 4; ++c) y[k].v[c] = mk[r][k].v[c] * fo[r].v[c];
        if (a.y_pred && ok) *reinterpret_cast<f4u*>(a.y_pred + ((size_t)b * K + k) * TF + (size_t)(t0 + r) * F + f) = y[k];
      }
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float d = y[k].v[c] - tg[r][j].v[c];
            s += d * d;
          }
          acc[k][j] += ok ? s : 0.f;
        }
    }
  }
  const int nt = F - F4;  // 0..3 bins at the end of every row
  if (nt && lane < ROWS_WAVE * nt) {
    const int r = lane / nt, f = F4 + lane - r * nt, t = tw + r;
    if (t < T) {
      const size_t row = (size_t)b * T + t;
      const float fo = a.f[row * a.ld_f + f];
      float y[K], tg[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        y[k] = a.masks[row * a.ld_mask + k * F + f] * fo;
        tg[k] = a.tgt[k][(size_t)b * TF + (size_t)t * F + f];
        if (a.y_pred) a.y_pred[((size_t)b * K + k) * TF + (size_t)t * F + f] = y[k];
      }
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float d = y[k] - tg[j];
          acc[k][j] += d * d;
        }
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float s = wave_sum(acc[k][j]);
      if (lane == 0) red[w][k * K + j] = s;
    }
  __syncthreads();
  if (tid < K * K)
    a.partial[((size_t)b * a.nchunk + chunk) * (K * K) + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// double-precision lane reductions on the data-parallel primitives (a double __shfl_xor is two ds_bpermute round trips of
// ~120 cycles in a dependent chain; this tiny kernel is nothing but such chains)
template <int CTRL>
DEV double dpp_mov_d(double v) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
DEV double sum8_d(double v) {  // over aligned groups of 8 lanes, result in all of them
  v += dpp_mov_d<DPP_XOR1>(v);
  v += dpp_mov_d<DPP_XOR2>(v);
  v += dpp_mov_d<DPP_HALF_MIRROR>(v);
  return v;
}
DEV double swap_sum_d(double v, bool row32) {  // v + the value of the lane 16 (or 32) away
  const unsigned long long u = __double_as_longlong(v);
  unsigned p[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const unsigned x = (unsigned)(u >> (32 * h));
    if (row32) {
      auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
      const unsigned a0 = r[0], a1 = r[1];
      p[h] = a0 ^ a1 ^ x;   // {a0, a1} = {own word, partner's word} in a lane-dependent order: xor with x leaves the partner's
    } else {
      auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
      const unsigned a0 = r[0], a1 = r[1];
      p[h] = a0 ^ a1 ^ x;
    }
  }
  return v + __longlong_as_double(((unsigned long long)p[1] << 32) | p[0]);
}
DEV double wave_sum_d(double v) {
  v = sum8_d(v);
  v += dpp_mov_d<DPP_MIRROR>(v);
  v = swap_sum_d(v, false);
  return swap_sum_d(v, true);
}

// Eight lanes per sample reduce its partials (double, fixed order), one of them scans the K! permutations in
// itertools.permutations order (first minimum wins, like torch.min), then thread 0 averages.  Single workgroup.
template <int K>
DEV void pit_finalize_body(const float* __restrict__ partial, int B, int nchunk, double norm, float* __restrict__ loss,
                           int* __restrict__ perm_out, float* __restrict__ pair_sse, float* __restrict__ per_sample,
                           double* best_sum) {
  double acc_best = 0.0;  // sum of this thread's samples' minima (threads with part == 0 only)
  for (int b0 = 0; b0 < B; b0 += 32) {
  const int b = b0 + (threadIdx.x >> 3), part = threadIdx.x & 7;
  double Dm[K][K];
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < K; ++j) Dm[k][j] = 0.0;
  // four chunks per lane are loaded before any is added; the additions keep the ascending-chunk order
  if (b < B)
    for (int c0 = part; c0 < nchunk; c0 += 32) {
      float v[4][K * K];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = min(c0 + 8 * u, nchunk - 1);
#pragma unroll
        for (int i = 0; i < K * K; ++i)
          v[u][i] = partial[((size_t)b * nchunk + c) * (K * K) + i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c0 + 8 * u < nchunk)
#pragma unroll
          for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < K; ++j) Dm[k][j] += (double)v[u][k * K + j];
    }
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < K; ++j) {
      Dm[k][j] = sum8_d(Dm[k][j]);
    }
  if (b < B && part == 0) {
    if (pair_sse)
      for (int k = 0; k < K; ++k)
        for (int j = 0; j < K; ++j) pair_sse[(size_t)b * K * K + k * K + j] = (float)Dm[k][j];
    // lexicographic permutations of 0..K-1
    int p[K], best_p[K];
    for (int i = 0; i < K; ++i) p[i] = i;
    float best = 0.f;
    bool first = true;
    while (true) {
      double s = 0.0;
      for (int i = 0; i < K; ++i) s += Dm[p[i]][i];
      const float cand = (float)(s / norm);  // candidates are float32 in the reference
      if (first || cand < best) {
        best = cand;
        for (int i = 0; i < K; ++i) best_p[i] = p[i];
        first = false;
      }
      int i = K - 2;
      while (i >= 0 && p[i] > p[i + 1]) --i;
      if (i < 0) break;
      int j = K - 1;
      while (p[j] < p[i]) --j;
      int tmp = p[i]; p[i] = p[j]; p[j] = tmp;
      for (int l = i + 1, r = K - 1; l < r; ++l, --r) { tmp = p[l]; p[l] = p[r]; p[r] = tmp; }
    }
    per_sample[b] = best;
    acc_best += (double)best;
    if (perm_out)
      for (int i = 0; i < K; ++i) perm_out[(size_t)b * K + i] = best_p[i];
  }
  }
  // batch mean: fixed-order tree (lanes, then waves) instead of one thread re-reading per_sample from memory
  acc_best = wave_sum_d(acc_best);
  if ((threadIdx.x & 63) == 0) best_sum[threadIdx.x >> 6] = acc_best;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (float)(((best_sum[0] + best_sum[1]) + (best_sum[2] + best_sum[3])) / (double)B);
}

template <int K>
__global__ __launch_bounds__(256) void pit_finalize_kernel(const float* __restrict__ partial, int B, int nchunk,
                                                           double norm, float* __restrict__ loss,
                                                           int* __restrict__ perm_out, float* __restrict__ pair_sse,
                                                           float* __restrict__ per_sample) {
  __shared__ double best_sum[4];
  pit_finalize_body<K>(partial, B, nchunk, norm, loss, perm_out, pair_sse, per_sample, best_sum);
}

template <int K>
__global__ __launch_bounds__(256) void pit_bwd_kernel(const float* __restrict__ masks, int64_t ld_mask,
                                                      const float* __restrict__ f, int64_t ld_f, PitArgs a,
                                                      const int* __restrict__ perm, int n_masks, float coef,
                                                      const float* __restrict__ gl_dev, float* __restrict__ grad) {
  const int b = blockIdx.y;
  const float gl = gl_dev ? *gl_dev : 1.f;   // the upstream gradient of the loss, read on the device (x * 1.f is exact)
  const int TF = a.T * a.F;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= TF) return;
  const int t = e / a.F, ff = e - t * a.F;
  const size_t row = (size_t)b * a.T + t;
  const float fo = f[row * ld_f + ff];
  // perm[b][i] = estimate matched to target i  ->  estimate k is compared with target inv[k]
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int k = perm[(size_t)b * K + i];
    const float y = masks[row * ld_mask + k * a.F + ff] * fo;
    grad[row * ld_mask + k * a.F + ff] = coef * (y - a.tgt[i][(size_t)b * TF + e]) * fo * gl;
  }
  for (int k = K; k < n_masks; ++k) grad[row * ld_mask + k * a.F + ff] = 0.f;
}

__global__ __launch_bounds__(256) void apply_masks_kernel(const float* __restrict__ masks, int64_t ld_mask,
                                                          const float* __restrict__ f, int64_t ld_f, int T, int F,
                                                          int K, float* __restrict__ y) {
  const int b = blockIdx.y;
  const int TF = T * F;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= TF) return;
  const int t = e / F, ff = e - t * F;
  const size_t row = (size_t)b * T + t;
  const float fo = f[row * ld_f + ff];
  for (int k = 0; k < K; ++k) y[((size_t)b * K + k) * TF + e] = masks[row * ld_mask + k * F + ff] * fo;
}

}  // namespace

extern "C" size_t csship_pit_workspace(int32_t B, int32_t T, int32_t F, int32_t K) {
  const size_t nchunk = ((size_t)T + ROWS_WG - 1) / ROWS_WG;
  return ((size_t)B * nchunk * K * K + (size_t)B) * sizeof(float);
}

extern "C" int csship_pit_mse(const float* masks, int64_t ld_mask, const float* f_orig, int64_t ld_f,
                              const float* const* targets_host, int32_t B, int32_t T, int32_t F, int32_t K,
                              float* y_pred, float* loss, int32_t* perm, float* pair_sse, void* workspace,
                              csship_stream_t stream) {
  if (!masks || !f_orig || !targets_host || !loss || !workspace) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || F <= 0 || B > 65535 || (int64_t)T * F > 0x7fffffff) return CSSHIP_EBADSHAPE;
  if (K < 1 || K > MAXK) return CSSHIP_EUNSUPPORTED;
  if (ld_mask < (int64_t)K * F || ld_f < F) return CSSHIP_EBADSHAPE;
  PitArgs a{};
  a.masks = masks; a.f = f_orig; a.y_pred = y_pred; a.ld_mask = ld_mask; a.ld_f = ld_f; a.T = T; a.F = F;
  for (int k = 0; k < K; ++k) {
    if (!targets_host[k]) return CSSHIP_ENULL;
    a.tgt[k] = targets_host[k];
  }
  a.nchunk = (T + ROWS_WG - 1) / ROWS_WG;
  a.partial = reinterpret_cast<float*>(workspace);
  float* per_sample = a.partial + (size_t)B * a.nchunk * K * K;
  const double norm = (double)K * T * F;
  hipStream_t s = as_stream(stream);
  dim3 grid(a.nchunk, B);
  const int fb = 256;  // finalize: one block (32 samples per pass) so the batch mean can follow a block barrier
#define PIT_CASE(KK)                                                                                          \
  case KK:                                                                                                    \
    {                                                                                                         \
      TimelineScope tl(CSSHIP_TL_PIT_SSE, s);                                                                 \
      hipLaunchKernelGGL(pit_sse_kernel<KK>, grid, dim3(256), 0, s, a);                                       \
    }                                                                                                         \
    CSSHIP_CHECK_LAUNCH();                                                                                    \
    {                                                                                                         \
      TimelineScope tl(CSSHIP_TL_PIT_FINALIZE, s);                                                            \
      hipLaunchKernelGGL(pit_finalize_kernel<KK>, dim3(1), dim3(fb), 0, s, a.partial, B, a.nchunk, norm, loss, \
                         perm, pair_sse, per_sample);                                                         \
    }                                                                                                         \
    break;
  switch (K) {
    PIT_CASE(1) PIT_CASE(2) PIT_CASE(3) PIT_CASE(4)
    default: return CSSHIP_EUNSUPPORTED;
  }
#undef PIT_CASE
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_pit_mse_bwd_dev(const float* masks, int64_t ld_mask, const float* f_orig, int64_t ld_f,
                                      const float* const* targets_host, const int32_t* perm, int32_t B, int32_t T,
                                      int32_t F, int32_t K, int32_t n_masks, float grad_loss,
                                      const float* grad_loss_dev, float* grad_masks, csship_stream_t stream);

extern "C" int csship_pit_mse_bwd(const float* masks, int64_t ld_mask, const float* f_orig, int64_t ld_f,
                                  const float* const* targets_host, const int32_t* perm, int32_t B, int32_t T,
                                  int32_t F, int32_t K, int32_t n_masks, float grad_loss, float* grad_masks,
                                  csship_stream_t stream) {
  return csship_pit_mse_bwd_dev(masks, ld_mask, f_orig, ld_f, targets_host, perm, B, T, F, K, n_masks, grad_loss, nullptr,
                                grad_masks, stream);
}

extern "C" int csship_pit_mse_bwd_dev(const float* masks, int64_t ld_mask, const float* f_orig, int64_t ld_f,
                                      const float* const* targets_host, const int32_t* perm, int32_t B, int32_t T,
                                      int32_t F, int32_t K, int32_t n_masks, float grad_loss,
                                      const float* grad_loss_dev, float* grad_masks, csship_stream_t stream) {
  if (!masks || !f_orig || !targets_host || !perm || !grad_masks) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || F <= 0 || B > 65535 || (int64_t)T * F > 0x7fffffff || n_masks < K) return CSSHIP_EBADSHAPE;
  if (K < 1 || K > MAXK) return CSSHIP_EUNSUPPORTED;
  if (ld_mask < (int64_t)n_masks * F || ld_f < F) return CSSHIP_EBADSHAPE;
  PitArgs a{};
  a.T = T; a.F = F;
  for (int k = 0; k < K; ++k) {
    if (!targets_host[k]) return CSSHIP_ENULL;
    a.tgt[k] = targets_host[k];
  }
  const float coef = 2.f * grad_loss / ((float)B * (float)K * (float)T * (float)F);
  dim3 grid((unsigned)(((size_t)T * F + 255) / 256), B);
  hipStream_t s = as_stream(stream);
  switch (K) {
    case 1: hipLaunchKernelGGL(pit_bwd_kernel<1>, grid, dim3(256), 0, s, masks, ld_mask, f_orig, ld_f, a, perm, n_masks, coef, grad_loss_dev, grad_masks); break;
    case 2: hipLaunchKernelGGL(pit_bwd_kernel<2>, grid, dim3(256), 0, s, masks, ld_mask, f_orig, ld_f, a, perm, n_masks, coef, grad_loss_dev, grad_masks); break;
    case 3: hipLaunchKernelGGL(pit_bwd_kernel<3>, grid, dim3(256), 0, s, masks, ld_mask, f_orig, ld_f, a, perm, n_masks, coef, grad_loss_dev, grad_masks); break;
    case 4: hipLaunchKernelGGL(pit_bwd_kernel<4>, grid, dim3(256), 0, s, masks, ld_mask, f_orig, ld_f, a, perm, n_masks, coef, grad_loss_dev, grad_masks); break;
    default: return CSSHIP_EUNSUPPORTED;
  }
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_apply_masks(const float* masks, int64_t ld_mask, const float* f_orig, int64_t ld_f, int32_t B,
                                  int32_t T, int32_t F, int32_t K, float* y_pred, csship_stream_t stream) {
  if (!masks || !f_orig || !y_pred) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || F <= 0 || K <= 0 || B > 65535 || (int64_t)T * F > 0x7fffffff) return CSSHIP_EBADSHAPE;
  if (ld_mask < (int64_t)K * F || ld_f < F) return CSSHIP_EBADSHAPE;
  dim3 grid((unsigned)(((size_t)T * F + 255) / 256), B);
  hipLaunchKernelGGL(apply_masks_kernel, grid, dim3(256), 0, as_stream(stream), masks, ld_mask, f_orig, ld_f, T, F, K,
                     y_pred);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
