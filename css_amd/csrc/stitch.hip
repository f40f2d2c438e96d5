// Mask stitching for long recordings (css/executor/stitcher.py:14-110).
//   similarity: for adjacent windows n, n+1 the 2x2 matrix  S[i][j] = -sum_{f, tau<margin}
//               | m_j(prev) |x|(prev)[f, T-margin+tau] - m_i(now) |x|(now)[f, tau] |^0.5      (stitcher.py:24-40)
//   connect   : speaker order per window from the running swap parity, winner-take-all (non-maximal
//               entries -> floor), overlap-add average over the windows covering each frame (:62-110).
// Both are small bandwidth-bound passes; sums are accumulated in double so the keep/swap decision does not
// depend on summation order.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void stitch_similarity_kernel(const float* __restrict__ mags,
                                                                const float* __restrict__ masks, int F, int T,
                                                                int margin, float* __restrict__ sim) {
  __shared__ double red[4][4];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* xp = mags + (size_t)n * F * T;
  const float* xn = xp + (size_t)F * T;
  const float* mp = masks + (size_t)n * F * T * 3;
  const float* mn = mp + (size_t)F * T * 3;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int e = tid; e < F * margin; e += 256) {
    const int f = e / margin, tau = e - f * margin;
    const size_t ip = (size_t)f * T + (T - margin + tau), in = (size_t)f * T + tau;
    const float ap = fabsf(xp[ip]), an = fabsf(xn[in]);
    const float ep0 = mp[ip * 3 + 0] * ap, ep1 = mp[ip * 3 + 1] * ap;
    const float en0 = mn[in * 3 + 0] * an, en1 = mn[in * 3 + 1] * an;
    acc[0] += (double)sqrtf(fabsf(ep0 - en0));  // i = 0, j = 0
    acc[1] += (double)sqrtf(fabsf(ep1 - en0));  // i = 0, j = 1
    acc[2] += (double)sqrtf(fabsf(ep0 - en1));  // i = 1, j = 0
    acc[3] += (double)sqrtf(fabsf(ep1 - en1));  // i = 1, j = 1
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double v = acc[k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[w][k] = v;
  }
  __syncthreads();
  if (tid < 4) sim[(size_t)n * 4 + tid] = -(float)(red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

__global__ __launch_bounds__(256) void stitch_connect_kernel(const float* __restrict__ masks,
                                                             const int* __restrict__ state, int W, int F, int T,
                                                             int hop, int total, float floor_v, float* __restrict__ out) {
  const int tau = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
  if (tau >= total) return;
  int w_hi = min(W - 1, tau / hop);
  int w_lo = max(0, (tau - T + hop) / hop);  // smallest w with tau - hop*w <= T-1
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  int cnt = 0;
  for (int w = w_lo; w <= w_hi; ++w) {
    const int t = tau - hop * w;
    if (t < 0 || t >= T) continue;
    const float* m = masks + (((size_t)w * F + f) * T + t) * 3;
    float s0 = m[0], s1 = m[1];
    const float nz = m[2];
    if (state[w]) { const float tmp = s0; s0 = s1; s1 = tmp; }
    const float mx = fmaxf(fmaxf(s0, s1), nz);
    a0 += (s0 == mx) ? s0 : floor_v;
    a1 += (s1 == mx) ? s1 : floor_v;
    a2 += (nz == mx) ? nz : floor_v;
    ++cnt;
  }
  const float inv = 1.f / (float)max(cnt, 1);
  out[((size_t)0 * F + f) * total + tau] = a0 * inv;
  out[((size_t)1 * F + f) * total + tau] = a1 * inv;
  out[((size_t)2 * F + f) * total + tau] = a2 * inv;
}

}  // namespace

extern "C" int csship_stitch_similarity(const float* mags, const float* masks, int32_t W, int32_t F, int32_t T,
                                        int32_t margin, float* sim, csship_stream_t stream) {
  if (!mags || !masks || !sim) return CSSHIP_ENULL;
  if (W < 2 || F <= 0 || T <= 0 || margin <= 0 || margin > T) return CSSHIP_EBADSHAPE;
  hipLaunchKernelGGL(stitch_similarity_kernel, dim3(W - 1), dim3(256), 0, as_stream(stream), mags, masks, F, T, margin, sim);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_stitch_connect(const float* masks, const int32_t* state, int32_t W, int32_t F, int32_t T,
                                     int32_t hop, float wta_floor, float* out, csship_stream_t stream) {
  if (!masks || !state || !out) return CSSHIP_ENULL;
  if (W < 1 || F <= 0 || T <= 0 || hop <= 0 || F > 65535) return CSSHIP_EBADSHAPE;
  const int total = hop * (W - 1) + T;
  hipLaunchKernelGGL(stitch_connect_kernel, dim3((total + 255) / 256, F), dim3(256), 0, as_stream(stream), masks, state,
                     W, F, T, hop, total, wta_floor, out);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
