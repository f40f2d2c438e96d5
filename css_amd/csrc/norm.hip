// LayerNorm, time-axis mean/variance normalisation statistics and the TF -> FT transpose.
// All HBM/L2-bound: one wave per row / per column group, float4 traffic where rows allow it.
#include "common.hpp"

namespace {

// ---- LayerNorm over D = 64*VPL*4 ... one wave per row, lane holds VPL float4 (D=256: 1, D=512: 2)
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t M,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        unsigned short* __restrict__ out_bf, float* __restrict__ out_f) {
  constexpr int D = 256 * VPL;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] = *reinterpret_cast<const float4*>(x + row * D + (i * 64 + lane) * 4);
    s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  const float mean = wave_sum(s) * (1.f / D);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
  }
  const float rstd = rsqrtf(wave_sum(ss) * (1.f / D) + 1e-5f);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
    const float4 bt = *reinterpret_cast<const float4*>(beta + c);
    float4 y;
    y.x = v[i].x * rstd * gm.x + bt.x;
    y.y = v[i].y * rstd * gm.y + bt.y;
    y.z = v[i].z * rstd * gm.z + bt.z;
    y.w = v[i].w * rstd * gm.w + bt.w;
    if (out_f) *reinterpret_cast<float4*>(out_f + row * D + c) = y;
    if (out_bf) *reinterpret_cast<uint2*>(out_bf + row * D + c) = make_uint2(pack_bf2(y.x, y.y), pack_bf2(y.z, y.w));
  }
}

// ---- LayerNorm for any D (BLSTM widths): one wave per row, strided scalar accesses, optional ReLU
__global__ __launch_bounds__(256) void layernorm_any_kernel(const float* __restrict__ x, int64_t M, int D,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int relu,
                                                            unsigned short* __restrict__ out_bf, float* __restrict__ out_f) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + row * D;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float d = xr[c] - mean;
    ss += d * d;
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + 1e-5f);
  for (int c = lane; c < D; c += 64) {
    float y = (xr[c] - mean) * rstd * gamma[c] + beta[c];
    if (relu) y = fmaxf(y, 0.f);
    if (out_f) out_f[row * D + c] = y;
    if (out_bf) out_bf[row * D + c] = f2bf(y);
  }
}

// ---- per-(b, column) mean and 1/(unbiased std + eps) over T.  Workgroup = (b, 64 columns), 8 waves;
// wave w walks rows t = w, w+8, ... with 8 independent loads in flight; two passes (mean, centred sum
// of squares) like torch, the second one served from L2.
constexpr int MV_W = 8;
__global__ __launch_bounds__(MV_W * 64) void mvn_stats_kernel(const float* __restrict__ x, int T, int F, int64_t ldx,
                                                             float clamp_min, float eps, float* __restrict__ mean,
                                                             float* __restrict__ rinv) {
  __shared__ float red[MV_W][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.y, f = blockIdx.x * 64 + lane;
  const bool ok = f < F;
  const float* xb = x + (size_t)b * T * ldx + (ok ? f : 0);
  float s = 0.f;
  int t = w;
  for (; t + 7 * MV_W < T; t += 8 * MV_W) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = xb[(size_t)(t + i * MV_W) * ldx];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += fmaxf(v[i], clamp_min);
  }
  for (; t < T; t += MV_W) s += fmaxf(xb[(size_t)t * ldx], clamp_min);
  red[w][lane] = s;
  __syncthreads();
  float mu = 0.f;
#pragma unroll
  for (int i = 0; i < MV_W; ++i) mu += red[i][lane];
  mu /= (float)T;
  __syncthreads();
  float ss = 0.f;
  t = w;
  for (; t + 7 * MV_W < T; t += 8 * MV_W) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = xb[(size_t)(t + i * MV_W) * ldx];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float d = fmaxf(v[i], clamp_min) - mu;
      ss += d * d;
    }
  }
  for (; t < T; t += MV_W) {
    const float d = fmaxf(xb[(size_t)t * ldx], clamp_min) - mu;
    ss += d * d;
  }
  red[w][lane] = ss;
  __syncthreads();
  if (w == 0 && ok) {
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < MV_W; ++i) var += red[i][lane];
    var /= (float)(T - 1);
    mean[(size_t)b * F + f] = mu;
    rinv[(size_t)b * F + f] = 1.f / (sqrtf(var) + eps);
  }
}

// ---- out[b, f_off + f, t] = (max(x[b,t,f], clamp) - mean[b,f]) * rinv[b,f];  32x32 LDS tile transpose
__global__ __launch_bounds__(256) void transpose_norm_kernel(const float* __restrict__ x, int T, int F, int64_t ldx,
                                                             float clamp_min, const float* __restrict__ mean,
                                                             const float* __restrict__ rinv, float* __restrict__ out,
                                                             int64_t out_bs, int64_t f_off) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int b = blockIdx.z, t0 = blockIdx.y * 32, f0 = blockIdx.x * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, f = f0 + tx;
    float v = 0.f;
    if (t < T && f < F) {
      v = fmaxf(x[((size_t)b * T + t) * ldx + f], clamp_min);
      if (mean) v = (v - mean[(size_t)b * F + f]) * rinv[(size_t)b * F + f];
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = f0 + ty + 8 * i, t = t0 + tx;
    if (t < T && f < F) out[(size_t)b * out_bs + (f_off + f) * (size_t)T + t] = tile[tx][ty + 8 * i];
  }
}

}  // namespace

extern "C" int csship_layernorm_relu(const float* x, int64_t M, int32_t D, const float* gamma, const float* beta,
                                     int32_t relu, uint16_t* out_bf16, float* out_f32, csship_stream_t stream) {
  if (!x || !gamma || !beta || (!out_bf16 && !out_f32)) return CSSHIP_ENULL;
  if (M <= 0 || D <= 0) return CSSHIP_EBADSHAPE;
  dim3 grid((unsigned)((M + 3) / 4));
  hipLaunchKernelGGL(layernorm_any_kernel, grid, dim3(256), 0, as_stream(stream), x, M, D, gamma, beta, relu, out_bf16,
                     out_f32);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_layernorm(const float* x, int64_t M, int32_t D, const float* gamma, const float* beta,
                                uint16_t* out_bf16, float* out_f32, csship_stream_t stream) {
  if (!x || !gamma || !beta || (!out_bf16 && !out_f32)) return CSSHIP_ENULL;
  if (M <= 0) return CSSHIP_EBADSHAPE;
  if (D != 256 && D != 512) return csship_layernorm_relu(x, M, D, gamma, beta, 0, out_bf16, out_f32, stream);
  if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || (out_f32 && !aligned16(out_f32)) ||
      (out_bf16 && !aligned8(out_bf16)))
    return CSSHIP_EALIGN;
  dim3 grid((unsigned)((M + 3) / 4));
  if (D == 256)
    hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, as_stream(stream), x, M, gamma, beta, out_bf16, out_f32);
  else if (D == 512)
    hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, as_stream(stream), x, M, gamma, beta, out_bf16, out_f32);
  else
    return CSSHIP_EUNSUPPORTED;
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_mvn_stats(const float* x, int64_t B, int64_t T, int64_t F, int64_t ldx, float clamp_min,
                                float eps, float* mean, float* rinv, csship_stream_t stream) {
  if (!x || !mean || !rinv) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || F <= 0 || ldx < F || B > 65535) return CSSHIP_EBADSHAPE;
  dim3 grid((unsigned)((F + 63) / 64), (unsigned)B);
  TimelineScope tl(CSSHIP_TL_MVN, as_stream(stream));
  hipLaunchKernelGGL(mvn_stats_kernel, grid, dim3(MV_W * 64), 0, as_stream(stream), x, (int)T, (int)F, ldx, clamp_min, eps,
                     mean, rinv);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_transpose_norm(const float* x, int64_t B, int64_t T, int64_t F, int64_t ldx, float clamp_min,
                                     const float* mean, const float* rinv, float* out, int64_t out_batch_stride,
                                     int64_t f_off, csship_stream_t stream) {
  if (!x || !out || ((mean == nullptr) != (rinv == nullptr))) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || F <= 0 || ldx < F || B > 65535 || (T + 31) / 32 > 65535) return CSSHIP_EBADSHAPE;
  dim3 grid((unsigned)((F + 31) / 32), (unsigned)((T + 31) / 32), (unsigned)B);
  hipLaunchKernelGGL(transpose_norm_kernel, grid, dim3(256), 0, as_stream(stream), x, (int)T, (int)F, ldx, clamp_min,
                     mean, rinv, out, out_batch_stride, f_off);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
