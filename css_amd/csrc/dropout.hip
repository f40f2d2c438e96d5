// Dropout on materialised tensors (training mode): csship_dropout_apply / _add / _mask.  The fused kernels (FeedForward,
// attention probabilities, ConvModule) generate the same masks in their epilogues from common.hpp's helpers; these
// elementwise forms serve the sites whose tensors exist in memory anyway (embedding output, attention output projection,
// BLSTM layers) and the tests.  HBM bound: one read + one write of the tensor.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                            int64_t ldy, int64_t rows, int cols, const unsigned* seed,
                                                            unsigned stream, float p) {
  const Drop d = drop_make(seed, stream, p);
  const int c4n = cols / 4;  // cols % 4 == 0 (checked on the host)
  const int64_t n4 = rows * c4n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
    *reinterpret_cast<float4*>(y + r * ldy + c) = drop_apply4(d, (unsigned)(r * cols + c), v);
  }
}

__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          float* __restrict__ out, float alpha, int64_t n4, int cols,
                                                          const unsigned* seed, unsigned stream, float p) {
  const Drop d = drop_make(seed, stream, p);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    const float4 yv = drop_apply4(d, (unsigned)(i * 4), reinterpret_cast<const float4*>(y)[i]);
    reinterpret_cast<float4*>(out)[i] = make_float4(xv.x + alpha * yv.x, xv.y + alpha * yv.y, xv.z + alpha * yv.z, xv.w + alpha * yv.w);
  }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* __restrict__ keep, int64_t n, const unsigned* seed,
                                                           unsigned stream, float p) {
  const Drop d = drop_make(seed, stream, p);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    keep[i] = d.thresh == 0u ? 1 : (drop_keep1(d, (unsigned)i) ? 1 : 0);
}

inline unsigned grid_for(int64_t work) { const int64_t g = (work + 255) / 256; return (unsigned)(g < 2048 ? g : 2048); }

}  // namespace

extern "C" int csship_dropout_apply(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int32_t cols,
                                    const csship_dropout_cfg* cfg, csship_stream_t stream) {
  if (!x || !y) return CSSHIP_ENULL;
  if (rows <= 0 || cols <= 0 || (cols % 4) != 0 || ldx < cols || ldy < cols || rows * (int64_t)cols > 0xffffffffLL)
    return CSSHIP_EBADSHAPE;
  if (!aligned16(x) || !aligned16(y) || (ldx % 4) != 0 || (ldy % 4) != 0) return CSSHIP_EALIGN;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(grid_for(rows * (cols / 4))), dim3(256), 0, as_stream(stream), x, ldx, y, ldy,
                     rows, cols, cfg ? cfg->seed : nullptr, cfg ? cfg->stream : 0u, cfg ? cfg->p : 0.f);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_dropout_add(const float* x, const float* y, float* out, float alpha, int64_t rows, int32_t cols,
                                  const csship_dropout_cfg* cfg, csship_stream_t stream) {
  if (!x || !y || !out) return CSSHIP_ENULL;
  if (rows <= 0 || cols <= 0 || (cols % 4) != 0 || rows * (int64_t)cols > 0xffffffffLL) return CSSHIP_EBADSHAPE;
  if (!aligned16(x) || !aligned16(y) || !aligned16(out)) return CSSHIP_EALIGN;
  const int64_t n4 = rows * cols / 4;
  hipLaunchKernelGGL(dropout_add_kernel, dim3(grid_for(n4)), dim3(256), 0, as_stream(stream), x, y, out, alpha, n4, cols,
                     cfg ? cfg->seed : nullptr, cfg ? cfg->stream : 0u, cfg ? cfg->p : 0.f);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_dropout_mask(uint8_t* keep, int64_t n, const csship_dropout_cfg* cfg, csship_stream_t stream) {
  if (!keep) return CSSHIP_ENULL;
  if (n <= 0 || n > 0xffffffffLL) return CSSHIP_EBADSHAPE;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), keep, n,
                     cfg ? cfg->seed : nullptr, cfg ? cfg->stream : 0u, cfg ? cfg->p : 0.f);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
