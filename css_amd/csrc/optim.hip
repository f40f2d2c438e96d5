// Gradient-norm clipping + Adam on flat buffers (the optimiser step of css/trainer/iteration_types.py:17-64:
// torch.nn.utils.clip_grad_norm_ then torch.optim.Adam.step in the reference, ~60 foreach launches and a host sync
// for the norm).  Two launches, no host round trip: the clip coefficient is read from device memory.
//   1. sqnorm: per-workgroup partial sums of g^2 (fixed order) -> one-workgroup reduce -> norm
//   2. adam:   c = min(1, max_norm / (norm + 1e-6)); g' = c g; m, v moments; p -= lr (m/bc1) / (sqrt(v/bc2) + eps)
//      The update is skipped entirely when the norm is not finite.  DEVIATION from the reference, which has no such
//      guard around optimizer.step() (it only refuses to SAVE a checkpoint after a NaN epoch loss, train.py:313): a
//      non-finite gradient there poisons the parameters, here it costs one step (the host-side step counter and
//      the bias corrections still advance).
#include <algorithm>

#include "common.hpp"

namespace {

constexpr int SQ_ELEMS = 8192;  // elements per workgroup of the norm pass

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float red[4];
  const int64_t base = (int64_t)blockIdx.x * SQ_ELEMS;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < SQ_ELEMS / 1024; ++i) {
    const int64_t e = base + (int64_t)(i * 256 + threadIdx.x) * 4;
    if (e + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(g + e);
      s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    } else {
      for (int64_t k = e; k < n; ++k) s += g[k] * g[k];
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(1024) void norm_finish_kernel(const float* __restrict__ part, int np, float* __restrict__ norm) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 1024) s += (double)part[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += red[i];
    *norm = (float)sqrt(t);
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, const float* __restrict__ norm,
                                                   float max_norm, float lr, float b1, float b2, float eps, float bc1,
                                                   float bc2, float weight_decay, const float* __restrict__ hyper) {
  if (hyper) {  // {lr, 1 - beta1^t, 1 - beta2^t} from device memory: a captured graph is replayed with new values
    lr = hyper[0];
    bc1 = hyper[1];
    bc2 = hyper[2];
  }
  const float nv = *norm;
  if (!(nv <= 3.0e38f)) return;  // NaN / inf gradient: skip the step (uniform)
  const float c = max_norm > 0.f ? fminf(1.f, max_norm / (nv + 1e-6f)) : 1.f;
  const float step = lr / bc1, rbc2 = 1.f / sqrtf(bc2);
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= n) return;
  if (e + 3 < n) {
    float4 pv = *reinterpret_cast<float4*>(p + e), mv = *reinterpret_cast<float4*>(m + e), vv = *reinterpret_cast<float4*>(v + e);
    const float4 gv = *reinterpret_cast<const float4*>(g + e);
    float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float gg = ga[q] * c + weight_decay * pa[q];
      ma[q] = b1 * ma[q] + (1.f - b1) * gg;
      va[q] = b2 * va[q] + (1.f - b2) * gg * gg;
      pa[q] -= step * ma[q] / (sqrtf(va[q]) * rbc2 + eps);
    }
    *reinterpret_cast<float4*>(p + e) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    *reinterpret_cast<float4*>(m + e) = make_float4(ma[0], ma[1], ma[2], ma[3]);
    *reinterpret_cast<float4*>(v + e) = make_float4(va[0], va[1], va[2], va[3]);
  } else {
    for (int64_t k = e; k < n; ++k) {
      const float gg = g[k] * c + weight_decay * p[k];
      m[k] = b1 * m[k] + (1.f - b1) * gg;
      v[k] = b2 * v[k] + (1.f - b2) * gg * gg;
      p[k] -= step * m[k] / (sqrtf(v[k]) * rbc2 + eps);
    }
  }
}

}  // namespace

// ---- many small gradient tensors added into their destinations by ONE launch (csship_grad_accumulate).  The explicit
// backward hands ~30 small gradients per encoder block (LayerNorm / conv parameters, q | k | v slices) to the optimiser's
// flat gradient buffer; as torch._foreach_add_ that is 7 multi-tensor launches of 15 us per step plus a contiguous()
// copy for every transposed view.  The job table travels in the kernel arguments (by value: a captured step replays it).
constexpr int GA_JOBS = 96;
struct GradAccTable {
  csship_grad_job job[GA_JOBS];
};
__global__ __launch_bounds__(256) void grad_accumulate_kernel(GradAccTable t) {
  const csship_grad_job j = t.job[blockIdx.y];
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= j.n) return;
  if (j.cols == 0) {  // contiguous source
    if (j.n - i0 >= 4 && ((((uintptr_t)j.src | (uintptr_t)j.dst) & 15) == 0)) {
      const float4 a = *reinterpret_cast<const float4*>(j.src + i0);
      float4 d = *reinterpret_cast<float4*>(j.dst + i0);
      d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
      *reinterpret_cast<float4*>(j.dst + i0) = d;
    } else {
      for (int64_t i = i0; i < min(j.n, i0 + 4); ++i) j.dst[i] += j.src[i];
    }
  } else if (j.cols < 0) {  // dst [n / C, C] (dense), C = -cols, += rows of src ld_src apart
    const int64_t C = -(int64_t)j.cols;
    for (int64_t i = i0; i < min(j.n, i0 + 4); ++i) {
      const int64_t r = i / C;
      j.dst[i] += j.src[r * j.ld_src + (i - r * C)];
    }
  } else {            // dst [n / cols, cols] (dense) += src^T: element (r, c) comes from src[c * ld_src + r]
    for (int64_t i = i0; i < min(j.n, i0 + 4); ++i) {
      const int64_t r = i / j.cols, c = i - r * j.cols;
      j.dst[i] += j.src[c * j.ld_src + r];
    }
  }
}

// dst = keep * dst + w * src over a table of dense tensors (the BatchNorm running statistics of all blocks: one launch)
__global__ __launch_bounds__(256) void ema_update_kernel(GradAccTable t, float keep, float w_head, int n_head, float w_tail) {
  const csship_grad_job j = t.job[blockIdx.y];
  const float w = (int)blockIdx.y < n_head ? w_head : w_tail;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < j.n) j.dst[i] = fmaf(w, j.src[i], keep * j.dst[i]);
}

extern "C" int csship_ema_update(const csship_grad_job* jobs_host, int32_t n_jobs, float keep, float w_head,
                                 int32_t n_head, float w_tail, csship_stream_t stream) {
  if (n_jobs < 0 || (n_jobs > 0 && !jobs_host)) return CSSHIP_ENULL;
  if (n_jobs > GA_JOBS || n_head < 0) return CSSHIP_EBADSHAPE;
  if (n_jobs == 0) return CSSHIP_OK;
  GradAccTable t{};
  int64_t maxn = 0;
  for (int k = 0; k < n_jobs; ++k) {
    const csship_grad_job& j = jobs_host[k];
    if (!j.src || !j.dst || j.n <= 0 || j.cols != 0) return CSSHIP_EBADSHAPE;
    t.job[k] = j;
    maxn = std::max(maxn, j.n);
  }
  hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)((maxn + 255) / 256), (unsigned)n_jobs), dim3(256), 0,
                     as_stream(stream), t, keep, w_head, (int)n_head, w_tail);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_grad_accumulate(const csship_grad_job* jobs_host, int32_t n_jobs, csship_stream_t stream) {
  if (n_jobs < 0 || (n_jobs > 0 && !jobs_host)) return CSSHIP_ENULL;
  for (int at = 0; at < n_jobs; at += GA_JOBS) {
    GradAccTable t{};
    const int m = std::min(GA_JOBS, n_jobs - at);
    int64_t maxn = 0;
    for (int k = 0; k < m; ++k) {
      const csship_grad_job& j = jobs_host[at + k];
      if (!j.src || !j.dst || j.n <= 0 || j.cols == INT32_MIN || (j.cols != 0 && (j.n % j.cols) != 0) ||
          (j.cols < 0 && j.ld_src < -(int64_t)j.cols))
        return CSSHIP_EBADSHAPE;
      t.job[k] = j;
      maxn = std::max(maxn, j.n);
    }
    hipLaunchKernelGGL(grad_accumulate_kernel, dim3((unsigned)((maxn + 1023) / 1024), (unsigned)m), dim3(256), 0, as_stream(stream), t);
    CSSHIP_CHECK_LAUNCH();
  }
  return CSSHIP_OK;
}

extern "C" size_t csship_adam_clip_workspace(int64_t n) {
  return (size_t)((n + SQ_ELEMS - 1) / SQ_ELEMS) * sizeof(float);
}

extern "C" int csship_adam_clip(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                float max_norm, float lr, float beta1, float beta2, float eps, int64_t step,
                                float weight_decay, const float* hyper_dev, float* grad_norm, void* workspace,
                                size_t workspace_bytes, csship_stream_t stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !grad_norm || !workspace) return CSSHIP_ENULL;
  if (n <= 0 || (step <= 0 && !hyper_dev)) return CSSHIP_EBADSHAPE;
  if (!aligned16(params) || !aligned16(grads) || !aligned16(exp_avg) || !aligned16(exp_avg_sq)) return CSSHIP_EALIGN;
  const int64_t np = (n + SQ_ELEMS - 1) / SQ_ELEMS;
  if (workspace_bytes < (size_t)np * sizeof(float)) return CSSHIP_EWORKSPACE;
  if (np > 0x7fffffff || (n + 1023) / 1024 > 0x7fffffff) return CSSHIP_EBADSHAPE;
  hipStream_t s = as_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)np), dim3(256), 0, s, grads, n, part);
  CSSHIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(norm_finish_kernel, dim3(1), dim3(1024), 0, s, part, (int)np, grad_norm);
  CSSHIP_CHECK_LAUNCH();
  const float bc1 = step > 0 ? 1.f - powf(beta1, (float)step) : 1.f, bc2 = step > 0 ? 1.f - powf(beta2, (float)step) : 1.f;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, n,
                     grad_norm, max_norm, lr, beta1, beta2, eps, bc1, bc2, weight_decay, hyper_dev);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}


// ---- csship_p2p_reduce: out[lo:hi] = scale * sum_j src_j[lo:hi] over up to 8 gradient buffers, summed in the order
// given (fixed: bit-reproducible).  The sources are the ranks' flat gradient buffers as this process sees them - its own
// and its peers' through hipIpc mappings (css_amd.parallel.P2PAllReduce): the reduce-scatter half of the direct
// (one-shot) all-reduce of SURVEY 5.8, which replaces the reference's checkpoint averaging
// (scripts/python/combine_models.py:55-82).  Every byte is read once with 16-byte loads - on an 8-GPU node the seven remote
// slices travel over seven different xGMI links at the same time.  `out` may be one of the sources (in place).
namespace {
struct P2PReduceArgs {
  const float* src[8];
  float* out;
  int world;
  long long lo, hi;
  float scale;
};
__global__ __launch_bounds__(256) void p2p_reduce_kernel(P2PReduceArgs a) {
  const long long stride = (long long)gridDim.x * 256 * 4;
  long long i = a.lo + ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  for (; i + 3 < a.hi; i += stride) {
    float4 acc = *reinterpret_cast<const float4*>(a.src[0] + i);
    for (int j = 1; j < a.world; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(a.src[j] + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(a.out + i) = make_float4(acc.x * a.scale, acc.y * a.scale, acc.z * a.scale, acc.w * a.scale);
  }
  if (i < a.hi) {  // at most one thread: the last 1..3 elements
    for (long long k = i; k < a.hi; ++k) {
      float acc = a.src[0][k];
      for (int j = 1; j < a.world; ++j) acc += a.src[j][k];
      a.out[k] = acc * a.scale;
    }
  }
}
}  // namespace

extern "C" int csship_p2p_reduce(const float* const* srcs_host, int32_t world, int64_t lo, int64_t hi, float scale,
                                 float* out, csship_stream_t stream) {
  if (!srcs_host || !out) return CSSHIP_ENULL;
  if (world < 1 || world > 8 || lo < 0 || hi < lo || (lo & 3)) return CSSHIP_EBADSHAPE;
  if (hi == lo) return CSSHIP_OK;
  P2PReduceArgs a{};
  for (int j = 0; j < world; ++j) {
    if (!srcs_host[j]) return CSSHIP_ENULL;
    if (!aligned16(srcs_host[j])) return CSSHIP_EALIGN;
    a.src[j] = srcs_host[j];
  }
  if (!aligned16(out)) return CSSHIP_EALIGN;
  a.out = out; a.world = world; a.lo = lo; a.hi = hi; a.scale = scale;
  const long long n4 = (hi - lo + 3) / 4;
  const unsigned grid = (unsigned)std::min<long long>((n4 + 255) / 256, 2048);
  hipLaunchKernelGGL(p2p_reduce_kernel, dim3(grid), dim3(256), 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
