// DOA likelihood for the mask merge of the 7-channel separator (css/executor/separator.py:206-250, softmax = False):
//   ret[b,c,a] = sum_{f in [lo,hi), t} binmask[b,f,t,c] * -( |x|^2 - |sv[f,a]^T conj(x)|^2 )^compression
// with binmask = mask > thr, plus the masked first-channel magnitude sums that pick the output to silence
// (separator.py:196-200).  Workgroup = (b, f): the steering row sv[f, :, :] is wave-uniform (scalar loads), threads
// walk the frames (re / im are [B, 7, F, T]: coalesced over T), 2 x A accumulators per thread, one shuffle reduction.
#include "common.hpp"

namespace {

constexpr int NCH = 7;

template <int A>
__global__ __launch_bounds__(64) void doa_like_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                      const float* __restrict__ mask, const float* __restrict__ sv_re,
                                                      const float* __restrict__ sv_im, int F, int T, int lo, int hi,
                                                      float thr, float compression, float* __restrict__ like_part,
                                                      float* __restrict__ en_part) {
  const int b = blockIdx.y, f = blockIdx.x, lane = threadIdx.x;
  const bool band = f >= lo && f < hi;
  float acc[2][A], en[2] = {0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int a = 0; a < A; ++a) acc[c][a] = 0.f;
  for (int t = lane; t < T; t += 64) {
    float xr[NCH], xi[NCH], xp = 0.f;
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
      const size_t o = (((size_t)b * NCH + m) * F + f) * T + t;
      xr[m] = re[o];
      xi[m] = im[o];
      xp += xr[m] * xr[m] + xi[m] * xi[m];
    }
    const float* mp = mask + (((size_t)b * F + f) * T + t) * 2;
    const float b0 = mp[0] > thr ? 1.f : 0.f, b1 = mp[1] > thr ? 1.f : 0.f;
    const float a0 = sqrtf(xr[0] * xr[0] + xi[0] * xi[0]);
    en[0] += b0 * a0;
    en[1] += b1 * a0;
    if (band && (b0 + b1) > 0.f) {
#pragma unroll
      for (int a = 0; a < A; ++a) {
        float hr = 0.f, hi_ = 0.f;
#pragma unroll
        for (int m = 0; m < NCH; ++m) {  // conj(x_m) * sv[f,a,m]
          const float sr = sv_re[((size_t)f * A + a) * NCH + m], si = sv_im[((size_t)f * A + a) * NCH + m];
          hr += xr[m] * sr + xi[m] * si;
          hi_ += xr[m] * si - xi[m] * sr;
        }
        // the reference divides by (1 + 1e-12), which is 1 in float32; rounding can leave a tiny negative base,
        // which torch.pow turns into NaN: clamp at 0 instead
        const float base = fmaxf(xp - (hr * hr + hi_ * hi_), 0.f);
        const float v = compression > 0.f ? -powf(base, compression) : -logf(base);
        acc[0][a] += b0 * v;
        acc[1][a] += b1 * v;
      }
    }
  }
  if (band) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const float s = wave_sum(acc[c][a]);
        if (lane == 0) like_part[(((size_t)b * (hi - lo) + (f - lo)) * 2 + c) * A + a] = s;
      }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float s = wave_sum(en[c]);
    if (lane == 0) en_part[((size_t)b * F + f) * 2 + c] = s;
  }
}

// like[b, c*A + a] = sum_f like_part ; energy[b, c] = sum_f en_part ; row_energy[b] = en_part row f = b (quirk 6)
__global__ __launch_bounds__(64) void doa_reduce_kernel(const float* __restrict__ like_part, const float* __restrict__ en_part,
                                                        int nband, int F, int CA, float* __restrict__ like,
                                                        float* __restrict__ energy) {
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int j = lane; j < CA; j += 64) {
    float s = 0.f;
    for (int f = 0; f < nband; ++f) s += like_part[((size_t)b * nband + f) * CA + j];
    like[(size_t)b * CA + j] = s;
  }
  if (lane < 2) {
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += en_part[((size_t)b * F + f) * 2 + lane];
    energy[b * 2 + lane] = s;
  }
}

}  // namespace

extern "C" size_t csship_doa_workspace(int32_t B, int32_t F, int32_t lo, int32_t hi, int32_t n_angles) {
  return ((size_t)B * (hi - lo) * 2 * n_angles + (size_t)B * F * 2) * sizeof(float);
}

extern "C" int csship_doa_likelihood(const float* x_re, const float* x_im, const float* mask, const float* sv_re,
                                     const float* sv_im, int32_t B, int32_t F, int32_t T, int32_t n_angles, int32_t lo,
                                     int32_t hi, float binarize_thr, float compression, float* likelihood,
                                     float* masked_energy, void* workspace, size_t workspace_bytes,
                                     csship_stream_t stream) {
  if (!x_re || !x_im || !mask || !sv_re || !sv_im || !likelihood || !masked_energy || !workspace) return CSSHIP_ENULL;
  if (B <= 0 || F <= 0 || T <= 0 || lo < 0 || hi > F || lo >= hi || B > 65535) return CSSHIP_EBADSHAPE;
  if (n_angles != 30) return CSSHIP_EUNSUPPORTED;  // separator.py:52 fixes nvecs = 30
  if (workspace_bytes < csship_doa_workspace(B, F, lo, hi, n_angles)) return CSSHIP_EWORKSPACE;
  float* like_part = reinterpret_cast<float*>(workspace);
  float* en_part = like_part + (size_t)B * (hi - lo) * 2 * n_angles;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(doa_like_kernel<30>, dim3(F, B), dim3(64), 0, s, x_re, x_im, mask, sv_re, sv_im, F, T, lo, hi,
                     binarize_thr, compression, like_part, en_part);
  CSSHIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(doa_reduce_kernel, dim3(B), dim3(64), 0, s, like_part, en_part, hi - lo, F, 2 * n_angles, likelihood,
                     masked_energy);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
