// Mask-based MVDR beamformer (css/executor/beamformer.py:126-183, arithmetic of asteroid's
// compute_scm + SoudenMVDRBeamformer, also stated in css/utils/mvdr_util.py:45-66).
//   scm_kernel      SCM[b,f] = sum_t m[b,t,f] x[b,:,t,f] x[b,:,t,f]^H  (D x D Hermitian) and sum_t m: HBM-bound,
//                   lane = frequency bin (coalesced over F), 4 waves split the frames, upper triangle in registers
//   souden_kernel   per (b,f): solve N Z = T (complex D x D, partial pivoting, fp64), w = Z[:, ref] / (tr Z + eps)
//   apply_kernel    y[b,f,t] = sum_d conj(w[b,f,d]) x[b,d,t,f] (TF -> FT through a 32x32 LDS tile) + the two energy
//                   sums of the rescale (beamformer.py:165-183)
//   finish_kernel   energy rescale and the deduplication gain of beamformer.py:67-80 for the two sources
// Spectra are separate re / im planes; x is in the STFT kernel's layout [B, D, T, F], masks are [B, T, F], results
// [B, F, T] (what csship_istft consumes).
#include "common.hpp"

namespace {

// up to four masks over the same spectra in one launch (blockIdx.z picks the mask and its outputs): the beamformer
// needs the noise and the two speech covariances of every batch, and one mask's grid is 5 x B workgroups
struct ScmSet {
  const float* mask[4];
  float* scm[4];
  float* msum[4];
};

template <int D>
__global__ __launch_bounds__(256) void scm_kernel(const float* __restrict__ xr, const float* __restrict__ xi, ScmSet set,
                                                  int64_t ldm, int64_t mask_bs, int T, int Tx, int F, float diag,
                                                  int normalize) {
  const float* __restrict__ mask = set.mask[blockIdx.z];
  float* __restrict__ scm = set.scm[blockIdx.z];
  float* __restrict__ msum = set.msum[blockIdx.z];
  constexpr int NTRI = D * (D + 1) / 2;
  __shared__ float red[4][2 * NTRI + 1][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.y, f = blockIdx.x * 64 + lane;
  const bool ok = f < F;
  const int fc = ok ? f : F - 1;
  float ar[NTRI], ai[NTRI], ms = 0.f;
#pragma unroll
  for (int i = 0; i < NTRI; ++i) ar[i] = ai[i] = 0.f;
  const size_t xb = (size_t)b * D * Tx * F;
  for (int t = w; t < T; t += 4) {
    const float m = mask[(size_t)b * mask_bs + (size_t)t * ldm + fc];
    float vr[D], vi[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      vr[d] = xr[xb + ((size_t)d * Tx + t) * F + fc];
      vi[d] = xi[xb + ((size_t)d * Tx + t) * F + fc];
    }
    ms += m;
    int k = 0;
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
      for (int q = p; q < D; ++q, ++k) {  // x_p conj(x_q)
        ar[k] += m * (vr[p] * vr[q] + vi[p] * vi[q]);
        ai[k] += m * (vi[p] * vr[q] - vr[p] * vi[q]);
      }
  }
#pragma unroll
  for (int i = 0; i < NTRI; ++i) {
    red[w][2 * i][lane] = ar[i];
    red[w][2 * i + 1][lane] = ai[i];
  }
  red[w][2 * NTRI][lane] = ms;
  __syncthreads();
  if (w != 0 || !ok) return;
  float tot = 0.f;
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) tot += red[ww][2 * NTRI][lane];
  msum[(size_t)b * F + f] = tot;
  const float sc = normalize ? 1.f / tot : 1.f;
  float* o = scm + ((size_t)b * F + f) * D * D * 2;
  int k = 0;
#pragma unroll
  for (int p = 0; p < D; ++p)
#pragma unroll
    for (int q = p; q < D; ++q, ++k) {
      float re = 0.f, im = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) {
        re += red[ww][2 * k][lane];
        im += red[ww][2 * k + 1][lane];
      }
      re *= sc;
      im *= sc;
      if (p == q) {
        re += diag;
        im = 0.f;
      }
      o[(p * D + q) * 2] = re;
      o[(p * D + q) * 2 + 1] = im;
      if (p != q) {  // Hermitian lower triangle
        o[(q * D + p) * 2] = re;
        o[(q * D + p) * 2 + 1] = -im;
      }
    }
}

// one thread per (b, f): Gaussian elimination with partial pivoting on [N | T] in fp64
template <int D>
__global__ __launch_bounds__(64) void souden_kernel(const float* __restrict__ tgt, const float* __restrict__ noi,
                                                    const float* __restrict__ noi2, int64_t n, int ref, double eps,
                                                    float* __restrict__ w) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  double ar[D][2 * D], ai[D][2 * D];
  const float* N = noi + i * D * D * 2;
  const float* N2 = noi2 ? noi2 + i * D * D * 2 : nullptr;
  const float* Tm = tgt + i * D * D * 2;
#pragma unroll
  for (int p = 0; p < D; ++p)
#pragma unroll
    for (int q = 0; q < D; ++q) {
      ar[p][q] = (double)N[(p * D + q) * 2] + (N2 ? (double)N2[(p * D + q) * 2] : 0.0);
      ai[p][q] = (double)N[(p * D + q) * 2 + 1] + (N2 ? (double)N2[(p * D + q) * 2 + 1] : 0.0);
      ar[p][D + q] = Tm[(p * D + q) * 2];
      ai[p][D + q] = Tm[(p * D + q) * 2 + 1];
    }
#pragma unroll
  for (int c = 0; c < D; ++c) {
    // pivot: largest |a[r][c]|, r >= c (branch-free row swap keeps the arrays in registers)
    int piv = c;
    double best = ar[c][c] * ar[c][c] + ai[c][c] * ai[c][c];
#pragma unroll
    for (int r = c + 1; r < D; ++r) {
      const double v = ar[r][c] * ar[r][c] + ai[r][c] * ai[r][c];
      if (v > best) {
        best = v;
        piv = r;
      }
    }
#pragma unroll
    for (int r = c + 1; r < D; ++r) {
      const bool sw = piv == r;
#pragma unroll
      for (int q = 0; q < 2 * D; ++q) {
        const double tr = ar[c][q], ti = ai[c][q];
        ar[c][q] = sw ? ar[r][q] : tr;
        ai[c][q] = sw ? ai[r][q] : ti;
        ar[r][q] = sw ? tr : ar[r][q];
        ai[r][q] = sw ? ti : ai[r][q];
      }
    }
    const double den = ar[c][c] * ar[c][c] + ai[c][c] * ai[c][c];
    const double ir = ar[c][c] / den, ii = -ai[c][c] / den;  // 1 / a[c][c]
#pragma unroll
    for (int q = c; q < 2 * D; ++q) {
      const double tr = ar[c][q] * ir - ai[c][q] * ii, ti = ar[c][q] * ii + ai[c][q] * ir;
      ar[c][q] = tr;
      ai[c][q] = ti;
    }
#pragma unroll
    for (int r = 0; r < D; ++r) {
      if (r == c) continue;
      const double fr = ar[r][c], fi = ai[r][c];
#pragma unroll
      for (int q = c; q < 2 * D; ++q) {
        ar[r][q] -= fr * ar[c][q] - fi * ai[c][q];
        ai[r][q] -= fr * ai[c][q] + fi * ar[c][q];
      }
    }
  }
  // Z = right half; w = Z[:, ref] / (trace(Z) + eps)
  double tr_r = eps, tr_i = 0.0;
#pragma unroll
  for (int p = 0; p < D; ++p) {
    tr_r += ar[p][D + p];
    tr_i += ai[p][D + p];
  }
  const double den = tr_r * tr_r + tr_i * tr_i;
  float* o = w + i * D * 2;
#pragma unroll
  for (int p = 0; p < D; ++p) {
    double zr = 0.0, zi = 0.0;
#pragma unroll
    for (int q = 0; q < D; ++q)
      if (q == ref) {
        zr = ar[p][D + q];
        zi = ai[p][D + q];
      }
    o[2 * p] = (float)((zr * tr_r + zi * tr_i) / den);
    o[2 * p + 1] = (float)((zi * tr_r - zr * tr_i) / den);
  }
}

// y[b,f,t] = sum_d conj(w[b,f,d]) x[b,d,t,f]; 32x32 (t, f) tiles transposed through LDS; energy partials per workgroup:
// part[(wg)*2 + 0] = sum |y|^2, + 1 = sum |mask * x_0|^2
template <int D>
__global__ __launch_bounds__(256) void bf_apply_kernel(const float* __restrict__ xr, const float* __restrict__ xi,
                                                       const float* __restrict__ w, const float* __restrict__ mask,
                                                       int64_t ldm, int64_t mask_bs, int T, int Tx, int F,
                                                       float* __restrict__ yr, float* __restrict__ yi,
                                                       float* __restrict__ part) {
  __shared__ float tr[32][33], ti[32][33];
  __shared__ float red[2][4];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int b = blockIdx.z, t0 = blockIdx.y * 32, f0 = blockIdx.x * 32;
  const int f = f0 + tx;
  float wr[D], wi[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    wr[d] = f < F ? w[(((size_t)b * F + f) * D + d) * 2] : 0.f;
    wi[d] = f < F ? w[(((size_t)b * F + f) * D + d) * 2 + 1] : 0.f;
  }
  float e_y = 0.f, e_m = 0.f;
  const size_t xb = (size_t)b * D * Tx * F;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i;
    float ar = 0.f, ai = 0.f;
    if (t < T && f < F) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const float vr = xr[xb + ((size_t)d * Tx + t) * F + f], vi = xi[xb + ((size_t)d * Tx + t) * F + f];
        ar += wr[d] * vr + wi[d] * vi;   // conj(w) * x
        ai += wr[d] * vi - wi[d] * vr;
        if (d == 0 && mask) {
          const float m = mask[(size_t)b * mask_bs + (size_t)t * ldm + f];
          e_m += m * m * (vr * vr + vi * vi);
        }
      }
      e_y += ar * ar + ai * ai;
    }
    tr[ty + 8 * i][tx] = ar;
    ti[ty + 8 * i][tx] = ai;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ff = f0 + ty + 8 * i, t = t0 + tx;
    if (t < T && ff < F) {
      yr[((size_t)b * F + ff) * T + t] = tr[tx][ty + 8 * i];
      yi[((size_t)b * F + ff) * T + t] = ti[tx][ty + 8 * i];
    }
  }
  e_y = wave_sum(e_y);
  e_m = wave_sum(e_m);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = e_y;
    red[1][threadIdx.x >> 6] = e_m;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const size_t wg = (size_t)blockIdx.z * gridDim.y * gridDim.x + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    part[wg * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[wg * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// energies[b][0..1] = sum over this batch item's workgroups (fixed order)
__global__ __launch_bounds__(64) void bf_energy_kernel(const float* __restrict__ part, int per_b, float* __restrict__ e) {
  const int b = blockIdx.x, lane = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int i = lane; i < per_b; i += 64) {
    s0 += (double)part[((size_t)b * per_b + i) * 2];
    s1 += (double)part[((size_t)b * per_b + i) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    s0 += __shfl_xor(s0, o);
    s1 += __shfl_xor(s1, o);
  }
  if (lane == 0) {
    e[b * 2] = (float)s0;
    e[b * 2 + 1] = (float)s1;
  }
}

// rescale (y *= sqrt(E_masked / E_y)) and deduplication of the two sources, in place
__global__ __launch_bounds__(256) void bf_finish_kernel(float* __restrict__ y0r, float* __restrict__ y0i,
                                                        float* __restrict__ y1r, float* __restrict__ y1i,
                                                        const float* __restrict__ e0, const float* __restrict__ e1,
                                                        int rescale, int dedup, int64_t per_b) {
  const int b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_b) return;
  const float s0 = rescale ? sqrtf(e0[b * 2 + 1] / e0[b * 2]) : 1.f, s1 = rescale ? sqrtf(e1[b * 2 + 1] / e1[b * 2]) : 1.f;
  const int64_t o = (int64_t)b * per_b + i;
  float ar = y0r[o] * s0, ai = y0i[o] * s0, br = y1r[o] * s1, bi = y1i[o] * s1;
  if (dedup) {
    // powers after the rescale, in dB; the quieter source is attenuated where the louder one dominates
    const float p0 = 10.f * log10f(e0[b * 2] * s0 * s0), p1 = 10.f * log10f(e1[b * 2] * s1 * s1);
    const float a0 = sqrtf(ar * ar + ai * ai), a1 = sqrtf(br * br + bi * bi);
    const float mx = fmaxf(a0, a1);
    if (p0 - p1 > 15.f) {
      const float gn = fmaxf(a1 / mx, 0.01f);
      br *= gn;
      bi *= gn;
    } else if (p1 - p0 > 15.f) {
      const float gn = fmaxf(a0 / mx, 0.01f);
      ar *= gn;
      ai *= gn;
    }
  }
  y0r[o] = ar; y0i[o] = ai; y1r[o] = br; y1i[o] = bi;
}

}  // namespace

#define MVDR_DISPATCH_D(D_, CALL) \
  switch (D_) {                   \
    case 2: { constexpr int DD = 2; CALL; } break; \
    case 3: { constexpr int DD = 3; CALL; } break; \
    case 4: { constexpr int DD = 4; CALL; } break; \
    case 5: { constexpr int DD = 5; CALL; } break; \
    case 6: { constexpr int DD = 6; CALL; } break; \
    case 7: { constexpr int DD = 7; CALL; } break; \
    case 8: { constexpr int DD = 8; CALL; } break; \
    default: return CSSHIP_EUNSUPPORTED;           \
  }

extern "C" int csship_mvdr_scm(const float* x_re, const float* x_im, const float* mask, int64_t ld_mask,
                               int64_t mask_batch_stride, int32_t B, int32_t D, int32_t T, int32_t T_x, int32_t F,
                               float diag_load, int32_t normalize, float* scm, float* mask_sum, csship_stream_t stream) {
  if (!x_re || !x_im || !mask || !scm || !mask_sum) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || T_x < T || F <= 0 || ld_mask < F || B > 65535) return CSSHIP_EBADSHAPE;
  const float* masks[1] = {mask};
  float* scms[1] = {scm};
  float* sums[1] = {mask_sum};
  return csship_mvdr_scm_multi(x_re, x_im, masks, 1, ld_mask, mask_batch_stride, B, D, T, T_x, F, diag_load, normalize, scms, sums,
                               stream);
}

extern "C" int csship_mvdr_scm_multi(const float* x_re, const float* x_im, const float* const* masks, int32_t n_masks,
                                     int64_t ld_mask, int64_t mask_batch_stride, int32_t B, int32_t D, int32_t T, int32_t T_x,
                                     int32_t F, float diag_load, int32_t normalize, float* const* scms, float* const* mask_sums,
                                     csship_stream_t stream) {
  if (!x_re || !x_im || !masks || !scms || !mask_sums) return CSSHIP_ENULL;
  if (n_masks < 1 || n_masks > 4) return CSSHIP_EUNSUPPORTED;
  if (B <= 0 || T <= 0 || T_x < T || F <= 0 || ld_mask < F || B > 65535) return CSSHIP_EBADSHAPE;
  ScmSet set{};
  for (int i = 0; i < n_masks; ++i) {
    if (!masks[i] || !scms[i] || !mask_sums[i]) return CSSHIP_ENULL;
    set.mask[i] = masks[i]; set.scm[i] = scms[i]; set.msum[i] = mask_sums[i];
  }
  dim3 grid((F + 63) / 64, B, n_masks);
  hipStream_t s = as_stream(stream);
  MVDR_DISPATCH_D(D, hipLaunchKernelGGL((scm_kernel<DD>), grid, dim3(256), 0, s, x_re, x_im, set, ld_mask, mask_batch_stride, T, T_x,
                                        F, diag_load, normalize))
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_mvdr_souden(const float* target_scm, const float* noise_scm, const float* noise_scm2, int64_t n_bins,
                                  int32_t D, int32_t ref_mic, float trace_eps, float* w, csship_stream_t stream) {
  if (!target_scm || !noise_scm || !w) return CSSHIP_ENULL;
  if (n_bins <= 0 || ref_mic < 0 || ref_mic >= D) return CSSHIP_EBADSHAPE;
  hipStream_t s = as_stream(stream);
  MVDR_DISPATCH_D(D, hipLaunchKernelGGL((souden_kernel<DD>), dim3((unsigned)((n_bins + 63) / 64)), dim3(64), 0, s, target_scm,
                                        noise_scm, noise_scm2, n_bins, ref_mic, (double)trace_eps, w))
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" size_t csship_mvdr_apply_workspace(int32_t B, int32_t T, int32_t F) {
  return (size_t)B * ((T + 31) / 32) * ((F + 31) / 32) * 2 * sizeof(float);
}

extern "C" int csship_mvdr_apply(const float* x_re, const float* x_im, const float* w, const float* mask, int64_t ld_mask,
                                 int64_t mask_batch_stride, int32_t B, int32_t D, int32_t T, int32_t T_x, int32_t F,
                                 float* y_re, float* y_im, float* energies, void* workspace, size_t workspace_bytes,
                                 csship_stream_t stream) {
  if (!x_re || !x_im || !w || !y_re || !y_im || !energies || !workspace) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || T_x < T || F <= 0 || B > 65535 || (T + 31) / 32 > 65535 || (mask && ld_mask < F))
    return CSSHIP_EBADSHAPE;
  if (workspace_bytes < csship_mvdr_apply_workspace(B, T, F)) return CSSHIP_EWORKSPACE;
  dim3 grid((F + 31) / 32, (T + 31) / 32, B);
  hipStream_t s = as_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
  MVDR_DISPATCH_D(D, hipLaunchKernelGGL((bf_apply_kernel<DD>), grid, dim3(256), 0, s, x_re, x_im, w, mask, ld_mask, mask_batch_stride, T, T_x, F,
                                        y_re, y_im, part))
  CSSHIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(bf_energy_kernel, dim3(B), dim3(64), 0, s, part, (int)(grid.x * grid.y), energies);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_mvdr_finish(float* y0_re, float* y0_im, float* y1_re, float* y1_im, const float* energies0,
                                  const float* energies1, int32_t rescale, int32_t dedup, int32_t B, int64_t per_batch,
                                  csship_stream_t stream) {
  if (!y0_re || !y0_im || !y1_re || !y1_im || !energies0 || !energies1) return CSSHIP_ENULL;
  if (B <= 0 || per_batch <= 0 || B > 65535) return CSSHIP_EBADSHAPE;
  hipLaunchKernelGGL(bf_finish_kernel, dim3((unsigned)((per_batch + 255) / 256), B), dim3(256), 0, as_stream(stream), y0_re,
                     y0_im, y1_re, y1_im, energies0, energies1, rescale, dedup, per_batch);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
