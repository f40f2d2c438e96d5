// Inter-channel phase difference features (css/executor/feature.py:105-134) and the inverse STFT used
// after beamforming (torch.istft(center=True, hann) at css/executor/beamformer.py:83-102).
#include "fft.hpp"

namespace {

constexpr int NBIN = 257;

// cos / sin of (phase_l - phase_r) from the complex spectra, without atan2/cos/sin round trips:
//   e^{i(pl - pr)} = u_l * conj(u_r),  u = (re, im)/|.|   (|.| = 0 -> phase 0 like atan2(0, 0))
DEV void unit_phase(float re, float im, float& c, float& s) {
  const float m2 = re * re + im * im;
  if (m2 > 0.f) {
    const float r = rsqrtf(m2);
    c = re * r;
    s = im * r;
  } else {
    c = 1.f;
    s = 0.f;
  }
}
DEV void phase_diff(float rl, float il, float rr, float ir, float& yr, float& yi) {
  float cl, sl, cr, sr;
  unit_phase(rl, il, cl, sl);
  unit_phase(rr, ir, cr, sr);
  yr = cl * cr + sl * sr;
  yi = sl * cr - cl * sr;
}

struct Pairs {
  int l[16], r[16];
};

// pass 1: time means of cos / sin per (b, pair, bin).  Workgroup = (64 bins, pair, b), 8 waves over t.
__global__ __launch_bounds__(512) void ipd_mean_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                       int C, int T, Pairs pr, float* __restrict__ means /*[B,P,2,257]*/) {
  __shared__ float red[2][8][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + lane, p = blockIdx.y, b = blockIdx.z, P = gridDim.y;
  const bool ok = f < NBIN;
  const size_t cl = ((size_t)b * C + pr.l[p]) * T, cr = ((size_t)b * C + pr.r[p]) * T;
  float sr = 0.f, si = 0.f;
  for (int t = w; t < T; t += 8) {
    if (ok) {
      const size_t ol = (cl + t) * NBIN + f, orr = (cr + t) * NBIN + f;
      float yr, yi;
      phase_diff(re[ol], im[ol], re[orr], im[orr], yr, yi);
      sr += yr;
      si += yi;
    }
  }
  red[0][w][lane] = sr;
  red[1][w][lane] = si;
  __syncthreads();
  if (w == 0 && ok) {
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a += red[0][i][lane];
      c += red[1][i][lane];
    }
    means[(((size_t)b * P + p) * 2 + 0) * NBIN + f] = a / (float)T;
    means[(((size_t)b * P + p) * 2 + 1) * NBIN + f] = c / (float)T;
  }
}

// pass 2: ipd = atan2(sin - mean sin, cos - mean cos), written transposed ([.., bin, t]) through a 32x32 tile
__global__ __launch_bounds__(256) void ipd_out_kernel(const float* __restrict__ re, const float* __restrict__ im, int C,
                                                      int T, Pairs pr, int P, const float* __restrict__ means,
                                                      float* __restrict__ out, int64_t out_bs, int64_t f_off) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int bp = blockIdx.z, b = bp / P, p = bp % P;
  const int t0 = blockIdx.y * 32, f0 = blockIdx.x * 32;
  const size_t cl = ((size_t)b * C + pr.l[p]) * T, cr = ((size_t)b * C + pr.r[p]) * T;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, f = f0 + tx;
    float v = 0.f;
    if (t < T && f < NBIN) {
      const size_t ol = (cl + t) * NBIN + f, orr = (cr + t) * NBIN + f;
      float yr, yi;
      phase_diff(re[ol], im[ol], re[orr], im[orr], yr, yi);
      const float mr = means[(((size_t)b * P + p) * 2 + 0) * NBIN + f];
      const float mi = means[(((size_t)b * P + p) * 2 + 1) * NBIN + f];
      v = atan2f(yi - mi, yr - mr);
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = f0 + ty + 8 * i, t = t0 + tx;
    if (t < T && f < NBIN) out[(size_t)b * out_bs + (f_off + (size_t)p * NBIN + f) * T + t] = tile[tx][ty + 8 * i];
  }
}

// ---- frame-major IPD (what the models consume: out[b, t, f_off + p*257 + f]).  One wave = one frame at a time: lane l
// takes bins 4l .. 4l+3 with ONE 16-byte load per channel and component (rows are 257 floats: 4-byte aligned), bin 256
// of the frame by lane 0.  Pass 1 leaves per-(batch, pair, time slice) partial sums, pass 2 adds them in a fixed order
// (bit-reproducible).
constexpr int IPD_TS = 8;  // time slices of pass 1

struct __attribute__((packed, aligned(4))) f4u {
  float v[4];
};
DEV f4u ld4(const float* p) { return *reinterpret_cast<const f4u*>(p); }

// pass 1: partial[b][p][slice][2][260] = sum over the slice's frames of cos / sin(phase_l - phase_r)
__global__ __launch_bounds__(256) void ipd_mean_tf_kernel(const float* __restrict__ re, const float* __restrict__ im, int C,
                                                          int T, Pairs pr, float* __restrict__ partial) {
  __shared__ float red[4][2][260];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sl = blockIdx.x, p = blockIdx.y, b = blockIdx.z, P = gridDim.y;
  const int per = (T + IPD_TS - 1) / IPD_TS, t_beg = sl * per, t_end = min(T, t_beg + per);
  const size_t cl = ((size_t)b * C + pr.l[p]) * T, cr = ((size_t)b * C + pr.r[p]) * T;
  float ac[4] = {0.f, 0.f, 0.f, 0.f}, as[4] = {0.f, 0.f, 0.f, 0.f}, tc = 0.f, ts = 0.f;
  for (int t = t_beg + w; t < t_end; t += 4) {
    const size_t ol = (cl + t) * NBIN + 4 * lane, orr = (cr + t) * NBIN + 4 * lane;
    const f4u rl = ld4(re + ol), il = ld4(im + ol), rr = ld4(re + orr), ir = ld4(im + orr);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float yr, yi;
      phase_diff(rl.v[e], il.v[e], rr.v[e], ir.v[e], yr, yi);
      ac[e] += yr;
      as[e] += yi;
    }
    if (lane == 0) {  // bin 256
      const size_t ql = (cl + t) * NBIN + 256, qr = (cr + t) * NBIN + 256;
      float yr, yi;
      phase_diff(re[ql], im[ql], re[qr], im[qr], yr, yi);
      tc += yr;
      ts += yi;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[w][0][4 * lane + e] = ac[e];
    red[w][1][4 * lane + e] = as[e];
  }
  if (lane == 0) {
    red[w][0][256] = tc;
    red[w][1][256] = ts;
  }
  __syncthreads();
  float* dst = partial + (((size_t)b * P + p) * IPD_TS + sl) * 2 * 260;
  for (int i = threadIdx.x; i < 2 * NBIN; i += 256) {
    const int row = i / NBIN, f = i - row * NBIN;
    dst[row * 260 + f] = (red[0][row][f] + red[1][row][f]) + (red[2][row][f] + red[3][row][f]);
  }
}

// pass 2: out[b, t, f_off + p*257 + f] = atan2(sin - mean sin, cos - mean cos); workgroup = (frames, pair, batch)
__global__ __launch_bounds__(256) void ipd_out_tf_kernel(const float* __restrict__ re, const float* __restrict__ im, int C,
                                                         int T, Pairs pr, const float* __restrict__ partial,
                                                         float* __restrict__ out, int64_t ld_out, int64_t f_off,
                                                         int frames_per_wg) {
  __shared__ float mean[2][260];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int p = blockIdx.y, b = blockIdx.z, P = gridDim.y, t0 = blockIdx.x * frames_per_wg;
  const float inv_t = 1.f / (float)T;
  for (int i = threadIdx.x; i < 2 * NBIN; i += 256) {
    const int row = i / NBIN, f = i - row * NBIN;
    float s = 0.f;
#pragma unroll
    for (int sl = 0; sl < IPD_TS; ++sl) s += partial[(((size_t)b * P + p) * IPD_TS + sl) * 2 * 260 + row * 260 + f];
    mean[row][f] = s * inv_t;
  }
  __syncthreads();
  const size_t cl = ((size_t)b * C + pr.l[p]) * T, cr = ((size_t)b * C + pr.r[p]) * T;
  float mc[4], ms[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mc[e] = mean[0][4 * lane + e];
    ms[e] = mean[1][4 * lane + e];
  }
  const float mc256 = mean[0][256], ms256 = mean[1][256];
  for (int t = t0 + w; t < min(T, t0 + frames_per_wg); t += 4) {
    const size_t ol = (cl + t) * NBIN + 4 * lane, orr = (cr + t) * NBIN + 4 * lane;
    const f4u rl = ld4(re + ol), il = ld4(im + ol), rr = ld4(re + orr), ir = ld4(im + orr);
    float* orow = out + ((size_t)b * T + t) * ld_out + f_off + (size_t)p * NBIN;
    f4u o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float yr, yi;
      phase_diff(rl.v[e], il.v[e], rr.v[e], ir.v[e], yr, yi);
      o.v[e] = atan2f(yi - ms[e], yr - mc[e]);
    }
    *reinterpret_cast<f4u*>(orow + 4 * lane) = o;
    if (lane == 0) {
      const size_t ql = (cl + t) * NBIN + 256, qr = (cr + t) * NBIN + 256;
      float yr, yi;
      phase_diff(re[ql], im[ql], re[qr], im[qr], yr, yi);
      orow[256] = atan2f(yi - ms256, yr - mc256);
    }
  }
}

// ------------------------------------------------------------------------------------------ iSTFT
// One wave per frame: 512-point inverse real FFT as conj(FFT256(conj(Z)))/256 with
//   Z[k] = Xe[k] + i Xo[k],  Xe = (X[k] + conj X[256-k])/2,  Xo = (X[k] - conj X[256-k])/2 * e^{+2 pi i k/512},
// using the same radix-4 wave FFT as the forward STFT (fft.hpp).  A workgroup
// owns output samples of FBI consecutive hops; it synthesises FBI+1 windowed frames into LDS, overlap-adds
// and divides by the Hann^2 envelope (torch.istft semantics, center=True: 256 samples trimmed on the left).
constexpr int FBI = 16;

__global__ __launch_bounds__(256) void istft_kernel(const float* __restrict__ re, const float* __restrict__ im, int T,
                                                    int64_t length, float* __restrict__ wav) {
  __shared__ float sre[NBIN][FBI + 2], sim[NBIN][FBI + 2];  // spectra of frames t0-1 .. t0+FBI-1, [bin][frame]
  __shared__ __attribute__((aligned(16))) float frames[FBI + 1][512];  // windowed time frames
  __shared__ __attribute__((aligned(16))) cplx sz[4][FFT_SCRATCH];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int64_t s = blockIdx.y;
  const int t0 = blockIdx.x * FBI;  // hop index: this workgroup writes padded samples [256*t0, 256*(t0+FBI))

  for (int i = tid; i < NBIN * (FBI + 1); i += 256) {
    const int k = i / (FBI + 1), j = i % (FBI + 1);
    const int t = t0 - 1 + j;
    const bool ok = t >= 0 && t < T;
    sre[k][j] = ok ? re[((size_t)s * NBIN + k) * T + t] : 0.f;
    sim[k][j] = ok ? im[((size_t)s * NBIN + k) * T + t] : 0.f;
  }
  cplx tw1[4], tws[6], twp[4];
  float win[8];
  fft_load_consts(1, l, win, tw1, tws, twp);
  const int ko = fft_out_index(l);
  __syncthreads();

  for (int j = w; j < FBI + 1; j += 4) {
    const int t = t0 - 1 + j;
    if (t < 0 || t >= T) {  // frame outside the signal contributes nothing
      for (int i = l; i < 512; i += 64) frames[j][i] = 0.f;
      continue;
    }
    // conj(Z[k]) for k = l + 64q
    cplx z[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = l + 64 * q, kc = 256 - k;  // kc in 1..256
      const cplx A = {sre[k][j], sim[k][j]}, Bc = {sre[kc][j], -sim[kc][j]};
      const cplx E = {0.5f * (A.x + Bc.x), 0.5f * (A.y + Bc.y)};
      const cplx O = cmul(cplx{0.5f * (A.x - Bc.x), 0.5f * (A.y - Bc.y)}, twp[q]);
      // Z = E + i O ;  conj(Z) = (E.x - O.y, -(E.y + O.x))
      z[q] = {E.x - O.y, -(E.y + O.x)};
    }
    cplx u[4];
    fft256_wave(z, u, tw1, tws, sz[w], l);
    // lane l, reg r: FFT(conj Z)[n], n = ko + 64 r ;  z[n] = conj(.)/256 ;  x[2n] = Re, x[2n+1] = Im
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float2*>(&frames[j][2 * (ko + 64 * r)]) =
          make_float2(u[r].x * (1.f / 256.f) * win[2 * r], -u[r].y * (1.f / 256.f) * win[2 * r + 1]);
  }
  __syncthreads();
  // overlap-add: padded sample p = 256*(t0 + h) + i gets frame (t0+h) [i] and frame (t0+h-1) [i + 256]
  for (int idx = tid; idx < FBI * 256; idx += 256) {
    const int h = idx >> 8, i = idx & 255;
    const int64_t n = (int64_t)256 * (t0 + h) + i - 256;  // output index after trimming the centre padding
    if (n < 0 || n >= length) continue;
    const int tc = t0 + h, tp = tc - 1;
    float acc = 0.f, env = 0.f;
    const float w0 = 0.5f - 0.5f * cospif((float)i * (1.f / 256.f));
    const float w1 = 0.5f - 0.5f * cospif((float)(i + 256) * (1.f / 256.f));
    if (tc < T) { acc += frames[h + 1][i]; env += w0 * w0; }
    if (tp >= 0 && tp < T) { acc += frames[h][i + 256]; env += w1 * w1; }
    wav[(size_t)s * length + n] = acc / env;
  }
}

}  // namespace

extern "C" int csship_ipd(const float* re, const float* im, int64_t B, int64_t C, int64_t T, const int32_t* pairs_l,
                          const int32_t* pairs_r, int32_t n_pairs, float* out, int64_t out_batch_stride, int64_t f_off,
                          float* workspace, csship_stream_t stream) {
  if (!re || !im || !pairs_l || !pairs_r || !out || !workspace) return CSSHIP_ENULL;
  if (B <= 0 || C <= 0 || T <= 0 || n_pairs <= 0 || B * n_pairs > 65535 || (T + 31) / 32 > 65535) return CSSHIP_EBADSHAPE;
  if (n_pairs > 16) return CSSHIP_EUNSUPPORTED;
  Pairs pr{};
  for (int i = 0; i < n_pairs; ++i) {
    if (pairs_l[i] < 0 || pairs_l[i] >= C || pairs_r[i] < 0 || pairs_r[i] >= C) return CSSHIP_EBADSHAPE;
    pr.l[i] = pairs_l[i];
    pr.r[i] = pairs_r[i];
  }
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(ipd_mean_kernel, dim3((NBIN + 63) / 64, n_pairs, (unsigned)B), dim3(512), 0, s, re, im, (int)C,
                     (int)T, pr, workspace);
  CSSHIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(ipd_out_kernel, dim3((NBIN + 31) / 32, (unsigned)((T + 31) / 32), (unsigned)(B * n_pairs)),
                     dim3(256), 0, s, re, im, (int)C, (int)T, pr, n_pairs, workspace, out, out_batch_stride, f_off);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" size_t csship_ipd_tf_workspace(int64_t B, int32_t n_pairs) {
  return (size_t)B * (size_t)n_pairs * IPD_TS * 2 * 260 * sizeof(float);
}

extern "C" int csship_ipd_tf(const float* re, const float* im, int64_t B, int64_t C, int64_t T, const int32_t* pairs_l,
                             const int32_t* pairs_r, int32_t n_pairs, float* out, int64_t ld_out, int64_t f_off,
                             float* workspace, size_t workspace_bytes, csship_stream_t stream) {
  if (!re || !im || !pairs_l || !pairs_r || !out || !workspace) return CSSHIP_ENULL;
  if (B <= 0 || C <= 0 || T <= 0 || n_pairs <= 0 || B > 65535) return CSSHIP_EBADSHAPE;
  if (n_pairs > 16) return CSSHIP_EUNSUPPORTED;
  if (ld_out < f_off + (int64_t)n_pairs * NBIN) return CSSHIP_EBADSHAPE;
  if (workspace_bytes < csship_ipd_tf_workspace(B, n_pairs)) return CSSHIP_EWORKSPACE;
  Pairs pr{};
  for (int i = 0; i < n_pairs; ++i) {
    if (pairs_l[i] < 0 || pairs_l[i] >= C || pairs_r[i] < 0 || pairs_r[i] >= C) return CSSHIP_EBADSHAPE;
    pr.l[i] = pairs_l[i];
    pr.r[i] = pairs_r[i];
  }
  hipStream_t s = as_stream(stream);
  const int fpw = 32;  // frames per workgroup in pass 2
  hipLaunchKernelGGL(ipd_mean_tf_kernel, dim3(IPD_TS, n_pairs, (unsigned)B), dim3(256), 0, s, re, im, (int)C, (int)T, pr,
                     workspace);
  CSSHIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(ipd_out_tf_kernel, dim3((unsigned)((T + fpw - 1) / fpw), n_pairs, (unsigned)B), dim3(256), 0, s, re,
                     im, (int)C, (int)T, pr, workspace, out, ld_out, f_off, fpw);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_istft(const float* re, const float* im, int64_t S, int64_t T, int64_t length, float* wav,
                            csship_stream_t stream) {
  if (!re || !im || !wav) return CSSHIP_ENULL;
  if (S <= 0 || T <= 0 || length <= 0 || length > 256 * (T - 1) || S > 65535) return CSSHIP_EBADSHAPE;
  // padded output index range [256, 256 + length): hops 1 .. (255 + length)/256
  const int64_t nhop = (256 + length + 255) / 256;
  hipLaunchKernelGGL(istft_kernel, dim3((unsigned)((nhop + FBI - 1) / FBI), (unsigned)S), dim3(256), 0, as_stream(stream),
                     re, im, (int)T, length, wav);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
