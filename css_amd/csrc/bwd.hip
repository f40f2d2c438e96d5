// Row kernels of the backward pass: LayerNorm backward (+ optional fused ReLU backward), ReLU backward,
// MVN materialisation for the embed weight gradient, and the relative-position softmax forward/backward that
// brackets the attention GEMMs of bgemm.hip.  All HBM/L2-bound, one wave per row.
#include "common.hpp"

namespace {

// ---- LayerNorm backward over D = 256*VPL.  y = (x - mu) * rstd * gamma + beta  [-> ReLU if relu]
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma   (+ dx_add)
//   dgamma = sum_rows dy * xhat,  dbeta = sum_rows dy   -> per-workgroup partials, summed by launch_slab_reduce
template <int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* dx_add, int relu, int64_t M, int rows_per_wg,
                                                     float* dx, float* __restrict__ part,
                                                     unsigned short* __restrict__ ybf = nullptr) {
  constexpr int D = 256 * VPL;
  __shared__ float red[4][2 * D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
  float4 gm[VPL], bt[VPL], ag[VPL], ab[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    gm[i] = *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 4);
    bt[i] = *reinterpret_cast<const float4*>(beta + (i * 64 + lane) * 4);
    ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row = r0 + w; row < r1; row += 4) {
    float4 v[VPL], d[VPL], addv[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i] = *reinterpret_cast<const float4*>(x + row * D + (i * 64 + lane) * 4);
      d[i] = *reinterpret_cast<const float4*>(dy + row * D + (i * 64 + lane) * 4);
      // the residual gradient with the row, not behind its four reductions (dx may alias dx_add: this row is read before it is written)
      addv[i] = dx_add ? *reinterpret_cast<const float4*>(dx_add + row * D + (i * 64 + lane) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
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
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;  // xhat
      if (ybf != nullptr)  // the LayerNorm's output itself, bf16: the second operand of the weight gradient behind it
        *reinterpret_cast<uint2*>(ybf + row * D + (i * 64 + lane) * 4) =
            make_uint2(pack_bf2(v[i].x * gm[i].x + bt[i].x, v[i].y * gm[i].y + bt[i].y),
                       pack_bf2(v[i].z * gm[i].z + bt[i].z, v[i].w * gm[i].w + bt[i].w));
      if (relu) {
        if (!(v[i].x * gm[i].x + bt[i].x > 0.f)) d[i].x = 0.f;
        if (!(v[i].y * gm[i].y + bt[i].y > 0.f)) d[i].y = 0.f;
        if (!(v[i].z * gm[i].z + bt[i].z > 0.f)) d[i].z = 0.f;
        if (!(v[i].w * gm[i].w + bt[i].w > 0.f)) d[i].w = 0.f;
      }
      ag[i].x += d[i].x * v[i].x; ag[i].y += d[i].y * v[i].y; ag[i].z += d[i].z * v[i].z; ag[i].w += d[i].w * v[i].w;
      ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
      d[i].x *= gm[i].x; d[i].y *= gm[i].y; d[i].z *= gm[i].z; d[i].w *= gm[i].w;  // g
      s1 += d[i].x + d[i].y + d[i].z + d[i].w;
      s2 += d[i].x * v[i].x + d[i].y * v[i].y + d[i].z * v[i].z + d[i].w * v[i].w;
    }
    s1 = wave_sum(s1) * (1.f / D);
    s2 = wave_sum(s2) * (1.f / D);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int64_t o = row * D + (i * 64 + lane) * 4;
      float4 r;
      r.x = rstd * (d[i].x - s1 - v[i].x * s2);
      r.y = rstd * (d[i].y - s1 - v[i].y * s2);
      r.z = rstd * (d[i].z - s1 - v[i].z * s2);
      r.w = rstd * (d[i].w - s1 - v[i].w * s2);
      if (dx_add) {
        const float4 a = addv[i];
        r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
      }
      *reinterpret_cast<float4*>(dx + o) = r;
    }
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    *reinterpret_cast<float4*>(&red[w][(i * 64 + lane) * 4]) = ag[i];
    *reinterpret_cast<float4*>(&red[w][D + (i * 64 + lane) * 4]) = ab[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256)
    part[(int64_t)blockIdx.x * 2 * D + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// ---- the same for any D <= 1024 (BLSTM widths): lane owns columns lane, lane + 64, ...; scalar accesses
__global__ __launch_bounds__(256) void ln_bwd_any_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* dx_add, int relu, int64_t M, int D, int rows_per_wg,
                                                         float* dx, float* __restrict__ part) {
  constexpr int MAXC = 16;
  extern __shared__ float red_any[];  // [4][2 * D]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
  float ag[MAXC], ab[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) ag[i] = ab[i] = 0.f;
  for (int64_t row = r0 + w; row < r1; row += 4) {
    const float* xr = x + row * D;
    const float* dr = dy + row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float d = xr[c] - mean;
      ss += d * d;
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = i * 64 + lane;
      if (c < D) {
        const float xh = (xr[c] - mean) * rstd;
        float d = dr[c];
        if (relu && !(xh * gamma[c] + beta[c] > 0.f)) d = 0.f;
        ag[i] += d * xh;
        ab[i] += d;
        const float gg = d * gamma[c];
        s1 += gg;
        s2 += gg * xh;
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    for (int c = lane; c < D; c += 64) {
      const float xh = (xr[c] - mean) * rstd;
      float d = dr[c];
      if (relu && !(xh * gamma[c] + beta[c] > 0.f)) d = 0.f;
      float r = rstd * (d * gamma[c] - s1 - xh * s2);
      if (dx_add) r += dx_add[row * D + c];
      dx[row * D + c] = r;
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = i * 64 + lane;
    if (c < D) {
      red_any[w * 2 * D + c] = ag[i];
      red_any[w * 2 * D + D + c] = ab[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256)
    part[(int64_t)blockIdx.x * 2 * D + c] = red_any[c] + red_any[2 * D + c] + red_any[4 * D + c] + red_any[6 * D + c];
}

// ---- dz = dy where y > 0 else 0 (2-D, row strides); workgroup = 256 columns of one row
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, int64_t ldd,
                                                       const float* __restrict__ y, int64_t ldy, int N, int cblocks,
                                                       float* __restrict__ out, int64_t ldo) {
  const int64_t row = blockIdx.x / cblocks;
  const int c = (blockIdx.x % cblocks) * 256 + threadIdx.x;
  if (c < N) out[row * ldo + c] = y[row * ldy + c] > 0.f ? dy[row * ldd + c] : 0.f;
}

// ---- the same into bf16 rows zero padded to ldo (even) columns; one wave per row, a column pair per lane and pass
__global__ __launch_bounds__(256) void relu_bwd_bf16_kernel(const float* __restrict__ dy, int64_t ldd,
                                                            const float* __restrict__ y, int64_t ldy, int64_t M, int N,
                                                            unsigned short* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* dr = dy + row * ldd;
  const float* yr = y + row * ldy;
  unsigned* orow = reinterpret_cast<unsigned*>(out + row * (int64_t)ldo);
  for (int c = 2 * lane; c < ldo; c += 128) {
    float v0 = 0.f, v1 = 0.f;
    if (c < N) v0 = yr[c] > 0.f ? dr[c] : 0.f;
    if (c + 1 < N) v1 = yr[c + 1] > 0.f ? dr[c + 1] : 0.f;
    orow[c >> 1] = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
  }
}

// ---- out_bf16[b*T + t, f] = (max(x, clamp) - mean[b, f]) * rinv[b, f], zero padded to ldo columns
__global__ __launch_bounds__(256) void mvn_apply_kernel(const float* __restrict__ x, int T, int F, int64_t ldx,
                                                        float clamp_min, const float* __restrict__ mean,
                                                        const float* __restrict__ rinv,
                                                        unsigned short* __restrict__ out, int ldo, int cblocks) {
  const int64_t row = blockIdx.x / cblocks;
  const int b = (int)(row / T);
  const int c = (blockIdx.x % cblocks) * 256 + threadIdx.x;
  if (c >= ldo) return;
  float v = 0.f;
  if (c < F) v = (fmaxf(x[row * ldx + c], clamp_min) - mean[(size_t)b * F + c]) * rinv[(size_t)b * F + c];
  out[row * ldo + c] = f2bf(v);
}

// ---- P[z, i, j] = softmax_j(S[z, i, j] + R[z, i, i - j + T - 1]); one wave per row; bf16 out, zero padded to ldp
__global__ __launch_bounds__(256) void relpos_softmax_kernel(const float* __restrict__ sqk, int64_t lds,
                                                             const float* __restrict__ rel, int64_t ldr, int64_t nrows,
                                                             int T, unsigned short* __restrict__ p, int64_t ldp) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // z*T + i
  if (row >= nrows) return;
  const int i = (int)(row % T);
  const float* s = sqk + row * lds;
  const float* r = rel + row * ldr + i + T - 1;
  float mx = -INFINITY;
  for (int j = lane; j < T; j += 64) mx = fmaxf(mx, s[j] + r[-j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) sum += __expf(s[j] + r[-j] - mx);
  const float inv = 1.f / wave_sum(sum);
  for (int j = lane; j < ldp; j += 64) p[row * ldp + j] = j < T ? f2bf(__expf(s[j] + r[-j] - mx) * inv) : (unsigned short)0;
}

// ---- dS = P o (dP - sum_j P dP); also its skewed copy dSk[b, i, h, r] = dS[b, h, i, i + T - 1 - r] (0 outside)
__global__ __launch_bounds__(256) void relpos_softmax_bwd_kernel(const unsigned short* __restrict__ p, int64_t ldp,
                                                                 const float* __restrict__ dp, int64_t ldd, int B, int H,
                                                                 int T, unsigned short* __restrict__ ds,
                                                                 unsigned short* __restrict__ dsk, int W) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b*H + h)*T + i
  if (row >= (int64_t)B * H * T) return;
  const int i = (int)(row % T), h = (int)((row / T) % H), b = (int)(row / ((int64_t)T * H));
  const unsigned short* pr = p + row * ldp;
  const float* dr = dp + row * ldd;
  float dot = 0.f;
  for (int j = lane; j < T; j += 64) dot += bf2f(pr[j]) * dr[j];
  dot = wave_sum(dot);
  for (int j = lane; j < ldp; j += 64) ds[row * ldp + j] = j < T ? f2bf(bf2f(pr[j]) * (dr[j] - dot)) : (unsigned short)0;
  unsigned short* sk = dsk + (((int64_t)b * T + i) * H + h) * W;
  for (int r = lane; r < W; r += 64) {
    const int j = i + T - 1 - r;
    sk[r] = (j >= 0 && j < T) ? f2bf(bf2f(pr[j]) * (dr[j] - dot)) : (unsigned short)0;
  }
}

// ---- D[b,h,i] = sum_d dO[b,i,h,d] * O[b,i,h,d]; 8 lanes per (b,i,h), 16-byte loads
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const unsigned short* __restrict__ dO,
                                                          const unsigned short* __restrict__ O, int B, int T, int H,
                                                          float* __restrict__ d) {
  const int64_t idx = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);  // (b*T + i)*H + h
  const int part = threadIdx.x & 7;
  const int64_t n = (int64_t)B * T * H;
  const int64_t c = min(idx, n - 1);
  const uint4 a = *reinterpret_cast<const uint4*>(dO + c * 64 + part * 8);
  const uint4 b = *reinterpret_cast<const uint4*>(O + c * 64 + part * 8);
  const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    s += bf2f((unsigned short)(av[q] & 0xffff)) * bf2f((unsigned short)(bv[q] & 0xffff)) +
         bf2f((unsigned short)(av[q] >> 16)) * bf2f((unsigned short)(bv[q] >> 16));
  s = sum8(s);
  if (part == 0 && idx < n) {
    const int h = (int)(idx % H);
    const int64_t bi = idx / H;
    d[((bi / T) * H + h) * T + bi % T] = s;
  }
}

// ---- weight shadows: f32 [R, K] (row stride lds) -> bf16 row-major [.., ld_rm] and/or MFMA fragment order, rows
// scaled by `scale`; one thread per 8 consecutive k.  Fragment layout as in include/csship.h:
// block(rt, ks)[16*g + m][j] = W[16*rt + m][32*ks + 8*g + j], rows beyond R are zero.
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, int R, int K, int64_t lds,
                                                          float scale, unsigned short* __restrict__ rm, int64_t ld_rm,
                                                          unsigned short* __restrict__ frag, int row0, int Kp) {
  const int kc = Kp / 8;  // chunks per (padded) row
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int Rp = (R + 15) / 16 * 16;
  if (i >= (int64_t)Rp * kc) return;
  const int r = (int)(i / kc), k = (int)(i % kc) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (r < R && k + e < K) ? w[(int64_t)r * lds + k + e] * scale : 0.f;
  const uint4 o = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
  if (rm && r < R && k < ld_rm) *reinterpret_cast<uint4*>(rm + (int64_t)(row0 + r) * ld_rm + k) = o;
  if (frag) {
    const int rr = row0 + r;
    const int64_t blk = (int64_t)(rr / 16) * (Kp / 32) + k / 32;
    *reinterpret_cast<uint4*>(frag + (blk * 64 + ((k % 32) / 8) * 16 + (rr % 16)) * 8) = o;
  }
}

// The same for a whole table of weights in ONE launch (blockIdx.y = job): a training step re-packs every Linear
// weight of the model (12 per Conformer block) and a launch each costs more than the packing itself.  A job may read
// its source transposed (logical W[r][k] = w[k * ld + r], e.g. the [K, R] view of a Linear weight whose transpose
// the FeedForward backward kernel wants fragment-packed): consecutive threads then walk r, so the reads stay
// coalesced and the 16-row fragment groups are written as contiguous 256-byte pieces.
__global__ __launch_bounds__(256) void pack_multi_kernel(const csship_pack_job* __restrict__ jobs) {
  const csship_pack_job j = jobs[blockIdx.y];
  const int kc = j.Kp / 8;
  const int Rp = (j.R + 15) / 16 * 16;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)Rp * kc) return;
  int r, k;
  if (j.transposed) {
    r = (int)(i % Rp);
    k = (int)(i / Rp) * 8;
  } else {
    r = (int)(i / kc);
    k = (int)(i % kc) * 8;
  }
  float v[8];
  if (!j.transposed && r < j.R && k + 8 <= j.K && ((j.ld_w & 3) | ((uintptr_t)j.w & 15)) == 0) {
    // whole 8-float group of a row-major source: two 16-byte loads instead of the eight predicated scalar loads below
    // (the 258-job re-pack of a Conformer-base training step: 58.4 -> 40.0 us, tools/probes/pack_time.py)
    const float4 a = *reinterpret_cast<const float4*>(j.w + (int64_t)r * j.ld_w + k);
    const float4 b = *reinterpret_cast<const float4*>(j.w + (int64_t)r * j.ld_w + k + 4);
    v[0] = a.x * j.scale, v[1] = a.y * j.scale, v[2] = a.z * j.scale, v[3] = a.w * j.scale;
    v[4] = b.x * j.scale, v[5] = b.y * j.scale, v[6] = b.z * j.scale, v[7] = b.w * j.scale;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool ok = r < j.R && k + e < j.K;
      const int64_t src = j.transposed ? (int64_t)(k + e) * j.ld_w + r : (int64_t)r * j.ld_w + k + e;
      v[e] = ok ? j.w[src] * j.scale : 0.f;
    }
  }
  const uint4 o = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
  if (j.row_major && r < j.R && k < j.ld_row_major)
    *reinterpret_cast<uint4*>(j.row_major + (int64_t)(j.dst_row0 + r) * j.ld_row_major + k) = o;
  if (j.frag) {
    const int rr = j.dst_row0 + r;
    const int kpd = j.Kp_dst > 0 ? j.Kp_dst : j.Kp;   // k-slice [k0, k0 + Kp) of a wider fragment table
    const int64_t blk = (int64_t)(rr / 16) * (kpd / 32) + (j.k0 + k) / 32;
    *reinterpret_cast<uint4*>(j.frag + (blk * 64 + ((k % 32) / 8) * 16 + (rr % 16)) * 8) = o;
  }
}

}  // namespace

extern "C" int csship_pack_weight(const float* w, int32_t R, int32_t K, int64_t ld_w, float scale,
                                  uint16_t* row_major_bf16, int64_t ld_row_major, uint16_t* frag_packed_bf16,
                                  int32_t dst_row0, csship_stream_t stream) {
  if (!w || (!row_major_bf16 && !frag_packed_bf16)) return CSSHIP_ENULL;
  if (R <= 0 || K <= 0 || ld_w < K || dst_row0 < 0) return CSSHIP_EBADSHAPE;
  const int Kp = frag_packed_bf16 ? (K + 31) / 32 * 32 : (K + 7) / 8 * 8;
  if (row_major_bf16 && (ld_row_major < Kp || (ld_row_major % 8) != 0 || !aligned16(row_major_bf16))) return CSSHIP_EALIGN;
  if (frag_packed_bf16 && ((dst_row0 % 16) != 0 || !aligned16(frag_packed_bf16))) return CSSHIP_EALIGN;
  const int64_t n = (int64_t)((R + 15) / 16 * 16) * (Kp / 8);
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), w, R, K, ld_w,
                     scale, row_major_bf16, ld_row_major, frag_packed_bf16, dst_row0, Kp);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_pack_weights_multi(const csship_pack_job* jobs_device, int32_t n_jobs, int64_t max_chunks,
                                         csship_stream_t stream) {
  if (!jobs_device) return CSSHIP_ENULL;
  if (n_jobs <= 0 || n_jobs > 65535 || max_chunks <= 0) return CSSHIP_EBADSHAPE;
  hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)((max_chunks + 255) / 256), (unsigned)n_jobs), dim3(256), 0,
                     as_stream(stream), jobs_device);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

// rows per workgroup of the LayerNorm backward: a row is a serial chain of four wave reductions, so the kernel wants
// many waves in flight (16 rows = 4 per wave: 3 waves per SIMD at 12 800 rows), not long row loops
static constexpr int LNB_ROWS = 16;

extern "C" size_t csship_layernorm_bwd_workspace(int64_t M, int32_t D) {
  return (size_t)((M + LNB_ROWS - 1) / LNB_ROWS) * 2 * D * sizeof(float);
}

extern "C" int csship_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* beta,
                                    const float* dx_add, int32_t relu, int64_t M, int32_t D, float* dx,
                                    float* dgamma_dbeta, void* workspace, size_t workspace_bytes,
                                    csship_stream_t stream) {
  return csship_layernorm_bwd_y(x, dy, gamma, beta, dx_add, relu, M, D, dx, dgamma_dbeta, nullptr, workspace, workspace_bytes,
                                stream);
}

extern "C" int csship_layernorm_bwd_y(const float* x, const float* dy, const float* gamma, const float* beta,
                                      const float* dx_add, int32_t relu, int64_t M, int32_t D, float* dx,
                                      float* dgamma_dbeta, uint16_t* y_bf16, void* workspace, size_t workspace_bytes,
                                      csship_stream_t stream) {
  if (!x || !dy || !gamma || !beta || !dx || !dgamma_dbeta || !workspace) return CSSHIP_ENULL;
  if (y_bf16 && (!(D == 256 || D == 512 || D == 1024) || !aligned8(y_bf16))) return CSSHIP_EUNSUPPORTED;
  if (M <= 0) return CSSHIP_EBADSHAPE;
  if (D <= 0 || D > 1024) return CSSHIP_EUNSUPPORTED;
  const bool fast = D == 256 || D == 512 || D == 1024;
  if (fast && (!aligned16(x) || !aligned16(dy) || !aligned16(dx) || !aligned16(gamma) || !aligned16(beta) ||
               (dx_add && !aligned16(dx_add))))
    return CSSHIP_EALIGN;
  const int nwg = (int)((M + LNB_ROWS - 1) / LNB_ROWS);
  if (workspace_bytes < (size_t)nwg * 2 * D * sizeof(float)) return CSSHIP_EWORKSPACE;
  hipStream_t s = as_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
  if (!fast)
    hipLaunchKernelGGL(ln_bwd_any_kernel, dim3(nwg), dim3(256), (size_t)8 * D * sizeof(float), s, x, dy, gamma, beta, dx_add,
                       relu, M, D, LNB_ROWS, dx, part);
  else if (D == 256)
    hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(nwg), dim3(256), 0, s, x, dy, gamma, beta, dx_add, relu, M, LNB_ROWS, dx, part, y_bf16);
  else if (D == 512)
    hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(nwg), dim3(256), 0, s, x, dy, gamma, beta, dx_add, relu, M, LNB_ROWS, dx, part, y_bf16);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(nwg), dim3(256), 0, s, x, dy, gamma, beta, dx_add, relu, M, LNB_ROWS, dx, part, y_bf16);
  CSSHIP_CHECK_LAUNCH();
  return launch_slab_reduce(part, nwg, 2 * (int64_t)D, 2 * D, 2 * (int64_t)D, 1.f, nullptr, nullptr, 0, dgamma_dbeta, s, true);
}

extern "C" int csship_relu_bwd(const float* dy, int64_t ld_dy, const float* y, int64_t ld_y, int64_t M, int32_t N,
                               float* out, int64_t ld_out, csship_stream_t stream) {
  if (!dy || !y || !out) return CSSHIP_ENULL;
  const int cblocks = (N + 255) / 256;
  if (M <= 0 || N <= 0 || ld_dy < N || ld_y < N || ld_out < N || M * cblocks > 0x7fffffff) return CSSHIP_EBADSHAPE;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)(M * cblocks)), dim3(256), 0, as_stream(stream), dy, ld_dy, y, ld_y,
                     N, cblocks, out, ld_out);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_relu_bwd_bf16(const float* dy, int64_t ld_dy, const float* y, int64_t ld_y, int64_t M, int32_t N,
                                    uint16_t* out_bf16, int64_t ld_out, csship_stream_t stream) {
  if (!dy || !y || !out_bf16) return CSSHIP_ENULL;
  if (M <= 0 || N <= 0 || ld_dy < N || ld_y < N || ld_out < N || (ld_out & 1) || ld_out > 0x7fffffff ||
      (M + 3) / 4 > 0x7fffffff || (reinterpret_cast<uintptr_t>(out_bf16) & 3))
    return CSSHIP_EBADSHAPE;
  hipLaunchKernelGGL(relu_bwd_bf16_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, as_stream(stream), dy, ld_dy, y,
                     ld_y, M, N, out_bf16, (int)ld_out);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_mvn_apply(const float* x, int64_t B, int64_t T, int64_t F, int64_t ldx, float clamp_min,
                                const float* mean, const float* rinv, uint16_t* out_bf16, int64_t ld_out,
                                csship_stream_t stream) {
  if (!x || !mean || !rinv || !out_bf16) return CSSHIP_ENULL;
  const int cblocks = (int)((ld_out + 255) / 256);
  if (B <= 0 || T <= 0 || F <= 0 || ldx < F || ld_out < F || B * T * cblocks > 0x7fffffff) return CSSHIP_EBADSHAPE;
  hipLaunchKernelGGL(mvn_apply_kernel, dim3((unsigned)(B * T * cblocks)), dim3(256), 0, as_stream(stream), x, (int)T,
                     (int)F, ldx, clamp_min, mean, rinv, out_bf16, (int)ld_out, cblocks);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_relpos_softmax(const float* s_qk, int64_t ld_s, const float* rel, int64_t ld_rel, int32_t B,
                                     int32_t H, int32_t T, uint16_t* p_bf16, int64_t ld_p, csship_stream_t stream) {
  if (!s_qk || !rel || !p_bf16) return CSSHIP_ENULL;
  if (B <= 0 || H <= 0 || T <= 0 || ld_s < T || ld_rel < 2 * T - 1 || ld_p < T) return CSSHIP_EBADSHAPE;
  const int64_t nrows = (int64_t)B * H * T;
  hipLaunchKernelGGL(relpos_softmax_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, as_stream(stream), s_qk,
                     ld_s, rel, ld_rel, nrows, T, p_bf16, ld_p);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_attn_rowdot(const uint16_t* d_out, const uint16_t* out, int32_t B, int32_t T, int32_t H, float* d,
                                  csship_stream_t stream) {
  if (!d_out || !out || !d) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0) return CSSHIP_EBADSHAPE;
  if (!aligned16(d_out) || !aligned16(out)) return CSSHIP_EALIGN;
  const int64_t n = (int64_t)B * T * H;
  hipLaunchKernelGGL(attn_rowdot_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, as_stream(stream), d_out, out, B, T, H,
                     d);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_relpos_softmax_bwd(const uint16_t* p_bf16, int64_t ld_p, const float* dp, int64_t ld_dp,
                                         int32_t B, int32_t H, int32_t T, uint16_t* ds_bf16, uint16_t* ds_skew_bf16,
                                         int32_t W, csship_stream_t stream) {
  if (!p_bf16 || !dp || !ds_bf16 || !ds_skew_bf16) return CSSHIP_ENULL;
  if (B <= 0 || H <= 0 || T <= 0 || ld_p < T || ld_dp < T || W < 2 * T - 1) return CSSHIP_EBADSHAPE;
  const int64_t nrows = (int64_t)B * H * T;
  hipLaunchKernelGGL(relpos_softmax_bwd_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, as_stream(stream),
                     p_bf16, ld_p, dp, ld_dp, B, H, T, ds_bf16, ds_skew_bf16, W);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
