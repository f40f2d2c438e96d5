// Flash-style relative-position attention for the Conformer (d_k = 64), gfx950.
//
// Replaces css/models/conformer.py:229-253 plus the [T,T,d_k] relative-position table
// of conformer.py:157-159,179-185:
//     score[i,j] = q_i.k_j + q_i.pe[clamp(i-j,-maxlen,maxlen-1)+maxlen]      (q pre-scaled)
//     out[i]     = softmax_j(score[i,:]) @ v
//
// One 256-thread workgroup = 4 waves per (b, h, 64 queries); wave w owns 16 queries.
//   * everything is computed TRANSPOSED so the softmax'd tile is already the MFMA B
//     operand of the P.V product (no LDS round trip for P):
//         S^T[j,i]  = K_tile[j,:]  . Q[i,:]      4 x (16 keys x 16 queries) per 64-key tile
//         Pb^T[c,i] = PE_band[c,:] . Q[i,:]      5 x (16 rel. offsets x 16 queries)
//         O^T[d,i] += V^T[d,j] P^T[j,i]          V^T fragments come from ds_read_b64_tr_b16
//   * the position term only depends on i-j: for a 16-query x 64-key block the 79 distinct
//     offsets form a band c = 15 - a + jl (a = query in block, jl = key in tile).  Because
//     query blocks and key tiles start at multiples of 16, every band tile is one ALIGNED,
//     REVERSED block of 16 consecutive pe rows; the host packs those blocks in MFMA fragment
//     order with the clamp baked in (css_amd.ops.pack_relpos), so the band operand is a
//     coalesced 1 KB global->VGPR load per fragment: no LDS staging, no per-element clamp.
//     The band product is skew-read from a per-wave fp32 scratch (the "rel-shift" without a
//     T x T x d_k table);
//   * K/V tiles are register-prefetched one tile ahead (issue-early / write-late), V is kept
//     in a 32-byte-group XOR swizzle that makes the transposed reads conflict free;
//   * online softmax over key tiles; key padding mask is always None on this path
//     (conformer.py:97), only the T tail of the last tile is masked.
#include "common.hpp"

namespace {

constexpr int PS = 84;  // per-wave band scratch row stride (floats): 336 B rows -> 8 consecutive rows hit 8 distinct 16-B slots
constexpr float LOG2E = 1.4426950408889634f;

DEV int v_off(int row, int chunk) {  // 16-byte chunk of a 128-byte V row; 32-byte groups XOR (row>>1)&3
  return row * 128 + ((((chunk >> 1) ^ ((row >> 1) & 3)) << 1) | (chunk & 1)) * 16;
}

// PROBS = false: the attention itself (optionally also the per-row log-sum-exp, for the backward pass).
// PROBS = true: no P.V; the normalised probabilities P[b,h,i,j] = exp(score - lse[b,h,i]) are written out as bf16
// [B,H,T,ldp] - the backward's recomputation of the softmax, with the same band machinery and no [T,T] f32 buffers.
template <bool PROBS>
__global__ __launch_bounds__(256, 4) void mha_relpos_kernel(const unsigned short* __restrict__ qkv,
                                                         const bf16x8* __restrict__ pe_packed, int q_min, int q_max,
                                                         unsigned short* __restrict__ out, int B, int T, int H,
                                                            unsigned long long* stamps, float* __restrict__ lse,
                                                            int64_t ldp) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sV[64 * 128];
  __shared__ __attribute__((aligned(16))) float sP[4 * 16 * PS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps block indices / addresses scalar
  const int g = lane >> 4, n = lane & 15;
  const int i0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int D = H * 64, D3 = 3 * D;
  const int i0w = i0 + 16 * w;
  const size_t base_b = (size_t)b * T;

  // Q fragments (MFMA B operand): lane (g,n) holds Q[i0w+n][ks*32 + 8g .. +7]
  bf16x8 qf[2];
  {
    const int qi = min(i0w + n, T - 1);
    const unsigned short* qrow = qkv + (base_b + qi) * D3 + h * 64;
    qf[0] = *reinterpret_cast<const bf16x8*>(qrow + 8 * g);
    qf[1] = *reinterpret_cast<const bf16x8*>(qrow + 32 + 8 * g);
  }

  // K/V staging: thread owns chunks (row = idx>>3, c = idx&7), idx = tid and tid + 256.  Named registers
  // (not arrays captured by a lambda: those are demoted to scratch memory and serialise the prefetch).
  // Per-lane base pointers are formed once; a key tile only adds a uniform offset (tail tile: clamped rows).
  const int srow0 = tid >> 3, srow1 = srow0 + 32, sc = tid & 7;
  const unsigned short* kv0 = qkv + (base_b + srow0) * D3 + h * 64 + sc * 8 + D;
  const unsigned short* kv1 = qkv + (base_b + srow1) * D3 + h * 64 + sc * 8 + D;
  uint4 kreg0, kreg1, vreg0, vreg1;
#define ATTN_LOAD_KV(J0)                                                                                  \
  {                                                                                                       \
    const unsigned short *s0_, *s1_;                                                                      \
    if ((J0) + 64 <= T) { /* uniform: whole tile inside the utterance */                                  \
      s0_ = kv0 + (size_t)(J0) * D3;                                                                      \
      s1_ = kv1 + (size_t)(J0) * D3;                                                                      \
    } else {                                                                                              \
      s0_ = qkv + (base_b + min((J0) + srow0, T - 1)) * D3 + h * 64 + sc * 8 + D;                         \
      s1_ = qkv + (base_b + min((J0) + srow1, T - 1)) * D3 + h * 64 + sc * 8 + D;                         \
    }                                                                                                     \
    kreg0 = *reinterpret_cast<const uint4*>(s0_);                                                         \
    kreg1 = *reinterpret_cast<const uint4*>(s1_);                                                         \
    if constexpr (!PROBS) {                                                                               \
      vreg0 = *reinterpret_cast<const uint4*>(s0_ + D);                                                   \
      vreg1 = *reinterpret_cast<const uint4*>(s1_ + D);                                                   \
    }                                                                                                     \
  }
#define ATTN_STORE_KV()                                                       \
  {                                                                           \
    *reinterpret_cast<uint4*>(sK + tile_off(srow0, sc)) = kreg0;              \
    *reinterpret_cast<uint4*>(sK + tile_off(srow1, sc)) = kreg1;              \
    if constexpr (!PROBS) {                                                   \
      *reinterpret_cast<uint4*>(sV + v_off(srow0, sc)) = vreg0;               \
      *reinterpret_cast<uint4*>(sV + v_off(srow1, sc)) = vreg1;               \
    }                                                                         \
  }
  // band fragments of key tile j0: band tile t covers rel = 16*q + 15 - m, q = (i0w - j0)/16 - t (scalar)
  bf16x8 pf0[5], pf1[5];
  const bf16x8* pe_lane = pe_packed + lane;
#define ATTN_LOAD_PE(PF, KSI, J0)                                                 \
  {                                                                               \
    const int qb_ = (i0w - (J0)) / 16; /* exact: both are multiples of 16 */      \
    _Pragma("unroll") for (int t = 0; t < 5; ++t) {                               \
      const int qi_ = min(max(qb_ - t, q_min), q_max) - q_min;                    \
      PF[t] = pe_lane[(size_t)(qi_ * 2 + (KSI)) * 64];                            \
    }                                                                             \
  }

  f32x4 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;
  float lse2 = 0.f;  // PROBS: this lane's query log-sum-exp in log2 units
  if constexpr (PROBS) lse2 = lse[((size_t)b * H + h) * T + min(i0w + n, T - 1)] * LOG2E;
  float* Pw = sP + w * 16 * PS;
  const float* Pr = Pw + n * PS + 15 - n + 4 * g;  // this lane's skewed read base: + 16t + r
  const int q4 = n >> 2, p4 = n & 3;

  stamp(stamps, 0);
  ATTN_LOAD_KV(0)
  ATTN_LOAD_PE(pf0, 0, 0)
  ATTN_STORE_KV()
  __syncthreads();

  stamp(stamps, 1);
  for (int j0 = 0; j0 < T; j0 += 64) {
    const bool more = j0 + 64 < T;
    ATTN_LOAD_PE(pf1, 1, j0)  // second half of this tile's band operands: lands during the first band MFMAs
    __builtin_amdgcn_sched_barrier(0);

    // ---- position band Pb^T = PE_band Q^T  -> per-wave scratch Pw[query n][c], c = 16t + 4g + r
    f32x4 pb[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) pb[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 5; ++t) pb[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf0[t], qf[0], pb[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 5; ++t) pb[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf1[t], qf[1], pb[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) ATTN_LOAD_KV(j0 + 64)  // next K/V tile: in flight during S, softmax and P.V; written to LDS after them
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 5; ++t) *reinterpret_cast<f32x4*>(Pw + n * PS + 16 * t + 4 * g) = pb[t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- S^T = bias (skew-read of the band, as the accumulator's initial value) + K Q^T
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = f32x4{Pr[16 * t], Pr[16 * t + 1], Pr[16 * t + 2], Pr[16 * t + 3]};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sK + tile_off(16 * t + n, ks * 4 + g));
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[ks], s[t], 0, 0, 0);
      }
    }
    if (!more) {  // only the last key tile has a tail: real branch (uniform), no selects on the main path
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j0 + 16 * t + 4 * g + r >= T) s[t][r] = -1e30f;
      asm volatile("" ::: "memory");
    }

    if constexpr (PROBS) {  // P = exp(score - lse), 4 consecutive keys per store; the T tail is exactly 0
      const int qi_ = i0w + n;
      if (qi_ < T) {
        unsigned short* prow = out + (((size_t)b * H + h) * T + qi_) * ldp + j0 + 4 * g;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (j0 + 16 * t + 4 * g + 3 < ldp) {
            uint2 v;
            v.x = pack_bf2(__builtin_amdgcn_exp2f(s[t][0] * LOG2E - lse2), __builtin_amdgcn_exp2f(s[t][1] * LOG2E - lse2));
            v.y = pack_bf2(__builtin_amdgcn_exp2f(s[t][2] * LOG2E - lse2), __builtin_amdgcn_exp2f(s[t][3] * LOG2E - lse2));
            *reinterpret_cast<uint2*>(prow + 16 * t) = v;
          }
      }
    } else {
    // ---- online softmax (query i = lane&15; keys spread over regs and g)
    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
    for (int t = 1; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float mscaled = m_new * LOG2E;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[t][r] * LOG2E - mscaled);
        s[t][r] = p;
        psum += p;
      }
    if (__any(m_new != m_run)) {  // rescale only when some query's running maximum moved
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] *= alpha;
      m_run = m_new;
    }
    l_run += psum;

    // ---- O^T += V^T P^T.  k index of the MFMA <-> key j = 32ks + 16*(e>>2) + 4g + (e&3)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 pbf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pbf[e] = (__bf16)s[2 * ks][e];
        pbf[4 + e] = (__bf16)s[2 * ks + 1][e];
      }
      const int r0 = 32 * ks + 4 * g + q4, r1 = r0 + 16;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // 4 bf16 (8 B) at column 16dt + 4p4 of rows r0 / r1: chunk 2dt + (p4>>1), half-chunk p4&1
        const unsigned char* a0 = sV + v_off(r0, 2 * dt + (p4 >> 1)) + (p4 & 1) * 8;
        const unsigned char* a1 = sV + v_off(r1, 2 * dt + (p4 >> 1)) + (p4 & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
        const s16x8 av = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), pbf, o[dt], 0, 0, 0);
      }
    }
    }  // !PROBS

    if (more) {
      ATTN_LOAD_PE(pf0, 0, j0 + 64)  // first half of the next tile's band operands; lands during the barriers
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // everyone is done reading this tile's K/V
      ATTN_STORE_KV()
      __syncthreads();
    }
  }

  stamp(stamps, 2);
  if constexpr (PROBS) return;
  float l = l_run;
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float inv = 1.f / l;
  const int qi = i0w + n;
  if (lse != nullptr && qi < T && g == 0) lse[((size_t)b * H + h) * T + qi] = m_run + __logf(l);
  if (qi < T) {
    unsigned short* orow = out + (base_b + qi) * D + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 v;
      v.x = pack_bf2(o[dt][0] * inv, o[dt][1] * inv);
      v.y = pack_bf2(o[dt][2] * inv, o[dt][3] * inv);
      *reinterpret_cast<uint2*>(orow + 16 * dt + 4 * g) = v;
    }
  }
}

// ---- softmax backward of the attention, attention-shaped (backward pass; replaces a [B,H,T,T] f32 dP buffer):
//   dS[b,h,i,j] = P[i,j] * (dO_i . v_j - D[b,h,i]),   D = rowsum(dO o O)
// Same tiling as the forward: workgroup = (b, h, 64 queries), wave = 16 queries, 64-key tiles of V staged in LDS in
// the K-tile image, dP^T = V dO^T by MFMA with the dO rows as the B operand (as q is in the forward).  dS goes out
// twice: row-major bf16 [B,H,T,ldp] (operand of dK = dS^T q, dq = dS k) and skewed onto the relative-position axis,
// ds_skew[b,i,h, i-j+T-1] (operand of the position-table gradients); entries of ds_skew outside the band are not
// written: the caller zeroes that buffer once.
__global__ __launch_bounds__(256, 4) void attn_ds_kernel(const unsigned short* __restrict__ qkv,
                                                         const unsigned short* __restrict__ dout,
                                                         const unsigned short* __restrict__ p, const float* __restrict__ dvec,
                                                         unsigned short* __restrict__ ds, unsigned short* __restrict__ dsk,
                                                         int B, int T, int H, int64_t ldp, int W) {
  __shared__ __attribute__((aligned(16))) unsigned char sV[64 * 128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int i0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int D = H * 64, D3 = 3 * D;
  const int i0w = i0 + 16 * w;
  const size_t base_b = (size_t)b * T;
  const int qi = min(i0w + n, T - 1);
  const bool qvalid = i0w + n < T;
  bf16x8 dof[2];
  {
    const unsigned short* drow = dout + (base_b + qi) * D + h * 64;
    dof[0] = *reinterpret_cast<const bf16x8*>(drow + 8 * g);
    dof[1] = *reinterpret_cast<const bf16x8*>(drow + 32 + 8 * g);
  }
  const float dval = dvec[((size_t)b * H + h) * T + qi];
  const size_t prow = (((size_t)b * H + h) * T + qi) * ldp;
  unsigned short* skrow = dsk + ((base_b + qi) * H + h) * (size_t)W + (qi + T - 1);  // + (-j)

  const int srow0 = tid >> 3, srow1 = srow0 + 32, sc = tid & 7;
  uint4 vreg0, vreg1;
#define DS_LOAD_V(J0)                                                                                          \
  {                                                                                                            \
    vreg0 = *reinterpret_cast<const uint4*>(qkv + (base_b + min((J0) + srow0, T - 1)) * D3 + 2 * D + h * 64 + sc * 8); \
    vreg1 = *reinterpret_cast<const uint4*>(qkv + (base_b + min((J0) + srow1, T - 1)) * D3 + 2 * D + h * 64 + sc * 8); \
  }
#define DS_STORE_V()                                                 \
  {                                                                  \
    *reinterpret_cast<uint4*>(sV + tile_off(srow0, sc)) = vreg0;     \
    *reinterpret_cast<uint4*>(sV + tile_off(srow1, sc)) = vreg1;     \
  }
  DS_LOAD_V(0)
  DS_STORE_V()
  __syncthreads();
  for (int j0 = 0; j0 < T; j0 += 64) {
    const bool more = j0 + 64 < T;
    if (more) DS_LOAD_V(j0 + 64)
    // P of this lane's 4 x 4 keys (zero beyond T by construction of the probabilities buffer)
    uint2 pv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = j0 + 16 * t + 4 * g;
      pv[t] = (j + 3 < ldp) ? *reinterpret_cast<const uint2*>(p + prow + j) : make_uint2(0u, 0u);
    }
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sV + tile_off(16 * t + n, ks * 4 + g));
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, dof[ks], s[t], 0, 0, 0);
      }
    if (qvalid) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = j0 + 16 * t + 4 * g;
        if (j + 3 >= ldp) continue;
        const float p0 = bf2f((unsigned short)(pv[t].x & 0xffff)), p1 = bf2f((unsigned short)(pv[t].x >> 16));
        const float p2 = bf2f((unsigned short)(pv[t].y & 0xffff)), p3 = bf2f((unsigned short)(pv[t].y >> 16));
        const unsigned short d0 = f2bf(p0 * (s[t][0] - dval)), d1 = f2bf(p1 * (s[t][1] - dval));
        const unsigned short d2 = f2bf(p2 * (s[t][2] - dval)), d3 = f2bf(p3 * (s[t][3] - dval));
        *reinterpret_cast<uint2*>(ds + prow + j) = make_uint2((unsigned)d0 | ((unsigned)d1 << 16), (unsigned)d2 | ((unsigned)d3 << 16));
        if (j < T) skrow[-j] = d0;
        if (j + 1 < T) skrow[-j - 1] = d1;
        if (j + 2 < T) skrow[-j - 2] = d2;
        if (j + 3 < T) skrow[-j - 3] = d3;
      }
    }
    if (more) {
      __syncthreads();
      DS_STORE_V()
      __syncthreads();
    }
  }
#undef DS_LOAD_V
#undef DS_STORE_V
}

}  // namespace

extern "C" int csship_mha_relpos(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                 uint16_t* out, int32_t B, int32_t T, int32_t H, csship_stream_t stream) {
  if (!qkv || !pe_packed || !out) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0 || q_max < q_min || B > 65535 || H > 65535) return CSSHIP_EBADSHAPE;
  if (!aligned16(qkv) || !aligned16(pe_packed) || !aligned8(out)) return CSSHIP_EALIGN;
  dim3 grid((T + 63) / 64, H, B);
  hipLaunchKernelGGL(mha_relpos_kernel<false>, grid, dim3(256), 0, as_stream(stream), qkv,
                     reinterpret_cast<const bf16x8*>(pe_packed), q_min, q_max, out, B, T, H, g_csship_stamps, nullptr,
                     (int64_t)0);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_mha_relpos_train(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                       uint16_t* out, float* lse, int32_t B, int32_t T, int32_t H,
                                       csship_stream_t stream) {
  if (!qkv || !pe_packed || !out || !lse) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0 || q_max < q_min || B > 65535 || H > 65535) return CSSHIP_EBADSHAPE;
  if (!aligned16(qkv) || !aligned16(pe_packed) || !aligned8(out)) return CSSHIP_EALIGN;
  dim3 grid((T + 63) / 64, H, B);
  hipLaunchKernelGGL(mha_relpos_kernel<false>, grid, dim3(256), 0, as_stream(stream), qkv,
                     reinterpret_cast<const bf16x8*>(pe_packed), q_min, q_max, out, B, T, H, nullptr, lse, (int64_t)0);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_mha_relpos_probs(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                       const float* lse, uint16_t* p_bf16, int64_t ld_p, int32_t B, int32_t T, int32_t H,
                                       csship_stream_t stream) {
  if (!qkv || !pe_packed || !lse || !p_bf16) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0 || q_max < q_min || B > 65535 || H > 65535 || ld_p < T) return CSSHIP_EBADSHAPE;
  if (!aligned16(qkv) || !aligned16(pe_packed) || !aligned8(p_bf16) || (ld_p % 4) != 0) return CSSHIP_EALIGN;
  dim3 grid((T + 63) / 64, H, B);
  hipLaunchKernelGGL(mha_relpos_kernel<true>, grid, dim3(256), 0, as_stream(stream), qkv,
                     reinterpret_cast<const bf16x8*>(pe_packed), q_min, q_max, p_bf16, B, T, H, nullptr,
                     const_cast<float*>(lse), ld_p);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_attn_softmax_bwd(const uint16_t* qkv, const uint16_t* d_out, const uint16_t* p_bf16, int64_t ld_p,
                                       const float* d, uint16_t* ds_bf16, uint16_t* ds_skew_bf16, int32_t W, int32_t B,
                                       int32_t T, int32_t H, csship_stream_t stream) {
  if (!qkv || !d_out || !p_bf16 || !d || !ds_bf16 || !ds_skew_bf16) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0 || B > 65535 || H > 65535 || ld_p < T || W < 2 * T - 1) return CSSHIP_EBADSHAPE;
  if (!aligned16(qkv) || !aligned16(d_out) || !aligned8(p_bf16) || !aligned8(ds_bf16) || (ld_p % 4) != 0) return CSSHIP_EALIGN;
  dim3 grid((T + 63) / 64, H, B);
  hipLaunchKernelGGL(attn_ds_kernel, grid, dim3(256), 0, as_stream(stream), qkv, d_out, p_bf16, d, ds_bf16, ds_skew_bf16, B, T,
                     H, ld_p, W);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
