// Key-stationary backward of the relative-position attention (training; css/models/conformer.py:229-253 under autograd):
// ONE pass that recomputes the scores and forms every product that does not need the window axis,
//     S = q k^T + q pe[i - j]^T          P = exp(S - lse)            P' = mask / (1 - p) o P       (the forward's dropout)
//     dP = mask / (1 - p) o (dO v^T)     dS = P o (dP - D),  D_i = dO_i . O_i
//     dV = P'^T dO        dK = dS^T q        dq1 = dS k        (dq = dq1 + dS_skew pe_window: csship_attn_pos_bwd_ds)
// and leaves exactly one [B, H, T, T] tensor in memory - the row-major dS the position pass skews - where rounds 3-5 wrote
// and re-read two (mha_fwd_kernel<PROBS, BWD> wrote P' and dS, attn_dvdk_kernel streamed both back: 72 us per block for the
// pair, 164 of their 205 MB of traffic).  VERDICT r5 item 2.
//
// Decomposition.  dV and dK reduce over the QUERIES, dq1 over the KEYS: the query-stationary kernel kept dq1 in registers
// and had to spill P' / dS to memory for a second, key-stationary pass.  Here a workgroup owns up to 256 KEYS of one
// (batch, head) - every wave 32 of them, its K / V rows as MFMA B fragments and its dV^T / dK^T tiles (64 x 32 each) in
// registers for the whole kernel - and walks the queries in blocks of 32 (Q and dO blocks + log-sum-exp + D staged in
// LDS, double buffered).  Scores are formed as S[query][key] (not transposed as in the forward), so that a wave's P' and
// dS tiles ARE the B operands of  dV^T += dO^T P'  and  dK^T += q^T dS  (reduction index = accumulator row = query; the
// A operands come from the staged blocks by transposing LDS reads).  dq1 needs the other index: every wave drops its dS
// tile into a shared [32 queries][keys of the workgroup] bf16 panel in LDS, and behind ONE barrier per block each wave
// multiplies one 16 x 16 tile of  dq1^T = K^T dS^T  over all the workgroup's keys (K^T by transposing reads of a K
// panel staged once) - no cross-wave float reduction, fixed order - while the same panel leaves for memory as whole
// 16-byte row pieces.  dq1 is a partial per key group (two at T = 400): the position pass adds them.
//
// The relative-position term in this orientation: for a 16-query x 16-key tile with bases I0, J0 the offsets i - j span
// two ALIGNED, reversed blocks of the packed table (ops.pack_relpos: block q, row m = pe[16 q + 15 - m]): "hi" = (I0 - J0)
// / 16 for i - j >= I0 - J0 and "lo" = hi - 1 below.  A wave's 32 queries x 32 keys need three blocks per query block;
// their products Pb[query][block row] go through a per-wave scratch row [hi | mid | lo] of 48 floats per query, from
// which score (a, b) of key block kb2 reads element 16 kb2 + 15 - a + b: the rel-shift is the read address, the score
// accumulators start from it.  The blocks live in an LDS ring shared by the workgroup (18 live, two new per query block).
#include <algorithm>

#include "common.hpp"

namespace {

#ifndef FB_ABL
#define FB_ABL 0   // diagnostic builds only (tools/attn_fb_ablate.sh): bit 0 no position band, 1 no dq1 product, 2 no dS write-out, 3 no dV / dK products, 4 no panel stores, 5 no exp / dS arithmetic
#endif
constexpr float LOG2E = 1.4426950408889634f;
constexpr int FB_KW = 256;                 // keys per workgroup at most (8 waves x 32)
constexpr int FB_QB = 32;                  // queries per block
constexpr int FB_QT = FB_QB * 128;         // a staged Q (or dO) block: 4 KB, rows of 64 bf16 in the tile_off layout
constexpr int FB_DSROW = FB_KW * 2;        // bytes of a row of the dS panel
constexpr int FB_DST = FB_QB * FB_DSROW;   // 16 KB
constexpr int FB_PS = 52;                  // floats per scratch row (48 used: 4 g x 52 spreads the four row groups over the banks)
constexpr int FB_PW = 16 * FB_PS;          // scratch floats per wave
constexpr int FB_R = 20;                   // ring slots: 18 blocks live + the two arriving
constexpr int FB_SMEM = 4 * FB_QT + 2 * 64 * 4 + FB_KW * 128 + 2 * FB_DST + 8 * FB_PW * 4 + FB_R * 2048;

struct FbArgs {
  const unsigned short* qkv;    // bf16 [B*T, 3*H*64], q rows pre-scaled by 1/sqrt(d_k)
  const bf16x8* pe;             // packed position blocks (ops.pack_relpos)
  int q_min, q_max;
  const float* lse;             // f32 [B, H, T]: the training forward's log-sum-exp
  const unsigned short* dout;   // bf16 [B*T, H*64]: gradient of the head-concatenated output
  const unsigned short* ao;     // bf16 [B*T, H*64]: the forward's output (D = rowsum(dO o O))
  unsigned short* ds;           // bf16 [B, H, T, ldp] out
  float* dq1;                   // f32 [G][B, H, T, 64] out: dS k over each key group
  unsigned short* dqkv;         // bf16 [B*T, 3*H*64]: the k and v thirds are written
  int64_t ldp;
  int B, T, H, G, KH;
  csship_dropout_cfg drop;
};

DEV int fb_ds_off(int row, int chunk) { return row * FB_DSROW + ((chunk ^ (row & 15)) << 4); }   // 16-byte chunk 0..31 of a panel row

template <int R>
DEV unsigned quad_bcast(unsigned v) {   // lane (quad, k) <- lane (quad, R)
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, R * 0x55, 0xf, 0xf, true);
}

template <bool DROP>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(FbArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[FB_SMEM];
  unsigned char* const sQ = smem;                                  // [2][FB_QT]
  unsigned char* const sO = smem + 2 * FB_QT;                      // [2][FB_QT]
  float* const sLD = reinterpret_cast<float*>(smem + 4 * FB_QT);   // [2][lse2 x 32 | D x 32]
  unsigned char* const sK = smem + 4 * FB_QT + 512;                // [256 keys][128 B], tile_off layout
  unsigned char* const sDS = sK + FB_KW * 128;                     // [2][FB_DST]
  float* const sP = reinterpret_cast<float*>(sDS + 2 * FB_DST);    // [8 waves][FB_PW]
  unsigned char* const sR = sDS + 2 * FB_DST + 8 * FB_PW * 4;      // [FB_R][2 k-steps][64 lanes x 16 B]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15, q4 = n >> 2, p4 = n & 3;
  const int BH = a.B * a.H, T = a.T;
  int bh, gi;
  {
    const int id = blockIdx.x;
    if ((BH & 7) == 0) {   // ids equal mod 8 share an XCD: the key groups of one (b, h) read the same Q / dO blocks from its L2
      const int per = 8 * a.G, blk = id / per, r = id - blk * per;
      bh = blk * 8 + (r & 7);
      gi = r >> 3;
    } else {
      bh = id / a.G;
      gi = id - bh * a.G;
    }
  }
  const int b = bh / a.H, h = bh - b * a.H;
  const int D = a.H * 64, D3 = 3 * D;
  const size_t base_b = (size_t)b * T;
  const int kb = gi * a.KH, kn = min(a.KH, T - kb);
  if (kn <= 0) return;   // (workgroup-uniform; the host launches no such workgroup)
  const bool active = 32 * w < kn;            // waves beyond the group's keys stage, keep the barriers and share the dq1 work
  const int nks = (kn + 31) >> 5;             // 32-key steps of the dq1 product
  const int jw = kb + 32 * w;                 // first key of the wave

  // ---- the wave's K and V rows as B fragments (lane (g, n): key n of block kb2, features 32 ks + 8 g .. + 7)
  bf16x8 kf[2][2], vf[2][2];
#pragma unroll
  for (int kb2 = 0; kb2 < 2; ++kb2) {
    const unsigned short* row = a.qkv + (base_b + min(jw + 16 * kb2 + n, T - 1)) * D3 + D + h * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[kb2][ks] = *reinterpret_cast<const bf16x8*>(row + 32 * ks + 8 * g);
      vf[kb2][ks] = *reinterpret_cast<const bf16x8*>(row + D + 32 * ks + 8 * g);
    }
  }
  // ---- the K panel of the whole group (zero rows beyond its keys): B operand of dq1^T = K^T dS^T
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int idx = tid + c * 512, row = idx >> 3, ch = idx & 7;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < kn) v = *reinterpret_cast<const uint4*>(a.qkv + (base_b + kb + row) * D3 + D + h * 64 + ch * 8);
    *reinterpret_cast<uint4*>(sK + tile_off(row, ch)) = v;
  }
  // ---- the position-block ring.  Query block `it` (queries 32 it ..) needs blocks lo(it) .. lo(it) + 17,
  // lo(it) = 2 it - kb / 16 - 16 (wave w, query half qb: blocks 2 it - kb / 16 - 2 w + qb - 2 .. + qb); block q sits in slot
  // q mod FB_R as two lane-linear 1-KB fragments.  Behind the barrier of block `it` waves 0-3 fetch blocks lo(it) + 20, + 21 -
  // block it + 2's newcomers - into the slots of lo(it), lo(it) + 1, which nobody reads any more (every wave has passed
  // block it's products; block it + 1 reads from lo(it) + 2 on) and which block it + 1's barrier publishes.
  const bf16x8* const pe_lane = a.pe + lane;
  auto ring_slot = [](int q) { const int m = q % FB_R; return m < 0 ? m + FB_R : m; };
  auto pe_src = [&](int q, int ks) {
    const int qi_ = min(max(q, a.q_min), a.q_max) - a.q_min;
    return pe_lane + (size_t)(qi_ * 2 + ks) * 64;
  };
  {
    const int lo0 = -(kb >> 4) - 16;   // blocks lo(0) .. lo(0) + 19: what query blocks 0 and 1 read
#pragma unroll
    for (int f0 = 0; f0 < 40; f0 += 8) {
      const int f = f0 + w;
      *reinterpret_cast<bf16x8*>(sR + (ring_slot(lo0 + (f >> 1)) * 2 + (f & 1)) * 1024 + lane * 16) = *pe_src(lo0 + (f >> 1), f & 1);
    }
  }

  // ---- staging of a query block: threads 0..255 the dO block (+ D from the forward's output), 256..511 the Q block (+ lse)
  const int sr = (tid & 255) >> 3, sc = tid & 7;
  const bool is_q = tid >= 256;
  uint4 st_a, st_o;
  float st_l;
  st_o = make_uint4(0u, 0u, 0u, 0u);
  st_l = 0.f;
#define FB_LOAD(I0, IT)                                                                                              \
  {                                                                                                                  \
    const size_t r_ = base_b + min((I0) + sr, T - 1);                                                               \
    if (is_q) {                                                                                                      \
      st_a = *reinterpret_cast<const uint4*>(a.qkv + r_ * D3 + h * 64 + sc * 8);                                     \
      st_l = a.lse[(size_t)bh * T + min((I0) + sr, T - 1)];                                                          \
    } else {                                                                                                         \
      st_a = *reinterpret_cast<const uint4*>(a.dout + r_ * D + h * 64 + sc * 8);                                     \
      st_o = *reinterpret_cast<const uint4*>(a.ao + r_ * D + h * 64 + sc * 8);                                       \
    }                                                                                                                \
  }
#define FB_STORE(I0, IT, BUF)                                                                                        \
  {                                                                                                                  \
    const bool ok_ = (I0) + sr < T;                                                                                  \
    const uint4 z_ = make_uint4(0u, 0u, 0u, 0u);                                                                     \
    if (is_q) {                                                                                                      \
      *reinterpret_cast<uint4*>(sQ + (BUF) * FB_QT + tile_off(sr, sc)) = ok_ ? st_a : z_;                            \
      if (sc == 0) sLD[(BUF) * 64 + sr] = st_l * LOG2E;                                                              \
    } else {                                                                                                         \
      *reinterpret_cast<uint4*>(sO + (BUF) * FB_QT + tile_off(sr, sc)) = ok_ ? st_a : z_;                            \
      const bf16x8 x_ = __builtin_bit_cast(bf16x8, st_a), y_ = __builtin_bit_cast(bf16x8, st_o);                     \
      float d_ = 0.f;                                                                                                \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) d_ += (float)x_[e] * (float)y_[e];                               \
      d_ = sum8(d_);                                                                                                 \
      if (sc == 0) sLD[(BUF) * 64 + 32 + sr] = ok_ ? d_ : 0.f;                                                       \
    }                                                                                                                \
  }
  FB_LOAD(0, -1)
  FB_STORE(0, -1, 0)   // (IT = -1: blocks -kb/16 + 0, + 1 - already in the ring: a harmless rewrite of the same data)
  const int nblk = (T + FB_QB - 1) / FB_QB;
  if (nblk > 1) FB_LOAD(FB_QB, 0)   // block 1 travels while block 0 is worked on
  __syncthreads();

  f32x4 dv[4][2], dk[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) dv[dt][kb2] = dk[dt][kb2] = f32x4{0.f, 0.f, 0.f, 0.f};
  const Drop dp = DROP ? drop_make(a.drop.seed, a.drop.stream, a.drop.p) : Drop{0u, 0u, 1.f, 0x80008000u};
  const unsigned Tp = (unsigned)a.ldp;
  float* const Pw = sP + w * FB_PW;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  // Every phase of a block is a chain  LDS read -> MFMA -> LDS write -> LDS read ...  inside one wave, and there are only two
  // waves per SIMD to overlap them (ablations of the first version, tools/attn_fb_ablate.sh: band 16 us, dV / dK 13, dq1 11,
  // dS write-out 9, panel stores 8 of 72 - each an order of magnitude above its instruction count).  So the reads of a
  // phase are issued in one batch ahead of everything that consumes them (sched_barrier keeps hipcc from sinking them back to
  // their first use), independent chains are interleaved by hand, and the block's global loads are issued BEFORE the
  // block's global stores (a wave's memory operations return in order: a load behind stores waits for them).
  for (int it = 0; it < nblk; ++it) {
    const int cur = it & 1, i0 = it * FB_QB;
    const unsigned char* const q_t = sQ + cur * FB_QT;
    const unsigned char* const o_t = sO + cur * FB_QT;
    const float* const ld_t = sLD + cur * 64;
    unsigned char* const ds_t = sDS + cur * FB_DST;

    if (active) {
      f32x4 s[2][2], dpt[2][2];
      {
        // ---- batch 1 of LDS reads: the Q and dO blocks as A fragments, the first query half's three position blocks
        bf16x8 qa[2][2], oa[2][2], pf[3][2];
        const int Q0 = ((i0 - jw) >> 4);   // (exact: both are multiples of 32)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) qa[qb][ks] = *reinterpret_cast<const bf16x8*>(q_t + tile_off(16 * qb + n, ks * 4 + g));
#define FB_RING(QB)                                                                        \
  _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                          \
    const unsigned char* fr = sR + ring_slot(Q0 + (QB) - u) * 2048 + lane * 16;            \
    pf[u][0] = *reinterpret_cast<const bf16x8*>(fr);                                       \
    pf[u][1] = *reinterpret_cast<const bf16x8*>(fr + 1024);                                \
  }
        FB_RING(0)
        __builtin_amdgcn_sched_barrier(0);
        // ---- position band, one query half after the other (three independent chains of two MFMAs each) through the scratch:
        // rows [hi | mid | lo]; score (a, b) of key block kb2 reads element 16 kb2 + 15 - a + b.  The second half's blocks are
        // requested before the first half's values go through the scratch.
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          if (FB_ABL & 1) { s[qb][0] = s[qb][1] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
          f32x4 pb[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) pb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int u = 0; u < 3; ++u) pb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[qb][ks], pf[u][ks], pb[u], 0, 0, 0);
          if (qb == 0) {
            FB_RING(1)
            __builtin_amdgcn_sched_barrier(0);
          } else {   // the dO fragments of dP: requested here, they arrive under the second half's scratch trip and the S products
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
              for (int ks = 0; ks < 2; ++ks) oa[q2][ks] = *reinterpret_cast<const bf16x8*>(o_t + tile_off(16 * q2 + n, ks * 4 + g));
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) Pw[(4 * g + r) * FB_PS + 16 * u + n] = pb[u][r];   // row a = 4 g + r, block row n
          // The reads below take values OTHER lanes of the wave stored (LDS operations of a wave execute in order, so no
          // hardware barrier is needed) - but to the compiler a lane's loads never alias the lane's own stores (rows 4 g + r
          // with r < 3 provably differ), and without the two compiler barriers it re-used the first query half's loads for
          // the second half
          asm volatile("" ::: "memory");
#pragma unroll
          for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[qb][kb2][r] = Pw[(4 * g + r) * (FB_PS - 1) + 16 * kb2 + 15 + n];   // [a][16 kb2 + 15 - a + n]
          asm volatile("" ::: "memory");
        }
#undef FB_RING
        // ---- S = band + Q K^T
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2)
              s[qb][kb2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[qb][ks], kf[kb2][ks], s[qb][kb2], 0, 0, 0);
        // ---- dP = dO V^T
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int kb2 = 0; kb2 < 2; ++kb2) dpt[qb][kb2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb2 = 0; kb2 < 2; ++kb2)
              dpt[qb][kb2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa[qb][ks], vf[kb2][ks], dpt[qb][kb2], 0, 0, 0);
      }
      // ---- P, P' and dS of the wave's 32 x 32 tile.  Lane (g, n) holds rows (queries) 4 g + r and column (key) n of each tile.
      u32x4 pw[2], dw[2];   // B operands: k index e = 4 qb + r <-> query 16 qb + 4 g + r
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        // log-sum-exp (log2 units) and D of the half's rows
        const f32x4 l2q = *reinterpret_cast<const f32x4*>(ld_t + 16 * qb + 4 * g);
        const f32x4 ddq = *reinterpret_cast<const f32x4*>(ld_t + 32 + 16 * qb + 4 * g);
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
          const int j = jw + 16 * kb2 + n;
          const bool jok = j < T;
          float pv[4], pd[4], dsv[4];
          unsigned w0 = 0u, w1 = 0u;
          if constexpr (DROP) {   // lane p4 of a quad hashes row p4's group of four keys; the quad shares the words
            const unsigned e0 = ((unsigned)bh * (unsigned)T + (unsigned)min(i0 + 16 * qb + 4 * g + p4, T - 1)) * Tp + (unsigned)(jw + 16 * kb2 + 4 * q4);
            w0 = hash32(dp.key + (e0 >> 2));
            w1 = hash_derive(w0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = i0 + 16 * qb + 4 * g + r;
            const float p = (jok && i < T) ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kb2][r], LOG2E, -l2q[r])) : 0.f;
            pv[r] = p;
            float dpv = dpt[qb][kb2][r];
            pd[r] = p;
            if constexpr (DROP) {
              const unsigned x0 = r == 0 ? quad_bcast<0>(w0) : r == 1 ? quad_bcast<1>(w0) : r == 2 ? quad_bcast<2>(w0) : quad_bcast<3>(w0);
              const unsigned x1 = r == 0 ? quad_bcast<0>(w1) : r == 1 ? quad_bcast<1>(w1) : r == 2 ? quad_bcast<2>(w1) : quad_bcast<3>(w1);
              const unsigned wd = (p4 & 2) ? x1 : x0;
              const bool keep = ((wd >> (16 * (p4 & 1))) & 0xffffu) >= dp.thresh;
              pd[r] = keep ? p * dp.scale : 0.f;
              dpv = keep ? dpv * dp.scale : 0.f;
            }
            dsv[r] = pv[r] * (dpv - ddq[r]);
          }
          pw[kb2][2 * qb] = pack_bf2(pd[0], pd[1]);
          pw[kb2][2 * qb + 1] = pack_bf2(pd[2], pd[3]);
          const unsigned d01 = pack_bf2(dsv[0], dsv[1]), d23 = pack_bf2(dsv[2], dsv[3]);
          dw[kb2][2 * qb] = d01;
          dw[kb2][2 * qb + 1] = d23;
          // dS tile -> the workgroup's panel (rows 16 qb + 4 g + r, key column 32 w + 16 kb2 + n) as DWORDS: neighbouring
          // lanes trade halves - the even lane takes rows 4 g + 0, 1 of the column pair (n, n + 1), the odd lane rows 4 g + 2, 3
          if (!(FB_ABL & 16)) {
            const bool odd = n & 1;
            const unsigned give = odd ? d01 : d23;
            const unsigned got = (unsigned)__builtin_amdgcn_update_dpp(0, (int)give, 0xB1, 0xf, 0xf, true);   // quad_perm [1, 0, 3, 2]
            const unsigned lo = odd ? got : d01, hi = odd ? d23 : got;       // column n & ~1, column n | 1 (two rows each)
            const unsigned wa = (lo & 0xffffu) | (hi << 16), wb = (lo >> 16) | (hi & 0xffff0000u);
            const int col = 32 * w + 16 * kb2 + (n & ~1), rowa = 16 * qb + 4 * g + (odd ? 2 : 0);
            *reinterpret_cast<unsigned*>(ds_t + fb_ds_off(rowa, col >> 3) + (col & 7) * 2) = wa;
            *reinterpret_cast<unsigned*>(ds_t + fb_ds_off(rowa + 1, col >> 3) + (col & 7) * 2) = wb;
          }
        }
      }
      // ---- dV^T += dO^T P',  dK^T += q^T dS: the eight A fragments (transposing reads of the staged blocks; k index e = 4 qb + r
      // <-> query 16 qb + 4 g + r) in ONE batch, then the sixteen products
      {
        bf16x8 ta_o[4], ta_q[4];
        const int r0 = 4 * g + q4, r1 = r0 + 16;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int co = 2 * dt + (p4 >> 1), ho = (p4 & 1) * 8;
          const s16x4 o_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(o_t + tile_off(r0, co) + ho));
          const s16x4 o_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(o_t + tile_off(r1, co) + ho));
          const s16x4 q_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q_t + tile_off(r0, co) + ho));
          const s16x4 q_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q_t + tile_off(r1, co) + ho));
          ta_o[dt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(o_lo, o_hi, 0, 1, 2, 3, 4, 5, 6, 7));
          ta_q[dt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(q_lo, q_hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int kb2 = 0; kb2 < ((FB_ABL & 8) ? 0 : 2); ++kb2) {
            dv[dt][kb2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta_o[dt], __builtin_bit_cast(bf16x8, pw[kb2]), dv[dt][kb2], 0, 0, 0);
            dk[dt][kb2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta_q[dt], __builtin_bit_cast(bf16x8, dw[kb2]), dk[dt][kb2], 0, 0, 0);
          }
      }
    }
    if (it + 1 < nblk) FB_STORE(i0 + FB_QB, it, cur ^ 1)
    // LDS-only barrier: __syncthreads() would also wait for the global stores of the previous block's tail (vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- behind the barrier: dq1^T tile (features 16 dt .. of queries 16 qb ..) over all the group's keys, and the panel's rows -> dS
    {
      const int qb = w >> 2, dt = w & 3;
      const int row = 16 * qb + n;
      const int co = 2 * dt + (p4 >> 1), ho = (p4 & 1) * 8;
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      const int nk = (FB_ABL & 2) ? 0 : nks;
      // the panel's rows for the write-out, requested with the first fragments
      const int orow = tid >> 4, oi = i0 + orow;
      uint4 ov[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int c = (tid & 15) + 16 * cc;
        if (8 * c < 32 * nks && kb + 8 * c < T) ov[cc] = *reinterpret_cast<const uint4*>(ds_t + fb_ds_off(orow, c));
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8 bv[4], av[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          const int ks = 4 * half + k4;
          if (ks < nk) {   // (workgroup-uniform)
            bv[k4] = *reinterpret_cast<const bf16x8*>(ds_t + fb_ds_off(row, 4 * ks + g));   // query n, keys 32 ks + 8 g .. + 7
            const int k0 = 32 * ks + 8 * g + q4;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sK + tile_off(k0, co) + ho));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sK + tile_off(k0 + 4, co) + ho));
            av[k4] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k4 = 0; k4 < 4; k4 += 2) {
          if (4 * half + k4 < nk) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[k4], bv[k4], acc0, 0, 0, 0);
          if (4 * half + k4 + 1 < nk) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[k4 + 1], bv[k4 + 1], acc1, 0, 0, 0);
        }
      }
      bf16x8 newcomer = bf16x8{};
      const int qnew = 2 * it - (kb >> 4) + 4 + (w >> 1);   // lo(it) + 20, + 21
      if (w < 4 && it + 2 < nblk) newcomer = *pe_src(qnew, w & 1);
      if (it + 2 < nblk) FB_LOAD(i0 + 2 * FB_QB, it + 1)   // ahead of this block's stores (a wave's memory operations return in order)
      if (oi < T && !(FB_ABL & 4)) {
        unsigned short* const drow = a.ds + ((size_t)bh * T + oi) * a.ldp + kb;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int c = (tid & 15) + 16 * cc;
          if (8 * c < a.KH && kb + 8 * c + 8 <= a.ldp) *reinterpret_cast<uint4*>(drow + 8 * c) = ov[cc];
        }
      }
      const int i = i0 + row;
      if (i < T) *reinterpret_cast<f32x4*>(a.dq1 + (((size_t)gi * BH + bh) * T + i) * 64 + 16 * dt + 4 * g) = acc0 + acc1;
      if (w < 4 && it + 2 < nblk) *reinterpret_cast<bf16x8*>(sR + (ring_slot(qnew) * 2 + (w & 1)) * 1024 + lane * 16) = newcomer;
    }
  }
#undef FB_LOAD
#undef FB_STORE

  if (!active) return;
  // dV^T / dK^T tiles: lane (g, n) holds features 16 dt + 4 g .. + 3 of key jw + 16 kb2 + n: one 8-byte store each
#pragma unroll
  for (int kb2 = 0; kb2 < 2; ++kb2) {
    const int j = jw + 16 * kb2 + n;
    if (j >= T || 32 * w + 16 * kb2 + n >= kn) continue;
    unsigned short* row = a.dqkv + (base_b + j) * D3 + h * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *reinterpret_cast<uint2*>(row + D + 16 * dt) = make_uint2(pack_bf2(dk[dt][kb2][0], dk[dt][kb2][1]), pack_bf2(dk[dt][kb2][2], dk[dt][kb2][3]));
      *reinterpret_cast<uint2*>(row + 2 * D + 16 * dt) = make_uint2(pack_bf2(dv[dt][kb2][0], dv[dt][kb2][1]), pack_bf2(dv[dt][kb2][2], dv[dt][kb2][3]));
    }
  }
}

// key groups per (b, h) and keys per group: as few groups as 256-key workgroups allow - two when that leaves CUs idle -
// keys dealt in 32s (32 x 4 heads x 400 frames: 2 groups of 224 / 176 keys, 256 workgroups)
void fb_groups(int B, int T, int H, int* G, int* KH) {
  int g0 = (T + FB_KW - 1) / FB_KW;
  if ((int64_t)B * H * g0 < 256 && T > 64) g0 *= 2;
  *KH = std::min(FB_KW, ((T + g0 - 1) / g0 + 31) / 32 * 32);
  *G = (T + *KH - 1) / *KH;
}

}  // namespace

extern "C" int csship_mha_relpos_bwd_fused_parts(int32_t B, int32_t T, int32_t H) {
  if (B <= 0 || T <= 0 || H <= 0) return 0;
  int G, KH;
  fb_groups(B, T, H, &G, &KH);
  return G;
}

extern "C" int csship_mha_relpos_bwd_fused(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                           const float* lse, const uint16_t* d_out, const uint16_t* out, uint16_t* ds_bf16,
                                           int64_t ld_p, float* dq1_parts, uint16_t* dqkv, int32_t B, int32_t T, int32_t H,
                                           const csship_dropout_cfg* drop, csship_stream_t stream) {
  if (!qkv || !pe_packed || !lse || !d_out || !out || !ds_bf16 || !dq1_parts || !dqkv) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0 || q_max < q_min || B > 65535 || H > 65535 || ld_p < T || (ld_p & 7) != 0) return CSSHIP_EBADSHAPE;
  const bool dropping = drop && drop->seed && drop->p > 0.f;
  if (dropping && (ld_p != (T + 7) / 8 * 8 || (int64_t)B * H * T * ld_p > 0xffffffffLL)) return CSSHIP_EBADSHAPE;   // 32-bit mask indices
  if (!aligned16(qkv) || !aligned16(pe_packed) || !aligned16(d_out) || !aligned16(out) || !aligned16(ds_bf16) ||
      !aligned16(dq1_parts) || !aligned8(dqkv))
    return CSSHIP_EALIGN;
  int G, KH;
  fb_groups(B, T, H, &G, &KH);
  const long long nwg = (long long)B * H * G;
  if (nwg > 0x7fffffffLL) return CSSHIP_EBADSHAPE;
  FbArgs a{qkv, reinterpret_cast<const bf16x8*>(pe_packed), q_min, q_max, lse, d_out, out, ds_bf16, dq1_parts, dqkv, ld_p, B, T, H, G, KH,
           dropping ? *drop : csship_dropout_cfg{nullptr, 0u, 0.f}};
  if (dropping) hipLaunchKernelGGL((attn_bwd_fused_kernel<true>), dim3((unsigned)nwg), dim3(512), 0, as_stream(stream), a);
  else hipLaunchKernelGGL((attn_bwd_fused_kernel<false>), dim3((unsigned)nwg), dim3(512), 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
