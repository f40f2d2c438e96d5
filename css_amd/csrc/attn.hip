// Flash-style relative-position attention for the Conformer (d_k = 64), gfx950.
//
// Replaces css/models/conformer.py:229-253 plus the [T,T,d_k] relative-position table
// of conformer.py:157-159,179-185:
//     score[i,j] = q_i.k_j + q_i.pe[clamp(i-j,-maxlen,maxlen-1)+maxlen]      (q pre-scaled)
//     out[i]     = softmax_j(score[i,:]) @ v
//
// mha_fwd_kernel below (8 or 4 waves per workgroup, a pair of 16-query blocks per wave):
//   * everything is computed TRANSPOSED so the softmax'd tile is already the MFMA B
//     operand of the P.V product (no LDS round trip for P):
//         S^T[j,i]  = K_tile[j,:]  . Q[i,:]      4 x (16 keys x 16 queries) per 64-key tile and query block
//         Pb^T[c,i] = PE_band[c,:] . Q[i,:]      5 x (16 rel. offsets x 16 queries)
//         O^T[d,i] += V^T[d,j] P^T[j,i]          V^T fragments come from ds_read_b64_tr_b16
//   * the position term only depends on i-j: for a 16-query x 64-key block the 79 distinct
//     offsets form a band c = 15 - a + jl (a = query in block, jl = key in tile).  Because
//     query blocks and key tiles start at multiples of 16, every band tile is one ALIGNED,
//     REVERSED block of 16 consecutive pe rows; the host packs those blocks in MFMA fragment
//     order with the clamp baked in (css_amd.ops.pack_relpos): no per-element clamp, no
//     T x T x d_k table.  The band product reaches the score accumulators through a per-wave
//     fp32 scratch whose STORES do the rel-shift;
//   * V is kept in a 32-byte-group XOR swizzle that makes the transposed reads conflict free;
//   * online softmax over key tiles; key padding mask is always None on this path
//     (conformer.py:97), only the T tail of the last tile is masked.
// The first-generation kernel of round 1 (4 waves per (b, h, 64 queries), band fragments straight from global memory,
// 29.9 us at 32 x 4 x 400 against 24.7) was removed once the backward's probability recompute moved here too (PROBS).
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr float LOG2E = 1.4426950408889634f;

DEV int v_off(int row, int chunk) {  // 16-byte chunk of a 128-byte V row; 32-byte groups XOR (row>>1)&3
  return row * 128 + ((((chunk >> 1) ^ ((row >> 1) & 3)) << 1) | (chunk & 1)) * 16;
}

// ---------------------------------------------------------------------------------------------------------------
// Forward attention, second generation (inference / training forward, and with PROBS the backward's recomputation of the
// probabilities).
//
// What changed against round 1's mha_relpos_kernel, and why (rocprof / PMC of round 1: 29.9 us per launch at 32 x 4 x 400,
// MFMA pipe 21 % busy, 4.0x the algorithmic traffic, two workgroup barriers per 64-key tile; ablation builds of this
// kernel, tools/attn_ablate.sh: the position band alone cost 8 of 25 us while its 12 fragment loads per wave and tile
// came straight from global memory - the CU's texture-address path takes ~18 cycles per wave-wide load):
//   * a workgroup is NW = 8 (4) waves and every wave owns a PAIR of 16-query blocks (32 queries): a workgroup covers up
//     to 256 queries of one (b, h), so each K/V tile is fetched by ceil(T/256) workgroups instead of ceil(T/64), and the
//     K and V^T fragments a wave reads from LDS feed two MFMAs each;
//   * the position blocks live in an LDS RING shared by the workgroup: at key tile j0 its waves need the 2 NW + 4
//     consecutive blocks (i0 - j0)/16 - 4 .. + 2 NW - 1, the next tile the same window moved down by four, so every
//     tile only FOUR new blocks (8 KB) enter the ring - one 1-KB fragment per wave - instead of 12 KB per wave;
//   * K/V tiles are double buffered in LDS: tile j+1 (and the ring's four new blocks) travel global -> registers during
//     tile j's work and are written at the end of the iteration: ONE workgroup barrier per key tile;
//   * the band's rel-shift is done by the STORE: every band value is written straight to the place in the wave's
//     scratch where its consumer's 16-byte accumulator load takes it from (a per-lane, per-register base that is fixed
//     for the whole kernel + an immediate per band tile), and the band of tile j+1 is computed during tile j's
//     exponentials (it depends on nothing but q and the ring);
//   * row sums come from one extra MFMA per k-step against a fragment of ones (the matrix pipe has slack, the vector
//     pipe does not), so the normaliser is the sum of the SAME bf16-rounded probabilities that multiply V; the cross
//     lane part of the row maximum uses v_permlane16/32_swap instead of two ds_bpermute round trips;
//   * workgroups of one (b, h) get ids that are equal mod 8 (the dispatcher deals ids round-robin over the XCDs), so
//     the second fetch of that head's K/V hits the same L2.  Speed only: nothing depends on the placement.
// Measured at 32 x 4 x 400 (graph replay, tools/bench_kernels.py attn): 24.2 us against 30.1 us.  What bounds it now
// (PMC + per-wave phase stamps, tools/attn_stamps.py, profiles/r02_attn_*): instruction ISSUE.  A wave issues for 39 %
// of its cycles (VALU incl. MFMA 57 %, scalar 23 %, LDS 17 % of that), the two waves of a SIMD are served strictly
// oldest first (waves 0-3 pass the exp + band phase in 1 775 cycles per tile, waves 4-7 in 3 035 and the older half
// waits for them at the barrier; s_setprio on the younger half just swaps the roles, running them half a tile out
// of phase changes nothing), so the SIMD's issue port is ~90 % busy.  The next step is fewer, fatter instructions:
// 32x32x16 MFMAs (half the matrix instructions, 12 + 16 LDS instructions for the rel-shift instead of 48), see DESIGN.md.
#ifndef A2_ABL
#define A2_ABL 0  // diagnostic builds only (tools/attn_ablate.sh): bit 0 no band, 1 no softmax, 2 no P.V, 3 no staging, 4 no S; backward: 6 no skewed-dS store, 7 no P' store, 8 no dS store
#endif
#ifdef A2_STAMPS  // diagnostic build: per-wave cycle sums of the loop's phases -> stamps[(wg * NW + wave) * 8 + phase]
#define A2_T(i)                                                                                          \
  {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    unsigned long long t_;                                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    acc_[i] += t_ - tl_;                                                                                 \
    tl_ = t_;                                                                                            \
  }
#else
#define A2_T(i)
#endif
constexpr int A2_KV = 64 * 128;   // one K (or V) tile in LDS
// band scratch floats per wave: pad | 4 key blocks x [16 queries][A2_PS] | pad.  A query's 16 keys of a block sit in a row
// of A2_PS = 20 floats: with 16 (64-byte rows) the skew's 40 single-float stores per tile hit every bank of their 32-lane
// group twice (address = 17 n + 4 g + const: n and n + 4, g and g - 1 collide) - 0.76 conflict cycles per active LDS
// cycle for the whole kernel (r03 PMC); 20 makes the stores conflict free and leaves the eight 16-byte loads two-way.
constexpr int A2_PS = 20;
constexpr int A2_PB = 16 * A2_PS;  // one key block of the scratch
constexpr int A2_PQ = 6 * A2_PB;

DEV float vmax(float a, float b) {  // bare v_max_f32 (fmaxf adds a canonicalising v_max per MFMA-produced operand)
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
DEV float vmax3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
DEV float xmax16(float x) {  // max over the four 16-lane rows of the wave (same column n), in all lanes
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const unsigned a0 = r[0], a1 = r[1];
  const float y = vmax(__uint_as_float(a0), __uint_as_float(a1));
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  const unsigned b0 = q[0], b1 = q[1];
  return vmax(__uint_as_float(b0), __uint_as_float(b1));
}

// DROP (training): dropout of the attention probabilities (conformer.py:249).  The mask multiplies the NORMALISED
// probabilities, so the row sums keep the undropped values and only the P.V operands are masked; the 1 / (1 - p) goes on
// the output.  Element index of P[b, h, i, j]: ((b*H + h)*T + i) * pad8(T) + j (the layout of the backward's probability
// buffer).
// PROBS (backward pass): the probabilities themselves, P = exp(score - lse) from the training forward's log-sum-exp, as
// bf16 rows of `out` = [B, H, T, ldp] (and with DROP the masked, rescaled copy `pdrop`, the operand of dV) - no online
// softmax, no P.V; same staging, band pipeline and tiling.
// BWD (with PROBS): the softmax backward and the first input gradient in the same pass, while the probabilities are in
// registers - dP^T = V dO^T (the V tile is already staged), dS = P o (mask / (1 - p) o dP - D), dS out row-major and skewed
// (as attn_ds_kernel, which read P back from memory), and dq1^T += K^T dS^T exactly as O^T += V^T P^T in the forward (K^T
// fragments by transposing reads of the staged K tile).  Of the probabilities only the operand of dV is written (the
// dropped ones with DROP).  One launch for four (probabilities, attn_rowdot's consumer attn_ds, dS k).
struct MhaBwdArgs {
  const unsigned short* dout;  // bf16 [B*T, H*64]
  const float* dvec;           // f32 [B, H, T]: rowsum(dO o O)
  unsigned short* ds;          // bf16 [B, H, T, ldp]
  unsigned short* dsk;         // bf16 [B, T, H, W], zero outside the band (caller)
  float* dq1;                  // f32 [B, H, T, 64] = dS k
  int W;
  const unsigned short* ao;    // bf16 [B*T, H*64]: the forward's output; non-null: D is computed here (dvec unused)
  unsigned short* q_rows;      // optional bf16 [B*T*H, 64]: the q rows in ds_skew's row order (operand of d(pe_window))
};

// LOG2Q (inference only: no DROP, no PROBS, no lse): q arrives pre-scaled by log2(e) / sqrt(d_k) (the caller folded the
// factor into the q rows of the projection), so scores and position band come out of the MFMAs in log2 units, and the
// band accumulators start at MINUS the running reference (m_run, in log2 units) instead of zero - the band of tile j+1 is
// computed after tile j's maximum is known.  The exponent argument is then the accumulator itself: the 32 fused
// multiply-adds per lane and tile are gone, and so is the cross-lane maximum unless some lane's scores exceed the
// reference (then, in a wave-uniform branch, the classic step: the reference moves up, what has been accumulated is
// rescaled and the tile's scores are brought onto the new reference).  Tile 0 establishes the reference explicitly.
template <int NW, bool DROP, bool PROBS = false, bool BWD = false, bool LOG2Q = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void mha_fwd_kernel(const unsigned short* __restrict__ qkv,
                                                            const bf16x8* __restrict__ pe_packed, int q_min, int q_max,
                                                            unsigned short* __restrict__ out, float* __restrict__ lse,
                                                            int BH, int T, int H, int G, int ppw,
                                                            unsigned long long* stamps, csship_dropout_cfg dcfg,
                                                            unsigned short* __restrict__ pdrop = nullptr, int ldp = 0,
                                                            MhaBwdArgs bw = MhaBwdArgs{}) {
  static_assert(!BWD || PROBS, "the backward pass recomputes the probabilities");
  static_assert(!LOG2Q || (!DROP && !PROBS), "LOG2Q is the inference form");
  constexpr int CH = 512 / (NW * 64);  // 16-byte chunks of a K (and of a V) tile staged per thread
  constexpr int R = 2 * NW + 8;        // ring slots (position blocks of 2 x 1 KB): 2 NW + 4 live + 4 arriving
  constexpr int NF = 8 / NW;           // new fragments staged per wave and tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * A2_KV + R * 2048 + NW * A2_PQ * 4];
  unsigned char* const sK = smem;                   // [2][64 x 128 B]
  unsigned char* const sV = smem + 2 * A2_KV;       // [2][64 x 128 B]
  unsigned char* const sR = smem + 4 * A2_KV;       // [R][2 k-steps][64 lanes x 16 B]
  float* const sP = reinterpret_cast<float*>(smem + 4 * A2_KV + R * 2048);
  (void)sP;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  int bh, gi;
  {
    const int id = blockIdx.x;
    if ((BH & 7) == 0) {  // ids equal mod 8 share an XCD: give them to one (b, h)
      const int per = 8 * G, blk = id / per, r = id - blk * per;
      bh = blk * 8 + (r & 7);
      gi = r >> 3;
    } else {
      bh = id / G;
      gi = id - bh * G;
    }
  }
  const int b = bh / H, h = bh - b * H;
  const int D = H * 64, D3 = 3 * D;
  const size_t base_b = (size_t)b * T;
  const int P = (T + 31) >> 5;
  const int pair = gi * ppw + w;
  const bool owner = (w < ppw) && (pair < P);                   // wave-uniform; the others repeat a neighbour's work
  const int i0g = 32 * gi * ppw;                                // first query of the workgroup
  const int i0w = 32 * min(gi * ppw + min(w, ppw - 1), P - 1);  // and only help with staging and barriers

  bf16x8 qf[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qi = min(i0w + 16 * qb + n, T - 1);
    const unsigned short* qrow = qkv + (base_b + qi) * D3 + h * 64;
    qf[qb][0] = *reinterpret_cast<const bf16x8*>(qrow + 8 * g);
    qf[qb][1] = *reinterpret_cast<const bf16x8*>(qrow + 32 + 8 * g);
  }

  bf16x8 dof[2][2];   // BWD: dO rows of the two query blocks (B operand, like q), D of this lane's two queries
  float dval[2] = {0.f, 0.f};
  f32x4 dq[2][4];
  if constexpr (BWD) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qi = min(i0w + 16 * qb + n, T - 1);
      const unsigned short* drow = bw.dout + (base_b + qi) * D + h * 64;
      dof[qb][0] = *reinterpret_cast<const bf16x8*>(drow + 8 * g);
      dof[qb][1] = *reinterpret_cast<const bf16x8*>(drow + 32 + 8 * g);
      if (bw.ao != nullptr) {  // D = dO . O over the 64 features of the head: 16 per lane, then the four lanes of the query
        const unsigned short* orow = bw.ao + (base_b + qi) * D + h * 64;
        const bf16x8 o0 = *reinterpret_cast<const bf16x8*>(orow + 8 * g), o1 = *reinterpret_cast<const bf16x8*>(orow + 32 + 8 * g);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)dof[qb][0][e] * (float)o0[e] + (float)dof[qb][1][e] * (float)o1[e];
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        dval[qb] = acc;
      } else {
        dval[qb] = bw.dvec[(size_t)bh * T + qi];
      }
      if (bw.q_rows != nullptr && owner && i0w + 16 * qb + n < T) {
        unsigned short* qr = bw.q_rows + ((base_b + qi) * H + h) * 64;
        *reinterpret_cast<bf16x8*>(qr + 8 * g) = qf[qb][0];
        *reinterpret_cast<bf16x8*>(qr + 32 + 8 * g) = qf[qb][1];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) dq[qb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  // K/V staging: chunk idx = tid + c * NW*64 -> (row = idx >> 3, 16-byte chunk = idx & 7)
  const int sc = tid & 7;
  uint4 kreg0, kreg1, vreg0, vreg1;  // named registers: small arrays written under `if (more)` are demoted to scratch
  kreg1 = vreg1 = make_uint4(0u, 0u, 0u, 0u);
#define A2_LOAD_KV(J0)                                                                                     \
  {                                                                                                        \
    const unsigned short* s_ = qkv + (base_b + min((J0) + (tid >> 3), T - 1)) * D3 + D + h * 64 + sc * 8;  \
    kreg0 = *reinterpret_cast<const uint4*>(s_);                                                           \
    vreg0 = *reinterpret_cast<const uint4*>(s_ + D);                                                       \
    if constexpr (CH > 1) {                                                                                \
      s_ = qkv + (base_b + min((J0) + (tid >> 3) + NW * 8, T - 1)) * D3 + D + h * 64 + sc * 8;             \
      kreg1 = *reinterpret_cast<const uint4*>(s_);                                                         \
      vreg1 = *reinterpret_cast<const uint4*>(s_ + D);                                                     \
    }                                                                                                      \
  }
#define A2_STORE_KV(BUF)                                                                                   \
  {                                                                                                        \
    *reinterpret_cast<uint4*>(sK + (BUF) * A2_KV + tile_off(tid >> 3, sc)) = kreg0;                        \
    *reinterpret_cast<uint4*>(sV + (BUF) * A2_KV + v_off(tid >> 3, sc)) = vreg0;                           \
    if constexpr (CH > 1) {                                                                                \
      *reinterpret_cast<uint4*>(sK + (BUF) * A2_KV + tile_off((tid >> 3) + NW * 8, sc)) = kreg1;           \
      *reinterpret_cast<uint4*>(sV + (BUF) * A2_KV + v_off((tid >> 3) + NW * 8, sc)) = vreg1;              \
    }                                                                                                      \
  }
  // ---- the position-block ring.  Block q (16 reversed rows of pe_k, clamp baked in; csship.h) sits in slot q mod R
  // as two lane-linear 1-KB fragments.  At key tile j0 the workgroup reads blocks qlo .. qlo + 2 NW + 3,
  // qlo = (i0g - j0)/16 - 4; blocks qlo - 4 .. qlo - 1 are the next tile's newcomers.
  const bf16x8* pe_lane = pe_packed + lane;
  auto ring_slot = [](int q) { const int m = q % R; return m < 0 ? m + R : m; };
  auto pe_src = [&](int q, int ks) {
    const int qi_ = min(max(q, q_min), q_max) - q_min;
    return pe_lane + (size_t)(qi_ * 2 + ks) * 64;
  };
  bf16x8 nf0, nf1;  // staged newcomers (NW = 8: one fragment per wave; NW = 4: one block = two fragments)
  nf1 = bf16x8{};
#define A2_LOAD_RING(J0) /* the four blocks below tile J0's window... i.e. tile J0's own lowest four */    \
  {                                                                                                        \
    const int qn_ = (i0g - (J0)) / 16 - 4;                                                                 \
    if constexpr (NF == 1) nf0 = *pe_src(qn_ + (w >> 1), w & 1);                                           \
    else { nf0 = *pe_src(qn_ + w, 0); nf1 = *pe_src(qn_ + w, 1); }                                         \
  }
#define A2_STORE_RING(J0)                                                                                  \
  {                                                                                                        \
    const int qn_ = (i0g - (J0)) / 16 - 4;                                                                 \
    if constexpr (NF == 1)                                                                                 \
      *reinterpret_cast<bf16x8*>(sR + (ring_slot(qn_ + (w >> 1)) * 2 + (w & 1)) * 1024 + lane * 16) = nf0; \
    else {                                                                                                 \
      *reinterpret_cast<bf16x8*>(sR + (ring_slot(qn_ + w) * 2) * 1024 + lane * 16) = nf0;                  \
      *reinterpret_cast<bf16x8*>(sR + (ring_slot(qn_ + w) * 2 + 1) * 1024 + lane * 16) = nf1;              \
    }                                                                                                      \
  }

  f32x4 o[2][4], lsum[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[qb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    lsum[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  Drop dp{0u, 0u, 1.f};
  unsigned ebase[2] = {0u, 0u};  // element index of P[b, h, this lane's query of block qb, 4g]
  if constexpr (DROP) {
    dp = drop_make(dcfg.seed, dcfg.stream, dcfg.p);
    const unsigned Tp = (unsigned)((T + 7) / 8 * 8);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
      ebase[qb] = ((unsigned)bh * (unsigned)T + (unsigned)min(i0w + 16 * qb + n, T - 1)) * Tp + 4u * (unsigned)g;
  }
  float m_run[2] = {LOG2Q ? 0.f : -1e30f, LOG2Q ? 0.f : -1e30f};
  float lse2[2] = {0.f, 0.f};  // PROBS: log-sum-exp of this lane's two queries, in log2 units
  if constexpr (PROBS) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) lse2[qb] = lse[(size_t)bh * T + min(i0w + 16 * qb + n, T - 1)] * LOG2E;
  }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  // Band scratch of this wave, laid out for the CONSUMER: query n's 64 values of a key tile as [key block tc][n][16],
  // so lane (g, n) reads its accumulator registers of key block tc with one 16-byte load at tc*A2_PB + A2_PS n + 4 g.
  // The producer (band tile t, lane (g, n), register r) holds relative offset c = 16 t + 4 g + r, which is key
  // k = c - (15 - n) of the tile: it stores to (k >> 4) * A2_PB + A2_PS n + (k & 15) = base_r + A2_PB t with a per-lane,
  // per-register base that is fixed for the whole kernel.  Of the five band tiles exactly one of t = 0 / t = 4 falls
  // outside the 64 keys for every (lane, r): those stores land in a pad block below / above the four key blocks.
  float* const Pw = sP + w * A2_PQ + A2_PB;
  float* pwr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k0 = 4 * g + r + n - 15;  // key (may be negative) of band tile 0's value
    pwr[r] = Pw + A2_PS * n + (k0 >= 0 ? k0 : k0 + 16 - A2_PB);
  }
  const float* const prd = Pw + A2_PS * n + 4 * g;
  const int q4 = n >> 2, p4 = n & 3;

  // band of key tile J0 for both query blocks -> scratch -> s (the S^T accumulators' initial value).  One query block
  // after the other through the SAME scratch: LDS operations of one wave execute in order, so the skew reads see the
  // stores before them and the second block's stores cannot overtake the first block's reads.
#define A2_BAND()                                                                                                        \
  {                                                                                                                      \
    /* block u = 0..5 of the pair sits in slot rs0 - u (mod R); query block qb uses u = t + 1 - qb */                    \
    bf16x8 pf[6][2];                                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 6; ++u) {                                                                      \
      const int sl_ = rs0 - u + ((rs0 - u) < 0 ? R : 0);                                                                 \
      const unsigned char* fr = sR + sl_ * 2048 + lane * 16;                                                             \
      pf[u][0] = *reinterpret_cast<const bf16x8*>(fr);                                                                   \
      pf[u][1] = *reinterpret_cast<const bf16x8*>(fr + 1024);                                                            \
    }                                                                                                                    \
    rs0 = rs0 - 4 + (rs0 < 4 ? R : 0); /* the next tile's window sits four blocks lower */                               \
    f32x4 pb[2][5];                                                                                                      \
    _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                                                   \
      const float bi_ = LOG2Q ? -m_run[qb] : 0.f;                                                                        \
      _Pragma("unroll") for (int t = 0; t < 5; ++t) pb[qb][t] = f32x4{bi_, bi_, bi_, bi_};                               \
      _Pragma("unroll") for (int t = 0; t < 5; ++t)                                                                      \
        pb[qb][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[t + 1 - qb][0], qf[qb][0], pb[qb][t], 0, 0, 0);           \
      _Pragma("unroll") for (int t = 0; t < 5; ++t)                                                                      \
        pb[qb][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[t + 1 - qb][1], qf[qb][1], pb[qb][t], 0, 0, 0);           \
    }                                                                                                                    \
    _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                                                   \
      /* the skew is done by the STORE: each band value goes straight to the place its consumer's 16-byte read takes */ \
      _Pragma("unroll") for (int t = 0; t < 5; ++t)                                                                      \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) pwr[r][A2_PB * t] = pb[qb][t][r];                                    \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) s[qb][t] = *reinterpret_cast<const f32x4*>(prd + A2_PB * t);          \
    }                                                                                                                    \
  }
#define A2_BAND_WAIT() {}

  stamp(stamps, 0);
  A2_LOAD_KV(0)
  {  // the whole ring: the windows of key tiles 0 and 64 (the band runs one tile ahead), 2 R fragments, NW at a time
    const int qlo = i0g / 16 - 8;
#pragma unroll
    for (int f0 = 0; f0 < 2 * R; f0 += NW) {
      const int f = f0 + w;
      *reinterpret_cast<bf16x8*>(sR + (ring_slot(qlo + (f >> 1)) * 2 + (f & 1)) * 1024 + lane * 16) = *pe_src(qlo + (f >> 1), f & 1);
    }
  }
  A2_STORE_KV(0)
  __syncthreads();
  stamp(stamps, 1);

  // Software pipeline: the band of tile j+1 depends on nothing but q and the ring, so it is computed DURING tile j's
  // softmax (matrix work beside vector work) and parked in s as the next S^T's initial value while P.V runs.
  if ((A2_ABL & 32) && w >= NW / 2) return;  // diagnostic (with bit 3): one wave per SIMD
#ifdef A2_DELAY
  if (w >= NW / 2) __builtin_amdgcn_s_sleep(A2_DELAY);  // diagnostic (with bit 3): the younger half runs out of phase
#endif
  f32x4 s[2][4];
  int rs0 = ring_slot(i0w / 16 + 1);  // slot of the pair's highest block at the tile the NEXT A2_BAND works on
  A2_BAND()
  int cur = 0;
#ifdef A2_STAMPS
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_)::"memory");
#endif
  // the tiles that have a successor, then the last one on its own copy of the body: at T = 400 it holds 16 keys of 64,
  // and three quarters of its S products, exponentials and P.V k-steps, and the band of a tile that does not exist,
  // were a seventh of the loop spent on zeros
  for (int j0 = 0; j0 + 64 < T; j0 += 64, cur ^= 1) {
#define A2_LAST 0
#include "attn_tile.inc"
#undef A2_LAST
  }
  {
    const int j0 = ((T - 1) / 64) * 64;
#define A2_LAST 1
#include "attn_tile.inc"
#undef A2_LAST
  }
#ifdef A2_STAMPS
  if (stamps != nullptr && lane == 0)
    for (int i = 0; i < 8; ++i) stamps[((size_t)blockIdx.x * NW + w) * 8 + i] = acc_[i];
#endif
#undef A2_BAND
#undef A2_LOAD_KV
#undef A2_STORE_KV
#undef A2_LOAD_RING
#undef A2_STORE_RING

  stamp(stamps, 2);
  if constexpr (BWD) {
    if (owner) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int qi = i0w + 16 * qb + n;
        if (qi < T) {
          float* row = bw.dq1 + ((size_t)bh * T + qi) * 64 + 4 * g;  // dq1^T rows d = 16 dt + 4 g + r of column i = n
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(row + 16 * dt) = dq[qb][dt];
        }
      }
    }
  }
  if (!owner || PROBS) return;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l = lsum[qb][0];
    const float inv = dp.scale / l;  // (dropout: survivors carry 1 / (1 - p))
    const int qi = i0w + 16 * qb + n;
    if (lse != nullptr && qi < T && g == 0) lse[((size_t)b * H + h) * T + qi] = m_run[qb] + __logf(l);
    if (qi < T) {
      unsigned short* orow = out + (base_b + qi) * D + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 v;
        v.x = pack_bf2(o[qb][dt][0] * inv, o[qb][dt][1] * inv);
        v.y = pack_bf2(o[qb][dt][2] * inv, o[qb][dt][3] * inv);
        *reinterpret_cast<uint2*>(orow + 16 * dt + 4 * g) = v;
      }
    }
  }
}

// Host side of mha_fwd_kernel: pairs of 16-query blocks, dealt to workgroups of 8 (or 4) waves.
static int launch_mha_fwd(const uint16_t* qkv, const uint16_t* pe_packed, int q_min, int q_max, uint16_t* out, float* lse,
                          int B, int T, int H, unsigned long long* stamps, hipStream_t s,
                          const csship_dropout_cfg* drop = nullptr, bool log2q = false) {
  const int P = (T + 31) / 32, BH = B * H;
  // 8 waves unless 4-wave workgroups are needed to give every CU one (short utterances, small batches)
  const int G8 = (P + 7) / 8, G4 = (P + 3) / 4;
  const bool four = (BH * G8 <= 128) && (G4 > G8);  // (a 4-wave workgroup's 84 KB of LDS leaves one workgroup per CU)
  const int G = four ? G4 : G8, ppw = (P + G - 1) / G;
  const long long nwg = (long long)BH * G;
  if (nwg > 0x7fffffffLL) return CSSHIP_EBADSHAPE;
  const bool dropping = drop != nullptr && drop->seed != nullptr && drop->p > 0.f;
  if (dropping && (int64_t)BH * T * ((T + 7) / 8 * 8) > 0xffffffffLL) return CSSHIP_EBADSHAPE;  // 32-bit element indices
  const csship_dropout_cfg dc = dropping ? *drop : csship_dropout_cfg{nullptr, 0u, 0.f};
  const bf16x8* pe = reinterpret_cast<const bf16x8*>(pe_packed);
  if (log2q && (dropping || lse != nullptr)) return CSSHIP_EUNSUPPORTED;  // the log2-domain form is the inference form
  if (log2q && four)
    hipLaunchKernelGGL((mha_fwd_kernel<4, false, false, false, true>), dim3((unsigned)nwg), dim3(256), 0, s, qkv, pe, q_min, q_max, out, lse, BH, T, H, G, ppw, stamps, dc);
  else if (log2q)
    hipLaunchKernelGGL((mha_fwd_kernel<8, false, false, false, true>), dim3((unsigned)nwg), dim3(512), 0, s, qkv, pe, q_min, q_max, out, lse, BH, T, H, G, ppw, stamps, dc);
  else if (four && dropping)
    hipLaunchKernelGGL((mha_fwd_kernel<4, true>), dim3((unsigned)nwg), dim3(256), 0, s, qkv, pe, q_min, q_max, out, lse, BH, T, H, G, ppw, stamps, dc);
  else if (four)
    hipLaunchKernelGGL((mha_fwd_kernel<4, false>), dim3((unsigned)nwg), dim3(256), 0, s, qkv, pe, q_min, q_max, out, lse, BH, T, H, G, ppw, stamps, dc);
  else if (dropping)
    hipLaunchKernelGGL((mha_fwd_kernel<8, true>), dim3((unsigned)nwg), dim3(512), 0, s, qkv, pe, q_min, q_max, out, lse, BH, T, H, G, ppw, stamps, dc);
  else
    hipLaunchKernelGGL((mha_fwd_kernel<8, false>), dim3((unsigned)nwg), dim3(512), 0, s, qkv, pe, q_min, q_max, out, lse, BH, T, H, G, ppw, stamps, dc);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

// the probabilities of the backward pass through the same kernel (PROBS)
static int launch_mha_probs(const uint16_t* qkv, const uint16_t* pe_packed, int q_min, int q_max, const float* lse,
                            uint16_t* p, uint16_t* pdrop, int64_t ldp, int B, int T, int H, hipStream_t s,
                            const csship_dropout_cfg* drop) {
  const int P = (T + 31) / 32, BH = B * H;
  const int G8 = (P + 7) / 8, G4 = (P + 3) / 4;
  const bool four = (BH * G8 <= 128) && (G4 > G8);
  const int G = four ? G4 : G8, ppw = (P + G - 1) / G;
  const long long nwg = (long long)BH * G;
  if (nwg > 0x7fffffffLL || ldp > 0x7fffffffLL) return CSSHIP_EBADSHAPE;
  const bool dropping = drop != nullptr && drop->seed != nullptr && drop->p > 0.f && pdrop != nullptr;
  if ((int64_t)BH * T * ldp > 0xffffffffLL && dropping) return CSSHIP_EBADSHAPE;  // 32-bit element indices of the mask
  const csship_dropout_cfg dc = dropping ? *drop : csship_dropout_cfg{nullptr, 0u, 0.f};
  const bf16x8* pe = reinterpret_cast<const bf16x8*>(pe_packed);
  float* lse_ = const_cast<float*>(lse);
#define A2_PROBS(NWV, DR) hipLaunchKernelGGL((mha_fwd_kernel<NWV, DR, true>), dim3((unsigned)nwg), dim3(NWV * 64), 0, s, qkv, pe, \
                                             q_min, q_max, p, lse_, BH, T, H, G, ppw, nullptr, dc, pdrop, (int)ldp)
  if (four && dropping) A2_PROBS(4, true);
  else if (four) A2_PROBS(4, false);
  else if (dropping) A2_PROBS(8, true);
  else A2_PROBS(8, false);
#undef A2_PROBS
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

// probabilities + softmax backward + dS k in one pass (BWD)
static int launch_mha_bwd(const uint16_t* qkv, const uint16_t* pe_packed, int q_min, int q_max, const float* lse,
                          const MhaBwdArgs& bw, uint16_t* p_v, int64_t ldp, int B, int T, int H, hipStream_t s,
                          const csship_dropout_cfg* drop) {
  const int P = (T + 31) / 32, BH = B * H;
  const int G8 = (P + 7) / 8, G4 = (P + 3) / 4;
  const bool four = (BH * G8 <= 128) && (G4 > G8);
  const int G = four ? G4 : G8, ppw = (P + G - 1) / G;
  const long long nwg = (long long)BH * G;
  if (nwg > 0x7fffffffLL || ldp > 0x7fffffffLL) return CSSHIP_EBADSHAPE;
  const bool dropping = drop != nullptr && drop->seed != nullptr && drop->p > 0.f;
  if ((int64_t)BH * T * ldp > 0xffffffffLL && dropping) return CSSHIP_EBADSHAPE;  // 32-bit element indices of the mask
  const csship_dropout_cfg dc = dropping ? *drop : csship_dropout_cfg{nullptr, 0u, 0.f};
  const bf16x8* pe = reinterpret_cast<const bf16x8*>(pe_packed);
  float* lse_ = const_cast<float*>(lse);
#define A2_BWD(NWV, DR) hipLaunchKernelGGL((mha_fwd_kernel<NWV, DR, true, true>), dim3((unsigned)nwg), dim3(NWV * 64), 0, s, qkv, pe,                                            q_min, q_max, p_v, lse_, BH, T, H, G, ppw, nullptr, dc, p_v, (int)ldp, bw)
  if (four && dropping) A2_BWD(4, true);
  else if (four) A2_BWD(4, false);
  else if (dropping) A2_BWD(8, true);
  else A2_BWD(8, false);
#undef A2_BWD
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
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
                                                         int B, int T, int H, int64_t ldp, int W, csship_dropout_cfg dcfg) {
  __shared__ __attribute__((aligned(16))) unsigned char sV[64 * 128];
  const Drop dp = drop_make(dcfg.seed, dcfg.stream, dcfg.p);  // dropout of the probabilities: dP = mask / (1 - p) o (dO v^T)
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
        const float4 dpv = drop_apply4(dp, (unsigned)(prow + j), make_float4(s[t][0], s[t][1], s[t][2], s[t][3]));
        const unsigned short d0 = f2bf(p0 * (dpv.x - dval)), d1 = f2bf(p1 * (dpv.y - dval));
        const unsigned short d2 = f2bf(p2 * (dpv.z - dval)), d3 = f2bf(p3 * (dpv.w - dval));
        *reinterpret_cast<uint2*>(ds + prow + j) = make_uint2((unsigned)d0 | ((unsigned)d1 << 16), (unsigned)d2 | ((unsigned)d3 << 16));
        // skewed copy: keys j .. j+3 are positions r .. r-3 of the query's row, i.e. ONE 8-byte piece (d3 d2 d1 d0) that is
        // only 2-byte aligned (its start is i + T - 4 - j): an unaligned global_store_dwordx2 instead of four 2-byte stores
        if (j + 3 < T) {
          struct __attribute__((packed, aligned(2))) h4u { unsigned short v[4]; };
          h4u sk;
          sk.v[0] = d3; sk.v[1] = d2; sk.v[2] = d1; sk.v[3] = d0;
          *reinterpret_cast<h4u*>(skrow - j - 3) = sk;
        } else {
          if (j < T) skrow[-j] = d0;
          if (j + 1 < T) skrow[-j - 1] = d1;
          if (j + 2 < T) skrow[-j - 2] = d2;
        }
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
  return csship_mha_relpos_ex(qkv, pe_packed, q_min, q_max, out, B, T, H, 0, stream);
}

extern "C" int csship_mha_relpos_ex(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                    uint16_t* out, int32_t B, int32_t T, int32_t H, int32_t flags, csship_stream_t stream) {
  if (!qkv || !pe_packed || !out) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0 || q_max < q_min || B > 65535 || H > 65535) return CSSHIP_EBADSHAPE;
  if (flags & ~CSSHIP_MHA_Q_LOG2E) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(qkv) || !aligned16(pe_packed) || !aligned8(out)) return CSSHIP_EALIGN;
  return launch_mha_fwd(qkv, pe_packed, q_min, q_max, out, nullptr, B, T, H, g_csship_stamps, as_stream(stream), nullptr,
                        (flags & CSSHIP_MHA_Q_LOG2E) != 0);
}

extern "C" int csship_mha_relpos_train(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                       uint16_t* out, float* lse, int32_t B, int32_t T, int32_t H,
                                       csship_stream_t stream) {
  return csship_mha_relpos_train_drop(qkv, pe_packed, q_min, q_max, out, lse, B, T, H, nullptr, stream);
}

extern "C" int csship_mha_relpos_train_drop(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                            uint16_t* out, float* lse, int32_t B, int32_t T, int32_t H,
                                            const csship_dropout_cfg* drop, csship_stream_t stream) {
  if (!qkv || !pe_packed || !out || !lse) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0 || q_max < q_min || B > 65535 || H > 65535) return CSSHIP_EBADSHAPE;
  if (!aligned16(qkv) || !aligned16(pe_packed) || !aligned8(out)) return CSSHIP_EALIGN;
  TimelineScope tl(CSSHIP_TL_ATTN, as_stream(stream));
  return launch_mha_fwd(qkv, pe_packed, q_min, q_max, out, lse, B, T, H, nullptr, as_stream(stream), drop);
}

extern "C" int csship_mha_relpos_probs(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                       const float* lse, uint16_t* p_bf16, int64_t ld_p, int32_t B, int32_t T, int32_t H,
                                       csship_stream_t stream) {
  return csship_mha_relpos_probs_drop(qkv, pe_packed, q_min, q_max, lse, p_bf16, nullptr, ld_p, B, T, H, nullptr, stream);
}

extern "C" int csship_mha_relpos_probs_drop(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                            const float* lse, uint16_t* p_bf16, uint16_t* p_dropped_bf16, int64_t ld_p,
                                            int32_t B, int32_t T, int32_t H, const csship_dropout_cfg* drop,
                                            csship_stream_t stream) {
  if (!qkv || !pe_packed || !lse || !p_bf16) return CSSHIP_ENULL;
  const bool dropping = drop && drop->seed && drop->p > 0.f;
  if (dropping && (!p_dropped_bf16 || ld_p != (T + 7) / 8 * 8 || (int64_t)B * H * T * ld_p > 0xffffffffLL)) return CSSHIP_EBADSHAPE;
  if (p_dropped_bf16 && !aligned8(p_dropped_bf16)) return CSSHIP_EALIGN;
  if (B <= 0 || T <= 0 || H <= 0 || q_max < q_min || B > 65535 || H > 65535 || ld_p < T) return CSSHIP_EBADSHAPE;
  if (!aligned16(qkv) || !aligned16(pe_packed) || !aligned8(p_bf16) || (ld_p % 4) != 0) return CSSHIP_EALIGN;
  return launch_mha_probs(qkv, pe_packed, q_min, q_max, lse, p_bf16, dropping ? p_dropped_bf16 : nullptr, ld_p, B, T, H,
                          as_stream(stream), drop);
}

extern "C" int csship_mha_relpos_bwd_core(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                          const float* lse, const uint16_t* d_out, const float* d, const uint16_t* out,
                                          uint16_t* p_v_bf16, uint16_t* ds_bf16, uint16_t* ds_skew_bf16, float* dq1,
                                          uint16_t* q_rows, int64_t ld_p, int32_t W, int32_t B, int32_t T, int32_t H,
                                          const csship_dropout_cfg* drop, csship_stream_t stream) {
  if (!qkv || !pe_packed || !lse || !d_out || (!d && !out) || !p_v_bf16 || !ds_bf16 || !dq1) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0 || q_max < q_min || B > 65535 || H > 65535 || ld_p < T || (ld_p & 3) ||
      (ds_skew_bf16 && W < 2 * T - 1))
    return CSSHIP_EBADSHAPE;
  if (drop && drop->seed && drop->p > 0.f && ld_p != (T + 7) / 8 * 8) return CSSHIP_EBADSHAPE;
  if (!aligned16(qkv) || !aligned16(pe_packed) || !aligned16(d_out) || !aligned8(p_v_bf16) || !aligned8(ds_bf16) ||
      !aligned16(dq1))
    return CSSHIP_EALIGN;
  if ((out && !aligned16(out)) || (q_rows && !aligned16(q_rows))) return CSSHIP_EALIGN;
  const MhaBwdArgs bw{d_out, d, ds_bf16, ds_skew_bf16, dq1, W, out, q_rows};
  return launch_mha_bwd(qkv, pe_packed, q_min, q_max, lse, bw, p_v_bf16, ld_p, B, T, H, as_stream(stream), drop);
}

extern "C" int csship_attn_softmax_bwd(const uint16_t* qkv, const uint16_t* d_out, const uint16_t* p_bf16, int64_t ld_p,
                                       const float* d, uint16_t* ds_bf16, uint16_t* ds_skew_bf16, int32_t W, int32_t B,
                                       int32_t T, int32_t H, csship_stream_t stream) {
  return csship_attn_softmax_bwd_drop(qkv, d_out, p_bf16, ld_p, d, ds_bf16, ds_skew_bf16, W, B, T, H, nullptr, stream);
}

extern "C" int csship_attn_softmax_bwd_drop(const uint16_t* qkv, const uint16_t* d_out, const uint16_t* p_bf16, int64_t ld_p,
                                            const float* d, uint16_t* ds_bf16, uint16_t* ds_skew_bf16, int32_t W, int32_t B,
                                            int32_t T, int32_t H, const csship_dropout_cfg* drop, csship_stream_t stream) {
  const bool dropping = drop && drop->seed && drop->p > 0.f;
  if (dropping && (ld_p != (T + 7) / 8 * 8 || (int64_t)B * H * T * ld_p > 0xffffffffLL)) return CSSHIP_EBADSHAPE;
  if (!qkv || !d_out || !p_bf16 || !d || !ds_bf16 || !ds_skew_bf16) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0 || B > 65535 || H > 65535 || ld_p < T || W < 2 * T - 1) return CSSHIP_EBADSHAPE;
  if (!aligned16(qkv) || !aligned16(d_out) || !aligned8(p_bf16) || !aligned8(ds_bf16) || (ld_p % 4) != 0) return CSSHIP_EALIGN;
  dim3 grid((T + 63) / 64, H, B);
  hipLaunchKernelGGL(attn_ds_kernel, grid, dim3(256), 0, as_stream(stream), qkv, d_out, p_bf16, d, ds_bf16, ds_skew_bf16, B, T,
                     H, ld_p, W, dropping ? *drop : csship_dropout_cfg{nullptr, 0u, 0.f});
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
