// Mask head + mask application + PIT-MSE as ONE kernel for the loss-only (evaluation / validation) path:
//     m_k = ReLU(Linear(x))[:, k*F:(k+1)*F]      (css/models/conformer.py:98-101)
//     y_k = m_k * f_orig                          (conformer.py:103)
//     D[b][k][j] = sum_{t,f} (y_k - target_j)^2   (css/objectives/pit.py:53-57 looped over the batch, mse.py:31-37)
// The K+1 mask tensors (39.5 MB at 32 x 400 x 771 f32) are never written: the unfused pair (panel_linear head +
// pit_sse) wrote them and read them straight back - 79 MB of round trip, more than everything else both kernels
// touch.  The noise mask does not enter the loss and is not computed at all (K * F of the (K + 1) * F head rows).
//
// Row-panel kernel like panel.hip (64 token rows per workgroup, transposed products, fragment-packed weights streamed
// global -> VGPR), with the head's rows PERMUTED on the host (css_amd.ops.pack_head_pit): pass p < K covers bins 0..255
// of speaker p, one last tile holds bin 256 of every speaker.  In the transposed accumulator a lane then owns 4
// CONSECUTIVE BINS of one token row per tile - exactly the 16-byte pieces of f_orig / the targets it needs, which it
// loads ONCE (the bins do not depend on the pass) - and the squared errors are accumulated in registers: no staging
// tile, no barrier in the pass loop.  K == 2, F == 257, d_model == 256, T >= 64 (a panel touches at most two samples).
// The kernel is bound by how fast a CU can pull its rows of x, f_orig and the targets (53 MB per launch at 32 x 400,
// ~11 B/clk per CU on 200 CUs), not by the 3.4 GFLOP of the head: 20 us, against 20.6 + 14.9 us for the two kernels it
// replaces.  Measured and rejected: 32-row panels under a 128-register budget (two workgroups per CU so that one's
// loads travel while the other multiplies): 25 us - the budget spills and every weight fragment feeds half the MFMAs.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"

namespace {

constexpr int RT = 4, BM = 16 * RT, DM = 256, KS = DM / 32, PD = 2, NF = 257, KSPK = 2;

DEV int panel_off(int row, int chunk) { return row * 512 + ((chunk ^ (row & 15)) << 4); }

struct __attribute__((packed, aligned(4))) f4u { float v[4]; };  // 16-byte access that is only 4-byte aligned (257-float rows)
DEV f4u load4(const float* p) { return *reinterpret_cast<const f4u*>(p); }

struct HeadPitArgs {
  const float* x;        // encoder output f32 [M, 256]
  const bf16x8* w;       // permuted head rows in fragment order: (K * 16 + 1) tiles x 8 k-steps x 64 lanes
  const float* bias;     // permuted the same way, (K * 16 + 1) * 16 floats
  const float* f;        // f_orig f32 [M, ldf]
  const float* tgt[KSPK];  // targets f32 [M, ldt]
  float* partial;        // [n_workgroups][2 slots][K * K]
  int64_t ldf, ldt;
  int M, T;
  int rpw;               // token rows per workgroup (<= BM; head_pit_split_kernel): rows rpw.. of the 64-row panel are padding
};

// One thread per sample adds its panels' partials (ascending panel order, double), scans the K! = 2 permutations in
// itertools order (first minimum wins, like torch.min), then thread 0 averages the samples in order.  Called by all
// nth threads of one workgroup; best_s: 1024 floats of LDS.
DEV void pit_finalize_body(const float* __restrict__ partial, int B, int T, int M, double norm, float* __restrict__ loss,
                           int* __restrict__ perm_out, float* __restrict__ pair_sse, int rpw, int tix, int nth, float* best_s) {
  constexpr int K = KSPK;
  for (int b = tix; b < B; b += nth) {
    double Dm[K][K] = {{0.0, 0.0}, {0.0, 0.0}};
    const int wg0 = (b * T) / rpw, wg1 = min(((b + 1) * T - 1) / rpw, (M - 1) / rpw);
    for (int w0 = wg0; w0 <= wg1; w0 += 8) {  // eight panels' partials in flight at a time, added in ascending order
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int wg = min(w0 + u, wg1), sl = b - (wg * rpw) / T;
        v[u] = *reinterpret_cast<const float4*>(partial + ((size_t)wg * 2 + sl) * K * K);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (w0 + u <= wg1) { Dm[0][0] += (double)v[u].x; Dm[0][1] += (double)v[u].y; Dm[1][0] += (double)v[u].z; Dm[1][1] += (double)v[u].w; }
    }
    if (pair_sse)
      for (int i = 0; i < K * K; ++i) pair_sse[(size_t)b * K * K + i] = (float)Dm[i / K][i % K];
    const float c0 = (float)((Dm[0][0] + Dm[1][1]) / norm), c1 = (float)((Dm[1][0] + Dm[0][1]) / norm);  // perms (0,1), (1,0)
    const bool swap = c1 < c0;
    best_s[b] = swap ? c1 : c0;
    if (perm_out) { perm_out[b * 2] = swap ? 1 : 0; perm_out[b * 2 + 1] = swap ? 0 : 1; }
  }
  __syncthreads();
  if (tix == 0) {
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += (double)best_s[b];
    *loss = (float)(s / (double)B);
  }
}


// -DHP_STAMPS: per-wave cycle stamps of this kernel's phases (tools/headpit_stamps.py, CSSHIP_HEADPIT_ONE_ROLE=1); not
// part of the product build
#ifdef HP_STAMPS
__device__ unsigned* g_hp_stamps = nullptr;
#define HP_STAMP(k) do { if (lane == 0) stamps[wave][k] = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define HP_STAMP(k) do {} while (0)
#endif

__global__ __launch_bounds__(512) void head_pit_kernel(HeadPitArgs a) {
  constexpr int NT = 2, K = KSPK;
  __shared__ __attribute__((aligned(16))) unsigned char Xs[BM * 512];
  __shared__ float red[8][2 * K * K];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef HP_STAMPS
  __shared__ unsigned stamps[8][16];
  if (tid < 128) stamps[tid >> 4][tid & 15] = 0;
#endif
  HP_STAMP(0);
  const int m0 = blockIdx.x * BM;
  const bf16x8* wbase = a.w + lane;
  bf16x8 wq[PD][NT];
#pragma unroll
  for (int k = 0; k < PD; ++k)
#pragma unroll
    for (int t = 0; t < NT; ++t) wq[k][t] = wbase[(size_t)((wave * NT + t) * KS + k) * 64];
  __builtin_amdgcn_sched_barrier(0);
  {  // encoder rows -> bf16 operand panel (LPR lanes per row: whole 128-byte lines per load instruction)
    constexpr int LPR = 512 / BM, NP = 64 / LPR;
    const int r = tid / LPR, part = tid % LPR;
    const float* xr = a.x + (size_t)min(m0 + r, a.M - 1) * DM;
    float4 v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (i * LPR + part) * 4);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c4 = i * LPR + part;
      *reinterpret_cast<uint2*>(Xs + panel_off(r, c4 >> 1) + (c4 & 1) * 8) = make_uint2(pack_bf2(v[i].x, v[i].y), pack_bf2(v[i].z, v[i].w));
    }
  }
  // this lane's pieces of f_orig and of the targets: bins (wave * 2 + t) * 16 + 4 g .. + 3 of rows m0 + 16 rt + n, the same
  // for every pass: 24 loads in flight during the first pass
  f4u fo[NT][RT], tg[K][NT][RT];
  int slot[RT];
  bool rok[RT];
  const int b0 = m0 / a.T;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = m0 + rt * 16 + n, rc = min(row, a.M - 1);
    rok[rt] = row < a.M;
    slot[rt] = rc / a.T - b0;  // 0 or 1 (T >= 64)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int bin = (wave * NT + t) * 16 + 4 * g;
      fo[t][rt] = load4(a.f + (size_t)rc * a.ldf + bin);
#pragma unroll
      for (int j = 0; j < K; ++j) tg[j][t][rt] = load4(a.tgt[j] + (size_t)rc * a.ldt + bin);
    }
  }
  // bare barrier behind lgkmcnt(0) only: __syncthreads() would also wait for the 24 loads above (vmcnt(0)) and put
  // their latency in front of the first pass instead of behind it
  HP_STAMP(1);   // x rows in LDS (this wave's part), f / target loads requested
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  HP_STAMP(2);

  float sse[2][K][K];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int j = 0; j < K; ++j) sse[s_][k][j] = 0.f;

#pragma unroll
  for (int p = 0; p < K; ++p) {
    const int tile0 = p * 16 + wave * NT;
    f32x4 acc[NT][RT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 bb = *reinterpret_cast<const float4*>(a.bias + (tile0 + t) * 16 + 4 * g);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
    }
    // next stream: the following pass's tiles, or (after the last pass) the single tail tile for wave 0
    const bf16x8* wc = wbase + (size_t)tile0 * KS * 64;
    const bf16x8* wn = p + 1 < K ? wc + (size_t)16 * KS * 64 : wbase + (size_t)(K * 16) * KS * 64;
    const int nnt = p + 1 < K ? NT : (wave == 0 ? 1 : 0);
    bf16x8 xa[2][RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) xa[0][rt] = *reinterpret_cast<const bf16x8*>(Xs + panel_off(rt * 16 + n, g));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 wcur[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) wcur[t] = wq[ks % PD][t];
      if (ks + PD < KS) {
#pragma unroll
        for (int t = 0; t < NT; ++t) wq[ks % PD][t] = wc[(size_t)(t * KS + ks + PD) * 64];
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t < nnt) wq[ks % PD][t] = wn[(size_t)(t * KS + ks + PD - KS) * 64];
      }
      if (ks + 1 < KS) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          xa[(ks + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(Xs + panel_off(rt * 16 + n, (ks + 1) * 4 + g));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcur[t], xa[ks & 1][rt], acc[t][rt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // mask p of this lane's bins -> its estimate -> squared errors against every target
    HP_STAMP(3 + 2 * p);   // pass p: MFMAs issued
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float e[K];
#pragma unroll
        for (int j = 0; j < K; ++j) e[j] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float y = fmaxf(acc[t][rt][c], 0.f) * fo[t][rt].v[c];
#pragma unroll
          for (int j = 0; j < K; ++j) {
            const float d = y - tg[j][t][rt].v[c];
            e[j] += d * d;
          }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float v0 = rok[rt] && slot[rt] == 0 ? e[j] : 0.f, v1 = rok[rt] && slot[rt] != 0 ? e[j] : 0.f;
          sse[0][p][j] += v0;
          sse[1][p][j] += v1;
        }
      }
    HP_STAMP(4 + 2 * p);   // pass p: squared errors done
  }
  // tail: bin 256 of every speaker = rows 0..K-1 of the last tile (wave 0; lanes g == 0 hold them as registers 0..K-1)
  if (wave == 0) {
    f32x4 acc[RT];
    const float4 bb = *reinterpret_cast<const float4*>(a.bias + K * 16 * 16 + 4 * g);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
    const bf16x8* wc = wbase + (size_t)(K * 16) * KS * 64;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 wcur = wq[ks % PD][0];
      if (ks + PD < KS) wq[ks % PD][0] = wc[(size_t)(ks + PD) * 64];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const bf16x8 xv = *reinterpret_cast<const bf16x8*>(Xs + panel_off(rt * 16 + n, ks * 4 + g));
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcur, xv, acc[rt], 0, 0, 0);
      }
    }
    if (g == 0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = m0 + rt * 16 + n;
        if (row < a.M) {
          const float fv = a.f[(size_t)row * a.ldf + 256];
          float tv[K];
#pragma unroll
          for (int j = 0; j < K; ++j) tv[j] = a.tgt[j][(size_t)row * a.ldt + 256];
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float y = fmaxf(acc[rt][k], 0.f) * fv;
#pragma unroll
            for (int j = 0; j < K; ++j) {
              const float d = y - tv[j];
              if (slot[rt] == 0) sse[0][k][j] += d * d;
              else sse[1][k][j] += d * d;
            }
          }
        }
      }
    }
  }
  HP_STAMP(7);     // tail tile (wave 0)
  // fixed-order reduction: lanes (DPP / permlane), waves (LDS), one partial per (workgroup, slot): no atomics
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float v = wave_sum(sse[s_][k][j]);
        if (lane == 0) red[wave][(s_ * K + k) * K + j] = v;
      }
  __syncthreads();
  if (tid < 2 * K * K) {
    float s_ = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s_ += red[w][tid];
    a.partial[(size_t)blockIdx.x * 2 * K * K + tid] = s_;
  }
  HP_STAMP(8);
#ifdef HP_STAMPS
  __syncthreads();
  if (g_hp_stamps && tid < 128) g_hp_stamps[(size_t)blockIdx.x * 128 + tid] = stamps[tid >> 4][tid & 15];
#endif
}

// ---- the same computation with the matrix work taken out from behind the loss operands (16 waves).  Phase stamps of
// the kernel above (tools/headpit_stamps.py): every wave requests its 24 pieces of f_orig / the targets and then streams
// weights - whose fragments return BEHIND those pieces (a wave's loads complete in order), so the matrix passes stand
// still until the 39.5 MB burst of all workgroups is through HBM (19 000 of a workgroup's 35 000 cycles), and the second
// pass, the tail tile and the reduction follow with the memory system idle.  Here
//   * the ReLU'd masks of both speakers go to an LDS tile [2][64 rows][256 bins] f32, and the squared errors are one
//     sweep over whole rows: every wave owns 4 rows, one 1 KB row of f_orig / a target per load instruction (instead of
//     16 rows x 64 bytes) and one 1 KB mask row per LDS read;
//   * waves 8..15 request their rows at kernel entry and do nothing else until the masks are there;
//   * waves 0..7 turn the encoder rows into the operand panel, run the two passes (two bin tiles each) and, wave 0,
//     the tail tile - and request their own rows only behind their last weight fragment.
// The matrix work runs under the burst; behind it are the error sweep and the reduction.
constexpr int MS_ROW = 1024;                       // bytes of a mask row in LDS (256 bins f32)
constexpr int MS_TILE = BM * MS_ROW;               // one speaker: 64 KB
DEV int ms_off(int row, int chunk) { return row * MS_ROW + ((chunk ^ (row & 7)) << 4); }   // 16-byte chunk 0..63

__global__ __launch_bounds__(1024) void head_pit_split_kernel(HeadPitArgs a) {
  constexpr int NT = 2, K = KSPK, LR = BM / 16;   // LR rows per wave in the error sweep
  // 32 KB operand panel + 128 KB of masks = the CU's 160 KB; the reduction's scratch lies in the dead panel
  __shared__ __attribute__((aligned(16))) unsigned char Xs[BM * 512];
  __shared__ __attribute__((aligned(16))) unsigned char Ms[K * MS_TILE];
  float (*red)[2 * K * K] = reinterpret_cast<float (*)[2 * K * K]>(Xs);
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // rpw of the panel's 64 rows are this workgroup's (12 800 rows: 50, so that 256 workgroups - one per CU - share the
  // loss operands instead of 200 pulling 64 rows each: the kernel is bound by what ONE CU can pull); the rest repeat the
  // last row in the matrix passes and are left out of the sweep
  const int rpw = a.rpw;
  const int m0 = blockIdx.x * rpw;
  const int bnd = (m0 / a.T + 1) * a.T;           // first row of the panel's second sample (T >= 64: at most two)
  float sse[2][K][K];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int j = 0; j < K; ++j) sse[s_][k][j] = 0.f;
  f4u fo[LR], tg[K][LR];                           // rows 4 wave .. 4 wave + 3, bins 4 lane .. 4 lane + 3
  auto request_rows = [&]() {
#pragma unroll
    for (int i = 0; i < LR; ++i) {
      if (wave * LR + i < rpw) {                    // wave-uniform
        const int rc = min(m0 + wave * LR + i, a.M - 1);
        fo[i] = load4(a.f + (size_t)rc * a.ldf + lane * 4);
#pragma unroll
        for (int j = 0; j < K; ++j) tg[j][i] = load4(a.tgt[j] + (size_t)rc * a.ldt + lane * 4);
      }
    }
  };

  if (wave >= 8) {
    request_rows();
    __builtin_amdgcn_s_barrier();                 // A: operand panel written (waves 0..7)
  } else {
    const bf16x8* wbase = a.w + lane;
    bf16x8 wq[PD][NT];
#pragma unroll
    for (int k = 0; k < PD; ++k)
#pragma unroll
      for (int t = 0; t < NT; ++t) wq[k][t] = wbase[(size_t)((wave * NT + t) * KS + k) * 64];
    {  // encoder rows -> bf16 operand panel (8 lanes per row: whole 128-byte lines per load instruction)
      constexpr int LPR = 512 / BM, NP = 64 / LPR;
      const int r = tid / LPR, part = tid % LPR;
      const float* xr = a.x + (size_t)min(m0 + min(r, rpw - 1), a.M - 1) * DM;
      float4 v[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (i * LPR + part) * 4);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int c4 = i * LPR + part;
        *reinterpret_cast<uint2*>(Xs + panel_off(r, c4 >> 1) + (c4 & 1) * 8) = make_uint2(pack_bf2(v[i].x, v[i].y), pack_bf2(v[i].z, v[i].w));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // A
#pragma unroll
    for (int p = 0; p < K; ++p) {
      const int tile0 = p * 16 + wave * NT;
      f32x4 acc[NT][RT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 bb = *reinterpret_cast<const float4*>(a.bias + (tile0 + t) * 16 + 4 * g);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
      }
      // next stream: the following pass's tiles, or (after the last pass) the single tail tile for wave 0
      const bf16x8* wc = wbase + (size_t)tile0 * KS * 64;
      const bf16x8* wn = p + 1 < K ? wc + (size_t)16 * KS * 64 : wbase + (size_t)(K * 16) * KS * 64;
      const int nnt = p + 1 < K ? NT : (wave == 0 ? 1 : 0);
      bf16x8 xa[2][RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) xa[0][rt] = *reinterpret_cast<const bf16x8*>(Xs + panel_off(rt * 16 + n, g));
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 wcur[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) wcur[t] = wq[ks % PD][t];
        if (ks + PD < KS) {
#pragma unroll
          for (int t = 0; t < NT; ++t) wq[ks % PD][t] = wc[(size_t)(t * KS + ks + PD) * 64];
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (t < nnt) wq[ks % PD][t] = wn[(size_t)(t * KS + ks + PD - KS) * 64];
        }
        if (ks + 1 < KS) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            xa[(ks + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(Xs + panel_off(rt * 16 + n, (ks + 1) * 4 + g));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcur[t], xa[ks & 1][rt], acc[t][rt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ReLU'd masks of speaker p: lane (n, g) owns bins (wave * 2 + t) * 16 + 4 g .. + 3 of rows 16 rt + n
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int r = rt * 16 + n;
          *reinterpret_cast<float4*>(Ms + p * MS_TILE + ms_off(r, (wave * NT + t) * 4 + g)) =
              make_float4(fmaxf(acc[t][rt][0], 0.f), fmaxf(acc[t][rt][1], 0.f), fmaxf(acc[t][rt][2], 0.f), fmaxf(acc[t][rt][3], 0.f));
        }
    }
    // tail: bin 256 of every speaker = rows 0..K-1 of the last tile (wave 0; lanes g == 0 hold them as registers 0..K-1)
    if (wave == 0) {
      float ftail[RT], ttail[K][RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int rc = min(m0 + min(rt * 16 + n, rpw - 1), a.M - 1);
        ftail[rt] = a.f[(size_t)rc * a.ldf + 256];
#pragma unroll
        for (int j = 0; j < K; ++j) ttail[j][rt] = a.tgt[j][(size_t)rc * a.ldt + 256];
      }
      f32x4 acc[RT];
      const float4 bb = *reinterpret_cast<const float4*>(a.bias + K * 16 * 16 + 4 * g);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
      const bf16x8* wc = wbase + (size_t)(K * 16) * KS * 64;
      bf16x8 wt[KS];                              // all of the tile's fragments in flight at once (a lone wave)
#pragma unroll
      for (int ks = 0; ks < PD; ++ks) wt[ks] = wq[ks][0];
#pragma unroll
      for (int ks = PD; ks < KS; ++ks) wt[ks] = wc[(size_t)ks * 64];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const bf16x8 xv = *reinterpret_cast<const bf16x8*>(Xs + panel_off(rt * 16 + n, ks * 4 + g));
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[ks], xv, acc[rt], 0, 0, 0);
        }
      }
      if (g == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = m0 + rt * 16 + n;
          if (row < a.M && rt * 16 + n < rpw) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const float y = fmaxf(acc[rt][k], 0.f) * ftail[rt];
#pragma unroll
              for (int j = 0; j < K; ++j) {
                const float d = y - ttail[j][rt];
                if (row < bnd) sse[0][k][j] += d * d;
                else sse[1][k][j] += d * d;
              }
            }
          }
        }
      }
    }
    request_rows();                               // behind every weight fragment of this wave
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // B: both speakers' masks are in LDS; the operand panel is dead
  // ---- the error sweep: rows 4 wave .. 4 wave + 3
#pragma unroll
  for (int i = 0; i < LR; ++i) {
    const int r = wave * LR + i, row = m0 + r;
    if (r >= rpw) continue;                        // wave-uniform: a padding row (nothing was loaded for it)
    float e[K][K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float4 mk = *reinterpret_cast<const float4*>(Ms + k * MS_TILE + ms_off(r, lane));
      const float mv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
      for (int j = 0; j < K; ++j) e[k][j] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float y = mv[c] * fo[i].v[c];
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float d = y - tg[j][i].v[c];
          e[k][j] += d * d;
        }
      }
    }
    const float w0 = row < min(bnd, a.M) ? 1.f : 0.f, w1 = (row >= bnd && row < a.M) ? 1.f : 0.f;   // wave-uniform
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        sse[0][k][j] += w0 * e[k][j];
        sse[1][k][j] += w1 * e[k][j];
      }
  }
  // fixed-order reduction: lanes, then waves, one partial per (workgroup, slot): no atomics
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float v = wave_sum(sse[s_][k][j]);
        if (lane == 0) red[wave][(s_ * K + k) * K + j] = v;
      }
  __syncthreads();
  if (tid < 2 * K * K) {
    float s_ = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s_ += red[w][tid];
    a.partial[(size_t)blockIdx.x * 2 * K * K + tid] = s_;
  }
}

// (Measured and rejected: the workgroup that arrives last - an agent-scope counter - doing this inside head_pit_split_kernel:
// 21.0 us against 15.0 + 3.5 for the two launches; the release / acquire fences and the serial tail cost more than the launch.)
__global__ __launch_bounds__(64) void head_pit_finalize_kernel(const float* __restrict__ partial, int B, int T, int M,
                                                               double norm, float* __restrict__ loss,
                                                               int* __restrict__ perm_out, float* __restrict__ pair_sse,
                                                               int rpw) {
  __shared__ float best_s[1024];
  pit_finalize_body(partial, B, T, M, norm, loss, perm_out, pair_sse, rpw, (int)threadIdx.x, 64, best_s);
}

}  // namespace

#ifdef HP_STAMPS
extern "C" int csship_debug_headpit_stamps(unsigned* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_hp_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t csship_head_pit_workspace(int32_t B, int32_t T) {
  if (B <= 0 || T <= 0) return 0;
  return ((size_t)B * T + 15) / 16 * 2 * KSPK * KSPK * sizeof(float);  // (at least 16 rows per workgroup)
}

extern "C" int csship_head_pit(const float* x_enc, const uint16_t* w_perm_packed, const float* bias_perm, const float* f_orig,
                               int64_t ld_f, const float* const* targets_host, int64_t ld_t, int32_t B, int32_t T, int32_t F, int32_t K,
                               int32_t d_model, float* loss, int32_t* perm, float* pair_sse, void* workspace,
                               size_t workspace_bytes, csship_stream_t stream) {
  if (!x_enc || !w_perm_packed || !bias_perm || !f_orig || !targets_host || !loss || !workspace) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || B > 1024 || (int64_t)B * T > 0x7fffffff / 1024 || ld_f < F || ld_t < F) return CSSHIP_EBADSHAPE;
  if (K != KSPK || F != NF || d_model != DM || T < BM) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x_enc) || !aligned16(w_perm_packed) || !aligned16(bias_perm)) return CSSHIP_EALIGN;
  if (workspace_bytes < csship_head_pit_workspace(B, T)) return CSSHIP_EWORKSPACE;
  HeadPitArgs a{};
  a.x = x_enc; a.w = reinterpret_cast<const bf16x8*>(w_perm_packed); a.bias = bias_perm; a.f = f_orig; a.ldf = ld_f; a.ldt = ld_t;
  for (int k = 0; k < K; ++k) {
    if (!targets_host[k]) return CSSHIP_ENULL;
    a.tgt[k] = targets_host[k];
  }
  a.partial = reinterpret_cast<float*>(workspace);
  a.M = B * T; a.T = T;
  hipStream_t s = as_stream(stream);
  static const bool one_role = getenv("CSSHIP_HEADPIT_ONE_ROLE") != nullptr;   // A/B: the 8-wave kernel, every wave both roles
  // rows per workgroup: the fewest rounds of one workgroup per CU that 64-row panels would take, the rows spread evenly
  // over all CUs of those rounds (12 800 rows on 256 CUs: 50 instead of 64 on 200 CUs with 56 idle)
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  static const int rpw_env = [] { const char* e = getenv("CSSHIP_HEADPIT_RPW"); return e ? atoi(e) : 0; }();   // A/B: 64 = whole panels
  int rpw = BM;
  if (!one_role) {
    const int64_t rounds = ((int64_t)a.M + (int64_t)BM * cus - 1) / ((int64_t)BM * cus);
    rpw = (int)((a.M + rounds * cus - 1) / (rounds * cus));
    if (rpw_env > 0) rpw = rpw_env;
    rpw = std::max(16, std::min(rpw, BM));
  }
  a.rpw = rpw;
  const unsigned nwg = (unsigned)((a.M + rpw - 1) / rpw);
  {
    TimelineScope tl(CSSHIP_TL_HEAD, s);
    if (one_role) hipLaunchKernelGGL(head_pit_kernel, dim3(nwg), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(head_pit_split_kernel, dim3(nwg), dim3(1024), 0, s, a);
  }
  CSSHIP_CHECK_LAUNCH();
  {
    TimelineScope tl(CSSHIP_TL_PIT_FINALIZE, s);
    hipLaunchKernelGGL(head_pit_finalize_kernel, dim3(1), dim3(64), 0, s, a.partial, B, T, a.M, (double)K * T * F, loss,
                       perm, pair_sse, rpw);
  }
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
