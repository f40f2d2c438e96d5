// dV and dK of the relative-position attention in ONE streaming pass over the probabilities and dS
// (backward of css/models/conformer.py:229-253 under torch.autograd in the reference):
//     dV[b,h,j,:] = sum_i P'[b,h,i,j] dO[b,i,h,:]        (P' = the operand of dV: the dropped, rescaled probabilities)
//     dK[b,h,j,:] = sum_i dS[b,h,i,j] q[b,i,h,:]
// Both are skinny batched products (N = 64, reduction over the T queries) whose whole cost is reading the two [B,H,T,T]
// bf16 operands once: 84 MB at 32 x 4 x 400, fresh in the Infinity Cache behind the one-pass kernel that wrote them.  The
// general batched GEMM took 17.8 us for each (one 128 x 64 tile per workgroup, 6 dependent 64-deep k-tiles: first-tile
// latency and a phase-serial loop around 0.2 us of matrix work).  Here a workgroup owns up to 256 KEYS of one (b, h) -
// every wave 32 of them, its dV^T and dK^T tiles (64 x 32 each) in 64 accumulator registers for the whole pass - and
// streams the query rows in blocks of 32: P and dS as [32 queries][keys of the workgroup] row segments, dO and q as
// [32][64] blocks, global -> registers four blocks ahead -> LDS (double buffered, one barrier per block).  All four
// operands are wanted with the reduction index (the query) inside a lane's fragment, i.e. transposed: they are staged
// exactly as they lie in memory and read with ds_read_b64_tr_b16 (the V^T read of the forward kernel).
// Measured (32 x 4 x 400, tools/bench_dvdk.py): 27.7 us against 35.6 us for the two products.  Streaming the same row
// segments with no other work takes 16.5 us (tools/probes/lds_vmem.hip: 5.0 TB/s at any prefetch depth - 400-byte
// halves of 800-byte rows; 5.6 TB/s with 512-byte halves of 1024-byte rows), the dO / q blocks add a third to the
// loads; the rest is the lockstep of the workgroup (tools/dvdk_stamps.py, per block and wave: loads issued 350..1300
// cycles behind the address unit's back pressure | fragments + MFMAs 520 | LDS stores 400 | barrier 500..1100).
// Tried and not kept: wave-private staging of P / dS with a barrier per four blocks (30.2 us - whole chunks cost three
// extra blocks at T = 400), all 24 fragment reads ahead of the MFMAs (no change), padded row strides and aligned key
// splits (no change), 128-key workgroups (34 us).
#include <algorithm>

#include "common.hpp"

namespace {

constexpr int DV_QB = 32;          // queries per block (one MFMA k-step)
constexpr int DV_KW = 256;         // keys per workgroup at most (8 waves x 32)
constexpr int DV_PROW = DV_KW * 2; // bytes of a staged P / dS row
constexpr int DV_PTILE = DV_QB * DV_PROW;   // 16 KB
constexpr int DV_ATILE = DV_QB * 128;       // dO / q block: 4 KB
constexpr int DV_STAGE = 2 * DV_PTILE + 2 * DV_ATILE;  // 40 KB

// 16-byte chunk `ch` (0..7) of a 128-byte group of row `row`: 32-byte groups XORed with (row >> 1) & 3, the swizzle of the
// forward kernel's V tile (attn.hip v_off): the transposing reads below are conflict free with it
DEV int dv_swz(int row, int ch) { return ((((ch >> 1) ^ ((row >> 1) & 3)) << 1) | (ch & 1)) * 16; }
// P / dS rows are 512 bytes apart - a multiple of the 256-byte bank span - so the 32-byte pair of chunks a transposing
// read takes from each of 16 rows must be spread by the row itself: pair index XOR (row & 7) inside the 256-byte span
// (with the V-tile swizzle above, made for 128-byte rows, rows 0..7 shared 4 slots: the fragment reads ran 4-way
// conflicted and were two thirds of the kernel's 28 us)
DEV int dv_poff(int row, int chunk) { return row * DV_PROW + (((chunk >> 1) ^ (row & 7)) << 5) + (chunk & 1) * 16; }
DEV int dv_aoff(int row, int chunk) { return row * 128 + dv_swz(row, chunk); }

// fragment X[m = m0 + n][k = the block's 32 rows] of a tile stored [k][m] (m contiguous): A and B operands alike
// (lane (n = lane & 15, g = lane >> 4) ends up with k = 4 g + 0..3 and 16 + 4 g + 0..3, the same for every tile)
DEV bf16x8 dv_frag_p(const unsigned char* tile, int m0, int g, int q4, int p4) {
  const int c = (m0 >> 3) + (p4 >> 1);
  const unsigned char* a0 = tile + dv_poff(4 * g + q4, c) + (p4 & 1) * 8;
  const unsigned char* a1 = tile + dv_poff(16 + 4 * g + q4, c) + (p4 & 1) * 8;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
DEV bf16x8 dv_frag_a(const unsigned char* tile, int m0, int g, int q4, int p4) {
  const int c = (m0 >> 3) + (p4 >> 1);
  const unsigned char* a0 = tile + dv_aoff(4 * g + q4, c) + (p4 & 1) * 8;
  const unsigned char* a1 = tile + dv_aoff(16 + 4 * g + q4, c) + (p4 & 1) * 8;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct DvDkArgs {
  const unsigned short* p;     // bf16 [B, H, T, ldp]: operand of dV
  const unsigned short* ds;    // bf16 [B, H, T, ldp]: operand of dK
  const unsigned short* dout;  // bf16 [B*T, H*64]
  const unsigned short* qkv;   // bf16 [B*T, 3*H*64]: q = the first third
  unsigned short* dqkv;        // bf16 [B*T, 3*H*64]: the k and v thirds are written
  int64_t ldp;
  int B, T, H, G, KH;          // G workgroups of KH keys per (b, h)
};

// -DDV_STAMPS: per-wave cycle stamps of the loop's phases (tools/dvdk_stamps.py); not part of the product build
#ifdef DV_STAMPS
__device__ unsigned* g_dv_stamps = nullptr;
#define DV_STAMP(j, k) do { if ((j) < 16 && lane == 0) stamps[w][(j) * 6 + (k)] = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define DV_STAMP(j, k) do {} while (0)
#endif

__global__ __launch_bounds__(512) void attn_dvdk_kernel(DvDkArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * DV_STAGE];
#ifdef DV_STAMPS
  __shared__ unsigned stamps[8][96];
#endif
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n = lane & 15, q4 = n >> 2, p4 = n & 3;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x / a.G, gi = blockIdx.x - bh * a.G;
  const int b = bh / a.H, h = bh - b * a.H;
  const int D = a.H * 64, D3 = 3 * D;
  const int kb = gi * a.KH;                       // first key of the workgroup
  const int kn = min(a.KH, a.T - kb);             // its keys
  if (kn <= 0) return;                            // (workgroup-uniform; the host never launches such a workgroup)
  const size_t prow0 = (size_t)bh * a.T;          // row index of query 0 in p / ds
  const size_t arow0 = (size_t)b * a.T;           // row index of query 0 in dout / qkv

  // ---- staging assignment.  P / dS: thread -> (row = tid >> 4, 16-byte chunks (tid & 15) and 16 + (tid & 15)) of the
  // block's [32][<= 256 keys] segment; dO (threads 0..255) and q (256..511): (row = t >> 3, chunk = t & 7) of [32][64]
  const int prow = tid >> 4, pc0 = tid & 15;
  const int t2 = tid & 255, arow = t2 >> 3, ac = t2 & 7;
  const bool is_q = tid >= 256;
  // Four register sets in flight (blocks j + 1 .. j + 4 while block j is multiplied): with one set ahead the loop ran at
  // one memory latency per 36 KB block (2.4 us, 2.65 TB/s over the chip); NAMED variables and a loop unrolled by four,
  // because arrays indexed by the loop counter would live in scratch memory.  Every load is issued unconditionally from a
  // clamped address, so the count of loads in flight (s_waitcnt vmcnt) is the same on every path: query rows past T
  // re-read row T - 1 and meet zero rows of dO / q; key columns outside the workgroup's range re-read its first chunk
  // and only feed key columns that are never stored.
  struct Regs { f32x4 pa, pb, da, db, a; };     // chunks pc0 and 16 + pc0 of the P and dS rows, one chunk of dO / q
  Regs s0, s1, s2, s3;
  const int col_a = pc0 * 8, col_b = (pc0 + 16) * 8;   // first key (relative) of the two chunks
  const int pcol_a = (col_a < kn && kb + col_a + 8 <= a.ldp) ? col_a : 0;
  const int pcol_b = (col_b < kn && kb + col_b + 8 <= a.ldp) ? col_b : 0;
  const int o_pa = dv_poff(prow, pc0), o_pb = dv_poff(prow, pc0 + 16);
  const int o_a = 2 * DV_PTILE + (is_q ? DV_ATILE : 0) + dv_aoff(arow, ac);
  const unsigned short* asrc = is_q ? a.qkv + arow0 * D3 + h * 64 + ac * 8 : a.dout + arow0 * D + h * 64 + ac * 8;
  const int alda = is_q ? D3 : D;
#define DV_LOAD(i0, R)                                                                      \
  {                                                                                         \
    const size_t qi_ = (prow0 + min((i0) + prow, a.T - 1)) * a.ldp + kb;                    \
    R.pa = *reinterpret_cast<const f32x4*>(a.p + qi_ + pcol_a);                             \
    R.da = *reinterpret_cast<const f32x4*>(a.ds + qi_ + pcol_a);                            \
    R.pb = *reinterpret_cast<const f32x4*>(a.p + qi_ + pcol_b);                             \
    R.db = *reinterpret_cast<const f32x4*>(a.ds + qi_ + pcol_b);                            \
    R.a = *reinterpret_cast<const f32x4*>(asrc + (size_t)min((i0) + arow, a.T - 1) * alda); \
  }
#define DV_STORE(i0, buf, R) /* (the zero rows are made here, next to the use: a select at the load would be sunk to the \
                                 loop's end by hipcc, behind a wait for every load in flight) */ \
  DV_STORE_BODY(i0, buf, R)
#define DV_STORE_BODY(i0, buf, R)                                       \
  {                                                                     \
    unsigned char* st_ = smem + (buf) * DV_STAGE;                       \
    *reinterpret_cast<f32x4*>(st_ + o_pa) = R.pa;                       \
    *reinterpret_cast<f32x4*>(st_ + DV_PTILE + o_pa) = R.da;            \
    *reinterpret_cast<f32x4*>(st_ + o_pb) = R.pb;                       \
    *reinterpret_cast<f32x4*>(st_ + DV_PTILE + o_pb) = R.db;            \
    *reinterpret_cast<f32x4*>(st_ + o_a) = (i0) + arow < a.T ? R.a : f32x4{0.f, 0.f, 0.f, 0.f}; \
  }

  f32x4 dv[4][2], dk[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) dv[dt][jt] = dk[dt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nblk = (a.T + DV_QB - 1) / DV_QB;
  const bool active = 32 * w < kn;                // waves beyond the workgroup's keys only stage
  auto compute = [&](int buf) {
    if (!active) return;
    const unsigned char* st = smem + buf * DV_STAGE;
    bf16x8 bp[2], bd[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      bp[jt] = dv_frag_p(st, 32 * w + 16 * jt, g, q4, p4);
      bd[jt] = dv_frag_p(st + DV_PTILE, 32 * w + 16 * jt, g, q4, p4);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16x8 ado = dv_frag_a(st + 2 * DV_PTILE, 16 * dt, g, q4, p4);
      const bf16x8 aq = dv_frag_a(st + 2 * DV_PTILE + DV_ATILE, 16 * dt, g, q4, p4);
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        dv[dt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ado, bp[jt], dv[dt][jt], 0, 0, 0);
        dk[dt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bd[jt], dk[dt][jt], 0, 0, 0);
      }
    }
  };
  DV_LOAD(0, s0)
  DV_LOAD(DV_QB, s1)
  DV_LOAD(2 * DV_QB, s2)
  DV_LOAD(3 * DV_QB, s3)
  DV_STORE(0, 0, s0)
  __syncthreads();
  // step j: block j + 4 -> the register set block j has left; block j from LDS stage j & 1; block j + 1 (requested
  // three steps ago) -> the other stage, which was last read at step j - 1, behind a barrier
#ifdef DV_STAMPS
#define DV_STEP(j, c, n, LOAD)             \
  DV_STAMP(j, 0);                          \
  LOAD(((j) + 4) * DV_QB, c)               \
  __builtin_amdgcn_sched_barrier(0);       \
  DV_STAMP(j, 1);                          \
  compute((j) & 1);                        \
  __builtin_amdgcn_sched_barrier(0);       \
  DV_STAMP(j, 2);                          \
  asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); \
  DV_STAMP(j, 3);                          \
  DV_STORE(((j) + 1) * DV_QB, ((j) + 1) & 1, n) \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  DV_STAMP(j, 4);                          \
  __syncthreads();                         \
  DV_STAMP(j, 5);
#else
#define DV_STEP(j, c, n, LOAD)             \
  LOAD(((j) + 4) * DV_QB, c)               \
  __builtin_amdgcn_sched_barrier(0);       \
  compute((j) & 1);                        \
  DV_STORE(((j) + 1) * DV_QB, ((j) + 1) & 1, n) \
  __syncthreads();
#endif
#define DV_NOLOAD(i0, R)
  // whole groups of four steps in a loop without exits (an exit between the steps is, to hipcc's wait-count pass, an edge
  // back to the loop's head with loads in flight into the registers the head reuses: vmcnt(0) once per group), then the
  // last one to three steps in line; the loads past the last block re-read the last row (nobody uses them)
  int it = 0;
  for (; it + 4 <= nblk; it += 4) {
    DV_STEP(it, s0, s1, DV_LOAD)
    DV_STEP(it + 1, s1, s2, DV_LOAD)
    DV_STEP(it + 2, s2, s3, DV_LOAD)
    DV_STEP(it + 3, s3, s0, DV_LOAD)
  }
  if (it < nblk) {
    DV_STEP(it, s0, s1, DV_NOLOAD)
    if (it + 1 < nblk) {
      DV_STEP(it + 1, s1, s2, DV_NOLOAD)
      if (it + 2 < nblk) { DV_STEP(it + 2, s2, s3, DV_NOLOAD) }
    }
  }
#undef DV_NOLOAD
#undef DV_STEP
#undef DV_LOAD
#undef DV_STORE
#undef DV_STORE_BODY
#ifdef DV_STAMPS
  __syncthreads();
  if (g_dv_stamps && tid < 96) {
    for (int ww = 0; ww < 8; ++ww) g_dv_stamps[((size_t)blockIdx.x * 8 + ww) * 96 + tid] = stamps[ww][tid];
  }
#endif
  if (!active) return;
  // dV^T / dK^T tiles: lane (g, n) holds features d = 16 dt + 4 g + 0..3 of key kb + 32 w + 16 jt + n: one 8-byte store each
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int j = kb + 32 * w + 16 * jt + n;
    if (j >= a.T || 32 * w + 16 * jt + n >= kn) continue;
    unsigned short* row = a.dqkv + (arow0 + j) * D3 + h * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *reinterpret_cast<uint2*>(row + D + 16 * dt) = make_uint2(pack_bf2(dk[dt][jt][0], dk[dt][jt][1]), pack_bf2(dk[dt][jt][2], dk[dt][jt][3]));
      *reinterpret_cast<uint2*>(row + 2 * D + 16 * dt) = make_uint2(pack_bf2(dv[dt][jt][0], dv[dt][jt][1]), pack_bf2(dv[dt][jt][2], dv[dt][jt][3]));
    }
  }
}

}  // namespace

#ifdef DV_STAMPS
extern "C" int csship_debug_dvdk_stamps(unsigned* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_dv_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int csship_attn_dvdk(const uint16_t* p_v, const uint16_t* ds, int64_t ld_p, const uint16_t* d_out,
                                const uint16_t* qkv, uint16_t* dqkv, int32_t B, int32_t T, int32_t H, csship_stream_t stream) {
  if (!p_v || !ds || !d_out || !qkv || !dqkv) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0 || ld_p < T || (ld_p & 7) != 0 || B > 65535 || H > 65535) return CSSHIP_EBADSHAPE;
  if (!aligned16(p_v) || !aligned16(ds) || !aligned16(d_out) || !aligned16(qkv) || !aligned8(dqkv)) return CSSHIP_EALIGN;
  // keys per workgroup: as few workgroups per (b, h) as 256-key workgroups allow, then the keys dealt evenly in 32s
  int g0 = (T + DV_KW - 1) / DV_KW;
  if ((int64_t)B * H * g0 < 256 && T > 64) g0 *= 2;   // (32 x 4 heads x 400 frames: 2 workgroups of 224 / 176 keys per head)
  const int KH = std::min(DV_KW, ((T + g0 - 1) / g0 + 31) / 32 * 32);
  const int G = (T + KH - 1) / KH;
  const long long nwg = (long long)B * H * G;
  if (nwg > 0x7fffffffLL) return CSSHIP_EBADSHAPE;
  DvDkArgs a{p_v, ds, d_out, qkv, dqkv, ld_p, B, T, H, G, KH};
  hipLaunchKernelGGL(attn_dvdk_kernel, dim3((unsigned)nwg), dim3(512), 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
