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


// =============================================================================== the two position products in one pass
// Behind the one-pass backward kernel the relative-position term leaves two products over the SKEWED dS
// (dS_skew[b, t, h, w] = dS[b, h, t, j] at w = t - j + T - 1; bf16 [B, T, H, W], zero outside the band w in [t, t + T - 1]):
//     dq[b, t, h, :]  = dq1[b, h, t, :] + sum_w dS_skew[b, t, h, w] pe_window[w, :]        (conformer.py:236-241)
//     d pe_window[w, :] += sum_{b, t, h} dS_skew[b, t, h, w] q[b, t, h, :]
// The tiled batched GEMM took 23.7 + 23.0 us for them at 32 x 4 x 400 (each reads the 41 MB band again at ~2 TB/s
// through its phase-serial tile loop, the second through 32 reduction slabs and a slab sum).  Here a workgroup owns 64
// frames of one sample, loads the band of each head ONCE as a [64 frames][<= 512 window rows] LDS tile (1 KB rows,
// 32-byte groups XORed with a key of the row: the 16-byte row reads of the first product and the transposing reads of
// the second are both conflict free) and runs both products from it:
//   * dq^T tiles = pe_window^T (fragments straight from global, the same for every head: loaded once, kept in registers)
//     x tile rows (ds_read_b128) - transposed product, so a lane owns four features of one frame;
//   * window-gradient tiles [window row][feature], accumulated in registers over the four heads (4 x 4 tiles per wave):
//     q^T (a [64][64] block staged like a k-major operand) x tile columns, both through ds_read_b64_tr_b16.
// The next head's tile travels global -> registers while this head's products run.  Every workgroup leaves its window
// gradient as one [<= 512][64] f32 partial (its window rows t0 .. t0 + 63 + T - 1); pos_window_reduce_kernel adds the
// partials that cover a window row in a fixed order (deterministic, no atomics) into the accumulator.
constexpr int PB_ROWS = 64;               // frames per workgroup
constexpr int PB_ROWB = 1024;             // bytes of a tile row: 512 window columns
constexpr int PB_TILE = PB_ROWS * PB_ROWB;  // 64 KB
constexpr int PB_QROWB = 256;             // bytes of a row of the staged q block (64 features used)
constexpr int PB_QTILE = PB_ROWS * PB_QROWB;  // 16 KB
constexpr int PB_MAXKS = 16;              // k-steps of 32 window columns at most

DEV int pb_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
DEV int pb_off(int row, int chunk) { return row * PB_ROWB + ((((chunk >> 1) ^ pb_key(row))) << 5) + (chunk & 1) * 16; }   // 16-byte chunk 0..63
DEV int pb_qoff(int row, int chunk) { return row * PB_QROWB + ((((chunk >> 1) ^ pb_key(row))) << 5) + (chunk & 1) * 16; }  // chunk 0..7

struct PosBwdArgs {
  const unsigned short* dsk;   // bf16 [B, T, H, ldw]; or (FROM_DS) dS row-major bf16 [B, H, T, ldw]
  const unsigned short* pet;   // bf16 [64, ldpt]: pe_window^T, zero beyond the window
  const unsigned short* q;     // bf16 [B*T*H, 64], rows in dS_skew's order
  const float* dq1;            // f32 [B, H, T, 64]
  int dq1_parts;               // dq1 arrives as this many partials, B*H*T*64 floats apart (csship_mha_relpos_bwd_fused: one per key group)
  int64_t ldq;                 // elements between the q rows of consecutive frames (H * 64 dense, 3 * H * 64 inside qkv)
  unsigned short* dqkv;        // bf16 [B*T, 3*H*64]: the q third is written
  float* partial;              // [B * nblk][nwt * 16][64]
  int64_t ldw, ldpt;
  int B, T, H, nblk, nks, nwt;
};

// FROM_DS: the tile is built from the ROW-MAJOR dS (bf16 [B, H, T, ldw = ld_p]) - no skewed copy exists in memory at all
// (writing it cost the one-pass backward kernel 11 of its 56 us: 8-byte pieces at 2-byte alignment).  Tile row r, window
// columns 8 c .. 8 c + 7 are keys j = r + T - 8 - 8 c + 7 .. down to + 0 of frame t0 + r: ONE 16-byte load at a 2-byte
// aligned address, the eight values reversed in registers (four v_alignbit), keys outside 0 .. T - 1 masked, one aligned
// 16-byte LDS store - the skew costs no extra pass and no scattered store.  The loads reach up to 14 bytes in front of
// and behind a row: the caller keeps 16 bytes of slack on both sides of the buffer.
template <bool FROM_DS>
__global__ __launch_bounds__(512) void attn_pos_bwd_kernel(PosBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char Ss[PB_TILE];
  __shared__ __attribute__((aligned(16))) unsigned char Qs[PB_QTILE];
  __shared__ __attribute__((aligned(16))) unsigned char Ps[PB_TILE];   // pe_window^T rows w0 .. of all 64 features: the same for every head
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n = lane & 15;
  const int v = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.nblk, blk = blockIdx.x - b * a.nblk;
  const int t0 = blk * PB_ROWS, w0 = t0;           // the block's band starts at window row t0
  const int D3 = 3 * a.H * 64;
  const int ncol = min(a.nks * 32, (int)(a.ldw - w0));   // window columns of the tile that exist in memory

  // ---- staging assignment: S tile chunk (row 8 j + (tid >> 6), 16-byte chunk tid & 63), q block chunk (tid >> 3, tid & 7)
  const int sc = tid & 63, sr0 = tid >> 6;
  const bool scol_ok = sc * 8 + 8 <= ncol;
  const int qr = tid >> 3, qc = tid & 7;
  const bool qrow_ok = t0 + qr < a.T;
  f32x4 sreg[8], qreg;
  struct __attribute__((packed, aligned(2))) U8 { unsigned short v[8]; };
  auto issue_loads = [&](int h) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 8 * j + sr0;
      if constexpr (FROM_DS) {
        // (wave-uniform 64-bit base of the head + a 32-bit offset per lane: eight 64-bit row addresses kept across the
        // heads were what spilled - and a scratch reload waits, in order, behind the tile prefetch)
        // The base sits 8 elements in front of the head's first row so that the offset is never negative (a chunk may start
        // up to 7 keys in front of its row: jl >= -7); an unsigned offset of -7 is 4 GB away.
        const unsigned short* const hb = a.dsk + ((size_t)b * a.H + h) * a.T * a.ldw - 8;
        const int js = r + a.T - 8 - 8 * sc;                      // key of the chunk's LAST column
        const int jl = (js > -8 && js < a.T) ? js : 0;            // chunks with no key in range re-read the row's start
        const unsigned off = (unsigned)(min(t0 + r, a.T - 1) * (int)a.ldw + jl + 8);
        sreg[j] = __builtin_bit_cast(f32x4, *reinterpret_cast<const U8*>(hb + off));
      } else {
        const size_t row = ((size_t)b * a.T + min(t0 + r, a.T - 1)) * a.H + h;
        sreg[j] = *reinterpret_cast<const f32x4*>(a.dsk + row * a.ldw + w0 + (scol_ok ? sc * 8 : 0));
      }
    }
    qreg = *reinterpret_cast<const f32x4*>(a.q + ((size_t)b * a.T + min(t0 + qr, a.T - 1)) * a.ldq + h * 64 + qc * 8);
  };
  // LDS offset of chunk sc of row 8 j + sr0: the swizzle key of that row is (sr0 & 3) | (j & 1) << 2 - two bases, j in the offset
  const int so_even = pb_off(sr0, sc), so_odd = pb_off(8 + sr0, sc) - 8 * PB_ROWB;
  auto stage = [&]() {
    const f32x4 zero{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 8 * j + sr0;
      const int so = ((j & 1) ? so_odd : so_even) + j * 8 * PB_ROWB;
      if constexpr (FROM_DS) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 x = __builtin_bit_cast(u32x4, sreg[j]);
        const int js = r + a.T - 8 - 8 * sc;
        const bool rok = t0 + r < a.T;
        u32x4 y;
#pragma unroll
        for (int d = 0; d < 4; ++d) {   // input elements 2 d (low half), 2 d + 1 (high half) are keys js + 2 d, js + 2 d + 1
          const unsigned mlo = (rok && (unsigned)(js + 2 * d) < (unsigned)a.T) ? 0x0000ffffu : 0u;
          const unsigned mhi = (rok && (unsigned)(js + 2 * d + 1) < (unsigned)a.T) ? 0xffff0000u : 0u;
          const unsigned m = x[d] & (mlo | mhi);
          y[3 - d] = (m >> 16) | (m << 16);   // reversed order: output element k = input element 7 - k
        }
        *reinterpret_cast<u32x4*>(Ss + so) = y;
      } else {
        *reinterpret_cast<f32x4*>(Ss + so) = (scol_ok && t0 + r < a.T) ? sreg[j] : zero;
      }
    }
    *reinterpret_cast<f32x4*>(Qs + pb_qoff(qr, qc)) = qrow_ok ? qreg : zero;
  };
  issue_loads(0);

  const int dt1 = v >> 1, tt0 = (v & 1) * 2;       // first product: feature tile dt1, frame tiles tt0, tt0 + 1
  {  // the window tile: [64 features][nks * 32 window rows from w0], staged once (held in registers across the heads - 64
     // per lane - it spilled)
    const bool pcol_ok = sc * 8 + 8 <= a.nks * 32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 8 * j + sr0;
      const f32x4 x = *reinterpret_cast<const f32x4*>(a.pet + (size_t)r * a.ldpt + w0 + (pcol_ok ? sc * 8 : 0));
      *reinterpret_cast<f32x4*>(Ps + pb_off(r, sc)) = x;
    }
  }
  f32x4 dw[4][4];                                   // window gradient: window tiles v, v + 8, v + 16, v + 24 x 4 feature tiles
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dw[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int h = 0; h < a.H; ++h) {
    __syncthreads();                                // every wave is done with the previous head's tile
    stage();
    // dq1 of this head's two frame tiles, then the next head's tile (the last head re-reads itself: the same number of
    // loads in flight on every trip)
    f32x4 d1[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int t = min(t0 + (tt0 + jj) * 16 + n, a.T - 1);
      d1[jj] = *reinterpret_cast<const f32x4*>(a.dq1 + (((size_t)b * a.H + h) * a.T + t) * 64 + dt1 * 16 + 4 * g);
      for (int pp = 1; pp < a.dq1_parts; ++pp)   // (kernel-uniform; partials added in order)
        d1[jj] += *reinterpret_cast<const f32x4*>(a.dq1 + (((size_t)(pp * a.B + b) * a.H + h) * a.T + t) * 64 + dt1 * 16 + 4 * g);
    }
    issue_loads(min(h + 1, a.H - 1));
    __syncthreads();
    // ---- first product: dq^T[feature][frame] = pe_window^T x tile rows
    f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < PB_MAXKS; ++ks) {
      if (ks < a.nks) {
        const bf16x8 pa = *reinterpret_cast<const bf16x8*>(Ps + pb_off(dt1 * 16 + n, ks * 4 + g));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const bf16x8 sb = *reinterpret_cast<const bf16x8*>(Ss + pb_off((tt0 + jj) * 16 + n, ks * 4 + g));
          dq[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, sb, dq[jj], 0, 0, 0);
        }
      }
    }
    // ---- dq = dq1 + position term -> the q third of dqkv (lane: features dt1 * 16 + 4 g .. + 3 of frame (tt0 + jj) * 16 + n)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int t = t0 + (tt0 + jj) * 16 + n;
      if (t < a.T) {
        const f32x4 r = dq[jj] + d1[jj];
        *reinterpret_cast<uint2*>(a.dqkv + ((size_t)b * a.T + t) * D3 + h * 64 + dt1 * 16 + 4 * g) =
            make_uint2(pack_bf2(r[0], r[1]), pack_bf2(r[2], r[3]));
      }
    }
    // ---- second product: window gradient [feature][window row] += q^T x tile columns, over the block's 64 frames
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = ks * 32 + 8 * g + (n >> 2);
      bf16x8 qa[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const unsigned char* p0 = Qs + r0 * PB_QROWB + ((dt ^ pb_key(r0)) << 5) + (n & 3) * 8;
        const unsigned char* p1 = Qs + (r0 + 4) * PB_QROWB + ((dt ^ pb_key(r0 + 4)) << 5) + (n & 3) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
        qa[dt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int wt = v + 8 * j;
        if (wt < a.nwt) {                           // wave-uniform
          const unsigned char* p0 = Ss + r0 * PB_ROWB + ((wt ^ pb_key(r0)) << 5) + (n & 3) * 8;
          const unsigned char* p1 = Ss + (r0 + 4) * PB_ROWB + ((wt ^ pb_key(r0 + 4)) << 5) + (n & 3) * 8;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
          const bf16x8 sb = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dw[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[dt], sb, dw[j][dt], 0, 0, 0);
        }
      }
    }
  }
  // ---- the block's window gradient: rows w0 + wt * 16 + n, features dt * 16 + 4 g .. + 3
  float* part = a.partial + (size_t)blockIdx.x * a.nwt * 16 * 64;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int wt = v + 8 * j;
    if (wt < a.nwt) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<f32x4*>(part + (size_t)(wt * 16 + n) * 64 + dt * 16 + 4 * g) = dw[j][dt];
    }
  }
}

// window row w of the accumulator += the partials of every (sample, block) whose band covers it, samples then blocks in
// ascending order.  Workgroup = one window row: 16 float4 columns x 16 sample groups, then a fixed-order LDS sum.
__global__ __launch_bounds__(256) void pos_window_reduce_kernel(const float* __restrict__ partial, int B, int nblk, int nwt,
                                                                float* __restrict__ dwin) {
  __shared__ float4 red[16][16];
  const int w = blockIdx.x, c4 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int span = nwt * 16;
  // blocks with 64 blk <= w < 64 blk + span
  const int hi = min(nblk - 1, w / PB_ROWS), lo = w >= span ? (w - span) / PB_ROWS + 1 : 0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = grp; b < B; b += 16)
    for (int blk = lo; blk <= hi; ++blk) {
      const float4 x = *reinterpret_cast<const float4*>(partial + (((size_t)b * nblk + blk) * span + (w - blk * PB_ROWS)) * 64 + c4 * 4);
      acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
    }
  red[grp][c4] = acc;
  __syncthreads();
  if (grp == 0) {
    float4 t = red[0][c4];
#pragma unroll
    for (int k = 1; k < 16; ++k) { t.x += red[k][c4].x; t.y += red[k][c4].y; t.z += red[k][c4].z; t.w += red[k][c4].w; }
    float4* o = reinterpret_cast<float4*>(dwin + (size_t)w * 64 + c4 * 4);
    float4 cur = *o;
    cur.x += t.x; cur.y += t.y; cur.z += t.z; cur.w += t.w;
    *o = cur;
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

extern "C" size_t csship_attn_pos_bwd_workspace(int32_t B, int32_t T) {
  if (B <= 0 || T <= 0) return 0;
  const size_t nblk = (size_t)(T + PB_ROWS - 1) / PB_ROWS, nwt = ((size_t)T + PB_ROWS - 1 + 15) / 16;
  return (size_t)B * nblk * nwt * 16 * 64 * sizeof(float);
}

// span = 63 + T window rows must fit PB_MAXKS k-steps of 32
extern "C" int csship_attn_pos_bwd_max_t(void) { return PB_MAXKS * 32 - (PB_ROWS - 1); }

static int pos_bwd_impl(const uint16_t* ds_skew, int64_t ld_w, const uint16_t* pe_window_t, int64_t ld_pt,
                        const uint16_t* q_rows, const float* dq1, uint16_t* dqkv, float* dwin_acc, int32_t W,
                        int32_t B, int32_t T, int32_t H, void* workspace, size_t workspace_bytes,
                        csship_stream_t stream, bool from_ds, int dq1_parts = 1, int64_t ld_q = 0);

extern "C" int csship_attn_pos_bwd(const uint16_t* ds_skew, int64_t ld_w, const uint16_t* pe_window_t, int64_t ld_pt,
                                   const uint16_t* q_rows, const float* dq1, uint16_t* dqkv, float* dwin_acc, int32_t W,
                                   int32_t B, int32_t T, int32_t H, void* workspace, size_t workspace_bytes,
                                   csship_stream_t stream) {
  return pos_bwd_impl(ds_skew, ld_w, pe_window_t, ld_pt, q_rows, dq1, dqkv, dwin_acc, W, B, T, H, workspace, workspace_bytes,
                      stream, false);
}

extern "C" int csship_attn_pos_bwd_ds(const uint16_t* ds, int64_t ld_p, const uint16_t* pe_window_t, int64_t ld_pt,
                                      const uint16_t* q_rows, const float* dq1, uint16_t* dqkv, float* dwin_acc, int32_t W,
                                      int32_t B, int32_t T, int32_t H, void* workspace, size_t workspace_bytes,
                                      csship_stream_t stream) {
  return pos_bwd_impl(ds, ld_p, pe_window_t, ld_pt, q_rows, dq1, dqkv, dwin_acc, W, B, T, H, workspace, workspace_bytes, stream,
                      true);
}

extern "C" int csship_attn_pos_bwd_ds_parts(const uint16_t* ds, int64_t ld_p, const uint16_t* pe_window_t, int64_t ld_pt,
                                            const uint16_t* q, int64_t ld_q, const float* dq1_parts, int32_t n_parts,
                                            uint16_t* dqkv, float* dwin_acc, int32_t W, int32_t B, int32_t T, int32_t H,
                                            void* workspace, size_t workspace_bytes, csship_stream_t stream) {
  if (n_parts < 1 || n_parts > 16) return CSSHIP_EBADSHAPE;
  if (ld_q < (int64_t)H * 64 || (ld_q & 7) != 0) return CSSHIP_EBADSHAPE;
  return pos_bwd_impl(ds, ld_p, pe_window_t, ld_pt, q, dq1_parts, dqkv, dwin_acc, W, B, T, H, workspace, workspace_bytes, stream, true,
                      n_parts, ld_q);
}

static int pos_bwd_impl(const uint16_t* ds_skew, int64_t ld_w, const uint16_t* pe_window_t, int64_t ld_pt,
                        const uint16_t* q_rows, const float* dq1, uint16_t* dqkv, float* dwin_acc, int32_t W,
                        int32_t B, int32_t T, int32_t H, void* workspace, size_t workspace_bytes,
                        csship_stream_t stream, bool from_ds, int dq1_parts, int64_t ld_q) {
  if (ld_q == 0) ld_q = (int64_t)H * 64;
  if (!ds_skew || !pe_window_t || !q_rows || !dq1 || !dqkv || !dwin_acc || !workspace) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0 || W < 2 * T - 1 || (ld_pt & 7) != 0) return CSSHIP_EBADSHAPE;
  if (from_ds ? (ld_w < T) : (ld_w < W || (ld_w & 7) != 0)) return CSSHIP_EBADSHAPE;
  const int nblk = (T + PB_ROWS - 1) / PB_ROWS;
  const int span = PB_ROWS - 1 + T;                 // window rows a block's band can touch
  const int nks = (span + 31) / 32, nwt = (span + 15) / 16;
  if (nks > PB_MAXKS) return CSSHIP_EUNSUPPORTED;   // T <= 449: the tile holds 512 window columns (csship_bgemm serves longer chunks)
  // the transposed window must cover every k-step of the last block (zero beyond the window); partial rows beyond the
  // accumulator's W rows are zero (outside every band) and are not added
  if (ld_pt < (int64_t)(nblk - 1) * PB_ROWS + nks * 32) return CSSHIP_EBADSHAPE;
  if ((int64_t)B * nblk > 0x7fffffffLL) return CSSHIP_EBADSHAPE;
  if ((!from_ds && !aligned16(ds_skew)) || ((uintptr_t)ds_skew & 1) || !aligned16(pe_window_t) || !aligned16(q_rows) ||
      !aligned16(dq1) || !aligned8(dqkv) || !aligned16(dwin_acc) || !aligned16(workspace))
    return CSSHIP_EALIGN;
  if (workspace_bytes < csship_attn_pos_bwd_workspace(B, T)) return CSSHIP_EWORKSPACE;
  PosBwdArgs a{ds_skew, pe_window_t, q_rows, dq1, dq1_parts, ld_q, dqkv, reinterpret_cast<float*>(workspace), ld_w, ld_pt, B, T, H, nblk, nks, nwt};
  hipStream_t s = as_stream(stream);
  side_join(s);   // (the previous call's window reduction may still read the partials this launch overwrites)
  if (from_ds) hipLaunchKernelGGL((attn_pos_bwd_kernel<true>), dim3((unsigned)(B * nblk)), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((attn_pos_bwd_kernel<false>), dim3((unsigned)(B * nblk)), dim3(512), 0, s, a);
  CSSHIP_CHECK_LAUNCH();
  const int rows = std::min(W, (nblk - 1) * PB_ROWS + nwt * 16);
  // the window gradient is a parameter gradient (nobody on the data-gradient chain reads it): beside the caller's stream when it has
  // a side stream; consecutive calls accumulate into dwin_acc in order, on that stream
  hipLaunchKernelGGL(pos_window_reduce_kernel, dim3((unsigned)rows), dim3(256), 0, side_fork(s), reinterpret_cast<const float*>(workspace), B,
                     nblk, nwt, dwin_acc);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
