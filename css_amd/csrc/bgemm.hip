// Batched bf16 MFMA GEMM for the backward pass:  C[z] = epi(alpha * opA(A[z]) . opB(B[z])).
//
// The backward of every Linear / attention product on the path (css/models/conformer.py:98,124,205-209,
// 229-253,315-318 under torch.autograd in the reference) is one of four operand layouts
//     dX = dY . W          A reduction-contiguous,  B reduction-STRIDED (W stays [N, K] row-major)
//     dW = dY^T . X        both operands reduction-strided (the reduction runs over the B*T rows)
//     dP = dO . V^T        both contiguous ("NT", like gemm.hip)
// so ONE kernel takes either layout per operand: a contiguous operand is staged as [row][k] (XOR-swizzled
// 128-byte rows, fragments by ds_read_b128); a strided operand is staged exactly as it lies in memory,
// [k][row] with 256-byte rows, and the MFMA fragment is produced by the LDS transposing read
// ds_read_b64_tr_b16 - no transposed copies of activations or weights exist anywhere.
// 128x128 output tile, 4 waves of 64x64 (4x4 v_mfma_f32_16x16x32_bf16), BK = 64, register-staged double
// buffering.  Long reductions (dW: K = B*T = 12 800) are split over blockIdx.z into fp32 slabs that a
// second kernel sums in a fixed order (deterministic, no atomics).  fp32 operands are rounded to bf16
// on load.  Batches use a two-level index z = z1 * nb2 + z2 (batch, head).
#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace {

struct BgemmArgs {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  const void* aux;
  int M, N, K;
  int64_t lda, ldb, ldc, ldaux;
  int nb2, splits, k_per_split;
  int64_t sA1, sA2, sB1, sB2, sC1, sC2, sX1, sX2, slab;
  int a_f32, b_f32, epi;
  float alpha;
  // COLSUM kernels (weight gradients): column sums of the kmajor A operand (= the bias gradient of the same Linear)
  int band;        // > 0: banded A (csship_bgemm_params.band / band_stride)
  int band_stride;
  float* cs_out;   // [M], alpha applied (splits == 1)
  float* cs_slab;  // [splits][M] partial sums (splits > 1), reduced with the C slabs
  float alpha_head;  // COLSUM kernels: output rows < head_rows take this factor instead of alpha (fused q | k | v weight
  int head_rows;     // gradient: the q rows carry the 1/sqrt(d_k) of the pre-scaled projection)
};

constexpr int BK = 64;
constexpr int STAGE = 256 * 128;  // A image (16 KB) + B image (16 KB)

// reduction-strided image: row = k (256 B = 128 operand rows), 32-byte group g of row k at ((g ^ key(k)) << 5).
// A half-wave's transposing read touches k rows {8a..8a+3, 8a+8..8a+11}: 8 distinct keys -> conflict free.
DEV int km_off(int k, int grp) { return k * 256 + ((grp ^ ((k & 3) | (((k >> 3) & 1) << 2))) << 5); }

// Raw image of 8 consecutive operand elements as loaded: bf16 -> x (one 16-byte load), f32 -> x, y (two).  The
// conversion to bf16 happens when the tile is written to LDS, AFTER the MFMAs of the current tile.
struct Raw8 {
  uint4 x, y;
};

template <bool F32>
DEV Raw8 load8(const void* base, int64_t off, int n_valid) {
  Raw8 r;
  r.y = make_uint4(0, 0, 0, 0);
  if constexpr (F32) {
    const float* p = reinterpret_cast<const float*>(base) + off;
    if (n_valid >= 8) {
      r.x = *reinterpret_cast<const uint4*>(p);
      r.y = *reinterpret_cast<const uint4*>(p + 4);
      return r;
    }
    unsigned v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = e < n_valid ? __float_as_uint(p[e]) : 0u;
    r.x = make_uint4(v[0], v[1], v[2], v[3]);
    r.y = make_uint4(v[4], v[5], v[6], v[7]);
    return r;
  } else {
    const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + off;
    if (n_valid >= 8) {
      r.x = *reinterpret_cast<const uint4*>(p);
      return r;
    }
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = e < n_valid ? p[e] : (unsigned short)0;
    r.x = make_uint4(v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16),
                     v[6] | ((unsigned)v[7] << 16));
    return r;
  }
}

DEV uint4 to_bf16(const Raw8& r, bool f32) {
  if (!f32) return r.x;
  return make_uint4(pack_bf2(__uint_as_float(r.x.x), __uint_as_float(r.x.y)), pack_bf2(__uint_as_float(r.x.z), __uint_as_float(r.x.w)),
                    pack_bf2(__uint_as_float(r.y.x), __uint_as_float(r.y.y)), pack_bf2(__uint_as_float(r.y.z), __uint_as_float(r.y.w)));
}

// Generic loader (any shape): one operand's 128 x 64 tile -> 4 raw groups per thread.  rows: tile's first operand row,
// nrows: operand rows, k0/kend: reduction range of this tile.  hipcc branches around every conditional load and
// waits for it inside the branch, so this loader's loads go out one after the other - it only serves shapes whose
// grouped dimension is not a multiple of 8.
template <bool KM>
DEV void load_operand(const void* base, bool f32, int64_t ld, int row0, int nrows, int k0, int kend, int tid,
                      Raw8 (&r)[4]) {
  const Raw8 zero = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  if constexpr (!KM) {
    const int srow = tid >> 3, k = k0 + (tid & 7) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + srow + 32 * i;
      const int nv = (row < nrows) ? kend - k : 0;
      if (nv <= 0) r[i] = zero;
      else r[i] = f32 ? load8<true>(base, (int64_t)row * ld + k, nv) : load8<false>(base, (int64_t)row * ld + k, nv);
    }
  } else {
    const int krow = tid >> 4, col = row0 + (tid & 15) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + krow + 16 * i;
      const int nv = (k < kend) ? nrows - col : 0;
      if (nv <= 0) r[i] = zero;
      else r[i] = f32 ? load8<true>(base, (int64_t)k * ld + col, nv) : load8<false>(base, (int64_t)k * ld + col, nv);
    }
  }
}

// Fast loader (whole 8-element groups, operand < 4 GB): buffer loads through a descriptor that covers exactly the
// operand.  A group outside the tile's rows or reduction range gets an offset beyond the descriptor's range and the
// hardware returns zeros: no branch, no select on the data, and the four (bf16) / eight (f32) loads of an operand go
// out back to back.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFF00u;

// ROWS = operand rows of the tile (128, or 64 for the narrow B tile): ROWS / 32 groups per thread
template <bool KM, bool F32, int ROWS = 128>
DEV void load_operand_buf(__amdgpu_buffer_rsrc_t rs, unsigned ld, int row0, int nrows, int k0, int kend, int tid,
                          Raw8 (&r)[4]) {
  constexpr unsigned esz = F32 ? 4 : 2;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    unsigned off;
    bool valid;
    if constexpr (!KM) {
      const int row = row0 + (tid >> 3) + 32 * i, k = k0 + (tid & 7) * 8;
      valid = row < nrows && k + 8 <= kend;
      off = (unsigned)row * ld + (unsigned)k;
    } else if constexpr (ROWS == 128) {
      const int k = k0 + (tid >> 4) + 16 * i, col = row0 + (tid & 15) * 8;
      valid = k < kend && col + 8 <= nrows;
      off = (unsigned)k * ld + (unsigned)col;
    } else {
      const int k = k0 + (tid >> 3) + 32 * i, col = row0 + (tid & 7) * 8;
      valid = k < kend && col + 8 <= nrows;
      off = (unsigned)k * ld + (unsigned)col;
    }
    const unsigned vo = valid ? off * esz : OOB;
    const v4u x = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
    r[i].x = make_uint4(x.x, x.y, x.z, x.w);
    if constexpr (F32) {
      const v4u y = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16, 0, 0);
      r[i].y = make_uint4(y.x, y.y, y.z, y.w);
    } else {
      r[i].y = make_uint4(0, 0, 0, 0);
    }
  }
}

template <bool KM, int ROWS = 128>
DEV void store_operand(unsigned char* img, int tid, const Raw8 (&r)[4], bool f32) {
  if constexpr (!KM) {
    const int srow = tid >> 3, sc = tid & 7;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) *reinterpret_cast<uint4*>(img + tile_off(srow + 32 * i, sc)) = to_bf16(r[i], f32);
  } else if constexpr (ROWS == 128) {
    const int krow = tid >> 4, ch = tid & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(img + km_off(krow + 16 * i, ch >> 1) + (ch & 1) * 16) = to_bf16(r[i], f32);
  } else {  // 64 operand rows: half of every 256-byte k row is used (same swizzle, same transposing reads)
    const int krow = tid >> 3, ch = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<uint4*>(img + km_off(krow + 32 * i, ch >> 1) + (ch & 1) * 16) = to_bf16(r[i], f32);
  }
}

// fragment of operand rows [r0, r0+16) for k-step ks (32 reduction indices) of the staged tile
template <bool KM>
DEV bf16x8 read_frag(const unsigned char* img, int r0, int ks, int lane) {
  const int lg = lane >> 4, ln = lane & 15;
  if constexpr (!KM) {
    return *reinterpret_cast<const bf16x8*>(img + tile_off(r0 + ln, ks * 4 + lg));
  } else {
    const int k = ks * 32 + 8 * lg + (ln >> 2);
    const unsigned char* a0 = img + km_off(k, r0 >> 4) + (ln & 3) * 8;  // key(k) == key(k + 4)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * 256));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  }
}

// The same k-major fragment read as inline assembly, for the LDS-DMA loop: behind the builtin hipcc waits vmcnt(0)
// before every LDS read that follows a buffer_load ... lds (it cannot tell the stages apart), which would serialise
// the loop again.  The caller counts lgkmcnt itself (FRAG_WAIT).  OFF = byte offset of the k-step inside the tile.
template <int OFF>
DEV bf16x8 read_frag_km_asm(unsigned lds_addr) {
  s16x4 lo, hi;
#ifdef GL_NO_READ
  asm volatile("" : "=v"(lo) : "v"(lds_addr), "n"(OFF));
  asm volatile("" : "=v"(hi) : "v"(lds_addr), "n"(OFF + 1024));
#else
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(lds_addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(lds_addr), "n"(OFF + 1024));
#endif
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
#define FRAG_WAIT(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]))

// one k-step of the LDS-DMA loop: fragments of both operands (OFF = the k-step's byte offset in the tile) ...
template <int OFF>
DEV void gl_read_step(bf16x8 (&a)[4], bf16x8 (&b)[4], unsigned st, const unsigned (&fa)[4], const unsigned (&fb)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = read_frag_km_asm<OFF>(st + fa[i]);
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = read_frag_km_asm<128 * 128 + OFF>(st + fb[j]);
}
// ... and its MFMAs, accumulating in place (transposed product, as the compiler-scheduled loop).  hipcc does not see
// an MFMA inside an asm statement, so the wait states it would insert are ours: an MFMA reading a register a VALU
// instruction wrote needs 2 - the fragments come from LDS (s_waitcnt), the accumulators are zeroed long before
// (s_nop after the zeroing), the vector of ones is a constant the compiler may re-create anywhere (s_nop in place).
// -DGL_NO_MFMA / -DGL_NO_READ / -DGL_NO_DMA: diagnostic builds of the LDS-DMA loop with one of its three streams knocked
// out (wrong results; tools/probes/wgrad_rate.py times what is left) - not part of the product build
#ifdef GL_NO_MFMA
#define GL_MFMA(c, x, y) asm volatile("" : "+a"(c) : "v"(x), "v"(y))
#else
#define GL_MFMA(c, x, y) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(x), "v"(y))
#endif
template <bool CS>
DEV void gl_mfma_step(f32x4 (&acc)[4][4], f32x4 (&cs)[4], const bf16x8& ones, const bf16x8 (&a)[4], const bf16x8 (&b)[4]) {
  asm volatile("s_nop 1");  // should hipcc ever copy a fragment (VALU) between its wait and these MFMAs: 2 wait states
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) GL_MFMA(acc[i][j], b[j], a[i]);
    if constexpr (CS) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(cs[i]) : "v"(ones), "v"(a[i]));
  }
}

// The MFMAs of one k-step with the fragment reads of the NEXT k-step between them: one fragment (two transposing reads)
// behind each of the first eight MFMAs, so the reads are all issued while the matrix pipe works and their latency runs
// under the last eight.  As two blocks - 16 reads, then 16 MFMAs - the pipe stood idle while a wave (one per SIMD,
// in-order issue) sent its reads and its LDS-DMA requests: 552 ns per 128 x 128 x 64 tile step on an otherwise idle
// GPU against 213 ns of MFMA work (tools/probes/wgrad_rate.py).
template <bool CS, int OFF>
DEV void gl_mfma_read_step(f32x4 (&acc)[4][4], f32x4 (&cs)[4], const bf16x8& ones, const bf16x8 (&a)[4], const bf16x8 (&b)[4],
                           bf16x8 (&an)[4], bf16x8 (&bn)[4], unsigned st, const unsigned (&fa)[4], const unsigned (&fb)[4]) {
  asm volatile("s_nop 1");
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      GL_MFMA(acc[i][j], b[j], a[i]);
      const int m = i * 4 + j;
      if (m < 4) an[m] = read_frag_km_asm<OFF>(st + fa[m]);
      else if (m < 8) bn[m - 4] = read_frag_km_asm<128 * 128 + OFF>(st + fb[m - 4]);
    }
    if constexpr (CS) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(cs[i]) : "v"(ones), "v"(a[i]));
  }
}

// -DBG_STAMPS: per-workgroup cycle stamps of the main loop's phases (tools/bgemm_stamps.py); not part of the product build
#ifdef BG_STAMPS
__device__ unsigned* g_bg_stamps = nullptr;
#define STAMP(i) do { if (tid == 0) stamps[(i)] = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
// FAST: whole 8-element groups and operands below 4 GB -> descriptor loads with compile-time operand types.
// COLSUM: the workgroups of the first tile column also sum the A operand over the reduction index with one more MFMA
// per A fragment against a fragment of ones (D[n][m] = sum_k 1 * A[k][m]: every row of the result is the column sum).
// TN: output columns per tile.  64 (FAST kernels only): products whose N is 64 (the per-head attention gradients) or
// whose 128-wide tiling leaves less than two workgroups per CU - a wave then owns 64 x 32 of the tile.
// GL (both operands bf16 and reduction-strided, whole tiles: the weight-gradient products): the tiles travel global ->
// LDS by LDS-DMA (buffer_load ... lds, 16 bytes per lane, no register hop) into GL_STAGES stages, GL_STAGES - 1 tiles
// ahead of the MFMAs, behind counted vmcnt waits and a bare s_barrier - the register-staged loop above it is phase-serial with
// one wave per SIMD (request loads 580 cycles, fragments 380, MFMAs 620, wait + LDS store 364, barrier 144 per
// 64-deep tile).  The LDS image is lane-linear per wave instruction, so the bank swizzle of km_off is applied to
// the SOURCE address (XOR is its own inverse).
constexpr int GL_STAGES = 4;  // 128 KB of the CU's 160 KB: one workgroup per CU, 96 KB of operands in flight (5 stages: no faster, measured)
#define GL_WAIT_TILE_(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define GL_WAIT_TILE() GL_WAIT_TILE_(16)
static_assert(8 * (GL_STAGES - 2) == 16, "GL_WAIT_TILE: 8 LDS-DMA instructions per wave and tile, GL_STAGES - 2 newer tiles in flight");
// one 128 x TN output tile (bx, by) of batch entry / reduction split bz; the kernels below pick the tile
template <bool A_KM, bool B_KM, bool FAST, bool A_F32, bool B_F32, bool COLSUM = false, int TN = 128, bool GL = false>
DEV void bgemm_tile(const BgemmArgs& g, const int bx, const int by, const int bz) {
  static_assert(TN == 128 || (TN == 64 && FAST), "narrow tiles use the descriptor loaders");
  static_assert(!GL || (A_KM && B_KM && FAST && !A_F32 && !B_F32 && TN == 128), "LDS-DMA path: bf16 k-major operands, 128 x 128 tiles");
  constexpr int NJ = TN / 32, WN = TN / 2;  // 16-column fragments per wave, columns per wave
  __shared__ __attribute__((aligned(16))) unsigned char smem[(GL ? GL_STAGES : 2) * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  // GL: 8 waves - waves 0..3 own the four 64 x 64 quarters of the tile (fragment reads and MFMAs only), waves 4..7 do
  // nothing but issue the LDS-DMA pieces (wave 4 + w the ones wave w used to issue) and wait for them.  One piece costs
  // its wave 60-180 cycles of issue; issued by the MFMA waves themselves (one wave per SIMD, in order) the eight pieces
  // per tile step stood in front of the MFMAs: 530 ns per step on an idle GPU against 213 of MFMA work, 492 with the
  // fragment reads knocked out, 350 with the MFMAs knocked out (tools/probes/wgrad_rate.py, -DGL_NO_*).  A loader
  // wave shares its SIMD with an MFMA wave and its issue stalls are that wave's alone.
  const int wave_all = GL ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
  const bool loader = GL && wave_all >= 4;   // wave-uniform
  const int wave = GL ? (wave_all & 3) : wave_all;
#ifdef BG_STAMPS
  __shared__ unsigned stamps[96];
  if (tid < 96) stamps[tid] = 0;
  __syncthreads();
  STAMP(0);
#endif
  const int wm = wave >> 1, wn = wave & 1;
  const int lg = lane >> 4, ln = lane & 15;
  const int m0 = by * 128, n0 = bx * TN;
  const int z = bz / g.splits, split = bz % g.splits;
  const int z1 = z / g.nb2, z2 = z % g.nb2;
  int kbeg = split * g.k_per_split, kend = min(g.K, kbeg + g.k_per_split);
  if (g.band > 0) {  // banded A: only the reduction tiles that meet rows m0 .. m0 + 127
    if constexpr (!A_KM) {
      kbeg = max(kbeg, (m0 / BK) * BK);
      kend = min(kend, (m0 + 128 + g.band + BK - 1) / BK * BK);
    } else {  // k = split * period + t * stride + h, nonzero for t in [m - band, m]
      const int lo = max(0, m0 - g.band) * g.band_stride, hi = (m0 + 128) * g.band_stride;
      const int base = split * g.k_per_split;
      kend = min(kend, base + (hi + BK - 1) / BK * BK);
      kbeg = min(kend, base + lo / BK * BK);
    }
  }
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const bool af32 = FAST ? A_F32 : g.a_f32 != 0, bf32 = FAST ? B_F32 : g.b_f32 != 0;
  const void* Ab = reinterpret_cast<const unsigned char*>(g.A) + (z1 * g.sA1 + z2 * g.sA2) * (af32 ? 4 : 2);
  const void* Bb = reinterpret_cast<const unsigned char*>(g.B) + (z1 * g.sB1 + z2 * g.sB2) * (bf32 ? 4 : 2);

  Raw8 ra[4], rb[4];
  // descriptors over exactly this batch entry's operands (wave-uniform: built from kernel arguments and blockIdx)
  const unsigned a_bytes = ((A_KM ? (unsigned)(g.K - 1) * (unsigned)g.lda + g.M : (unsigned)(g.M - 1) * (unsigned)g.lda + g.K)) * (A_F32 ? 4u : 2u);
  const unsigned b_bytes = ((B_KM ? (unsigned)(g.K - 1) * (unsigned)g.ldb + g.N : (unsigned)(g.N - 1) * (unsigned)g.ldb + g.K)) * (B_F32 ? 4u : 2u);
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(Ab), 0, FAST ? a_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(Bb), 0, FAST ? b_bytes : 0, 0x00020000);
  auto load_tile = [&](int k0) {
    if constexpr (FAST) {
      load_operand_buf<A_KM, A_F32>(rsa, (unsigned)g.lda, m0, g.M, k0, kend, tid, ra);
      load_operand_buf<B_KM, B_F32, TN>(rsb, (unsigned)g.ldb, n0, g.N, k0, kend, tid, rb);
    } else {
      load_operand<A_KM>(Ab, af32, g.lda, m0, g.M, k0, kend, tid, ra);
      load_operand<B_KM>(Bb, bf32, g.ldb, n0, g.N, k0, kend, tid, rb);
    }
  };
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 cs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_cs = COLSUM && bx == 0 && wn == 0;  // wave-uniform
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  if constexpr (GL) {
    // one tile = 2 x 16 wave instructions of 1 KB (4 k rows of 256 B); wave w issues instructions 4w .. 4w + 3 of each
    // operand: lane l lands at row 4q + (l >> 4), 16-byte piece l & 15 of the image, and fetches the piece the swizzle
    // wants there
    // M, N are whole tiles and the split's reduction range whole 64-deep tiles here (host), so a tile is either all
    // there or all beyond the range: per-lane byte offsets inside a tile once, the tile's offset (or OOB) per issue
    unsigned ga[4], gb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = wave * 4 + j;
      const int kl = q * 4 + (lane >> 4), cp = lane & 15;
      const int grp = (cp >> 1) ^ ((kl & 3) | (((kl >> 3) & 1) << 2));
      const int e0 = (2 * grp + (cp & 1)) * 8;
      ga[j] = ((unsigned)kl * (unsigned)g.lda + (unsigned)(m0 + e0)) * 2u;
      gb[j] = ((unsigned)kl * (unsigned)g.ldb + (unsigned)(n0 + e0)) * 2u;
    }
    auto glds_tile = [&](int k0, unsigned so) {
#ifdef GL_NO_DMA
      const bool valid = false;   // every request beyond the descriptor: zeros, no memory traffic
#else
      const bool valid = k0 < kend;
#endif
      const unsigned ka = (unsigned)k0 * (unsigned)g.lda * 2u, kb = (unsigned)k0 * (unsigned)g.ldb * 2u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = wave * 4 + j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void*)(smem + so + q * 1024), 16, (int)(valid ? ga[j] + ka : OOB), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (__attribute__((address_space(3))) void*)(smem + so + 128 * 128 + q * 1024), 16, (int)(valid ? gb[j] + kb : OOB), 0, 0, 0);
      }
    };
    // per-lane part of the fragment addresses (read_frag<true>): k = 8 (lane >> 4) + ((lane & 15) >> 2) within the
    // k-step, 8 bytes per lane & 3; km_off's swizzle key depends on k & 3 and bit 3 of k only, so the k-step (a
    // multiple of 32) and the stage add linearly and ride in the instruction's offset field
    const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned fa[4], fb[4];
    {
      const int kf = 8 * (lane >> 4) + ((lane & 15) >> 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = km_off(kf, wm * 4 + i) + (lane & 3) * 8;
        fb[i] = km_off(kf, wn * 4 + i) + (lane & 3) * 8;
      }
    }
    // The loop in inline assembly statements, which hipcc keeps in source order.  Left to the compiler, the second
    // k-step's MFMAs get destination tuples shifted against their sources and 48 accumulators travel through register
    // copies per tile; and the LDS reads would wait for vmcnt(0) (read_frag_km_asm).  Schedule per tile, one barrier:
    //   wait my part of tile kt | barrier | read k-step 0 of tile kt | MFMAs of k-step 1 of tile kt - 1 (they hide the
    //   barrier and the LDS latency) | issue tile kt + 2 | read k-step 1 | MFMAs of k-step 0 | fragments landed
    // The body exists twice, with and without the column-sum MFMAs (a wave-uniform branch inside would sit between asm
    // statements).  Every iteration issues a tile - beyond the reduction range the offsets are outside the descriptor:
    // zeros, no memory traffic - so the vmcnt count is the same everywhere.
    if (loader) {
      // tile kt lives in stage kt % GL_STAGES; tiles kt + 1 .. kt + GL_STAGES - 2 may still be in flight when tile kt is
      // needed (8 pieces per loader wave each), and tile kt + GL_STAGES - 1 goes into the stage tile kt - 1 was read from.
      // Every iteration issues a tile - beyond the reduction range the offsets are outside the descriptor: zeros, no
      // memory traffic - so the vmcnt count is the same everywhere.  One barrier per tile step, shared with the MFMA waves.
      if (nk > 0) {
#pragma unroll
        for (int t = 0; t < GL_STAGES - 1; ++t) glds_tile(kbeg + t * BK, t * STAGE);
        unsigned cur = 0, prev = (GL_STAGES - 1) * STAGE;
        GL_WAIT_TILE();                // my part of tile 0 has landed
        __builtin_amdgcn_s_barrier();
        glds_tile(kbeg + (GL_STAGES - 1) * BK, prev);
        for (int kt = 1; kt < nk; ++kt) {
          prev = cur;
          cur = (cur == (GL_STAGES - 1) * STAGE) ? 0 : cur + STAGE;
          GL_WAIT_TILE();                // my part of tile kt has landed
          __builtin_amdgcn_s_barrier();  // all of tile kt is in LDS, and every MFMA wave is done reading tile kt - 1
          glds_tile(kbeg + (kt + GL_STAGES - 1) * BK, prev);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zero tiles behind the last one
      }
      return;  // the loader waves take no part in the epilogue (no barrier follows)
    }
    // The MFMA waves' loop in inline assembly statements, which hipcc keeps in source order.  Left to the compiler, the
    // second k-step's MFMAs get destination tuples shifted against their sources and 48 accumulators travel through
    // register copies per tile; and the LDS reads would wait for vmcnt(0) (read_frag_km_asm).  Per tile, one barrier:
    //   barrier | MFMAs of k-step 1 of tile kt - 1 with the reads of k-step 0 of tile kt between them | fragments landed
    //   | MFMAs of k-step 0 with the reads of k-step 1 between them | fragments landed
    // The body exists twice, with and without the column-sum MFMAs (a wave-uniform branch inside would sit between asm
    // statements).
    auto main_loop = [&](auto with_cs) {
      constexpr bool CS = decltype(with_cs)::value;
      bf16x8 a0[4], b0[4], a1[4], b1[4];
      auto read_step0 = [&](unsigned st) { gl_read_step<0>(a0, b0, st, fa, fb); };
      auto mfma_step = [&](bf16x8 (&a)[4], bf16x8 (&b)[4]) { gl_mfma_step<CS>(acc, cs, ones, a, b); };
      // MFMAs of the k-step held in (a1, b1) while k-step 0 of the tile at st is read into (a0, b0), and the other way round
      auto mfma1_read0 = [&](unsigned st) { gl_mfma_read_step<CS, 0>(acc, cs, ones, a1, b1, a0, b0, st, fa, fb); };
      auto mfma0_read1 = [&](unsigned st) { gl_mfma_read_step<CS, 32 * 256>(acc, cs, ones, a0, b0, a1, b1, st, fa, fb); };
      unsigned cur = 0;
      __builtin_amdgcn_s_barrier();    // tile 0 is in LDS
      read_step0(smem_lds);
      FRAG_WAIT(0, a0);
      FRAG_WAIT(0, b0);
      asm volatile("s_nop 4");  // accumulators zeroed by VALU moves
      mfma0_read1(smem_lds);
      FRAG_WAIT(0, a1);
      FRAG_WAIT(0, b1);
      for (int kt = 1; kt < nk; ++kt) {
        cur = (cur == (GL_STAGES - 1) * STAGE) ? 0 : cur + STAGE;
        __builtin_amdgcn_s_barrier();  // all of tile kt is in LDS, and every wave is done reading tile kt - 1
        mfma1_read0(smem_lds + cur);   // k-step 1 of tile kt - 1 | fragments of k-step 0 of tile kt
        FRAG_WAIT(0, a0);
        FRAG_WAIT(0, b0);
        mfma0_read1(smem_lds + cur);
        FRAG_WAIT(0, a1);
        FRAG_WAIT(0, b1);
      }
      mfma_step(a1, b1);
      // the MFMA results are read by ordinary instructions next, which the compiler cannot know to delay
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    };
    if (nk > 0) {
      if (COLSUM && do_cs) main_loop(std::true_type{});
      else main_loop(std::false_type{});
    }
  } else {
  if (nk > 0) {
    load_tile(kbeg);
    store_operand<A_KM>(smem, tid, ra, af32);
    store_operand<B_KM, TN>(smem + 128 * 128, tid, rb, bf32);
  }
  __syncthreads();
  STAMP(1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      load_tile(kbeg + (kt + 1) * BK);
    }
    __builtin_amdgcn_sched_barrier(0);  // hipcc would sink the loads / hoist their conversion next to each other
    if (kt < 12) STAMP(2 + kt * 5);  // loads requested
    const unsigned char* as = smem + (kt & 1) * STAGE;
    const unsigned char* bs = as + 128 * 128;
    // both k-steps' fragments are requested before the first MFMA: the second half's LDS reads run under the first
    // half's MFMAs (with one wave per SIMD nothing else would overlap them)
    bf16x8 af[2][4], bf[2][NJ];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) af[ks][i] = read_frag<A_KM>(as, wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[ks][j] = read_frag<B_KM>(bs, wn * WN + j * 16, ks, lane);
    }
#ifdef BG_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (kt < 12) STAMP(3 + kt * 5);  // fragments read
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);  // C^T tile
    if constexpr (COLSUM) {
      if (do_cs) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 4; ++i) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[ks][i], cs[i], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt < 12) STAMP(4 + kt * 5);  // MFMAs issued
    if (kt + 1 < nk) {
      unsigned char* nx = smem + ((kt + 1) & 1) * STAGE;
      store_operand<A_KM>(nx, tid, ra, af32);
      store_operand<B_KM, TN>(nx + 128 * 128, tid, rb, bf32);
    }
    if (kt < 12) STAMP(5 + kt * 5);  // next tile in LDS
    __syncthreads();
    if (kt < 12) STAMP(6 + kt * 5);
  }
  }  // !GL
  STAMP(62);

  // ---- epilogue.  The MFMAs compute the transposed tile (B fragment as the A operand), so acc[i][j][r] is
  //   row = m0 + wm*64 + i*16 + ln,  col = n0 + wn*64 + j*16 + lg*4 + r :
  // a lane owns FOUR CONSECUTIVE COLUMNS of one row per (i, j) and stores them as one 16-byte (f32) / 8-byte (bf16)
  // piece.  With one element per store the epilogue of a K = 512 product cost as much as its main loop (1 300 VALU
  // instructions of address arithmetic and 64 store instructions per lane against 256 MFMAs).
  if constexpr (COLSUM) {
    if (do_cs && lg == 0) {  // rows 0..3 of the 16 x 16 result are all the column sum: lane ln of the first row group stores it
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + ln;
        if (m < g.M) {
          if (g.splits > 1) g.cs_slab[(int64_t)split * g.M + m] = cs[i][0];
          else g.cs_out[m] = (m < g.head_rows ? g.alpha_head : g.alpha) * cs[i][0];
        }
      }
    }
  }
  const int64_t coff = z1 * g.sC1 + z2 * g.sC2 + (int64_t)split * g.slab;
  const int64_t xo = z1 * g.sX1 + z2 * g.sX2;
  const bool to_slab = g.splits > 1;
  const bool obf = (g.epi & CSSHIP_BEPI_OUT_BF16) != 0;
  // whole 4-column groups and 16- / 8-byte aligned rows (checked once per launch: wave-uniform)
  const bool vec = (g.N & 3) == 0 &&
                   (to_slab || ((g.ldc & 3) == 0 && (coff & 3) == 0 && ((uintptr_t)g.C & 15) == 0 &&
                                (!g.bias || ((uintptr_t)g.bias & 15) == 0) &&
                                (!(g.epi & (CSSHIP_BEPI_MASK_POS | CSSHIP_BEPI_ADD)) ||
                                 ((g.ldaux & 3) == 0 && (xo & 3) == 0 && ((uintptr_t)g.aux & 15) == 0))));
  if (!to_slab && vec) {
    // Vector epilogue in TWO phases: every operand the epilogue reads (bias, mask or addend: 16 bytes per piece) is
    // requested first, then all pieces are finished and stored.  Interleaved - load, use, store, per piece - every piece
    // was a round trip of its own: the loads sit behind run-time epilogue flags, hipcc's s_waitcnt insertion then assumes
    // nothing about what is in flight and waits for vmcnt(0) before each use, which also waits for the previous piece's
    // STORE to be acknowledged (16 dependent round trips per lane in the two position products of the attention backward).
    float4 bq[NJ], ax[4][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      bq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) ax[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (g.bias) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wn * WN + j * 16 + lg * 4;
        if (col < g.N) bq[j] = *reinterpret_cast<const float4*>(g.bias + col);
      }
    }
    if (g.epi & (CSSHIP_BEPI_MASK_POS | CSSHIP_BEPI_ADD)) {
      const bool msk = (g.epi & CSSHIP_BEPI_MASK_POS) != 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + wm * 64 + i * 16 + ln;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = n0 + wn * WN + j * 16 + lg * 4;
          if (row < g.M && col < g.N) {
            const int64_t ao = xo + (int64_t)row * g.ldaux + col;
            if (msk) {
              const uint2 m = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(g.aux) + ao);
              ax[i][j].x = __uint_as_float(m.x);
              ax[i][j].y = __uint_as_float(m.y);
            } else {
              ax[i][j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.aux) + ao);
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + wm * 64 + i * 16 + ln;
      if (row >= g.M) continue;
      const float al = (COLSUM && row < g.head_rows) ? g.alpha_head : g.alpha;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wn * WN + j * 16 + lg * 4;
        if (col >= g.N) continue;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        const float b[4] = {bq[j].x, bq[j].y, bq[j].z, bq[j].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (g.epi & CSSHIP_BEPI_SCALE_BIAS) ? al * (v[r] + b[r]) : al * v[r] + b[r];
        if (g.epi & CSSHIP_BEPI_MASK_POS) {
          const unsigned mx = __float_as_uint(ax[i][j].x), my = __float_as_uint(ax[i][j].y);
          if (!(__uint_as_float(mx << 16) > 0.f)) v[0] = 0.f;
          if (!(__uint_as_float(mx & 0xffff0000u) > 0.f)) v[1] = 0.f;
          if (!(__uint_as_float(my << 16) > 0.f)) v[2] = 0.f;
          if (!(__uint_as_float(my & 0xffff0000u) > 0.f)) v[3] = 0.f;
        }
        if (g.epi & CSSHIP_BEPI_ADD) {
          v[0] += ax[i][j].x; v[1] += ax[i][j].y; v[2] += ax[i][j].z; v[3] += ax[i][j].w;
        }
        if (g.epi & CSSHIP_BEPI_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        const int64_t co = coff + (int64_t)row * g.ldc + col;
        if (obf)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(g.C) + co) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        else
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + co) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  } else {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + wm * 64 + i * 16 + ln;
    if (row >= g.M) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + wn * WN + j * 16 + lg * 4;
      if (col >= g.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (to_slab) {
        float* dst = reinterpret_cast<float*>(g.C) + coff + (int64_t)row * g.N + col;
        if (vec) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (col + r < g.N) dst[r] = v[r];
        }
        continue;
      }
      const int64_t co = coff + (int64_t)row * g.ldc + col, ao = xo + (int64_t)row * g.ldaux + col;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (col + r >= g.N) continue;
        const float bias = g.bias ? g.bias[col + r] : 0.f;
        const float al = (COLSUM && row < g.head_rows) ? g.alpha_head : g.alpha;
        float x = (g.epi & CSSHIP_BEPI_SCALE_BIAS) ? al * (v[r] + bias) : al * v[r] + bias;
        if (g.epi & CSSHIP_BEPI_MASK_POS) {
          if (!(bf2f(reinterpret_cast<const unsigned short*>(g.aux)[ao + r]) > 0.f)) x = 0.f;
        }
        if (g.epi & CSSHIP_BEPI_ADD) x += reinterpret_cast<const float*>(g.aux)[ao + r];
        if (g.epi & CSSHIP_BEPI_RELU) x = fmaxf(x, 0.f);
        if (obf) reinterpret_cast<unsigned short*>(g.C)[co + r] = f2bf(x);
        else reinterpret_cast<float*>(g.C)[co + r] = x;
      }
    }
  }
  }
#ifdef BG_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(63);  // result stores drained
  if (g_bg_stamps && tid < 64) {  // wave 0 wrote them itself: no barrier (the loader waves of the GL form are gone by now)
    const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    g_bg_stamps[lin * 96 + tid] = stamps[tid];
    if (tid < 32) g_bg_stamps[lin * 96 + 64 + tid] = stamps[64 + tid];
  }
#endif
}

template <bool A_KM, bool B_KM, bool FAST, bool A_F32, bool B_F32, bool COLSUM = false, int TN = 128, bool GL = false>
__global__ __launch_bounds__(GL ? 512 : 256) void bgemm_kernel(BgemmArgs g) {
  // XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin by their linear id, and every XCD has its own L2:
  // in the natural order the tiles that share operand rows (all output tiles of one K-split, or of one batch entry)
  // land on 8 different XCDs and each pulls the same rows through its own L2 (166 MB of L2 fills for 39 MB of
  // operands in the FeedForward dW product).  Here the workgroups of one XCD walk all tiles of one (batch, split)
  // slice before the next, so the re-reads hit that XCD's L2.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const unsigned tiles = gridDim.x * gridDim.y, nz = gridDim.z;
    if ((nz & 7) == 0) {
      const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      const unsigned xcd = lin & 7, j = lin >> 3;
      const unsigned t = j % tiles;
      bz = (j / tiles) * 8 + xcd;
      bx = t % gridDim.x;
      by = t / gridDim.x;
    }
  }
  bgemm_tile<A_KM, B_KM, FAST, A_F32, B_F32, COLSUM, TN, GL>(g, bx, by, bz);
}

// ---- the weight gradients of a whole encoder block as ONE launch (csship_wgrad_hold / csship_wgrad_flush).
// Launched one by one, each of the block's six products (four FeedForward weights, q | k | v, the output projection:
// 64 + 12 + 4 tiles of 128 x 128, all reducing over the same B*T rows) has to cut its reduction into 16-64 pieces to
// occupy 256 CUs: a workgroup runs 3-12 tiles of k, most of its time is pipeline fill and the 64 KB slab it writes, and
// the block moves ~200 MB of slabs.  Together they are 80 tiles: three pieces of the reduction fill the chip once,
// every workgroup streams 67 tiles of k and the slabs shrink to 16 MB.  The table maps workgroup -> (product, split,
// tile) so that the workgroups of one XCD (linear id mod 8) walk whole (product, split) units: tiles of one unit share
// operand rows and run in step, so the re-reads hit that XCD's L2.
constexpr int WG_JOBS = 12, WG_MAXWG = 1024;
constexpr unsigned short WG_IDLE = 0xFFFFu;
struct WgJob {
  const void* A;  // dY  bf16 [K, M] (row stride lda)
  const void* B;  // X   bf16 [K, N] (row stride ldb)
  float* C;       // [splits][M * N] slabs
  float* cs;      // [splits][M] column-sum slabs
  int M, N, lda, ldb;
};
struct WgGroup {
  WgJob job[WG_JOBS];
  int K, splits, k_per_split, pad;
  unsigned short map[WG_MAXWG];  // job << 12 | split << 8 | tile
};

__global__ __launch_bounds__(512) void bgemm_wgrad_group_kernel(WgGroup G) {
  const unsigned e = G.map[blockIdx.x];
  if (e == WG_IDLE) return;
  const WgJob& J = G.job[e >> 12];
  BgemmArgs g{};
  g.A = J.A; g.B = J.B; g.C = J.C;
  g.M = J.M; g.N = J.N; g.K = G.K;
  g.lda = J.lda; g.ldb = J.ldb; g.ldc = J.N;
  g.nb2 = 1; g.splits = G.splits; g.k_per_split = G.k_per_split;
  g.slab = (int64_t)J.M * J.N;
  g.alpha = 1.f;
  g.cs_slab = J.cs;
  const int tx = J.N >> 7, tile = (int)(e & 255u);
  bgemm_tile<true, true, true, false, false, true, 128, true>(g, tile % tx, tile / tx, (int)((e >> 8) & 15u));
}

// out[r, c] = alpha * sum_s slab[s][r, c] (+ bias[c]) (+ add[r, c]); fixed summation order.
// Workgroup = 64 lanes x V consecutive elements x NW waves; wave w adds slabs w, w+NW, ... with eight loads in flight,
// then one LDS reduction - short outputs with many slabs (LayerNorm / bias partials) do not serialise on one thread,
// long ones (V = 4: N and the slab size are multiples of 4) move 16 bytes per load instruction.
// one virtual workgroup of 64 * NW threads (thread index tix) reducing 64 * V elements; red: its NW x 64 vectors of LDS.
// tail: the short second output of a weight-gradient product (head_rows counts its elements, not rows).
template <int V, int NW, int U>
DEV void slab_reduce_body(const float* __restrict__ slabs, int S, int64_t n, int N, int64_t ldc, float alpha,
                          const float* __restrict__ bias, const float* add, int64_t ldadd, float* out, float alpha_head,
                          int head_rows, bool tail, int bid, int tix, float (*red_)[64][V]) {
  typedef float vec __attribute__((ext_vector_type(V)));
  vec(*red)[64] = reinterpret_cast<vec(*)[64]>(red_);
  const int lane = tix & 63, w = tix >> 6;
  const int64_t i = ((int64_t)bid * 64 + lane) * V;
  vec s[U];
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = 0.f;
  if (i < n) {
    int k = w;
    for (; k + (U - 1) * NW < S; k += U * NW) {  // U independent loads in flight per lane
#pragma unroll
      for (int u = 0; u < U; ++u) s[u] += *reinterpret_cast<const vec*>(slabs + (int64_t)(k + u * NW) * n + i);
    }
    if (k < S) {  // last, partial round: no branch around the loads (out-of-range slabs re-read the last one, masked)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kk = k + u * NW;
        const vec v = *reinterpret_cast<const vec*>(slabs + (int64_t)min(kk, S - 1) * n + i);
        s[u] += kk < S ? v : (vec)0.f;
      }
    }
  }
  vec tot = s[0];
#pragma unroll
  for (int u = 1; u < U; ++u) tot += s[u];
  red[w][lane] = tot;
  __syncthreads();
  if (w != 0 || i >= n) return;
  const int64_t r = i / N, c = i % N;
  vec t = red[0][lane];
#pragma unroll
  for (int j = 1; j < NW; ++j) t += red[j][lane];
#pragma unroll
  for (int e = 0; e < V; ++e) {  // V = 4: the four elements share a row (N % 4 == 0)
    // head_rows: the first rows of the main output (first elements of the tail vector) take alpha_head
    float v = ((tail ? c + e : r) < head_rows ? alpha_head : alpha) * t[e];
    if (bias) v += bias[c + e];
    if (add) v += add[r * ldadd + c + e];
    out[r * ldc + c + e] = v;
  }
}

template <int V, int NW, int U>
__global__ __launch_bounds__(64 * NW) void slab_reduce_kernel(const float* __restrict__ slabs, int S, int64_t n, int N,
                                                              int64_t ldc, float alpha, const float* __restrict__ bias,
                                                              const float* add, int64_t ldadd, float* out, int nb_main,
                                                              const float* __restrict__ slabs2, int64_t n2,
                                                              float* __restrict__ out2, float alpha_head, int head_rows) {
  __shared__ __attribute__((aligned(16))) float red[NW][64][V];
  int bid = blockIdx.x;
  bool tail = false;
  if (bid >= nb_main) {  // second, short output reduced by the same launch (the bias gradient next to its weight gradient)
    bid -= nb_main;
    tail = true;
    slabs = slabs2; n = n2; N = (int)n2; ldc = n2; bias = nullptr; add = nullptr; out = out2;
  }
  slab_reduce_body<V, NW, U>(slabs, S, n, N, ldc, alpha, bias, add, ldadd, out, alpha_head, head_rows, tail, bid,
                             (int)threadIdx.x, red);
}

// ---- deferred reductions: up to RB_MAX queued jobs in ONE launch of 1024-thread workgroups.  Every job keeps the
// summation structure (V, NW, U) its own launch would use, so the results are the same bit for bit: a 16-wave job takes a
// whole workgroup per virtual workgroup, a 4-wave job four virtual workgroups per workgroup.
constexpr int RB_MAX = 36;
struct ReduceJob {
  const float* slabs;
  const float* bias;
  const float* add;
  float* out;
  int64_t n, ldc, ldadd;
  int S, N, head_rows, kind;  // kind: 0 = <4,4,8>, 1 = <1,16,8>, 2 = <1,4,4>, 3 = flat (S <= 4); | 4 = tail output
  float alpha, alpha_head;
  int nvb, pad_;              // virtual workgroups
};
struct ReduceBatch {
  ReduceJob job[RB_MAX];
  int first[RB_MAX + 1];  // first workgroup of every job
  int njobs;
};

// at most four slabs of a long output (the grouped weight gradients: 2-3 pieces of the reduction): one thread per four
// consecutive elements sums them with all loads in flight; the wave-per-slab structure above would issue 32 loads per
// element group for the three it needs.
DEV void slab_sum_flat(const ReduceJob& q, int wg, int t, bool tail) {
  const int64_t i = ((int64_t)wg * 1024 + t) * 4;
  if (i >= q.n) return;
  float4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(q.slabs + (int64_t)min(k, q.S - 1) * q.n + i);
  float4 a = v[0];
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (k < q.S) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
  const int64_t r = i / q.N, c = i % q.N;
  const float al = (tail ? c : r) < q.head_rows ? q.alpha_head : q.alpha;  // (tail: head_rows % 4 == 0 on this path)
  float4 o = make_float4(al * a.x, al * a.y, al * a.z, al * a.w);
  if (q.bias) {
    const float4 bq = *reinterpret_cast<const float4*>(q.bias + c);
    o.x += bq.x; o.y += bq.y; o.z += bq.z; o.w += bq.w;
  }
  if (q.add) {
    const float4 ad = *reinterpret_cast<const float4*>(q.add + r * q.ldadd + c);
    o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
  }
  *reinterpret_cast<float4*>(q.out + r * q.ldc + c) = o;
}

__global__ __launch_bounds__(1024) void slab_reduce_batch_kernel(ReduceBatch b) {
  __shared__ __attribute__((aligned(16))) float red[4][4][64][4];
  int j = 0;
  while (j + 1 < b.njobs && (int)blockIdx.x >= b.first[j + 1]) ++j;  // wave-uniform
  const ReduceJob& q = b.job[j];
  const int wg = blockIdx.x - b.first[j], t = threadIdx.x;
  const int kind = q.kind & 3;
  const bool tail = (q.kind & 4) != 0;
  if (kind == 3) {
    slab_sum_flat(q, wg, t, tail);
  } else if (kind == 1) {
    slab_reduce_body<1, 16, 8>(q.slabs, q.S, q.n, q.N, q.ldc, q.alpha, q.bias, q.add, q.ldadd, q.out, q.alpha_head,
                               q.head_rows, tail, wg, t, reinterpret_cast<float(*)[64][1]>(&red[0][0][0][0]));
  } else {
    // virtual workgroups beyond the job's last one find i >= n and only take part in the barrier
    const int sub = t >> 8, vb = wg * 4 + sub;
    if (kind == 0)
      slab_reduce_body<4, 4, 8>(q.slabs, q.S, q.n, q.N, q.ldc, q.alpha, q.bias, q.add, q.ldadd, q.out, q.alpha_head,
                                q.head_rows, tail, vb, t & 255, reinterpret_cast<float(*)[64][4]>(&red[sub][0][0][0]));
    else
      slab_reduce_body<1, 4, 4>(q.slabs, q.S, q.n, q.N, q.ldc, q.alpha, q.bias, q.add, q.ldadd, q.out, q.alpha_head,
                                q.head_rows, tail, vb, t & 255, reinterpret_cast<float(*)[64][1]>(&red[sub][0][0][0]));
  }
}

// ---- column sums of a [M, N] f32 / bf16 matrix (bias gradients): stage 1 = per-slice partials, stage 2 = slab_reduce
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x, int f32, int64_t M, int N, int64_t ld,
                                                     int rows_per_wg, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
  float s = 0.f;
  if (c < N) {  // 8 independent loads in flight per lane
    if (f32) {
      const float* p = reinterpret_cast<const float*>(x) + c;
      int64_t r = r0 + w;
      for (; r + 28 < r1; r += 32) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[(r + 4 * i) * ld];
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
      for (; r < r1; r += 4) s += p[r * ld];
    } else {
      const unsigned short* p = reinterpret_cast<const unsigned short*>(x) + c;
      int64_t r = r0 + w;
      for (; r + 28 < r1; r += 32) {
        unsigned short v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[(r + 4 * i) * ld];
        s += ((bf2f(v[0]) + bf2f(v[1])) + (bf2f(v[2]) + bf2f(v[3]))) + ((bf2f(v[4]) + bf2f(v[5])) + (bf2f(v[6]) + bf2f(v[7])));
      }
      for (; r < r1; r += 4) s += bf2f(p[r * ld]);
    }
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < N) part[(int64_t)blockIdx.y * N + c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// narrow (128 x 64) tiles: N of 64 or less (no half-empty B tile), needs the descriptor loaders
bool narrow_tile(const csship_bgemm_params* p) {
  static const bool off = getenv("CSSHIP_BGEMM_WIDE") != nullptr;  // A/B timing
  static const int maxn = [] { const char* e = getenv("CSSHIP_BGEMM_NARROW_MAXN"); return e ? atoi(e) : 256; }();  // 256: the 1024 x 256 FeedForward weight gradient takes 32 narrow tiles x 12 slabs instead of 16 x 24 (half the slab traffic): training step 13.76 -> 13.56 ms
  return !off && p->N <= maxn && (((p->a_kmajor ? p->M : p->K) | (p->b_kmajor ? p->N : p->K)) & 7) == 0;
}

// the LDS-DMA loop (bgemm_kernel<..., GL>): weight-gradient products of bf16 operands in whole tiles
bool glds_path(const csship_bgemm_params* p, bool colsum) {
  static const bool off = getenv("CSSHIP_BGEMM_NOGLDS") != nullptr;  // A/B timing
  return !off && colsum && p->a_kmajor && p->b_kmajor && !p->a_f32 && !p->b_f32 && (p->M % 128) == 0 && (p->N % 128) == 0 &&
         p->batch1 == 1 && p->batch2 == 1;   // (any K: the rows of the last tile beyond K lie outside the descriptor - zeros)
}

int pick_splits(const csship_bgemm_params* p, bool colsum = false) {
  if (p->splits > 0) return p->splits;
  if (glds_path(p, colsum)) {
    // one workgroup per CU (128 KB of LDS): as many slabs as fill the 256 CUs once, in eights for the XCD-aware order
    // (measured, 12 800 rows: 256 x 1024 -> 16 slabs 21 us against 30 us with 24; tools/wgrad_sweep.py)
    const int tiles = (p->M / 128) * (p->N / 128), kt = (p->K + BK - 1) / BK;
    int s = 256 / tiles;
    if (s >= 8) s &= ~7;
    if (s > kt / 2) s = kt / 2;
    return s < 1 ? 1 : s;
  }
  const int tn = narrow_tile(p) ? 64 : 128;
  const int64_t tiles = (int64_t)((p->M + 127) / 128) * ((p->N + tn - 1) / tn) * p->batch1 * p->batch2;
  const int kt = (p->K + BK - 1) / BK;
  if (tiles >= 128 || kt < 8 || p->batch1 * p->batch2 != 1 || (p->epilogue & ~CSSHIP_BEPI_ADD) != 0) return 1;
  static const int target = [] { const char* e = getenv("CSSHIP_BGEMM_TARGET_WGS"); return e ? atoi(e) : 384; }();  // A/B timing
  int s = (int)((target + tiles - 1) / tiles);
  if (s > kt / 2) s = kt / 2;
  return s < 1 ? 1 : s;
}

}  // namespace

// slabs2 / n2 / out2 (optional): a second output of n2 elements with the same S and alpha, reduced by the same launch
// the host thread's queue of deferred reductions (csship_reduce_defer / csship_reduce_flush)
struct DeferQueue {
  bool on = false;
  std::vector<ReduceJob> jobs;
};
static thread_local DeferQueue t_defer;
static thread_local bool t_hold = false;   // csship_reduce_hold: queue EVERY slab reduction, whatever its size
static thread_local bool t_flat = false;   // csship_wgrad_flush: long outputs of at most four slabs take slab_sum_flat

static void queue_job(const float* slabs, int S, int64_t n, int N, int64_t ldc, float alpha, const float* bias,
                      const float* add, int64_t ldadd, float* out, float alpha_head, int head_rows, int kind) {
  ReduceJob j{};
  j.slabs = slabs; j.bias = bias; j.add = add; j.out = out; j.n = n; j.ldc = ldc; j.ldadd = ldadd; j.S = S; j.N = N;
  j.head_rows = head_rows; j.kind = kind; j.alpha = alpha; j.alpha_head = alpha_head;
  const int v = ((kind & 3) == 0 || (kind & 3) == 3) ? 4 : 1;
  j.nvb = (int)((n / v + 63) / 64);
  t_defer.jobs.push_back(j);
}

// ---- side stream for parameter-gradient reductions (see common.hpp)
static thread_local hipStream_t t_side = nullptr;
static thread_local hipEvent_t t_side_fork = nullptr, t_side_join = nullptr;
static thread_local bool t_side_dirty = false;

hipStream_t side_fork(hipStream_t main) {
  if (t_side == nullptr || t_side == main) return main;
  if (hipEventRecord(t_side_fork, main) != hipSuccess || hipStreamWaitEvent(t_side, t_side_fork, 0) != hipSuccess) return main;
  t_side_dirty = true;
  return t_side;
}

void side_join(hipStream_t main) {
  if (t_side == nullptr || !t_side_dirty || t_side == main) return;
  if (hipEventRecord(t_side_join, t_side) == hipSuccess && hipStreamWaitEvent(main, t_side_join, 0) == hipSuccess) t_side_dirty = false;
}

extern "C" int csship_reduce_side_stream(csship_stream_t side) {
  if (side != nullptr && t_side_fork == nullptr) {
    if (hipEventCreateWithFlags(&t_side_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&t_side_join, hipEventDisableTiming) != hipSuccess)
      return CSSHIP_ELAUNCH;
  }
  t_side = as_stream(side);
  if (side == nullptr) t_side_dirty = false;
  return CSSHIP_OK;
}

extern "C" int csship_reduce_side_join(csship_stream_t main) {
  side_join(as_stream(main));
  return CSSHIP_OK;
}

static int launch_slab_reduce2(const float* slabs, int S, int64_t n, int N, int64_t ldc, float alpha, const float* bias,
                               const float* add, int64_t ldadd, float* out, const float* slabs2, int64_t n2, float* out2,
                               hipStream_t s, float alpha_head = 0.f, int head_rows = 0, bool deferrable = false) {
  // long outputs: 16-byte loads, 4 waves per 256 elements; short outputs with many slabs (bias / LayerNorm partials): 16
  // waves share the slabs of 64 elements
  const bool wide = (N & 3) == 0 && (n & 3) == 0 && n >= 4096 && aligned16(slabs) && (n2 & 3) == 0 && aligned16(slabs2);
  // only SMALL slab sets are ever marked deferrable (LayerNorm / FeedForward bias partials).  A weight gradient's 16 MB of
  // slabs are summed right behind the product, out of the 256 MB MALL, and its scratch is rewritten by the next
  // product without reaching HBM; queued for the end of the block they came back from HBM: 112 us per block for 125 MB
  // against 14 launches x 5.7 us (measured).  The size test keeps that true whatever a caller marks.
  // (round 6 re-measured it with ONE set: the conv backward's 14.5 MB of tap-gradient partials queued for the block's batch launch
  // instead of their own 8 us launch - training step 7.65 -> 7.98 ms)
  static const int64_t defer_max = (int64_t)8 << 20;
  if (t_hold || (deferrable && t_defer.on && (int64_t)S * (n + n2) * 4 <= defer_max)) {  // the same kernel shape as the launch below would pick, later
    const bool flat = t_flat && wide && S <= 4 && (ldc & 3) == 0 && aligned16(out) && (!bias || aligned16(bias)) &&
                      (!add || (aligned16(add) && (ldadd & 3) == 0)) && (head_rows & 3) == 0 && aligned16(out2);
    const int kind = flat ? 3 : wide ? 0 : (S >= 32 ? 1 : 2);
    queue_job(slabs, S, n, N, ldc, alpha, bias, add, ldadd, out, alpha_head, head_rows, kind);
    if (n2 > 0) queue_job(slabs2, S, n2, (int)n2, n2, alpha, nullptr, nullptr, 0, out2, alpha_head, head_rows, kind | 4);
    return CSSHIP_OK;
  }
  if (deferrable) s = side_fork(s);   // a parameter gradient too large to queue: beside the caller's stream when it has a side stream
  if (wide) {
    const unsigned nb = (unsigned)((n / 4 + 63) / 64), nb2 = (unsigned)((n2 / 4 + 63) / 64);
    hipLaunchKernelGGL((slab_reduce_kernel<4, 4, 8>), dim3(nb + nb2), dim3(256), 0, s, slabs, S, n, N, ldc, alpha, bias, add,
                       ldadd, out, (int)nb, slabs2, n2, out2, alpha_head, head_rows);
  } else {
    const unsigned nb = (unsigned)((n + 63) / 64), nb2 = (unsigned)((n2 + 63) / 64);
    if (S >= 32)
      hipLaunchKernelGGL((slab_reduce_kernel<1, 16, 8>), dim3(nb + nb2), dim3(1024), 0, s, slabs, S, n, N, ldc, alpha, bias,
                         add, ldadd, out, (int)nb, slabs2, n2, out2, alpha_head, head_rows);
    else
      hipLaunchKernelGGL((slab_reduce_kernel<1, 4, 4>), dim3(nb + nb2), dim3(256), 0, s, slabs, S, n, N, ldc, alpha, bias, add,
                         ldadd, out, (int)nb, slabs2, n2, out2, alpha_head, head_rows);
  }
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

int launch_slab_reduce(const float* slabs, int S, int64_t n, int N, int64_t ldc, float alpha, const float* bias,
                       const float* add, int64_t ldadd, float* out, hipStream_t s, bool deferrable) {
  return launch_slab_reduce2(slabs, S, n, N, ldc, alpha, bias, add, ldadd, out, nullptr, 0, nullptr, s, 0.f, 0, deferrable);
}

extern "C" int csship_reduce_defer(int32_t on) {
  const int was = t_defer.on ? 1 : 0;
  t_defer.on = on != 0;
  if (!on) t_defer.jobs.clear();  // leaving the mode without a flush (error paths) drops what was queued
  return was;
}

extern "C" int csship_reduce_hold(int32_t on) {
  const int was = t_hold ? 1 : 0;
  t_hold = on != 0;
  return was;
}

extern "C" int csship_reduce_pending(void) { return (int)t_defer.jobs.size(); }

// ---- grouped weight gradients (bgemm_wgrad_group_kernel): csship_bgemm_wgrad calls made while the hold is on are
// queued when they fit the LDS-DMA loop (bf16 k-major operands in whole tiles, at most 96 tiles) and share the queue's
// reduction length; anything else runs at once, as without the hold.
struct WgradQ {
  const void* A; const void* B; float* dW; float* db; const void* aux;
  int M, N, K; int64_t lda, ldb, ldc, ldaux;
  float alpha, alpha_head; int head_rows, epi;
};
static thread_local bool t_wq_on = false;
static thread_local std::vector<WgradQ> t_wq;

extern "C" int csship_wgrad_hold(int32_t on) {
  const int was = t_wq_on ? 1 : 0;
  t_wq_on = on != 0;
  if (!on) t_wq.clear();  // leaving the mode without a flush (error paths) drops what was queued
  return was;
}

extern "C" int csship_wgrad_pending(void) { return (int)t_wq.size(); }

// (the table holds slots x 8 entries, slots = the fullest XCD's workgroups: at most the average plus one unit - the deal
// is largest first, each unit to the emptiest XCD - so `tiles x splits + 8 x largest unit` entries always suffice)
static int wgrad_group_splits(size_t first, size_t count) {
  int tiles = 0, maxu = 0;
  for (size_t i = first; i < first + count; ++i) {
    const int t = (t_wq[i].M / 128) * (t_wq[i].N / 128);
    tiles += t;
    maxu = std::max(maxu, t);
  }
  const int kt = (t_wq[first].K + BK - 1) / BK;
  int s = tiles > 0 ? 256 / tiles : 2;
  s = std::max(2, std::min(s, 16));
  while (s > 2 && (tiles * s + 8 * maxu > WG_MAXWG || kt / s < 8)) --s;
  return std::min(s, std::max(1, kt));
}

// the queue is flushed in chunks of at most WG_JOBS products whose tiles x 2 (the fewest splits) fit the table
static size_t wgrad_chunk(size_t first) {
  size_t n = 0;
  int tiles = 0, maxu = 0;
  while (first + n < t_wq.size() && n < (size_t)WG_JOBS) {
    const int t = (t_wq[first + n].M / 128) * (t_wq[first + n].N / 128);
    if (n > 0 && (tiles + t) * 2 + 8 * std::max(maxu, t) > WG_MAXWG) break;
    tiles += t;
    maxu = std::max(maxu, t);
    ++n;
  }
  return n;
}

extern "C" size_t csship_wgrad_group_workspace(void) {
  size_t need = 0;
  for (size_t at = 0; at < t_wq.size();) {
    const size_t n = wgrad_chunk(at);
    const int s = wgrad_group_splits(at, n);
    size_t b = 0;
    for (size_t i = at; i < at + n; ++i) b += (size_t)s * ((size_t)t_wq[i].M * t_wq[i].N + t_wq[i].M) * sizeof(float);
    need += b;   // chunks keep their own slabs: one reduction launch at the end reads them all
    at += n;
  }
  return need;
}

extern "C" int csship_wgrad_flush(void* workspace, size_t workspace_bytes, csship_stream_t stream) {
  side_join(as_stream(stream));   // (the previous flush's reduction may still read the slabs this one's products overwrite)
  if (t_wq.empty()) return CSSHIP_OK;
  if (!workspace || workspace_bytes < csship_wgrad_group_workspace()) {
    t_wq.clear();
    return CSSHIP_EWORKSPACE;
  }
  hipStream_t s = as_stream(stream);
  float* ws = reinterpret_cast<float*>(workspace);
  const bool held = t_hold;
  int rc = CSSHIP_OK;
  for (size_t at = 0; at < t_wq.size() && rc == CSSHIP_OK;) {
    const size_t n = wgrad_chunk(at);
    const int S = wgrad_group_splits(at, n);
    WgGroup G{};
    G.K = t_wq[at].K;
    G.splits = S;
    const int kt = (G.K + BK - 1) / BK;
    G.k_per_split = ((kt + S - 1) / S) * BK;
    // (product, split) units, largest first, each to the XCD with the fewest workgroups so far
    struct Unit { int job, split, tiles; };
    std::vector<Unit> units;
    for (size_t i = 0; i < n; ++i) {
      const WgradQ& q = t_wq[at + i];
      WgJob& J = G.job[i];
      J.A = q.A; J.B = q.B; J.M = q.M; J.N = q.N; J.lda = (int)q.lda; J.ldb = (int)q.ldb;
      J.C = ws;
      ws += (size_t)S * q.M * q.N;
      J.cs = ws;
      ws += (size_t)S * q.M;
      for (int sp = 0; sp < S; ++sp) units.push_back(Unit{(int)i, sp, (q.M / 128) * (q.N / 128)});
    }
    std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.tiles > b.tiles; });
    std::vector<unsigned short> xcd[8];
    for (const Unit& u : units) {
      int best = 0;
      for (int x = 1; x < 8; ++x)
        if (xcd[x].size() < xcd[best].size()) best = x;
      for (int t = 0; t < u.tiles; ++t) xcd[best].push_back((unsigned short)(u.job << 12 | u.split << 8 | t));
    }
    size_t slots = 0;
    for (int x = 0; x < 8; ++x) slots = std::max(slots, xcd[x].size());
    if (slots * 8 > (size_t)WG_MAXWG) {
      rc = CSSHIP_EBADSHAPE;
      break;
    }
    for (size_t sl = 0; sl < slots; ++sl)
      for (int x = 0; x < 8; ++x) G.map[sl * 8 + x] = sl < xcd[x].size() ? xcd[x][sl] : WG_IDLE;
    hipLaunchKernelGGL(bgemm_wgrad_group_kernel, dim3((unsigned)(slots * 8)), dim3(512), 0, s, G);
    if (hipGetLastError() != hipSuccess) {
      rc = CSSHIP_ELAUNCH;
      break;
    }
    // the slab sums: queued behind whatever the caller has deferred, one launch for all of them below
    t_hold = true;
    t_flat = true;
    for (size_t i = 0; i < n && rc == CSSHIP_OK; ++i) {
      const WgradQ& q = t_wq[at + i];
      rc = launch_slab_reduce2(G.job[i].C, S, (int64_t)q.M * q.N, q.N, q.ldc, q.alpha, nullptr,
                               (q.epi & CSSHIP_BEPI_ADD) ? reinterpret_cast<const float*>(q.aux) : nullptr, q.ldaux, q.dW,
                               G.job[i].cs, (int64_t)q.M, q.db, s, q.alpha_head, q.head_rows);
    }
    t_hold = held;
    t_flat = false;
    at += n;
  }
  t_wq.clear();
  if (rc != CSSHIP_OK) {
    t_defer.jobs.clear();
    return rc;
  }
  return csship_reduce_flush(stream);
}

extern "C" int csship_reduce_flush(csship_stream_t stream) {
  std::vector<ReduceJob>& q = t_defer.jobs;
  hipStream_t s = q.empty() ? as_stream(stream) : side_fork(as_stream(stream));   // parameter gradients only: beside the caller's stream
  for (size_t at = 0; at < q.size(); at += RB_MAX) {
    ReduceBatch b{};
    b.njobs = (int)std::min<size_t>(RB_MAX, q.size() - at);
    int wgs = 0;
    for (int j = 0; j < b.njobs; ++j) {
      b.job[j] = q[at + j];
      b.first[j] = wgs;
      const int k3 = b.job[j].kind & 3;
      wgs += k3 == 1 ? b.job[j].nvb : k3 == 3 ? (b.job[j].nvb + 15) / 16 : (b.job[j].nvb + 3) / 4;
    }
    b.first[b.njobs] = wgs;
    if (wgs > 0) {
      hipLaunchKernelGGL(slab_reduce_batch_kernel, dim3((unsigned)wgs), dim3(1024), 0, s, b);
      if (hipGetLastError() != hipSuccess) {
        q.clear();
        return CSSHIP_ELAUNCH;
      }
    }
  }
  q.clear();
  return CSSHIP_OK;
}

#ifdef BG_STAMPS
extern "C" int csship_debug_bgemm_stamps(unsigned* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_bg_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t csship_bgemm_workspace(const csship_bgemm_params* p) {
  if (!p) return 0;
  const int s = pick_splits(p);
  return s > 1 ? (size_t)s * p->M * p->N * sizeof(float) : 0;
}

static int bgemm_impl(const void* A, const void* B, void* C, const float* bias, const void* aux,
                      const csship_bgemm_params* p, void* workspace, size_t workspace_bytes, csship_stream_t stream,
                      float* colsum_out, float alpha_head = 0.f, int head_rows = 0);

extern "C" int csship_bgemm(const void* A, const void* B, void* C, const float* bias, const void* aux,
                            const csship_bgemm_params* p, void* workspace, size_t workspace_bytes,
                            csship_stream_t stream) {
  return bgemm_impl(A, B, C, bias, aux, p, workspace, workspace_bytes, stream, nullptr);
}

static bool wgrad_shape_ok(const csship_bgemm_params* p) {
  return p->a_kmajor && p->b_kmajor && p->batch1 == 1 && p->batch2 == 1 && (p->epilogue & ~CSSHIP_BEPI_ADD) == 0 &&
         ((p->M | p->N) & 7) == 0;
}

extern "C" size_t csship_bgemm_wgrad_workspace(const csship_bgemm_params* p) {
  if (!p) return 0;
  const int s = pick_splits(p, true);
  return s > 1 ? (size_t)s * ((size_t)p->M * p->N + p->M) * sizeof(float) : 0;
}

extern "C" int csship_bgemm_wgrad(const void* dY, const void* X, float* dW, float* db, const void* aux,
                                  const csship_bgemm_params* p, float alpha_head, int32_t head_rows, void* workspace,
                                  size_t workspace_bytes, csship_stream_t stream) {
  if (!db || !p) return CSSHIP_ENULL;
  if (!wgrad_shape_ok(p) || head_rows < 0 || head_rows > p->M) return CSSHIP_EUNSUPPORTED;
  return bgemm_impl(dY, X, dW, nullptr, aux, p, workspace, workspace_bytes, stream, db, alpha_head, head_rows);
}

static int bgemm_impl(const void* A, const void* B, void* C, const float* bias, const void* aux,
                      const csship_bgemm_params* p, void* workspace, size_t workspace_bytes, csship_stream_t stream,
                      float* colsum_out, float alpha_head, int head_rows) {
  if (!A || !B || !C || !p) return CSSHIP_ENULL;
  if (p->M <= 0 || p->N <= 0 || p->K <= 0 || p->batch1 <= 0 || p->batch2 <= 0) return CSSHIP_EBADSHAPE;
  if ((p->epilogue & (CSSHIP_BEPI_MASK_POS | CSSHIP_BEPI_ADD)) && !aux) return CSSHIP_ENULL;
  if ((p->epilogue & CSSHIP_BEPI_MASK_POS) && (p->epilogue & CSSHIP_BEPI_ADD)) return CSSHIP_EUNSUPPORTED;
  // vector loads: rows of 8 bf16 / 4+4 f32 must be 16-byte aligned
  const int64_t amod = p->a_f32 ? 4 : 8, bmod = p->b_f32 ? 4 : 8;
  if (!aligned16(A) || !aligned16(B) || (p->lda % amod) || (p->ldb % bmod) || (p->sA1 % amod) || (p->sA2 % amod) ||
      (p->sB1 % bmod) || (p->sB2 % bmod))
    return CSSHIP_EALIGN;
  if (p->lda < (p->a_kmajor ? p->M : p->K) || p->ldb < (p->b_kmajor ? p->N : p->K) || p->ldc < p->N)
    return CSSHIP_EBADSHAPE;
  if (t_wq_on && colsum_out && !bias && p->splits == 0 && glds_path(p, true) && (p->M / 128) * (p->N / 128) <= 96 &&
      p->K >= 16 * BK && (p->ldc & 3) == 0 && aligned16(C) && (t_wq.empty() || t_wq[0].K == p->K)) {
    t_wq.push_back(WgradQ{A, B, reinterpret_cast<float*>(C), colsum_out, aux, p->M, p->N, p->K, p->lda, p->ldb, p->ldc, p->ldaux,
                          p->alpha, alpha_head, head_rows, p->epilogue});
    return CSSHIP_OK;
  }
  const int splits = pick_splits(p, colsum_out != nullptr);
  const int64_t nz = (int64_t)p->batch1 * p->batch2 * splits;
  if (nz > 65535) return CSSHIP_EBADSHAPE;
  if (splits > 1) {
    if (p->batch1 * p->batch2 != 1 || (p->epilogue & ~CSSHIP_BEPI_ADD) != 0) return CSSHIP_EUNSUPPORTED;
    if (!workspace || workspace_bytes < (size_t)splits * ((size_t)p->M * p->N + (colsum_out ? p->M : 0)) * sizeof(float))
      return CSSHIP_EWORKSPACE;
  }
  BgemmArgs a{};
  a.A = A; a.B = B; a.bias = bias; a.aux = aux;
  a.C = splits > 1 ? workspace : C;
  a.M = p->M; a.N = p->N; a.K = p->K;
  a.lda = p->lda; a.ldb = p->ldb; a.ldc = p->ldc; a.ldaux = p->ldaux;
  a.nb2 = p->batch2; a.splits = splits;
  const int kt = (p->K + BK - 1) / BK;
  a.k_per_split = ((kt + splits - 1) / splits) * BK;
  a.sA1 = p->sA1; a.sA2 = p->sA2; a.sB1 = p->sB1; a.sB2 = p->sB2; a.sC1 = p->sC1; a.sC2 = p->sC2; a.sX1 = p->sX1; a.sX2 = p->sX2;
  a.slab = (int64_t)p->M * p->N;
  a.a_f32 = p->a_f32; a.b_f32 = p->b_f32; a.epi = p->epilogue; a.alpha = p->alpha;
  a.cs_out = colsum_out;
  a.band = 0;
  a.band_stride = 0;
  if (p->band > 0 && !p->a_kmajor && splits == 1) a.band = p->band;
  // kmajor form: one K-slab per period (the caller gave splits = K / period), whole 64-deep tiles
  if (p->band > 0 && p->a_kmajor && p->band_stride > 0 && p->splits == splits && splits > 1 && p->K % (splits * BK) == 0) {
    a.band = p->band;
    a.band_stride = p->band_stride;
  }
  a.alpha_head = alpha_head; a.head_rows = head_rows;
  a.cs_slab = splits > 1 ? reinterpret_cast<float*>(workspace) + (size_t)splits * p->M * p->N : nullptr;
  const bool narrow = narrow_tile(p) && !colsum_out;
  dim3 grid(narrow ? (p->N + 63) / 64 : (p->N + 127) / 128, (p->M + 127) / 128, (unsigned)nz);
  hipStream_t s = as_stream(stream);
  // descriptor loads need whole 8-element groups and 32-bit byte offsets inside one batch entry
  const int64_t a_elems = p->a_kmajor ? (int64_t)(p->K - 1) * p->lda + p->M : (int64_t)(p->M - 1) * p->lda + p->K;
  const int64_t b_elems = p->b_kmajor ? (int64_t)(p->K - 1) * p->ldb + p->N : (int64_t)(p->N - 1) * p->ldb + p->K;
  const bool fast = (((p->a_kmajor ? p->M : p->K) | (p->b_kmajor ? p->N : p->K)) & 7) == 0 &&
                    a_elems * (p->a_f32 ? 4 : 2) < 0xFFFFFF00ll && b_elems * (p->b_f32 ? 4 : 2) < 0xFFFFFF00ll;
#define BG_LAUNCH(AK, BK_)                                                                                      \
  do {                                                                                                          \
    if (fast && narrow) {                                                                                       \
      if (p->a_f32 && p->b_f32) hipLaunchKernelGGL((bgemm_kernel<AK, BK_, true, true, true, false, 64>), grid, dim3(256), 0, s, a);   \
      else if (p->a_f32) hipLaunchKernelGGL((bgemm_kernel<AK, BK_, true, true, false, false, 64>), grid, dim3(256), 0, s, a);         \
      else if (p->b_f32) hipLaunchKernelGGL((bgemm_kernel<AK, BK_, true, false, true, false, 64>), grid, dim3(256), 0, s, a);         \
      else hipLaunchKernelGGL((bgemm_kernel<AK, BK_, true, false, false, false, 64>), grid, dim3(256), 0, s, a);                      \
    } else                                                                                                      \
    if (!fast) hipLaunchKernelGGL((bgemm_kernel<AK, BK_, false, false, false>), grid, dim3(256), 0, s, a);      \
    else if (p->a_f32 && p->b_f32) hipLaunchKernelGGL((bgemm_kernel<AK, BK_, true, true, true>), grid, dim3(256), 0, s, a);   \
    else if (p->a_f32) hipLaunchKernelGGL((bgemm_kernel<AK, BK_, true, true, false>), grid, dim3(256), 0, s, a);              \
    else if (p->b_f32) hipLaunchKernelGGL((bgemm_kernel<AK, BK_, true, false, true>), grid, dim3(256), 0, s, a);              \
    else hipLaunchKernelGGL((bgemm_kernel<AK, BK_, true, false, false>), grid, dim3(256), 0, s, a);                           \
  } while (0)
  if (colsum_out) {
    if (!fast) return CSSHIP_EUNSUPPORTED;
    const bool gl = glds_path(p, true);
    if (p->a_f32 && p->b_f32) hipLaunchKernelGGL((bgemm_kernel<true, true, true, true, true, true>), grid, dim3(256), 0, s, a);
    else if (p->a_f32) hipLaunchKernelGGL((bgemm_kernel<true, true, true, true, false, true>), grid, dim3(256), 0, s, a);
    else if (p->b_f32) hipLaunchKernelGGL((bgemm_kernel<true, true, true, false, true, true>), grid, dim3(256), 0, s, a);
    else if (gl) hipLaunchKernelGGL((bgemm_kernel<true, true, true, false, false, true, 128, true>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((bgemm_kernel<true, true, true, false, false, true>), grid, dim3(256), 0, s, a);
  } else if (p->a_kmajor && p->b_kmajor) BG_LAUNCH(true, true);
  else if (p->a_kmajor) BG_LAUNCH(true, false);
  else if (p->b_kmajor) BG_LAUNCH(false, true);
  else BG_LAUNCH(false, false);
#undef BG_LAUNCH
  CSSHIP_CHECK_LAUNCH();
  if (splits > 1) {
    const int64_t n = (int64_t)p->M * p->N;
    return launch_slab_reduce2(reinterpret_cast<const float*>(workspace), splits, n, p->N, p->ldc, p->alpha, bias,
                               (p->epilogue & CSSHIP_BEPI_ADD) ? reinterpret_cast<const float*>(aux) : nullptr,
                               p->ldaux, reinterpret_cast<float*>(C), colsum_out ? a.cs_slab : nullptr,
                               colsum_out ? (int64_t)p->M : 0, colsum_out, s, alpha_head, head_rows,
                               // a product accumulated in place (C += ...) is a gradient accumulator nobody reads before
                               // the block's flush (the relative-position window gradient: 6.5 MB of slabs per block)
                               !colsum_out && (p->epilogue & CSSHIP_BEPI_ADD) && aux == C);
  }
  return CSSHIP_OK;
}

extern "C" size_t csship_colsum_workspace(int64_t M, int32_t N) {
  const int slices = (int)((M + 127) / 128);
  return (size_t)slices * N * sizeof(float);
}

extern "C" int csship_colsum(const void* x, int32_t x_is_f32, int64_t M, int32_t N, int64_t ld, float alpha, float* out,
                             void* workspace, size_t workspace_bytes, csship_stream_t stream) {
  if (!x || !out || !workspace) return CSSHIP_ENULL;
  if (M <= 0 || N <= 0 || ld < N) return CSSHIP_EBADSHAPE;
  const int slices = (int)((M + 127) / 128);
  if (slices > 65535) return CSSHIP_EBADSHAPE;
  if (workspace_bytes < (size_t)slices * N * sizeof(float)) return CSSHIP_EWORKSPACE;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, slices), dim3(256), 0, s, x, x_is_f32, M, N, ld, 128,
                     reinterpret_cast<float*>(workspace));
  CSSHIP_CHECK_LAUNCH();
  return launch_slab_reduce(reinterpret_cast<const float*>(workspace), slices, (int64_t)N, N, (int64_t)N, alpha, nullptr,
                            nullptr, 0, out, s);
}
