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
template <bool A_KM, bool B_KM, bool FAST, bool A_F32, bool B_F32, bool COLSUM = false, int TN = 128>
__global__ __launch_bounds__(256) void bgemm_kernel(BgemmArgs g) {
  static_assert(TN == 128 || (TN == 64 && FAST), "narrow tiles use the descriptor loaders");
  constexpr int NJ = TN / 32, WN = TN / 2;  // 16-column fragments per wave, columns per wave
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef BG_STAMPS
  __shared__ unsigned stamps[96];
  if (tid < 96) stamps[tid] = 0;
  __syncthreads();
  STAMP(0);
#endif
  const int wm = wave >> 1, wn = wave & 1;
  const int lg = lane >> 4, ln = lane & 15;
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
  const int m0 = by * 128, n0 = bx * TN;
  const int z = bz / g.splits, split = bz % g.splits;
  const int z1 = z / g.nb2, z2 = z % g.nb2;
  const int kbeg = split * g.k_per_split, kend = min(g.K, kbeg + g.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;
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
      if (vec) {
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
          const float4 bv = *reinterpret_cast<const float4*>(g.bias + col);
          b[0] = bv.x; b[1] = bv.y; b[2] = bv.z; b[3] = bv.w;
        }
        const float al = (COLSUM && row < g.head_rows) ? g.alpha_head : g.alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (g.epi & CSSHIP_BEPI_SCALE_BIAS) ? al * (v[r] + b[r]) : al * v[r] + b[r];
        if (g.epi & CSSHIP_BEPI_MASK_POS) {
          const uint2 m = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(g.aux) + ao);
          if (!(__uint_as_float(m.x << 16) > 0.f)) v[0] = 0.f;
          if (!(__uint_as_float(m.x & 0xffff0000u) > 0.f)) v[1] = 0.f;
          if (!(__uint_as_float(m.y << 16) > 0.f)) v[2] = 0.f;
          if (!(__uint_as_float(m.y & 0xffff0000u) > 0.f)) v[3] = 0.f;
        }
        if (g.epi & CSSHIP_BEPI_ADD) {
          const float4 a4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.aux) + ao);
          v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
        }
        if (g.epi & CSSHIP_BEPI_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (obf)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(g.C) + co) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        else
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + co) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
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
  __syncthreads();
  if (g_bg_stamps && tid < 96) {
    const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    g_bg_stamps[lin * 96 + tid] = stamps[tid];
  }
#endif
}

// out[r, c] = alpha * sum_s slab[s][r, c] (+ bias[c]) (+ add[r, c]); fixed summation order.
// Workgroup = 64 lanes x V consecutive elements x NW waves; wave w adds slabs w, w+NW, ... with eight loads in flight,
// then one LDS reduction - short outputs with many slabs (LayerNorm / bias partials) do not serialise on one thread,
// long ones (V = 4: N and the slab size are multiples of 4) move 16 bytes per load instruction.
template <int V, int NW, int U>
__global__ __launch_bounds__(64 * NW) void slab_reduce_kernel(const float* __restrict__ slabs, int S, int64_t n, int N,
                                                              int64_t ldc, float alpha, const float* __restrict__ bias,
                                                              const float* add, int64_t ldadd, float* out, int nb_main,
                                                              const float* __restrict__ slabs2, int64_t n2,
                                                              float* __restrict__ out2, float alpha_head, int head_rows) {
  typedef float vec __attribute__((ext_vector_type(V)));
  __shared__ vec red[NW][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int bid = blockIdx.x;
  bool tail = false;
  if (bid >= nb_main) {  // second, short output reduced by the same launch (the bias gradient next to its weight gradient)
    bid -= nb_main;
    tail = true;
    slabs = slabs2; n = n2; N = (int)n2; ldc = n2; bias = nullptr; add = nullptr; out = out2;
  }
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

int pick_splits(const csship_bgemm_params* p) {
  if (p->splits > 0) return p->splits;
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
static int launch_slab_reduce2(const float* slabs, int S, int64_t n, int N, int64_t ldc, float alpha, const float* bias,
                               const float* add, int64_t ldadd, float* out, const float* slabs2, int64_t n2, float* out2,
                               hipStream_t s, float alpha_head = 0.f, int head_rows = 0) {
  // long outputs: 16-byte loads, 4 waves per 256 elements; short outputs with many slabs (bias / LayerNorm partials): 16
  // waves share the slabs of 64 elements
  if ((N & 3) == 0 && (n & 3) == 0 && n >= 4096 && aligned16(slabs) && (n2 & 3) == 0 && aligned16(slabs2)) {
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
                       const float* add, int64_t ldadd, float* out, hipStream_t s) {
  return launch_slab_reduce2(slabs, S, n, N, ldc, alpha, bias, add, ldadd, out, nullptr, 0, nullptr, s);
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
  const int s = pick_splits(p);
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
  const int splits = pick_splits(p);
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
    if (p->a_f32 && p->b_f32) hipLaunchKernelGGL((bgemm_kernel<true, true, true, true, true, true>), grid, dim3(256), 0, s, a);
    else if (p->a_f32) hipLaunchKernelGGL((bgemm_kernel<true, true, true, true, false, true>), grid, dim3(256), 0, s, a);
    else if (p->b_f32) hipLaunchKernelGGL((bgemm_kernel<true, true, true, false, true, true>), grid, dim3(256), 0, s, a);
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
                               colsum_out ? (int64_t)p->M : 0, colsum_out, s, alpha_head, head_rows);
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
