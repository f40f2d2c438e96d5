// bf16 MFMA GEMM with fused prologue / epilogue:  C = epi(A[M,K] @ W[N,K]^T + bias).
//
// Replaces every torch.nn.Linear on the hot path (css/models/conformer.py:98,124,
// 205-209,315-318).  CDNA4 design: 256-thread workgroup = 4 waves, each wave owns a
// 64x64 accumulator block as 4x4 v_mfma_f32_16x16x32_bf16 tiles; A and W tiles of 64
// K-columns are staged global -> registers -> LDS (XOR-swizzled 128-byte rows, conflict
// free ds_read_b128) with the next tile's global loads in flight during the MFMAs.
// Both operands are K-contiguous ("NT" GEMM), so the A and B fragments use the same
// LDS image and read pattern.
#include "common.hpp"

namespace {

struct GemmArgs {
  const void* A;
  const unsigned short* W;
  const float* bias;
  const float* res;
  void* C;
  const float* mean;
  const float* rinv;
  const float* ln_g;
  const float* ln_b;
  int M, N, K, lda, ldw, ldc, ldr, rows_per_batch;
  float alpha;
};

constexpr int BK = 64;

template <int WAVES_M, int WAVES_N, int AMODE, bool KFULL>
DEV void gemm_load_tiles(const GemmArgs& g, int kt, int m0, int n0, int srow, int sc,
                         uint4 (&ra)[WAVES_M * 2], uint4 (&rb)[WAVES_N * 2],
                         uint4 (&ra2)[WAVES_M * 2]) {
  constexpr int A_CH = WAVES_M * 2, B_CH = WAVES_N * 2;
  constexpr bool RAW_F32 = KFULL && AMODE == CSSHIP_A_F32;
    const int k = kt * BK + sc * 8;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int row = min(m0 + srow + 32 * i, g.M - 1);
      if constexpr (KFULL && AMODE == CSSHIP_A_BF16) {
        ra[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(g.A) + (size_t)row * g.lda + k);
      } else if constexpr (RAW_F32) {
        const float* p = reinterpret_cast<const float*>(g.A) + (size_t)row * g.lda + k;
        ra[i] = *reinterpret_cast<const uint4*>(p);
        ra2[i] = *reinterpret_cast<const uint4*>(p + 4);
      } else if constexpr (AMODE == CSSHIP_A_BF16) {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(g.A) + (size_t)row * g.lda + k;
        ra[i] = (k < g.K) ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
      } else if constexpr (AMODE == CSSHIP_A_F32) {
        const float* p = reinterpret_cast<const float*>(g.A) + (size_t)row * g.lda + k;
        if (k < g.K) {
          const float4 x0 = *reinterpret_cast<const float4*>(p);
          const float4 x1 = *reinterpret_cast<const float4*>(p + 4);
          ra[i] = make_uint4(pack_bf2(x0.x, x0.y), pack_bf2(x0.z, x0.w), pack_bf2(x1.x, x1.y), pack_bf2(x1.z, x1.w));
        } else {
          ra[i] = make_uint4(0, 0, 0, 0);
        }
      } else {  // fp32 rows of arbitrary length, (x - mean[b,k]) * rinv[b,k] applied on load
        const float* p = reinterpret_cast<const float*>(g.A) + (size_t)row * g.lda;
        const size_t sb = (size_t)(row / g.rows_per_batch) * g.K;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int kk = k + e;
          v[e] = (kk < g.K) ? (p[kk] - g.mean[sb + kk]) * g.rinv[sb + kk] : 0.f;
        }
        ra[i] = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int row = min(n0 + srow + 32 * i, g.N - 1);
      rb[i] = *reinterpret_cast<const uint4*>(g.W + (size_t)row * g.ldw + k);  // ldw padded to nk*64
    }
}

template <int WAVES_M, int WAVES_N, int AMODE, bool KFULL>
DEV void gemm_store_tiles(unsigned char* smem, int stage, int srow, int sc, uint4 (&ra)[WAVES_M * 2], const uint4 (&rb)[WAVES_N * 2],
                          const uint4 (&ra2)[WAVES_M * 2]) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64;
  constexpr int A_CH = WAVES_M * 2, B_CH = WAVES_N * 2;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr bool RAW_F32 = KFULL && AMODE == CSSHIP_A_F32;
    unsigned char* as = smem + stage * STAGE;
    unsigned char* bs = as + BM * 128;
    if constexpr (RAW_F32) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i)
        ra[i] = make_uint4(pack_bf2(__uint_as_float(ra[i].x), __uint_as_float(ra[i].y)),
                           pack_bf2(__uint_as_float(ra[i].z), __uint_as_float(ra[i].w)),
                           pack_bf2(__uint_as_float(ra2[i].x), __uint_as_float(ra2[i].y)),
                           pack_bf2(__uint_as_float(ra2[i].z), __uint_as_float(ra2[i].w)));
    }
#pragma unroll
    for (int i = 0; i < A_CH; ++i) *reinterpret_cast<uint4*>(as + tile_off(srow + 32 * i, sc)) = ra[i];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) *reinterpret_cast<uint4*>(bs + tile_off(srow + 32 * i, sc)) = rb[i];
}

// KFULL: K is a multiple of the 64-deep tile (every large Linear on the path) - the tile loads are straight-line code and
// f32 activations stay raw in registers until the tile is written to LDS, after the MFMAs.  hipcc waits for a conditional
// load inside its branch, which made the generic loader's eight loads go out one memory round trip after the other.
template <int WAVES_M, int WAVES_N, int AMODE, int EPI, bool KFULL>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64;
  constexpr int A_CH = BM / 32, B_CH = BN / 32;  // 16-byte chunks per thread per tile
  constexpr int STAGE = (BM + BN) * 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lg = lane >> 4, ln = lane & 15;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int nk = (g.K + BK - 1) / BK;
  const int srow = tid >> 3, sc = tid & 7;  // staging row (0..31) and chunk (0..7)

  uint4 ra[A_CH], rb[B_CH], ra2[A_CH];  // ra2: second half of raw f32 groups (KFULL f32 activations only)

  // (tile loads / LDS stores are free functions on the register arrays: arrays captured by a lambda are demoted to
  // scratch memory by hipcc, which turned the prefetch into load -> wait -> scratch store)
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  gemm_load_tiles<WAVES_M, WAVES_N, AMODE, KFULL>(g, 0, m0, n0, srow, sc, ra, rb, ra2);
  gemm_store_tiles<WAVES_M, WAVES_N, AMODE, KFULL>(smem, 0, srow, sc, ra, rb, ra2);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gemm_load_tiles<WAVES_M, WAVES_N, AMODE, KFULL>(g, kt + 1, m0, n0, srow, sc, ra, rb, ra2);
    if constexpr (KFULL) __builtin_amdgcn_sched_barrier(0);  // loads stay up here, conversion + LDS stores below the MFMAs
    const unsigned char* as = smem + (kt & 1) * STAGE;
    const unsigned char* bs = as + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(as + tile_off(wm * 64 + i * 16 + ln, ks * 4 + lg));
        bf[i] = *reinterpret_cast<const bf16x8*>(bs + tile_off(wn * 64 + i * 16 + ln, ks * 4 + lg));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if constexpr (KFULL) __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) gemm_store_tiles<WAVES_M, WAVES_N, AMODE, KFULL>(smem, (kt + 1) & 1, srow, sc, ra, rb, ra2);
    __syncthreads();
  }

  // ---- epilogue.  acc[i][j][r]: row = m0 + wm*64 + i*16 + lg*4 + r, col = n0 + wn*64 + j*16 + ln
  float bias_v[4];
  int col_v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    col_v[j] = n0 + wn * 64 + j * 16 + ln;
    bias_v[j] = (g.bias != nullptr && col_v[j] < g.N) ? g.bias[col_v[j]] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm * 64 + i * 16 + lg * 4 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[i][j][r] + bias_v[j];
        if constexpr ((EPI & CSSHIP_EPI_RESIDUAL) != 0) {
          if (row < g.M && col_v[j] < g.N) v = g.res[(size_t)row * g.ldr + col_v[j]] + g.alpha * v;
        }
        acc[i][j][r] = v;
      }
    }

  if constexpr ((EPI & CSSHIP_EPI_LAYERNORM) != 0) {
    // Full output rows live in this workgroup (WAVES_M == 1, WAVES_N == 4, N == 256).
    static_assert((EPI & CSSHIP_EPI_LAYERNORM) == 0 || (WAVES_M == 1 && WAVES_N == 4), "LN epilogue needs BN == 256");
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][64 rows]; K loop is finished
    float mean[4][4] = {}, rstd[4][4] = {};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = pass == 0 ? acc[i][j][r] : acc[i][j][r] - mean[i][r];
            s += pass == 0 ? d : d * d;
          }
          s = sum8(s);
          s += dpp_mov<DPP_MIRROR>(s);
          if (ln == 0) red[wn * 64 + i * 16 + lg * 4 + r] = s;
        }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = i * 16 + lg * 4 + r;
          const float t = (red[rr] + red[64 + rr] + red[128 + rr] + red[192 + rr]) * (1.f / 256.f);
          if (pass == 0) mean[i][r] = t; else rstd[i][r] = rsqrtf(t + 1e-5f);
        }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = g.ln_g[col_v[j]], bb = g.ln_b[col_v[j]];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = (acc[i][j][r] - mean[i][r]) * rstd[i][r] * gg + bb;
    }
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm * 64 + i * 16 + lg * 4 + r;
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (col_v[j] >= g.N) continue;
        float v = acc[i][j][r];
        if constexpr ((EPI & CSSHIP_EPI_RELU) != 0) v = fmaxf(v, 0.f);
        if constexpr ((EPI & CSSHIP_EPI_OUT_BF16) != 0)
          reinterpret_cast<unsigned short*>(g.C)[(size_t)row * g.ldc + col_v[j]] = f2bf(v);
        else
          reinterpret_cast<float*>(g.C)[(size_t)row * g.ldc + col_v[j]] = v;
      }
    }
}

template <int WAVES_M, int WAVES_N, int AMODE, int EPI>
int launch(const GemmArgs& a, hipStream_t s) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64;
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM);
  // (only for f32 activations: with bf16 activations hipcc parks the staged registers in scratch memory for the
  // straight-line form and the kernel gets slower, 129 vs 112 us at 12800 x 2048 x 512)
  if (AMODE == CSSHIP_A_F32 && (a.K % BK) == 0)
    hipLaunchKernelGGL((gemm_kernel<WAVES_M, WAVES_N, AMODE, EPI, AMODE == CSSHIP_A_F32>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((gemm_kernel<WAVES_M, WAVES_N, AMODE, EPI, false>), grid, dim3(256), 0, s, a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

}  // namespace

extern "C" int csship_gemm(const void* A, const uint16_t* W, const float* bias, const float* residual,
                           void* C, const float* mvn_mean, const float* mvn_rinv, const float* ln_gamma,
                           const float* ln_beta, const csship_gemm_params* p, csship_stream_t stream) {
  if (!A || !W || !C || !p) return CSSHIP_ENULL;
  if (p->M <= 0 || p->N <= 0 || p->K <= 0) return CSSHIP_EBADSHAPE;
  const int nk = (p->K + BK - 1) / BK;
  if (p->ldw < nk * BK || (p->ldw % 8) != 0 || !aligned16(W)) return CSSHIP_EALIGN;
  if (p->lda < p->K || p->ldc < p->N) return CSSHIP_EBADSHAPE;
  const int epi = p->epilogue;
  if ((epi & CSSHIP_EPI_RESIDUAL) && (!residual || p->ldr < p->N)) return CSSHIP_ENULL;
  if ((epi & CSSHIP_EPI_LAYERNORM) && (p->N != 256 || !ln_gamma || !ln_beta)) return CSSHIP_EBADSHAPE;
  if (p->a_mode == CSSHIP_A_BF16 && ((p->lda % 8) != 0 || (p->K % 8) != 0 || !aligned16(A))) return CSSHIP_EALIGN;
  if (p->a_mode == CSSHIP_A_F32 && ((p->lda % 4) != 0 || (p->K % 8) != 0 || !aligned16(A))) return CSSHIP_EALIGN;
  if (p->a_mode == CSSHIP_A_F32_MVN && (!mvn_mean || !mvn_rinv || p->rows_per_batch <= 0)) return CSSHIP_ENULL;

  GemmArgs a{A, W, bias, residual, C, mvn_mean, mvn_rinv, ln_gamma, ln_beta,
             p->M, p->N, p->K, p->lda, p->ldw, p->ldc, p->ldr, p->rows_per_batch, p->alpha};
  hipStream_t s = as_stream(stream);
  const int key = p->a_mode * 16 + epi;
  switch (key) {
    // A bf16
    case CSSHIP_A_BF16 * 16 + 0: return launch<2, 2, CSSHIP_A_BF16, 0>(a, s);
    case CSSHIP_A_BF16 * 16 + CSSHIP_EPI_OUT_BF16: return launch<2, 2, CSSHIP_A_BF16, CSSHIP_EPI_OUT_BF16>(a, s);
    case CSSHIP_A_BF16 * 16 + (CSSHIP_EPI_RELU | CSSHIP_EPI_OUT_BF16):
      return launch<2, 2, CSSHIP_A_BF16, CSSHIP_EPI_RELU | CSSHIP_EPI_OUT_BF16>(a, s);
    case CSSHIP_A_BF16 * 16 + CSSHIP_EPI_RELU: return launch<2, 2, CSSHIP_A_BF16, CSSHIP_EPI_RELU>(a, s);
    case CSSHIP_A_BF16 * 16 + CSSHIP_EPI_RESIDUAL:  // no row-wide epilogue: the square tile does more MFMA work per byte staged
      return launch<2, 2, CSSHIP_A_BF16, CSSHIP_EPI_RESIDUAL>(a, s);
    case CSSHIP_A_BF16 * 16 + (CSSHIP_EPI_RESIDUAL | CSSHIP_EPI_LAYERNORM):
      return launch<1, 4, CSSHIP_A_BF16, CSSHIP_EPI_RESIDUAL | CSSHIP_EPI_LAYERNORM>(a, s);
    // A fp32 (converted on load)
    case CSSHIP_A_F32 * 16 + 0: return launch<2, 2, CSSHIP_A_F32, 0>(a, s);
    case CSSHIP_A_F32 * 16 + CSSHIP_EPI_RELU: return launch<2, 2, CSSHIP_A_F32, CSSHIP_EPI_RELU>(a, s);
    case CSSHIP_A_F32 * 16 + CSSHIP_EPI_OUT_BF16: return launch<2, 2, CSSHIP_A_F32, CSSHIP_EPI_OUT_BF16>(a, s);
    // A fp32 with MVN applied on load (embed)
    case CSSHIP_A_F32_MVN * 16 + (CSSHIP_EPI_LAYERNORM | CSSHIP_EPI_RELU):
      return launch<1, 4, CSSHIP_A_F32_MVN, CSSHIP_EPI_LAYERNORM | CSSHIP_EPI_RELU>(a, s);
    case CSSHIP_A_F32_MVN * 16 + 0: return launch<2, 2, CSSHIP_A_F32_MVN, 0>(a, s);
    default: return CSSHIP_EUNSUPPORTED;
  }
}
