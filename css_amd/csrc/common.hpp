// Shared device helpers for libcsship (gfx950 only: wave = 64 lanes, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/csship.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DEV static __device__ __forceinline__

DEV unsigned short f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
DEV float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
DEV unsigned pack_bf2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }

DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// LDS tile of bf16 rows of 64 elements (128 B, eight 16-byte chunks).  Chunk c of row r
// lives at r*128 + ((c ^ ((r>>1)&7))<<4): 16 consecutive rows reading one logical chunk
// with ds_read_b128 touch 16 distinct 16-byte slots of the 256-byte bank row.
DEV int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

static inline hipStream_t as_stream(csship_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define CSSHIP_CHECK_LAUNCH()                          \
  do {                                                 \
    if (hipGetLastError() != hipSuccess) return CSSHIP_ELAUNCH; \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

// out[r, c] = alpha * sum_s slabs[s][r*N + c] (+ bias[c]) (+ add[r*ldadd + c]), fixed order (bgemm.hip)
int launch_slab_reduce(const float* slabs, int S, int64_t n, int N, int64_t ldc, float alpha, const float* bias,
                       const float* add, int64_t ldadd, float* out, hipStream_t s);

// Optional in-kernel phase stamps (diagnostic only; nullptr in normal operation): thread 0 of every
// workgroup stores s_memrealtime (100 MHz) at up to 8 phase boundaries.  Set by csship_debug_stamps().
extern unsigned long long* g_csship_stamps;
DEV void stamp(unsigned long long* st, int slot) {
  if (st != nullptr && threadIdx.x == 0) {
    const size_t wg = (size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
    st[wg * 8 + slot] = __builtin_amdgcn_s_memrealtime();
  }
}
