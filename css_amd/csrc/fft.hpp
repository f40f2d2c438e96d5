// 256-point complex FFT on one 64-lane wave (4 points per lane), shared by the STFT and iSTFT kernels.
// Radix-4 decimation in frequency, 4 stages, every butterfly in registers.  With n = n0 + 4 n1 + 16 n2 + 64 n3 and
// k = k0 + 4 k1 + 16 k2 + 64 k3, lane l starts as (n0, n1, n2) holding register q = n3, i.e. z[l + 64 q].  Between
// stages the register index is exchanged with one base-4 digit of the lane index (a 4x4 transpose across 4 lanes):
//   digit n2 (lane bits 4,5): v_permlane32_swap + v_permlane16_swap = 8 instructions, no LDS, no selects;
//   digits n1, n0: through a 2 KB per-wave LDS scratch (4 ds_write_b64 + 4 ds_read_b64 each, padded row strides
//   68 / 65 make both conflict free).
// 3 twiddle multiplies per stage from the build-time table.  Replaces the earlier radix-4 + 6 radix-2 shuffle stages
// (~225 -> ~110 VALU instructions per frame for the transform itself).
// Output: lane l = (k2, k1, k0) (k2 in bits 0-1), register r = k3 holds Z[fft_out_index(l) + 64 r].
#pragma once
#include "common.hpp"

// complex numbers are 2-vectors so that the butterflies compile to packed FP32 (v_pk_add_f32 / v_pk_fma_f32 with
// op_sel / neg modifiers): one instruction per complex add, two per complex multiply
typedef float cplx __attribute__((ext_vector_type(2)));
DEV cplx cmul(cplx a, cplx b) {  // a.xx * b + a.yy * (-b.y, b.x)
  return __builtin_elementwise_fma(a.xx, b, a.yy * cplx{-b.y, b.x});
}
DEV cplx cis_pi(float t) {  // exp(i*pi*t)
  float sn, cs;
  sincospif(t, &sn, &cs);
  return {cs, sn};
}

// dpp_mov<CTRL>: common.hpp

template <int H>
DEV float lane_xor(float x, int lane) {
  if constexpr (H == 1) return dpp_mov<0xB1>(x);                     // quad_perm [1,0,3,2]
  else if constexpr (H == 2) return dpp_mov<0x4E>(x);                // quad_perm [2,3,0,1]
  else if constexpr (H == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(x));  // row_half_mirror, then quad_perm [3,2,1,0]
  else if constexpr (H == 8) return dpp_mov<0x128>(x);               // row_ror:8
  else if constexpr (H == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return __builtin_bit_cast(float, ((lane >> 4) & 1) ? r[0] : r[1]);
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return __builtin_bit_cast(float, (lane >> 5) ? r[0] : r[1]);
  }
}

#include "fft_tables.inc"

// per-lane constants from the build-time table (gen_tables.py): window, radix-4 twiddles, stage twiddles,
// real-FFT post/pre twiddles.  which = 0 forward, 1 inverse.
DEV void fft_load_consts(int which, int l, float (&win)[8], cplx (&tw1)[4], cplx (&tws)[6], cplx (&twp)[4]) {
  float c[36];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const float4 v = kFftTab[which][i][l];
    c[4 * i] = v.x; c[4 * i + 1] = v.y; c[4 * i + 2] = v.z; c[4 * i + 3] = v.w;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) win[i] = c[i];
  tw1[0] = {1.f, 0.f};
#pragma unroll
  for (int q = 1; q < 4; ++q) tw1[q] = {c[8 + 2 * (q - 1)], c[9 + 2 * (q - 1)]};
#pragma unroll
  for (int st = 0; st < 6; ++st) tws[st] = {c[14 + 2 * st], c[15 + 2 * st]};
#pragma unroll
  for (int q = 0; q < 4; ++q) twp[q] = {c[26 + 2 * q], c[27 + 2 * q]};
}

DEV int fft_out_index(int l) { return ((l >> 4) & 3) | (l & 12) | ((l & 3) << 4); }  // k0 + 4 k1 + 16 k2

constexpr int FFT_SCRATCH = 4 * 68;  // cplx entries of per-wave LDS scratch

DEV void radix4_dif(cplx (&u)[4]) {  // y_k = sum_n x_n (-i)^(n k)
  const cplx a02 = u[0] + u[2], s02 = u[0] - u[2];
  const cplx a13 = u[1] + u[3], s13 = u[1] - u[3];
  const cplx r = {s13.y, -s13.x};  // -i (x1-x3)
  u[0] = a02 + a13;
  u[1] = s02 + r;
  u[2] = a02 - a13;
  u[3] = s02 - r;
}

// 2x2 block transposes between a register pair and one lane bit: a' = [a.low | b.low], b' = [a.high | b.high]
DEV void swap32f(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  const unsigned r0 = r[0], r1 = r[1];  // extract first: bit_cast applied directly to r[1] reads element 0 (hipcc 7.2)
  a = __uint_as_float(r0);
  b = __uint_as_float(r1);
}
DEV void swap16f(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __uint_as_float(r0);
  b = __uint_as_float(r1);
}

DEV void swap32(cplx& a, cplx& b) {
  float ax = a.x, ay = a.y, bx = b.x, by = b.y;
  swap32f(ax, bx);
  swap32f(ay, by);
  a = {ax, ay};
  b = {bx, by};
}
DEV void swap16(cplx& a, cplx& b) {
  float ax = a.x, ay = a.y, bx = b.x, by = b.y;
  swap16f(ax, bx);
  swap16f(ay, by);
  a = {ax, ay};
  b = {bx, by};
}

DEV void fft_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// u[c][q] = input point l + 64q of transform c (in place); tw1[q] = W256^(l q); tws[0..2] = W64^((l & 15) k),
// tws[3..5] = W16^((l & 3) k), k = 1..3; zs: this wave's LDS scratch (NC * FFT_SCRATCH entries).  NC independent
// transforms advance together: one wave-level sync per exchange serves all of them and the scheduler overlaps one
// transform's LDS round trip with the other's butterflies (a lone dependent chain leaves the SIMD idle).
template <int NC>
DEV void fft256_wave_n(cplx (&u)[NC][4], const cplx (&tw1)[4], const cplx (&tws)[6], cplx* zs, int l) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    radix4_dif(u[c]);  // over n3 -> k0
#pragma unroll
    for (int q = 1; q < 4; ++q) u[c][q] = cmul(u[c][q], tw1[q]);
    // register k0 <-> lane digit n2
    swap32(u[c][0], u[c][2]); swap32(u[c][1], u[c][3]);
    swap16(u[c][0], u[c][1]); swap16(u[c][2], u[c][3]);
    radix4_dif(u[c]);  // over n2 -> k1
#pragma unroll
    for (int q = 1; q < 4; ++q) u[c][q] = cmul(u[c][q], tws[q - 1]);
    // register k1 <-> lane digit n1
#pragma unroll
    for (int r = 0; r < 4; ++r) zs[c * FFT_SCRATCH + r * 68 + l] = u[c][r];
  }
  fft_wave_sync();
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int base = ((l >> 2) & 3) * 68 + (l & ~12);
#pragma unroll
    for (int j = 0; j < 4; ++j) u[c][j] = zs[c * FFT_SCRATCH + base + 4 * j];
  }
  fft_wave_sync();
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    radix4_dif(u[c]);  // over n1 -> k2
#pragma unroll
    for (int q = 1; q < 4; ++q) u[c][q] = cmul(u[c][q], tws[2 + q]);
    // register k2 <-> lane digit n0
#pragma unroll
    for (int r = 0; r < 4; ++r) zs[c * FFT_SCRATCH + r * 65 + l] = u[c][r];
  }
  fft_wave_sync();
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int base = (l & 3) * 65 + (l & ~3);
#pragma unroll
    for (int j = 0; j < 4; ++j) u[c][j] = zs[c * FFT_SCRATCH + base + j];
  }
  fft_wave_sync();
#pragma unroll
  for (int c = 0; c < NC; ++c) radix4_dif(u[c]);  // over n0 -> k3
}

// single transform: u may alias z
DEV void fft256_wave(const cplx (&z)[4], cplx (&u)[4], const cplx (&tw1)[4], const cplx (&tws)[6], cplx* zs, int l) {
  cplx v[1][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[0][q] = z[q];
  fft256_wave_n<1>(v, tw1, tws, zs, l);
#pragma unroll
  for (int q = 0; q < 4; ++q) u[q] = v[0][q];
}
