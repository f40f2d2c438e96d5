// 256-point complex FFT on one 64-lane wave (4 points per lane), shared by the STFT and iSTFT kernels.
// Lane l holds z[l], z[l+64], z[l+128], z[l+192]: one radix-4 butterfly in registers, then four independent
// 64-point FFTs ACROSS the lanes as 6 radix-2 DIF stages.  The partner exchange lane ^ h is done in the
// VALU/crossbar without LDS addresses: DPP quad_perm (h = 1, 2), row_half_mirror + quad reverse (h = 4),
// row_ror:8 (h = 8) and the gfx950 v_permlane16_swap / v_permlane32_swap (h = 16, 32).
// Output: lane l, register r holds Z[4*brev6(l) + r].
#pragma once
#include "common.hpp"

struct cplx {
  float x, y;
};
DEV cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
DEV cplx cis_pi(float t) {  // exp(i*pi*t)
  cplx r;
  sincospif(t, &r.y, &r.x);
  return r;
}

template <int CTRL>
DEV float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

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

// per-lane twiddles for the 6 cross-lane stages (block size m = 64, 32, ..., 2)
DEV void fft_stage_twiddles(cplx (&tws)[6], int l) {
#pragma unroll
  for (int st = 0; st < 6; ++st) {
    const int half = 32 >> st, m = 2 * half, p = l & (m - 1);
    tws[st] = (p >= half) ? cis_pi(-2.f * (float)(p - half) / (float)m) : cplx{1.f, 0.f};
  }
}

template <int H>
DEV void fft_stage(cplx (&u)[4], cplx tw, int l) {
  const float sg = (l & H) ? -1.f : 1.f;  // lower lane: a + b ; upper lane: (a - b) seen from b's side = other - mine
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ox = lane_xor<H>(u[r].x, l), oy = lane_xor<H>(u[r].y, l);
    const cplx d = {fmaf(u[r].x, sg, ox), fmaf(u[r].y, sg, oy)};
    if constexpr (H == 1) {
      u[r] = d;  // block size 2: twiddle W_2^0 = 1
    } else if constexpr (H == 2) {
      // block size 4: twiddle is 1 except at position 3 (l & 3 == 3), where it is W_4^1 = -i: (x, y) -> (y, -x)
      const bool rot = (l & 3) == 3;
      u[r] = {rot ? d.y : d.x, rot ? -d.x : d.y};
    } else {
      u[r] = cmul(d, tw);
    }
  }
}

// z[q] = input point l + 64q; tw1[q] = W256^(l*q)
DEV void fft256_wave(const cplx (&z)[4], cplx (&u)[4], const cplx (&tw1)[4], const cplx (&tws)[6], int l) {
  const cplx a02 = {z[0].x + z[2].x, z[0].y + z[2].y}, s02 = {z[0].x - z[2].x, z[0].y - z[2].y};
  const cplx a13 = {z[1].x + z[3].x, z[1].y + z[3].y}, s13 = {z[1].x - z[3].x, z[1].y - z[3].y};
  u[0] = {a02.x + a13.x, a02.y + a13.y};
  u[1] = cmul({s02.x + s13.y, s02.y - s13.x}, tw1[1]);  // (z0-z2) - i(z1-z3)
  u[2] = cmul({a02.x - a13.x, a02.y - a13.y}, tw1[2]);
  u[3] = cmul({s02.x - s13.y, s02.y + s13.x}, tw1[3]);  // (z0-z2) + i(z1-z3)
  fft_stage<32>(u, tws[0], l);
  fft_stage<16>(u, tws[1], l);
  fft_stage<8>(u, tws[2], l);
  fft_stage<4>(u, tws[3], l);
  fft_stage<2>(u, tws[4], l);
  fft_stage<1>(u, tws[5], l);
}
