// Diagnostic (not product code): operand / scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands,
// checked with exact integer data.  Hypothesis printed as PASS/FAIL:
//   D[i][j] = C[i][j] + sum_{b=0..3} 2^(sa[i+16b]-127) * 2^(sb[j+16b]-127) * sum_{k in [32b, 32b+32)} A[i][k] * B[k][j]
// where lane l = (row-or-col l&15, g = l>>4) holds k = 16g + [0,16) in bytes 0..15 and k = 64 + 16g + [0,16) in bytes
// 16..31 (two half blocks, NOT 32 consecutive k: found with tools/micro/mx_probe.hip), the scale of block b of row i is
// byte `opsel` of the scale VGPR of lane i + 16b, accumulator layout as the bf16 16x16 forms (col = lane&15,
// row = 4*(lane>>4) + reg).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mx_layout.hip -o /tmp/mx_layout && /tmp/mx_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const unsigned char* a, const unsigned char* b, const unsigned* sa, const unsigned* sb, float* d, int opsel) {
  const int l = threadIdx.x;
  i32x8 av, bv;
  for (int i = 0; i < 8; ++i) {
    av[i] = reinterpret_cast<const int*>(a + l * 32)[i];
    bv[i] = reinterpret_cast<const int*>(b + l * 32)[i];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, (int)sa[l], 0, (int)sb[l]);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 2, (int)sa[l], 1, (int)sb[l]);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

static float e4m3(unsigned char v) {  // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}
int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0;  // 1: scales all 127; 2: data all ones
  std::vector<unsigned char> a(64 * 32), b(64 * 32);
  std::vector<unsigned> sa(64), sb(64);
  srand(1);
  // values: sign x {0, 1, 2, 3, 4, 6, 8} -> exact small integers in e4m3 (1 = 0x38, 2 = 0x40, 3 = 0x44, 4 = 0x48, 6 = 0x4c, 8 = 0x50)
  const unsigned char tab[7] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4c, 0x50};
  for (auto& v : a) v = tab[rand() % 7] | ((rand() & 1) << 7);
  for (auto& v : b) v = tab[rand() % 7] | ((rand() & 1) << 7);
  for (int l = 0; l < 64; ++l) {
    // byte 0 used with opsel 0; bytes 2 (A) / 1 (B) with the second launch; other bytes hold decoys
    const unsigned ea = 124 + rand() % 7, eb = 124 + rand() % 7, ea2 = 124 + rand() % 7, eb1 = 124 + rand() % 7;
    sa[l] = ea | (0x7fu << 8) | (ea2 << 16) | (0x81u << 24);
    sb[l] = eb | (eb1 << 8) | (0x7du << 16) | (0x83u << 24);
    if (variant == 1) sa[l] = sb[l] = 0x7f7f7f7fu;
  }
  if (variant == 2) { for (auto& v : a) v = 0x38; for (auto& v : b) v = 0x38; }
  if (variant == 3) { for (auto& v : b) v = 0x38; for (auto& s : sb) s = 0x7f7f7f7fu; }
  if (variant == 4) { for (auto& v : a) v = 0x38; for (auto& s : sa) s = 0x7f7f7f7fu; }
  unsigned char *da, *db; unsigned *dsa, *dsb; float* dd;
  hipMalloc(&da, a.size()); hipMalloc(&db, b.size()); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 1024);
  hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int opsel = 0; opsel < 2; ++opsel) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd, opsel);
    std::vector<float> d(256);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        // block b = k / 32 (scale from lane row + 16 b); element k of a row sits in lane group g = (k % 64) / 16,
        // byte e = k % 16 + 16 * (k / 64): a lane holds TWO 16-element pieces, k = 16 g + [0, 16) and 64 + 16 g + [0, 16)
        double ref = 0;
        for (int blk = 0; blk < 4; ++blk) {
          double s = 0;
          for (int r = 0; r < 32; ++r) {
            const int kk = 32 * blk + r, g = (kk % 64) / 16, e = kk % 16 + 16 * (kk / 64);
            s += (double)e4m3(a[(i + 16 * g) * 32 + e]) * e4m3(b[(j + 16 * g) * 32 + e]);
          }
          const int la = i + 16 * blk, lb = j + 16 * blk;
          const int xa = opsel ? (sa[la] >> 16) & 255 : sa[la] & 255, xb = opsel ? (sb[lb] >> 8) & 255 : sb[lb] & 255;
          ref += ldexp(s, xa - 127 + xb - 127);
        }
        const float got = d[(j + 16 * (i >> 2)) * 4 + (i & 3)];  // row i -> lane group i>>2, reg i&3; col j -> lane j
        if (fabs(got - ref) > 1e-6 * fabs(ref) + 1e-9) { if (bad < 5) printf("  opsel %d D[%d][%d]: got %g want %g\n", opsel, i, j, got, ref); ++bad; }
      }
    printf("opsel %s: %s (%d of 256 wrong)\n", opsel ? "A=2,B=1" : "0", bad ? "FAIL" : "PASS", bad);
    bad_total += bad;
  }
  return bad_total != 0;
}
