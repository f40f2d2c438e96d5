// Diagnostic (not product code): probes which lane / byte of the scale operands and which operand bytes of
// v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3) pair up.  Prints raw findings.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int OA, int OB>
__global__ void k(const unsigned char* a, const unsigned char* b, const unsigned* sa, const unsigned* sb, float* d) {
  const int l = threadIdx.x;
  i32x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = reinterpret_cast<const int*>(a + l * 32)[i]; bv[i] = reinterpret_cast<const int*>(b + l * 32)[i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, OA, (int)sa[l], OB, (int)sb[l]);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
static unsigned char *da, *db; static unsigned *dsa, *dsb; static float* dd;
static std::vector<float> run(const std::vector<unsigned char>& a, const std::vector<unsigned char>& b, const std::vector<unsigned>& sa,
                              const std::vector<unsigned>& sb, int oa = 0, int ob = 0) {
  hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  if (oa == 0 && ob == 0) hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  else if (oa == 1) hipLaunchKernelGGL((k<1, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  else if (oa == 2) hipLaunchKernelGGL((k<2, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  else hipLaunchKernelGGL((k<3, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  std::vector<float> d(256), D(256);
  hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = d[l * 4 + r];  // D[row][col]
  return D;
}
int main() {
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 1024);
  std::vector<unsigned char> ones(2048, 0x38), zeros(2048, 0);
  std::vector<unsigned> s127(64, 0x7f7f7f7fu);
  auto base = run(ones, ones, s127, s127);
  printf("all ones, scales 127: D[0][0] = %g D[5][9] = %g (expect 128)\n", base[0], base[5 * 16 + 9]);
  for (int byte = 0; byte < 4; ++byte)
    for (int opsel = 0; opsel < 4; ++opsel) {
      printf("A-scale byte %d doubled, opsel_a %d: lanes that change rows:", byte, opsel);
      for (int L = 0; L < 64; ++L) {
        auto sa = s127; sa[L] = (sa[L] & ~(0xffu << (8 * byte))) | (0x80u << (8 * byte));
        auto D = run(ones, ones, sa, s127, opsel, 0);
        int nrow = 0, row = -1; float delta = 0;
        for (int i = 0; i < 16; ++i) if (D[i * 16] != 128.f) { ++nrow; row = i; delta = D[i * 16] - 128.f; }
        if (nrow) printf(" L%d->r%d(%+g)%s", L, row, delta, nrow > 1 ? "*" : "");
      }
      printf("\n");
    }
  // B scale, byte 0, opsel 0
  printf("B-scale byte 0 doubled: ");
  for (int L = 0; L < 64; ++L) {
    auto sb = s127; sb[L] = (sb[L] & ~0xffu) | 0x80u;
    auto D = run(ones, ones, s127, sb);
    int ncol = 0, col = -1; float delta = 0;
    for (int j = 0; j < 16; ++j) if (D[j] != 128.f) { ++ncol; col = j; delta = D[j] - 128.f; }
    if (ncol) printf(" L%d->c%d(%+g)%s", L, col, delta, ncol > 1 ? "*" : "");
  }
  printf("\n");
  // which scale lane-group applies to which data lane-group
  for (int which = 0; which < 2; ++which) {
    printf("%s: data group g (rows) x scale group g' (cols): D[0][0] (32 = plain, 64 = this scale applies)\n", which ? "B" : "A");
    for (int g = 0; g < 4; ++g) {
      auto x = zeros;
      for (int l = 16 * g; l < 16 * g + 16; ++l) for (int e = 0; e < 32; ++e) x[l * 32 + e] = 0x38;
      printf("  g=%d:", g);
      for (int gp = 0; gp < 4; ++gp) {
        auto sc = s127;
        for (int l = 16 * gp; l < 16 * gp + 16; ++l) sc[l] = 0x80808080u;
        auto D = which ? run(ones, x, s127, sc) : run(x, ones, sc, s127);
        printf(" %g", D[0]);
      }
      printf("\n");
    }
  }
  // data pairing: A one-hot at (la, ea); B lane lb element e = code
  const unsigned char code16[16] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50, 0x51, 0x52, 0x53, 0x54, 0x55, 0x56, 0x57};  // 0..15
  for (int t = 0; t < 12; ++t) {
    const int la = (t * 23 + 5) % 64, ea = (t * 7 + 3) % 32;
    auto a = zeros; a[la * 32 + ea] = 0x38;
    std::vector<unsigned char> b1(2048), b2(2048);
    for (int lb = 0; lb < 64; ++lb) for (int e = 0; e < 32; ++e) { b1[lb * 32 + e] = code16[e % 16]; b2[lb * 32 + e] = code16[(e / 16) + 2 * (lb >> 4) + 1]; }
    auto D1 = run(a, b1, s127, s127), D2 = run(a, b2, s127, s127);
    int row = -1;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (D1[i * 16 + j] != 0 || D2[i * 16 + j] != 0) row = i;
    printf("A one-hot lane %d (row-lane %d, g %d) elem %d -> D row %d: paired B elem%%16 = %g, (elem/16 + 2g + 1) = %g  [expect %d, %d]\n", la, la & 15, la >> 4, ea,
           row, row >= 0 ? D1[row * 16 + 3] : -1.f, row >= 0 ? D2[row * 16 + 3] : -1.f, ea % 16, ea / 16 + 2 * (la >> 4) + 1);
  }
  return 0;
}
