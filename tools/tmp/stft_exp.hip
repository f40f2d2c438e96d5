// experiment: which part of the STFT kernel costs what.  MODE 0 full, 1 no FFT (load -> window -> store), 2 FFT but no stores
#include "../../css_amd/csrc/fft.hpp"
#include <cstdio>
constexpr int NBIN = 257;
template <int MODE, int FW, int NW>
__global__ __launch_bounds__(64 * NW) void k(const float* __restrict__ audio, int64_t n_samples, int T, float* __restrict__ mag, int S) {
  __shared__ __attribute__((aligned(16))) cplx sz[NW][FFT_SCRATCH];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int wps = (T + FW - 1) / FW;
  const int gw = blockIdx.x * NW + w;
  const int64_t s = gw / wps;
  if (s >= S) return;
  const int tw0 = (gw % wps) * FW;
  const int nf = min(FW, T - tw0);
  const float* sig = audio + s * n_samples + (int64_t)tw0 * 256;
  float win[8]; cplx tw1[4], tws[6], twp[4];
  fft_load_consts(0, l, win, tw1, tws, twp);
  const int ko = fft_out_index(l);
  float2 raw[4], nxt[2];
  for (int q = 0; q < 4; ++q) raw[q] = *reinterpret_cast<const float2*>(sig + 2 * (l + 64 * q));
  cplx* zs = sz[w];
  float sink = 0.f;
  for (int fi = 0; fi < nf; ++fi) {
    const int t = tw0 + fi;
    if (fi + 1 < nf) {
      nxt[0] = *reinterpret_cast<const float2*>(sig + (int64_t)(fi + 1) * 256 + 256 + 2 * l);
      nxt[1] = *reinterpret_cast<const float2*>(sig + (int64_t)(fi + 1) * 256 + 256 + 2 * (l + 64));
    }
    cplx z[4];
    for (int q = 0; q < 4; ++q) z[q] = {raw[q].x * win[2 * q], raw[q].y * win[2 * q + 1]};
    const size_t orow = ((size_t)s * T + t) * NBIN;
    if constexpr (MODE == 1) {
      for (int q = 0; q < 4; ++q) mag[orow + l + 64 * q] = z[q].x + z[q].y;
    } else {
      cplx u[4];
      fft256_wave(z, u, tw1, tws, zs, l);
      for (int r = 0; r < 4; ++r) zs[ko + 64 * r] = u[r];
      fft_wave_sync();
      for (int q = 0; q < 4; ++q) {
        const int kk = l + 64 * q;
        const cplx A = zs[kk], Bv = zs[(256 - kk) & 255];
        const cplx E = {0.5f * (A.x + Bv.x), 0.5f * (A.y - Bv.y)};
        const cplx Dd = {A.x - Bv.x, A.y + Bv.y};
        const cplx O = {0.5f * Dd.y, -0.5f * Dd.x};
        const cplx WO = cmul(twp[q], O);
        const float xr = E.x + WO.x, xi = E.y + WO.y;
        const float m = __builtin_amdgcn_sqrtf(xr * xr + xi * xi);
        if constexpr (MODE == 0) mag[orow + kk] = m; else sink += m;
      }
      fft_wave_sync();
    }
    raw[0] = raw[2]; raw[1] = raw[3]; raw[2] = nxt[0]; raw[3] = nxt[1];
  }
  if (MODE == 2 && sink == 12345.678f) mag[0] = sink;
}
template <int MODE, int FW, int NW> void run(const char* nm, float* a, float* m, int S, int64_t n, int T) {
  const int wps = (T + FW - 1) / FW;
  dim3 grid((S * wps + NW - 1) / NW, 1);
  const int Sg = S;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, FW, NW>), grid, dim3(64 * NW), 0, 0, a, n, T, m, Sg);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, FW, NW>), grid, dim3(64 * NW), 0, 0, a, n, T, m, Sg);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.1f us\n", nm, ms * 1000 / 20);
}
int main() {
  const int S = 96, T = 400; const int64_t n = 512 + 256 * (T - 1);
  float *a, *m; hipMalloc(&a, S * n * 4); hipMalloc(&m, (size_t)S * T * NBIN * 4 + 4096);
  hipMemset(a, 0, S * n * 4);
  run<0, 8, 4>("FW8 NW4", a, m, S, n, T); run<0, 8, 8>("FW8 NW8", a, m, S, n, T); run<0, 8, 16>("FW8 NW16", a, m, S, n, T); run<0, 10, 8>("FW10 NW8", a, m, S, n, T); run<0, 5, 16>("FW5 NW16", a, m, S, n, T); run<0, 4, 16>("FW4 NW16", a, m, S, n, T); run<1, 8, 4>("copy NW4", a, m, S, n, T); run<1, 8, 16>("copy NW16", a, m, S, n, T);
  return 0;
}
