// Bidirectional LSTM recurrence for the BLSTM mask estimator (css/models/blstm.py:139-160, torch.nn.LSTM
// gate order i, f, g, o, zero initial state).
//
// The input projection x W_ih^T + b_ih + b_hh for all T steps is one big MFMA GEMM done beforehand
// (csship_gemm); this kernel does the sequential part.  One workgroup = (16 batch rows, direction); wave w
// owns H/NW hidden units and ALL FOUR gates of those units, so the cell update is lane-local:
//   gates^T[unit][batch] = Gx^T (initial accumulator, bias folded in) + W_hh[unit rows][:] . h_{t-1}^T
// with W_hh as the MFMA A operand, streamed every step from L2 in fragment-packed order (it does not fit
// on one CU: 2 MB at H = 512), and h_{t-1} as a bf16 [16][H] panel in LDS (double buffered, one barrier
// per step).  c_t stays in f32 registers.  Latency bound by construction (T sequential steps).
#include "common.hpp"

namespace {


struct LstmArgs {
  const float* gx;     // [B, T, 2, 4H]
  const bf16x8* whh;   // packed per direction: [2][4H/16][Hp/32][64]
  float* out;          // [B, T, 2H]
  int B, T, H, Hp, nw; // Hp = H rounded up to 32
};

DEV float sigmoidf_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
DEV float tanhf_fast(float x) { return 2.f * sigmoidf_fast(2.f * x) - 1.f; }

template <int UT>  // 16-unit tiles per wave
__global__ __launch_bounds__(512) void lstm_kernel(LstmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];  // 2 panels [16][Hp] bf16, rows padded by 16 B
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, n = lane & 15;
  const int dir = blockIdx.y, b0 = blockIdx.x * 16;
  const int H = a.H, KSH = a.Hp / 32, rstride = a.Hp * 2 + 16;
  const int panel_bytes = 16 * rstride;
  const int brow = min(b0 + n, a.B - 1);
  const bool bvalid = b0 + n < a.B;
  const int unit0 = w * UT * 16;  // first hidden unit of this wave

  for (int i = tid; i < 2 * panel_bytes / 4; i += blockDim.x) reinterpret_cast<unsigned*>(hsm)[i] = 0u;  // h_{-1} = 0
  float c[UT][4];
#pragma unroll
  for (int u = 0; u < UT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[u][r] = 0.f;
  // A-operand stream of this wave: tiles (gate q, unit tile u) -> packed tile index q*(H/16) + unit0/16 + u
  const bf16x8* wbase = a.whh + (size_t)dir * (4 * H / 16) * KSH * 64 + lane;
  __syncthreads();

  for (int step = 0; step < a.T; ++step) {
    const int t = dir == 0 ? step : a.T - 1 - step;
    const unsigned char* hin = hsm + (step & 1) * panel_bytes;
    unsigned char* hout = hsm + ((step + 1) & 1) * panel_bytes;
    const float* gxr = a.gx + (((size_t)brow * a.T + t) * 2 + dir) * 4 * H;
    f32x4 acc[UT][4];
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(gxr + q * H + unit0 + 16 * u + 4 * g);
        acc[u][q] = f32x4{v.x, v.y, v.z, v.w};
      }
    for (int ks = 0; ks < KSH; ++ks) {
      const bf16x8 hb = *reinterpret_cast<const bf16x8*>(hin + n * rstride + (ks * 4 + g) * 16);
#pragma unroll
      for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bf16x8 wf = wbase[((size_t)(q * (H / 16) + unit0 / 16 + u) * KSH + ks) * 64];
          acc[u][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hb, acc[u][q], 0, 0, 0);
        }
    }
    float* orow = a.out + ((size_t)brow * a.T + t) * 2 * H + dir * H;
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      float hv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = sigmoidf_fast(acc[u][0][r]), fg = sigmoidf_fast(acc[u][1][r]);
        const float gg = tanhf_fast(acc[u][2][r]), og = sigmoidf_fast(acc[u][3][r]);
        c[u][r] = fg * c[u][r] + ig * gg;
        hv[r] = og * tanhf_fast(c[u][r]);
      }
      const int un = unit0 + 16 * u + 4 * g;
      *reinterpret_cast<uint2*>(hout + n * rstride + un * 2) = make_uint2(pack_bf2(hv[0], hv[1]), pack_bf2(hv[2], hv[3]));
      if (bvalid) *reinterpret_cast<float4*>(orow + un) = make_float4(hv[0], hv[1], hv[2], hv[3]);
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int csship_lstm_bidir(const float* gx, const uint16_t* whh_packed, float* out, int32_t B, int32_t T,
                                 int32_t H, csship_stream_t stream) {
  if (!gx || !whh_packed || !out) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0) return CSSHIP_EBADSHAPE;
  if ((H % 16) != 0) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(gx) || !aligned16(whh_packed) || !aligned16(out)) return CSSHIP_EALIGN;
  int nw = H / 16, ut = 1;
  if (nw > 8) {
    if (H % 128) return CSSHIP_EUNSUPPORTED;
    nw = 8;
    ut = H / 128;
  }
  const int Hp = (H + 31) / 32 * 32;
  LstmArgs a{gx, reinterpret_cast<const bf16x8*>(whh_packed), out, B, T, H, Hp, nw};
  const size_t lds = 2 * 16 * (size_t)(Hp * 2 + 16);
  dim3 grid((B + 15) / 16, 2), block(nw * 64);
  hipStream_t s = as_stream(stream);
  switch (ut) {
    case 1: hipLaunchKernelGGL(lstm_kernel<1>, grid, block, lds, s, a); break;
    case 2: hipLaunchKernelGGL(lstm_kernel<2>, grid, block, lds, s, a); break;
    case 4: hipLaunchKernelGGL(lstm_kernel<4>, grid, block, lds, s, a); break;
    default: return CSSHIP_EUNSUPPORTED;
  }
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
