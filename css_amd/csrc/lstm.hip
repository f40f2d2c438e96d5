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
  float* gates_out;    // optional (training): activated gates i,f,g,o  [B, T, 2, 4H]
  float* c_out;        // optional (training): cell states            [B, T, 2, H]
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
    const size_t srow = ((size_t)brow * a.T + t) * 2 + dir;
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      float hv[4];
      const int un = unit0 + 16 * u + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = sigmoidf_fast(acc[u][0][r]), fg = sigmoidf_fast(acc[u][1][r]);
        const float gg = tanhf_fast(acc[u][2][r]), og = sigmoidf_fast(acc[u][3][r]);
        c[u][r] = fg * c[u][r] + ig * gg;
        hv[r] = og * tanhf_fast(c[u][r]);
        acc[u][0][r] = ig; acc[u][1][r] = fg; acc[u][2][r] = gg; acc[u][3][r] = og;
      }
      if (a.gates_out && bvalid) {  // saved for the backward pass
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(a.gates_out + srow * 4 * H + q * H + un) =
              make_float4(acc[u][q][0], acc[u][q][1], acc[u][q][2], acc[u][q][3]);
        *reinterpret_cast<float4*>(a.c_out + srow * H + un) = make_float4(c[u][0], c[u][1], c[u][2], c[u][3]);
      }
      *reinterpret_cast<uint2*>(hout + n * rstride + un * 2) = make_uint2(pack_bf2(hv[0], hv[1]), pack_bf2(hv[2], hv[3]));
      if (bvalid) *reinterpret_cast<float4*>(orow + un) = make_float4(hv[0], hv[1], hv[2], hv[3]);
    }
    __syncthreads();
  }
}

// ---- the same recurrence spread over H/32 CUs per (16 batch rows, direction).
// The kernel above is bound by one CU pulling all of W_hh (2 MB at H = 512) through its L1 every step (13 us per step,
// 8.9 ms per layer at T = 400, 4 workgroups busy).  Here a workgroup of two waves owns 32 hidden units: each wave keeps
// the W_hh rows of its 16 units (4 gates x H: 64 KB at H = 512) IN REGISTERS for the whole sequence, so a step is 4*H/32
// MFMAs on register operands.  What a step needs from the other workgroups is h_{t-1} (16 rows x H bf16 = 16 KB):
//   publish: every lane stores its 4 units of h_t write-through (sc1) into a double-buffered global panel, the waves
//            drain their stores (vmcnt 0), barrier, one lane stores the workgroup's epoch flag (agent-scope atomic);
//   consume: each wave polls the H/32 flags of its group (relaxed, bounded), then loads its MFMA B fragments of the
//            panel with sc1 loads - every load of a handed-off byte is an sc1 load, so no acquire fence is needed.
// Nothing depends on dispatch order or placement; all workgroups of a group must be resident together (64 workgroups of
// 128 threads at B = 32, H = 512), and every spin is bounded: on a timeout the error word is set and the kernel exits.
constexpr int LSTM_NW = 2;                 // waves per forward workgroup, 16 hidden units each (4 measured: 2.71 vs 2.66 us per step)
constexpr int FLAG_STRIDE = 16;            // 32-bit words between flags (one 64-byte line each)
constexpr unsigned SPIN_LIMIT = 1u << 21;  // ~1 s

struct LstmDistArgs {
  const float* gx;      // [B, T, 2, 4H]
  const bf16x8* whh;    // packed per direction: [2][4H/16][H/32][64]
  float* out;           // [B, T, 2H]
  float* gates_out;     // optional (training)
  float* c_out;         // optional (training)
  unsigned char* hx;    // [2 parity][2 dir][tiles][16][H] bf16
  unsigned* flags;      // [2 dir][tiles][H/32] * FLAG_STRIDE, zero at launch
  unsigned* err;        // STICKY: outside the per-launch memset, only ever set (1 = a spin timed out); read and cleared
                        // by the host at its own synchronisation points (css_amd.ops.lstm_dist_check)
  int B, T, tiles, allow_l2;
  unsigned tag0;        // per-launch salt << 16: a granule left by an earlier launch can never carry this launch's tag
  unsigned* stamps;     // -DLSTM_STAMPS only
};

typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

// AUX: cache-coherence bits of the hand-off STORE.  The loads are always sc1 (miss the CU's L1, served by the XCD's L2
// or, for lines it does not hold, by memory).  16 = sc1 store: written through to memory, visible on every XCD.
// 0 = plain store: stays in the producer's L2, which only the CUs of the same XCD read - a third of the round trip
// (tools/micro/pingpong.hip: 0.34 us against 0.95 us), valid only when the whole group runs on one XCD.
template <int KSH, int AUX>  // H = 32 * KSH
DEV void lstm_dist_body(const LstmDistArgs& a, const int cu, const int tile, const int dir, const int tiles) {
  constexpr int H = 32 * KSH;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, n = lane & 15;
  const int ut = cu * LSTM_NW + w, unit0 = 16 * ut;
  const int b0 = tile * 16;
  const int brow = min(b0 + n, a.B - 1);
  const bool bvalid = b0 + n < a.B;

  // this wave's rows of W_hh: gate q, k-step ks -> packed tile (q*(H/16) + ut, ks); resident for all T steps
  bf16x8 wf[4][KSH];
  {
    const bf16x8* wbase = a.whh + (size_t)dir * (4 * H / 16) * KSH * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks) wf[q][ks] = wbase[((size_t)(q * (H / 16) + ut) * KSH + ks) * 64];
  }
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  __shared__ bf16x8 xch[2 * KSH * 64];  // the step's h_{t-1} fragments, shared by the two waves (2 parities)
  // The exchanged panel carries its own validity: every two bf16 values travel as an 8-byte granule {epoch, value}
  // (epoch = tag0 + step + 1, never 0; tag0 = a per-launch salt in the high half, and the panels are zeroed per launch
  // as well), so the consumer polls the data itself - no flag,
  // no store drain, no barrier: one fabric round trip per step instead of three.  Row stride 4*H bytes.
  const unsigned row_bytes = 4u * H;
  const unsigned group = (unsigned)(dir * tiles + tile) * 16u * row_bytes;
  const unsigned parity_bytes = 2u * (unsigned)tiles * 16u * row_bytes;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.hx, 0, (int)(2u * parity_bytes), 0x00020000);

  // the input projections of step s + 1 are requested during step s (they come from HBM: 2 us if asked for on demand)
  float4 gxn[4];
  {
    const int t0 = dir == 0 ? 0 : a.T - 1;
    const float* gxr = a.gx + (((size_t)brow * a.T + t0) * 2 + dir) * 4 * H + unit0 + 4 * g;
#pragma unroll
    for (int q = 0; q < 4; ++q) gxn[q] = *reinterpret_cast<const float4*>(gxr + q * H);
  }
#ifdef LSTM_STAMPS
  __shared__ unsigned stamps[LSTM_NW][32][6];
#define LSTAMP(i, v) do { if (lane == 0 && step >= 100 && step < 132) stamps[w][step - 100][(i)] = (v); } while (0)
#else
#define LSTAMP(i, v) do {} while (0)
#endif
  for (int step = 0; step < a.T; ++step) {
    const int t = dir == 0 ? step : a.T - 1 - step;
    LSTAMP(0, (unsigned)__builtin_readcyclecounter());
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{gxn[q].x, gxn[q].y, gxn[q].z, gxn[q].w};
    if (step + 1 < a.T) {
      const int tn = dir == 0 ? step + 1 : a.T - 2 - step;
      const float* gxr = a.gx + (((size_t)brow * a.T + tn) * 2 + dir) * 4 * H + unit0 + 4 * g;
#pragma unroll
      for (int q = 0; q < 4; ++q) gxn[q] = *reinterpret_cast<const float4*>(gxr + q * H);
    }
    if (step > 0) {
      // ---- h_{t-1}: this lane's B fragments are 8 consecutive units of row n per k-step = 4 granules = 32 bytes;
      // re-read (bounded) until every granule of the wave carries this step's epoch
      // fragment order: piece (k-step, half) of all 64 lanes is 1 KB contiguous - one load instruction = 8 whole lines
      const unsigned pbase = (unsigned)(step & 1) * parity_bytes + group + (unsigned)lane * 16u;
      // The two waves need the same panel: each fetches half of its k-steps (the sc1 loads are the step's longest
      // phase: ~2 us for 32 KB per wave), the halves meet in LDS (double-buffered by step parity, one bare barrier).
      constexpr int KH = KSH / LSTM_NW;
      bf16x8 hh[KH];
      bool ok = false;
      LSTAMP(1, (unsigned)__builtin_readcyclecounter());
      const unsigned hbase = pbase + (unsigned)(w * KH) * 2048u;
      for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
        bool mine = true;
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
          const v4u_t lo = __builtin_amdgcn_raw_buffer_load_b128(rs, hbase + (unsigned)ks * 2048u, 0, 16);
          const v4u_t hi = __builtin_amdgcn_raw_buffer_load_b128(rs, hbase + (unsigned)ks * 2048u + 1024u, 0, 16);
          const unsigned want = a.tag0 + (unsigned)step;
          mine = mine && lo.x == want && lo.z == want && hi.x == want && hi.z == want;
          const v4u_t fr = {lo.y, lo.w, hi.y, hi.w};
          hh[ks] = __builtin_bit_cast(bf16x8, fr);
        }
#ifdef LSTM_NOEXCH
        mine = true;
#endif
        if (__all(mine)) {
          ok = true;
          LSTAMP(5, spins);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) {
        if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;  // (a wave that has ended no longer counts at the other wave's barrier)
      }
      bf16x8* xs = xch + (size_t)(step & 1) * KSH * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) xs[(w * KH + ks) * 64] = hh[ks];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // bare: __syncthreads() would also wait for the output stores and the gx prefetch
      bf16x8 hb[KSH];
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks) hb[ks] = xs[ks * 64];
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q][ks], hb[ks], acc[q], 0, 0, 0);
    }
    LSTAMP(2, (unsigned)__builtin_readcyclecounter());
    float hv[4];
    const int un = unit0 + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ig = sigmoidf_fast(acc[0][r]), fg = sigmoidf_fast(acc[1][r]);
      const float gg = tanhf_fast(acc[2][r]), og = sigmoidf_fast(acc[3][r]);
      c[r] = fg * c[r] + ig * gg;
      hv[r] = og * tanhf_fast(c[r]);
      acc[0][r] = ig; acc[1][r] = fg; acc[2][r] = gg; acc[3][r] = og;
    }
    // ---- publish h_t: two granules {epoch = step + 1, 2 x bf16}, one 16-byte write-through store
    if (step + 1 < a.T) {
      const unsigned ep = a.tag0 + (unsigned)(step + 1);
      const v4u_t hp = {ep, pack_bf2(hv[0], hv[1]), ep, pack_bf2(hv[2], hv[3])};
      // units un..un+3 of row n = half (un % 8) / 4 of the consumer lane (g = (un % 32) / 8, n)'s fragment of k-step un / 32
      const unsigned piece = (unsigned)(un >> 5) * 2u + (unsigned)((un >> 2) & 1);
      const unsigned off = (unsigned)((step + 1) & 1) * parity_bytes + group + piece * 1024u +
                           ((unsigned)((un & 31) >> 3) * 16u + (unsigned)n) * 16u;
      __builtin_amdgcn_raw_buffer_store_b128(hp, rs, off, 0, AUX);  // sc1: write-through
    }
    LSTAMP(3, (unsigned)__builtin_readcyclecounter());
    const size_t srow = ((size_t)brow * a.T + t) * 2 + dir;
    if (bvalid) {
      *reinterpret_cast<float4*>(a.out + ((size_t)brow * a.T + t) * 2 * H + dir * H + un) =
          make_float4(hv[0], hv[1], hv[2], hv[3]);
      if (a.gates_out) {  // saved for the backward pass
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(a.gates_out + srow * 4 * H + q * H + un) =
              make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
        *reinterpret_cast<float4*>(a.c_out + srow * H + un) = make_float4(c[0], c[1], c[2], c[3]);
      }
    }
    LSTAMP(4, (unsigned)__builtin_readcyclecounter());
  }
#ifdef LSTM_STAMPS
  if (a.stamps && lane < 32) {
    unsigned* o = a.stamps + (((size_t)(dir * tiles + tile) * (H / (16 * LSTM_NW)) + cu) * LSTM_NW + w) * 32 * 6 + lane * 6;
    for (int i = 0; i < 6; ++i) o[i] = stamps[w][lane][i];
  }
#endif
}

// Placement.  Workgroups are dealt to the 8 XCDs round-robin by linear id, and an XCD's CUs share one L2: a group
// (one direction x 16 batch rows, H/32 workgroups) whose members all sit on ONE XCD can hand its state over through
// that L2 (sc0 loads, ~4x shorter round trip) instead of through memory.  The 1-D grid is laid out so that group g
// owns the ids = g (mod 8); nothing relies on it: every workgroup publishes the XCC_ID it actually runs on, the
// group compares (bounded wait), and only a group that is provably co-located takes the L2 path - all members see
// the same ids, so they all take the same path.
DEV int lstm_group_colocated(unsigned* place, int slot, int n, unsigned* err) {
  const int lane = threadIdx.x & 63;
  const unsigned mine = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) + 1u;  // HW_REG_XCC_ID[3:0] + 1
  if (threadIdx.x == 0) __hip_atomic_store(place + slot * FLAG_STRIDE + 1, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
    unsigned v = mine;
    if (lane < n) v = __hip_atomic_load(place + lane * FLAG_STRIDE + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__all(v != 0u)) return __all(v == mine) ? 1 : 0;
    __builtin_amdgcn_s_sleep(2);
  }
  if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return -1;
}

template <int KSH>
__global__ __launch_bounds__(64 * LSTM_NW) void lstm_dist_kernel(LstmDistArgs a) {
  constexpr int NCU = 2 * KSH / LSTM_NW;  // H / (16 * LSTM_NW)
  const int tiles = a.tiles, groups = 2 * tiles;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int g = xcd + 8 * (slot / NCU), cu = slot % NCU;
  if (g >= groups) return;
  const int dir = g / tiles, tile = g % tiles;
  const int same = a.allow_l2 ? lstm_group_colocated(a.flags + (size_t)g * NCU * FLAG_STRIDE, cu, NCU, a.err) : 0;
  if (same < 0) return;
  if (same) lstm_dist_body<KSH, 0>(a, cu, tile, dir, tiles);
  else lstm_dist_body<KSH, 16>(a, cu, tile, dir, tiles);
}

// ---- backward of the recurrence (BPTT).  Mirrors the forward: workgroup = (16 batch rows, direction), wave w owns
// H/NW hidden units; per step (walking time against the forward order)
//   dh^T[unit][batch] = dy_t + W_hh^T[unit rows][4H] . dgates_{next}^T      (MFMA, W_hh^T streamed, dgates panel in LDS)
//   cell / gate derivatives lane-local from the saved activated gates and cell states,
//   dgates_t (pre-activation gradients of i,f,g,o) -> LDS panel for the next step and global for the weight GEMMs
// (dW_ih, dW_hh, db, dx are plain GEMMs / column sums over dgates afterwards).
struct LstmBwdArgs {
  const float* gates;  // [B, T, 2, 4H] activated
  const float* c;      // [B, T, 2, H]
  const float* dy;     // [B, T, 2H]
  const bf16x8* whhT;  // packed per direction from W_hh^T [H, 4H]: [2][H/16][4H/32][64]
  float* dgates;       // [B, T, 2, 4H]
  int B, T, H;
};

template <int UT>
__global__ __launch_bounds__(512) void lstm_bwd_kernel(LstmBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];  // 2 panels [16][4H] bf16, rows padded by 16 B
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, n = lane & 15;
  const int dir = blockIdx.y, b0 = blockIdx.x * 16;
  const int H = a.H, KS4 = 4 * H / 32, rstride = 4 * H * 2 + 16;
  const int panel_bytes = 16 * rstride;
  const int brow = min(b0 + n, a.B - 1);
  const bool bvalid = b0 + n < a.B;
  const int unit0 = w * UT * 16;
  for (int i = tid; i < 2 * panel_bytes / 4; i += blockDim.x) reinterpret_cast<unsigned*>(hsm)[i] = 0u;
  float dc[UT][4];
#pragma unroll
  for (int u = 0; u < UT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) dc[u][r] = 0.f;
  const bf16x8* wbase = a.whhT + (size_t)dir * (H / 16) * KS4 * 64 + lane;
  __syncthreads();

  for (int step = 0; step < a.T; ++step) {
    const int t = dir == 0 ? a.T - 1 - step : step;            // against the forward order
    const int tp = dir == 0 ? t - 1 : t + 1;                   // the forward pass's previous step
    const bool has_prev = tp >= 0 && tp < a.T;
    const unsigned char* din = hsm + (step & 1) * panel_bytes;
    unsigned char* dout = hsm + ((step + 1) & 1) * panel_bytes;
    const size_t srow = ((size_t)brow * a.T + t) * 2 + dir;
    const size_t prow = ((size_t)brow * a.T + (has_prev ? tp : t)) * 2 + dir;
    f32x4 acc[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const float4 v = *reinterpret_cast<const float4*>(a.dy + ((size_t)brow * a.T + t) * 2 * H + dir * H + unit0 + 16 * u + 4 * g);
      acc[u] = f32x4{v.x, v.y, v.z, v.w};
    }
    for (int ks = 0; ks < KS4; ++ks) {
      const bf16x8 db = *reinterpret_cast<const bf16x8*>(din + n * rstride + (ks * 4 + g) * 16);
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const bf16x8 wf = wbase[((size_t)(unit0 / 16 + u) * KS4 + ks) * 64];
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, db, acc[u], 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int un = unit0 + 16 * u + 4 * g;
      float4 gq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) gq[q] = *reinterpret_cast<const float4*>(a.gates + srow * 4 * H + q * H + un);
      const float4 ct = *reinterpret_cast<const float4*>(a.c + srow * H + un);
      float4 cp = *reinterpret_cast<const float4*>(a.c + prow * H + un);
      if (!has_prev) cp = make_float4(0.f, 0.f, 0.f, 0.f);
      const float iv[4] = {gq[0].x, gq[0].y, gq[0].z, gq[0].w}, fv[4] = {gq[1].x, gq[1].y, gq[1].z, gq[1].w};
      const float gv[4] = {gq[2].x, gq[2].y, gq[2].z, gq[2].w}, ov[4] = {gq[3].x, gq[3].y, gq[3].z, gq[3].w};
      const float cv[4] = {ct.x, ct.y, ct.z, ct.w}, pv[4] = {cp.x, cp.y, cp.z, cp.w};
      float da[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dh = acc[u][r];
        const float tc = tanhf_fast(cv[r]);
        const float dcc = dc[u][r] + dh * ov[r] * (1.f - tc * tc);
        da[0][r] = dcc * gv[r] * iv[r] * (1.f - iv[r]);
        da[1][r] = dcc * pv[r] * fv[r] * (1.f - fv[r]);
        da[2][r] = dcc * iv[r] * (1.f - gv[r] * gv[r]);
        da[3][r] = dh * tc * ov[r] * (1.f - ov[r]);
        dc[u][r] = dcc * fv[r];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<uint2*>(dout + n * rstride + (q * H + un) * 2) =
            make_uint2(pack_bf2(da[q][0], da[q][1]), pack_bf2(da[q][2], da[q][3]));
        if (bvalid)
          *reinterpret_cast<float4*>(a.dgates + srow * 4 * H + q * H + un) = make_float4(da[q][0], da[q][1], da[q][2], da[q][3]);
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int csship_lstm_bidir_bwd(const float* gates, const float* c, const float* dy, const uint16_t* whhT_packed,
                                     float* dgates, int32_t B, int32_t T, int32_t H, csship_stream_t stream) {
  if (!gates || !c || !dy || !whhT_packed || !dgates) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || H <= 0) return CSSHIP_EBADSHAPE;
  if ((H % 16) != 0) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(gates) || !aligned16(c) || !aligned16(dy) || !aligned16(whhT_packed) || !aligned16(dgates)) return CSSHIP_EALIGN;
  int nw = H / 16, ut = 1;
  if (nw > 8) {
    if (H % 128) return CSSHIP_EUNSUPPORTED;
    nw = 8;
    ut = H / 128;
  }
  LstmBwdArgs a{gates, c, dy, reinterpret_cast<const bf16x8*>(whhT_packed), dgates, B, T, H};
  const size_t lds = 2 * 16 * (size_t)(4 * H * 2 + 16);
  if (lds > 160 * 1024) return CSSHIP_EUNSUPPORTED;
  dim3 grid((B + 15) / 16, 2), block(nw * 64);
  hipStream_t s = as_stream(stream);
#define LSTM_BWD_CASE(U)                                                                                     \
  case U:                                                                                                    \
    if (lds > 64 * 1024 &&                                                                                   \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_kernel<U>),                              \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)             \
      return CSSHIP_ELAUNCH;                                                                                 \
    hipLaunchKernelGGL(lstm_bwd_kernel<U>, grid, block, lds, s, a);                                          \
    break;
  switch (ut) {
    LSTM_BWD_CASE(1) LSTM_BWD_CASE(2) LSTM_BWD_CASE(4)
    default: return CSSHIP_EUNSUPPORTED;
  }
#undef LSTM_BWD_CASE
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

namespace {

// ---- BPTT spread the same way: a wave keeps the W_hh^T rows of its 16 units (16 x 4H bf16 = 64 KB at H = 512) in
// registers; per step the group exchanges the pre-activation gate gradients of the previous step (16 rows x 4H bf16 =
// 64 KB), fetched as MFMA B fragments 16 k-steps at a time, the next 16 in flight under the current MFMAs.
struct LstmBwdDistArgs {
  const float* gates;   // [B, T, 2, 4H] activated
  const float* c;       // [B, T, 2, H]
  const float* dy;      // [B, T, 2H]
  const bf16x8* whhT;   // packed per direction from W_hh^T [H, 4H]: [2][H/16][4H/32][64]
  float* dgates;        // [B, T, 2, 4H]
  unsigned char* dx;    // exchange panels [2 parity][2 dir][tiles][16][4H] bf16
  unsigned* flags;
  unsigned* err;        // sticky, as in LstmDistArgs
  int B, T, tiles, allow_l2;
  unsigned tag0;        // per-launch salt << 16 added to the epoch flags
};

template <int KSH, int AUX>  // AUX as in lstm_dist_body: 16 = through memory, 0 = through the XCD's L2 (co-located group)
DEV void lstm_bwd_dist_body(const LstmBwdDistArgs& a, const int cu, const int tile, const int dir, const int tiles) {
  // one wave per workgroup, H/16 workgroups per group: a CU fetches the panel once (two waves of a workgroup would
  // each fetch all of it), and the hand-off needs no barrier
  constexpr int H = 32 * KSH, NCU = H / 16, KS4 = 4 * KSH;
  const int lane = threadIdx.x, g = lane >> 4, n = lane & 15;
  const int ut = cu, un = 16 * ut + 4 * g;
  const int b0 = tile * 16;
  const int brow = min(b0 + n, a.B - 1);
  const bool bvalid = b0 + n < a.B;

  // W_hh^T fragments live in LDS (each lane re-reads only what it wrote, so no barrier): the registers
  // they would take hold the whole incoming panel instead, so a step costs one memory round trip.
  extern __shared__ bf16x8 wlds[];
  bf16x8* wf = wlds + lane;
  {
    const bf16x8* wbase = a.whhT + (size_t)dir * (H / 16) * KS4 * 64 + lane;
#pragma unroll 8
    for (int ks = 0; ks < KS4; ++ks) wf[ks * 64] = wbase[((size_t)ut * KS4 + ks) * 64];
  }
  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  const unsigned row_bytes = 4u * H * 2u;
  const unsigned group = (unsigned)(dir * tiles + tile) * 16u * row_bytes;
  const unsigned parity_bytes = 2u * (unsigned)tiles * 16u * row_bytes;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.dx, 0, (int)(2u * parity_bytes), 0x00020000);
  unsigned* fl = a.flags + (size_t)(dir * tiles + tile) * NCU * FLAG_STRIDE;

  for (int step = 0; step < a.T; ++step) {
    const int t = dir == 0 ? a.T - 1 - step : step;            // against the forward order
    const int tp = dir == 0 ? t - 1 : t + 1;                   // the forward pass's previous step
    const bool has_prev = tp >= 0 && tp < a.T;
    const size_t srow = ((size_t)brow * a.T + t) * 2 + dir;
    const size_t prow = ((size_t)brow * a.T + (has_prev ? tp : t)) * 2 + dir;
    const float4 dyv = *reinterpret_cast<const float4*>(a.dy + ((size_t)brow * a.T + t) * 2 * H + dir * H + un);
    f32x4 acc = {dyv.x, dyv.y, dyv.z, dyv.w};
    // this step's saved forward values (independent of the hand-off: requested before the wait)
    float4 gq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gq[q] = *reinterpret_cast<const float4*>(a.gates + srow * 4 * H + q * H + un);
    const float4 ct = *reinterpret_cast<const float4*>(a.c + srow * H + un);
    float4 cp = *reinterpret_cast<const float4*>(a.c + prow * H + un);
    if (!has_prev) cp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (step > 0) {
      bool ok = false;
      for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
        const unsigned want = a.tag0 + (unsigned)step;  // flags are zero at launch and only grow: tag0 + 1 .. tag0 + T
        unsigned v = want;
        if (lane < NCU) v = __hip_atomic_load(fl + lane * FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= want && v - a.tag0 <= (unsigned)a.T)) {
          ok = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) {
        if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler ordering only: the loads below are sc1
      // fragment order: the 64 lanes' fragments of one k-step are 1 KB contiguous (one load instruction = 8 whole lines)
      const unsigned pbase = (unsigned)(step & 1) * parity_bytes + group + (unsigned)lane * 16u;
      bf16x8 db[KS4];
#pragma unroll
      for (int i = 0; i < KS4; ++i)
        db[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, pbase + (unsigned)i * 1024u, 0, 16));
      __builtin_amdgcn_sched_barrier(0);  // every load in flight before the first MFMA waits
#pragma unroll
      for (int i = 0; i < KS4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i * 64], db[i], acc, 0, 0, 0);
    }
    const float iv[4] = {gq[0].x, gq[0].y, gq[0].z, gq[0].w}, fv[4] = {gq[1].x, gq[1].y, gq[1].z, gq[1].w};
    const float gv[4] = {gq[2].x, gq[2].y, gq[2].z, gq[2].w}, ov[4] = {gq[3].x, gq[3].y, gq[3].z, gq[3].w};
    const float cv[4] = {ct.x, ct.y, ct.z, ct.w}, pv[4] = {cp.x, cp.y, cp.z, cp.w};
    float da[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dh = acc[r];
      const float tc = tanhf_fast(cv[r]);
      const float dcc = dc[r] + dh * ov[r] * (1.f - tc * tc);
      da[0][r] = dcc * gv[r] * iv[r] * (1.f - iv[r]);
      da[1][r] = dcc * pv[r] * fv[r] * (1.f - fv[r]);
      da[2][r] = dcc * iv[r] * (1.f - gv[r] * gv[r]);
      da[3][r] = dh * tc * ov[r] * (1.f - ov[r]);
      dc[r] = dcc * fv[r];
    }
    if (step + 1 < a.T) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v2u_t dp = {pack_bf2(da[q][0], da[q][1]), pack_bf2(da[q][2], da[q][3])};
        // reduction index k = q*H + un .. +3 of row n: half (k % 8) / 4 of consumer lane (g = (k % 32) / 8, n), k-step k / 32
        const unsigned k = (unsigned)(q * H + un);
        const unsigned off = (unsigned)((step + 1) & 1) * parity_bytes + group + (k >> 5) * 1024u +
                             (((k & 31u) >> 3) * 16u + (unsigned)n) * 16u + ((k >> 2) & 1u) * 8u;
        __builtin_amdgcn_raw_buffer_store_b64(dp, rs, off, 0, AUX);  // sc1: write-through
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        if constexpr (AUX == 16) __hip_atomic_store(fl + cu * FLAG_STRIDE, a.tag0 + (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(fl + cu * FLAG_STRIDE, a.tag0 + (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    // the kernel's own output goes out after the hand-off, off the other workgroups' critical path
    if (bvalid) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(a.dgates + srow * 4 * H + q * H + un) = make_float4(da[q][0], da[q][1], da[q][2], da[q][3]);
    }
  }
}

template <int KSH>
__global__ __launch_bounds__(64) void lstm_bwd_dist_kernel(LstmBwdDistArgs a) {
  constexpr int NCU = 2 * KSH;  // H / 16
  const int tiles = a.tiles, groups = 2 * tiles;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int g = xcd + 8 * (slot / NCU), cu = slot % NCU;
  if (g >= groups) return;
  const int dir = g / tiles, tile = g % tiles;
  const int same = a.allow_l2 ? lstm_group_colocated(a.flags + (size_t)g * NCU * FLAG_STRIDE, cu, NCU, a.err) : 0;
  if (same < 0) return;
  if (same) lstm_bwd_dist_body<KSH, 0>(a, cu, tile, dir, tiles);
  else lstm_bwd_dist_body<KSH, 16>(a, cu, tile, dir, tiles);
}

}  // namespace

#ifdef LSTM_STAMPS
static unsigned* g_lstm_stamps = nullptr;
extern "C" void csship_debug_lstm_stamps(unsigned* buf) { g_lstm_stamps = buf; }
#endif

extern "C" size_t csship_lstm_dist_workspace(int32_t B, int32_t H) {
  if (B <= 0 || (H != 256 && H != 512)) return 0;
  const size_t tiles = (size_t)(B + 15) / 16, ncu = (size_t)H / 16;  // the backward runs H/16 workgroups per group
  const size_t sync_bytes = (2 * tiles * ncu * FLAG_STRIDE + 16) * sizeof(unsigned);  // flags + error word, 64-byte multiple
  return sync_bytes + 2 * 2 * tiles * 16 * (size_t)(4 * H) * 2 + 64;  // the backward's 4H-wide panels + the sticky error line
}

// The error word is STICKY: it lives in the last 64 bytes of the workspace, which no launch ever clears (the caller
// zeroes the workspace once when it allocates it).  A timed-out hand-off in any launch sets it; the host reads and
// clears it at its own synchronisation points.
extern "C" size_t csship_lstm_dist_workspace_error_offset(int32_t B, int32_t H) {
  const size_t n = csship_lstm_dist_workspace(B, H);
  return n ? n - 64 : 0;
}

static unsigned lstm_launch_tag() {  // 1 .. 0x7fff in the high half; eager launches never repeat a recent tag
  static unsigned counter = 0;
  counter = counter % 0x7fffu + 1u;
  return counter << 16;
}

extern "C" int csship_lstm_bidir_dist(const float* gx, const uint16_t* whh_packed, float* out, float* gates_out,
                                      float* c_out, int32_t B, int32_t T, int32_t H, void* workspace,
                                      size_t workspace_bytes, csship_stream_t stream) {
  if (!gx || !whh_packed || !out || !workspace) return CSSHIP_ENULL;
  if ((gates_out == nullptr) != (c_out == nullptr)) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || T > 65534) return CSSHIP_EBADSHAPE;
  if (H != 256 && H != 512) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(gx) || !aligned16(whh_packed) || !aligned16(out) || !aligned16(workspace) ||
      (gates_out && (!aligned16(gates_out) || !aligned16(c_out))))
    return CSSHIP_EALIGN;
  const size_t need = csship_lstm_dist_workspace(B, H);
  if (workspace_bytes < need) return CSSHIP_EWORKSPACE;
  const int tiles = (B + 15) / 16, ncu = H / (16 * LSTM_NW), nslot = H / 16;  // sync region: sized for the backward's H/16 workgroups
  if (tiles > 65535 || 2 * (size_t)tiles * ncu > 2048) return CSSHIP_EBADSHAPE;  // all workgroups must be resident together
  const size_t sync_bytes = (2 * (size_t)tiles * nslot * FLAG_STRIDE + 16) * sizeof(unsigned);
  hipStream_t s = as_stream(stream);
  // error word + this call's tagged panels (2 parities x 2 directions x tiles x 16 rows x 4H bytes): epoch tags
  // of an earlier call must not be mistaken for this one's
  const size_t panel_bytes = 2 * 2 * (size_t)tiles * 16 * (size_t)(4 * H);
  if (hipMemsetAsync(workspace, 0, sync_bytes + panel_bytes, s) != hipSuccess) return CSSHIP_ELAUNCH;
  LstmDistArgs a{};
  a.gx = gx; a.whh = reinterpret_cast<const bf16x8*>(whh_packed); a.out = out; a.gates_out = gates_out; a.c_out = c_out;
  a.flags = reinterpret_cast<unsigned*>(workspace);
  a.err = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(workspace) + csship_lstm_dist_workspace_error_offset(B, H));
  a.hx = reinterpret_cast<unsigned char*>(workspace) + sync_bytes;
  a.B = B; a.T = T; a.tiles = tiles;
  a.tag0 = lstm_launch_tag();
  static const bool l2_off = [] { const char* e = getenv("CSSHIP_LSTM_L2"); return e && e[0] == '0'; }();
  a.allow_l2 = l2_off ? 0 : 1;
#ifdef LSTM_STAMPS
  a.stamps = g_lstm_stamps;
#endif
  dim3 grid(8 * ncu * ((2 * tiles + 7) / 8));  // group g = ids congruent to g mod 8 (see lstm_group_colocated)
  if (H == 512) hipLaunchKernelGGL(lstm_dist_kernel<16>, grid, dim3(64 * LSTM_NW), 0, s, a);
  else hipLaunchKernelGGL(lstm_dist_kernel<8>, grid, dim3(64 * LSTM_NW), 0, s, a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_lstm_bidir_bwd_dist(const float* gates, const float* c, const float* dy, const uint16_t* whhT_packed,
                                          float* dgates, int32_t B, int32_t T, int32_t H, void* workspace,
                                          size_t workspace_bytes, csship_stream_t stream) {
  if (!gates || !c || !dy || !whhT_packed || !dgates || !workspace) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || T > 65534) return CSSHIP_EBADSHAPE;
  if (H != 256 && H != 512) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(gates) || !aligned16(c) || !aligned16(dy) || !aligned16(whhT_packed) || !aligned16(dgates) ||
      !aligned16(workspace))
    return CSSHIP_EALIGN;
  if (workspace_bytes < csship_lstm_dist_workspace(B, H)) return CSSHIP_EWORKSPACE;
  const int tiles = (B + 15) / 16, ncu = H / 16;  // one-wave workgroups of 16 hidden units
  if (tiles > 65535 || 2 * (size_t)tiles * ncu > 4096) return CSSHIP_EBADSHAPE;
  const size_t sync_bytes = (2 * (size_t)tiles * ncu * FLAG_STRIDE + 16) * sizeof(unsigned);
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(workspace, 0, sync_bytes, s) != hipSuccess) return CSSHIP_ELAUNCH;
  LstmBwdDistArgs a{};
  a.gates = gates; a.c = c; a.dy = dy; a.whhT = reinterpret_cast<const bf16x8*>(whhT_packed); a.dgates = dgates;
  a.flags = reinterpret_cast<unsigned*>(workspace);
  a.err = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(workspace) + csship_lstm_dist_workspace_error_offset(B, H));
  a.dx = reinterpret_cast<unsigned char*>(workspace) + sync_bytes;
  a.B = B; a.T = T; a.tiles = tiles;
  a.tag0 = lstm_launch_tag();
  static const bool l2_off = [] { const char* e = getenv("CSSHIP_LSTM_L2"); return e && e[0] == '0'; }();
  a.allow_l2 = l2_off ? 0 : 1;
  dim3 grid(8 * ncu * ((2 * tiles + 7) / 8));  // group g = ids congruent to g mod 8 (see lstm_group_colocated)
  const size_t lds = (size_t)(H / 8) * 64 * 16;  // the wave's W_hh^T fragments (64 KB at H = 512)
  if (H == 512) hipLaunchKernelGGL(lstm_bwd_dist_kernel<16>, grid, dim3(64), lds, s, a);
  else hipLaunchKernelGGL(lstm_bwd_dist_kernel<8>, grid, dim3(64), lds, s, a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

static int lstm_forward(const float* gx, const uint16_t* whh_packed, float* out, float* gates_out, float* c_out,
                        int32_t B, int32_t T, int32_t H, csship_stream_t stream);

extern "C" int csship_lstm_bidir(const float* gx, const uint16_t* whh_packed, float* out, int32_t B, int32_t T,
                                 int32_t H, csship_stream_t stream) {
  return lstm_forward(gx, whh_packed, out, nullptr, nullptr, B, T, H, stream);
}

extern "C" int csship_lstm_bidir_train(const float* gx, const uint16_t* whh_packed, float* out, float* gates_out,
                                       float* c_out, int32_t B, int32_t T, int32_t H, csship_stream_t stream) {
  if (!gates_out || !c_out) return CSSHIP_ENULL;
  if (!aligned16(gates_out) || !aligned16(c_out)) return CSSHIP_EALIGN;
  return lstm_forward(gx, whh_packed, out, gates_out, c_out, B, T, H, stream);
}

static int lstm_forward(const float* gx, const uint16_t* whh_packed, float* out, float* gates_out, float* c_out,
                        int32_t B, int32_t T, int32_t H, csship_stream_t stream) {
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
  LstmArgs a{gx, reinterpret_cast<const bf16x8*>(whh_packed), out, gates_out, c_out, B, T, H, Hp, nw};
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
