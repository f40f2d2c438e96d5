// Round-trip latency of a two-workgroup hand-off through global memory, by cache-coherence mode and placement.
//   hipcc --offload-arch=gfx950 -O3 -o build/pingpong tools/micro/pingpong.hip && build/pingpong
// Workgroup 0 and workgroup PEER bounce an epoch counter ROUNDS times: each waits (bounded) for the other's slot to
// reach the epoch, then stores epoch + 1 into its own.  PEER = 8 lands on the same XCD as workgroup 0 when workgroups
// are dealt round-robin over 8 XCDs (checked: both report HW_REG_XCC_ID), PEER = 1 on another one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int ROUNDS = 200;
constexpr unsigned SPIN = 1u << 16;

template <int LD, int ST, int INV>  // aux bits of the load / store (1 = sc0, 16 = sc1), INV: 0 none, 1 buffer_inv sc0, 2 buffer_inv sc1
__global__ void pingpong(unsigned* slots, int peer, unsigned long long* out) {
  const int me = blockIdx.x;
  if (me != 0 && me != peer) return;
  if (threadIdx.x != 0) return;
  const int a = me == 0 ? 0 : 1;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slots, 0, 4096, 0x00020000);
  const unsigned mine = a * 256u, other = (1 - a) * 256u;  // separate 256-byte regions
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned fails = 0;
  for (unsigned r = 1; r <= ROUNDS; ++r) {
    if (a == 0) __builtin_amdgcn_raw_buffer_store_b32(r, rs, mine, 0, ST);
    bool ok = false;
    for (unsigned s = 0; s < SPIN; ++s) {
      if (INV == 1) asm volatile("buffer_inv sc0" ::: "memory");
      if (INV == 2) asm volatile("buffer_inv sc1" ::: "memory");
      asm volatile("" ::: "memory");  // the load below is re-issued every iteration
      const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, other, 0, LD);
      if (v >= r) { ok = true; break; }
    }
    if (!ok) { fails = r; break; }
    if (a == 1) __builtin_amdgcn_raw_buffer_store_b32(r, rs, mine, 0, ST);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  out[a * 4 + 0] = t1 - t0;
  out[a * 4 + 1] = fails;
  out[a * 4 + 2] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
}

template <int LD, int ST, int INV>
void run(const char* name, unsigned* slots, unsigned long long* out) {
  for (int peer : {8, 1}) {
    hipMemset(slots, 0, 4096);
    hipMemset(out, 0, 64);
    hipLaunchKernelGGL((pingpong<LD, ST, INV>), dim3(16), dim3(64), 0, 0, slots, peer, out);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    // s_memrealtime ticks at 100 MHz
    printf("%-34s peer %d  xcc %llu/%llu  %s  %.2f us per round trip\n", name, peer, h[2], h[6],
           (h[1] || h[5]) ? "TIMEOUT" : "ok     ", (double)h[0] * 10.0 / 1000.0 / ROUNDS);
  }
}

int main() {
  unsigned* slots; unsigned long long* out;
  hipMalloc(&slots, 4096); hipMalloc(&out, 64);
  run<16, 16, 0>("ld sc1, st sc1", slots, out);
  run<17, 17, 0>("ld sc0 sc1, st sc0 sc1", slots, out);
  run<1, 1, 0>("ld sc0, st sc0", slots, out);
  run<1, 0, 0>("ld sc0, st plain", slots, out);
  run<0, 0, 1>("ld plain + inv sc0, st plain", slots, out);
  run<0, 0, 2>("ld plain + inv sc1, st plain", slots, out);
  run<0, 16, 2>("ld plain + inv sc1, st sc1", slots, out);
  run<1, 16, 0>("ld sc0, st sc1", slots, out);
  run<16, 0, 0>("ld sc1, st plain", slots, out);
  run<16, 1, 0>("ld sc1, st sc0", slots, out);
  return 0;
}
