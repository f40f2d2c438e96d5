// What v_permlane16_swap_b32 / v_permlane32_swap_b32 do on gfx950 (lane -> value maps), for the store-widening epilogues.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/permlane.hip -o build/permlane && build/permlane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  unsigned a = 1000 + lane, b = 2000 + lane;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
  auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + lane] = q[0];
  out[192 + lane] = q[1];
}
int main() {
  unsigned* d;
  unsigned h[256];
  (void)hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap: new a", "permlane16_swap: new b", "permlane32_swap: new a", "permlane32_swap: new b"};
  for (int v = 0; v < 4; ++v) {
    printf("%s (a = 1000 + lane, b = 2000 + lane), first lane of each 16-lane row:", names[v]);
    for (int r = 0; r < 4; ++r) printf("  row %d: %u", r, h[v * 64 + r * 16]);
    printf("\n");
  }
  return 0;
}
