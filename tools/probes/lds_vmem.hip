// Micro-benchmarks behind the design of csship_attn_dvdk (css_amd/csrc/attnbwd.hip); not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/probes/lds_vmem.hip -o build/lds_vmem && build/lds_vmem
// 1. rowseg: the kernel's global access pattern alone (row segments of two [B*H][T][T] bf16 arrays, two workgroups per
//    matrix) - what streaming P and dS costs with no other work.  MI355X: 16.5 us = 5.0 TB/s for 128 x 400 x 400.
// 2. probe<G, L>: do LDS reads and global loads share a path into the VGPRs?  One workgroup of 512 threads per CU; per
//    iteration every wave issues G global_load_dwordx4 (streaming, each workgroup its own slice) and L LDS reads
//    (ds_read_b64 or ds_read_b128, linear = conflict free), then waits for all of them.  MI355X: no - 5 loads alone
//    1436 ns per iteration (7.3 TB/s), with 24 ds_read_b64 1511 ns; the 24 reads alone 240 ns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int G, int L, bool B128>
__global__ __launch_bounds__(512) void probe(const f32x4* __restrict__ src, size_t slice_vec, int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  const int tid = threadIdx.x;
  for (int i = tid; i < 65536 / 16; i += 512) reinterpret_cast<f32x4*>(smem)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const f32x4* base = src + (size_t)blockIdx.x * slice_vec;
  float acc = 0.f;
  size_t pos = tid;
  for (int it = 0; it < iters; ++it) {
    f32x4 gv[G > 0 ? G : 1];
    f32x4 lv[L > 0 ? L : 1];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      gv[u] = base[pos];
      pos += 512;
      if (pos >= slice_vec) pos -= slice_vec;
    }
#pragma unroll
    for (int u = 0; u < L; ++u) {
      if constexpr (B128) {
        lv[u] = *reinterpret_cast<const f32x4*>(smem + ((tid * 16 + u * 8192 + it * 512) & 65535 & ~15));
      } else {
        const f32x2 t = *reinterpret_cast<const f32x2*>(smem + ((tid * 8 + u * 4096 + it * 512) & 65535 & ~7));
        lv[u] = f32x4{t[0], t[1], 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) asm volatile("" ::"v"(gv[u]));
#pragma unroll
    for (int u = 0; u < L; ++u) asm volatile("" ::"v"(lv[u]));
    if (it == iters - 1) {
#pragma unroll
      for (int u = 0; u < G; ++u) acc += gv[u][0];
#pragma unroll
      for (int u = 0; u < L; ++u) acc += lv[u][0];
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int G, int L, bool B128>
void run(const f32x4* src, size_t slice_vec, float* sink, int nwg) {
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<G, L, B128><<<nwg, 512>>>(src, slice_vec, iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) probe<G, L, B128><<<nwg, 512>>>(src, slice_vec, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 5;
  const double gbytes = (double)nwg * iters * G * 512 * 16, lbytes = (double)nwg * iters * L * 512 * (B128 ? 16 : 8);
  printf("G=%d L=%2d %s: %8.1f us  | per iteration %6.1f ns | global %6.2f TB/s (%5.1f B/ns/CU) | LDS %7.2f TB/s (%6.1f B/ns/CU)\n", G, L,
         B128 ? "b128" : "b64 ", us, us * 1e3 / iters, gbytes / us * 1e-6, gbytes / nwg / (us * 1e3), lbytes / us * 1e-6,
         lbytes / nwg / (us * 1e3));
}

// Row-segment streaming as csship_attn_dvdk does it: `nmat` matrices of T rows x `stride` bytes in each of two arrays, two
// workgroups per matrix (column halves of `seg` bytes), 32 rows per step, thread -> (row tid >> 4, 16-byte chunks tid & 15
// and 16 + (tid & 15)), DEPTH steps in flight.
template <int DEPTH>
__global__ __launch_bounds__(512) void rowseg(const unsigned char* __restrict__ pa, const unsigned char* __restrict__ pb, int T,
                                              size_t stride, int seg, float* sink) {
  const int tid = threadIdx.x, row = tid >> 4, ch = tid & 15;
  const int mat = blockIdx.x >> 1, half = blockIdx.x & 1;
  const size_t base = (size_t)mat * T * stride + (size_t)half * seg;
  const int c0 = ch * 16 < seg ? ch * 16 : 0, c1 = (ch + 16) * 16 < seg ? (ch + 16) * 16 : 0;
  f32x4 r[DEPTH][4];
  const int nblk = (T + 31) / 32;
  auto load = [&](int j, f32x4 (&v)[4]) {
    const size_t o = base + (size_t)min(32 * j + row, T - 1) * stride;
    v[0] = *reinterpret_cast<const f32x4*>(pa + o + c0);
    v[1] = *reinterpret_cast<const f32x4*>(pb + o + c0);
    v[2] = *reinterpret_cast<const f32x4*>(pa + o + c1);
    v[3] = *reinterpret_cast<const f32x4*>(pb + o + c1);
  };
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(d, r[d]);
  for (int j = 0; j < nblk; j += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      asm volatile("" ::"v"(r[d][0]), "v"(r[d][1]), "v"(r[d][2]), "v"(r[d][3]));
      acc += r[d][0][0];
      load(j + d + DEPTH, r[d]);
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int DEPTH>
void run_rowseg(const unsigned char* pa, const unsigned char* pb, int nmat, int T, size_t stride, int seg, float* sink, int nsets) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t set = (size_t)nmat * T * stride;
  rowseg<DEPTH><<<2 * nmat, 512>>>(pa, pb, T, stride, seg, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 40;
  for (int r = 0; r < reps; ++r) rowseg<DEPTH><<<2 * nmat, 512>>>(pa + (r % nsets) * set, pb + (r % nsets) * set, T, stride, seg, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("rowseg depth %d: %d x %d rows x %zu B, halves of %d B, %d sets: %7.1f us = %5.2f TB/s\n", DEPTH, nmat, T, stride, seg, nsets, us,
         2.0 * nmat * T * 2 * seg / us * 1e-6);
}

int main() {
  {
    const int nmat = 128, T = 400;
    unsigned char *pa, *pb;
    float* sink;
    const size_t cap = (size_t)nmat * 512 * 1024 * 4;   // up to 4 sets of 1024-byte rows
    hipMalloc(&pa, cap);
    hipMalloc(&pb, cap);
    hipMalloc(&sink, 4);
    hipMemset(pa, 0, cap);
    hipMemset(pb, 0, cap);
    run_rowseg<1>(pa, pb, nmat, T, 800, 400, sink, 4);
    run_rowseg<2>(pa, pb, nmat, T, 800, 400, sink, 4);
    run_rowseg<4>(pa, pb, nmat, T, 800, 400, sink, 4);
    run_rowseg<4>(pa, pb, nmat, T, 800, 400, sink, 1);
    run_rowseg<4>(pa, pb, nmat, T, 1024, 512, sink, 4);
    run_rowseg<4>(pa, pb, nmat, 512, 1024, 512, sink, 4);
    run_rowseg<4>(pa, pb, nmat, 512, 1024, 512, sink, 1);
    hipFree(pa);
    hipFree(pb);
  }
  const int nwg = 256;
  const size_t slice_bytes = 256 << 10;           // 64 MB in all: beyond the L2s, inside the Infinity Cache
  const size_t slice_vec = slice_bytes / 16;
  f32x4* src;
  float* sink;
  hipMalloc(&src, slice_bytes * nwg);
  hipMalloc(&sink, 4);
  hipMemset(src, 0, slice_bytes * nwg);
  run<5, 0, false>(src, slice_vec, sink, nwg);
  run<0, 24, false>(src, slice_vec, sink, nwg);
  run<5, 24, false>(src, slice_vec, sink, nwg);
  run<5, 12, false>(src, slice_vec, sink, nwg);
  run<5, 12, true>(src, slice_vec, sink, nwg);
  run<0, 12, true>(src, slice_vec, sink, nwg);
  run<5, 6, true>(src, slice_vec, sink, nwg);
  run<8, 0, false>(src, slice_vec, sink, nwg);
  run<8, 24, false>(src, slice_vec, sink, nwg);
  run<2, 24, false>(src, slice_vec, sink, nwg);
  return 0;
}
