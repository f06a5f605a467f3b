// Micro-benchmark: what one CU's memory pipe sustains (B/clk) for the two access patterns of the step kernel.
//   mode 0: LDS-DMA stream of a private region into an LDS ring, `depth` slots of `slot_kb` KiB in flight
//   mode 1: Adam-like register stream: load 3 arrays (16 B/lane), fma, store 3 arrays; `depth` tiles of 1 KiB x 3 in flight per wave
// Each workgroup (512 threads) walks its own region of `region_kb` KiB `reps` times.  Prints bytes / cycle / CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define GAS __attribute__((address_space(1)))
typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int DEPTH>
__global__ __launch_bounds__(512) void k_dma(const char* base, int region_kb, int slot_kb, int reps, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const GAS char* src = (const GAS char*)base + (size_t)blockIdx.x * region_kb * 1024;
  const int nslots = region_kb / slot_kb;
  auto issue = [&](int s) {
    const GAS char* p = src + (size_t)(s % nslots) * slot_kb * 1024;
    char* d = smem + (s % DEPTH) * slot_kb * 1024;
    for (int q = wave; q < slot_kb; q += 8)
      __builtin_amdgcn_global_load_lds((const GAS void*)(p + q * 1024 + lane * 16), (lds_vp)(d + q * 1024), 16, 0, 0);
  };
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int total = nslots * reps;
  for (int s = 0; s < DEPTH - 1 && s < total; ++s) issue(s);
  float acc = 0.f;
  for (int s = 0; s < total; ++s) {
    if (s + DEPTH - 1 < total) issue(s + DEPTH - 1);
    // wait for slot s: everything but the youngest (DEPTH-1) slots' instructions of this wave
    const int per = (slot_kb - wave + 7) / 8;
    const int young = (s + DEPTH - 1 < total ? DEPTH - 1 : total - 1 - s) * per;
    if (young <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (young <= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // conservative buckets
    else if (young <= 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (young <= 16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc += ((float*)(smem + (s % DEPTH) * slot_kb * 1024))[t];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 512 + t] = acc;
}

// IL: p / m / v of a tile adjacent in memory ([tile][3][256 floats]: one 3-KiB burst per tile) instead of three arrays
template <int DEPTH, bool NT, bool IL = false>
__global__ __launch_bounds__(512) void k_adam(float* base, int region_kb, int reps, unsigned long long* cyc) {
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // region = 3 arrays of region_kb/3 KiB; tiles of 1 KiB per array; wave w takes tiles w, w+8, ...
  const int tiles = region_kb / 3;
  GAS float* P = (GAS float*)base + (size_t)blockIdx.x * region_kb * 256;
  GAS float* M = P + (IL ? (size_t)256 : (size_t)tiles * 256);
  GAS float* V = M + (IL ? (size_t)256 : (size_t)tiles * 256);
  constexpr int TS = IL ? 768 : 256;                 // floats between consecutive tiles of one array
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    f32x4 p[DEPTH], m[DEPTH], v[DEPTH];
    int tl = wave;
    auto req = [&](int i, int tile) {
      const size_t o = (size_t)tile * TS + lane * 4;
      if (NT) { p[i] = __builtin_nontemporal_load((const GAS f32x4*)(P + o)); m[i] = __builtin_nontemporal_load((const GAS f32x4*)(M + o)); v[i] = __builtin_nontemporal_load((const GAS f32x4*)(V + o)); }
      else { p[i] = *(const GAS f32x4*)(P + o); m[i] = *(const GAS f32x4*)(M + o); v[i] = *(const GAS f32x4*)(V + o); }
    };
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i) if (tl + 8 * i < tiles) req(i, tl + 8 * i);
    for (; tl < tiles; tl += 8 * DEPTH) {
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) {
        const int cur = tl + 8 * i;
        if (cur >= tiles) break;
        const int nx = cur + 8 * (DEPTH - 1);
        if (nx < tiles) req((i + DEPTH - 1) % DEPTH, nx);
        const size_t o = (size_t)cur * TS + lane * 4;
        f32x4 pp = p[i], mm = m[i], vv = v[i];
        mm = mm * 0.9f + 0.1f; vv = vv * 0.999f + 0.001f; pp = pp - 1e-4f * mm;
        if (NT) { __builtin_nontemporal_store(pp, (GAS f32x4*)(P + o)); __builtin_nontemporal_store(mm, (GAS f32x4*)(M + o)); __builtin_nontemporal_store(vv, (GAS f32x4*)(V + o)); }
        else { *(GAS f32x4*)(P + o) = pp; *(GAS f32x4*)(M + o) = mm; *(GAS f32x4*)(V + o) = vv; }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

static double med(unsigned long long* c, int n) {
  unsigned long long* a = (unsigned long long*)malloc(n * 8);
  for (int i = 0; i < n; ++i) a[i] = c[i];
  for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) if (a[j] < a[i]) { unsigned long long x = a[i]; a[i] = a[j]; a[j] = x; }
  double r = (double)a[n / 2]; free(a); return r;
}

int main() {
  const int maxwg = 256;
  const size_t region_kb = 4320;            // 4.2 MiB per workgroup (~ p/m/v of one SE-3 model)
  char* buf; hipMalloc(&buf, (size_t)maxwg * region_kb * 1024);
  hipMemset(buf, 0, (size_t)maxwg * region_kb * 1024);
  unsigned long long* cyc; hipMalloc(&cyc, maxwg * 8);
  float* sink; hipMalloc(&sink, maxwg * 512 * 4);
  unsigned long long h[maxwg];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {1, 256}) {
    for (int rk : {432, 4320}) {              // region per WG: 432 KiB (cache resident when alone) / 4.2 MiB
      const int reps = rk == 432 ? 40 : 4;
#define RUN_DMA(D, SLOT) { \
      hipFuncSetAttribute((const void*)k_dma<D>, hipFuncAttributeMaxDynamicSharedMemorySize, D * SLOT * 1024); \
      hipLaunchKernelGGL(k_dma<D>, dim3(wgs), dim3(512), D * SLOT * 1024, 0, buf, rk, SLOT, reps, cyc, sink); hipDeviceSynchronize(); \
      hipEventRecord(e0); hipLaunchKernelGGL(k_dma<D>, dim3(wgs), dim3(512), D * SLOT * 1024, 0, buf, rk, SLOT, reps, cyc, sink); hipEventRecord(e1); hipDeviceSynchronize(); \
      float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, wgs * 8, hipMemcpyDeviceToHost); \
      printf("dma   wgs %3d region %4d KiB slot %2d KiB depth %d : %6.2f B/clk/CU  (%.1f GB/s/CU, %.2f TB/s chip)\n", wgs, rk, SLOT, D, \
             (double)rk * 1024 * reps / med(h, wgs), (double)rk * 1024 * reps / (ms * 1e-3) / 1e9, (double)rk * 1024 * reps * wgs / (ms * 1e-3) / 1e12); }
      RUN_DMA(2, 36) RUN_DMA(3, 36) RUN_DMA(4, 36) RUN_DMA(2, 72) RUN_DMA(8, 18)
#define RUN_ADAM(D, NT) { \
      hipLaunchKernelGGL((k_adam<D, NT>), dim3(wgs), dim3(512), 0, 0, (float*)buf, rk, reps, cyc); hipDeviceSynchronize(); \
      hipEventRecord(e0); hipLaunchKernelGGL((k_adam<D, NT>), dim3(wgs), dim3(512), 0, 0, (float*)buf, rk, reps, cyc); hipEventRecord(e1); hipDeviceSynchronize(); \
      float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, wgs * 8, hipMemcpyDeviceToHost); \
      printf("adam  wgs %3d region %4d KiB depth %d nt %d : %6.2f B/clk/CU r+w (%.1f GB/s/CU, %.2f TB/s chip)\n", wgs, rk, D, NT, \
             2.0 * rk * 1024 * reps / med(h, wgs), 2.0 * rk * 1024 * reps / (ms * 1e-3) / 1e9, 2.0 * rk * 1024 * reps * wgs / (ms * 1e-3) / 1e12); }
      RUN_ADAM(1, false) RUN_ADAM(2, false) RUN_ADAM(2, true) RUN_ADAM(4, true) RUN_ADAM(8, true)
#define RUN_ADAM_IL(D, NT) { \
      hipLaunchKernelGGL((k_adam<D, NT, true>), dim3(wgs), dim3(512), 0, 0, (float*)buf, rk, reps, cyc); hipDeviceSynchronize(); \
      hipEventRecord(e0); hipLaunchKernelGGL((k_adam<D, NT, true>), dim3(wgs), dim3(512), 0, 0, (float*)buf, rk, reps, cyc); hipEventRecord(e1); hipDeviceSynchronize(); \
      float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, wgs * 8, hipMemcpyDeviceToHost); \
      printf("adamIL wgs %3d region %4d KiB depth %d nt %d : %6.2f B/clk/CU r+w (%.1f GB/s/CU, %.2f TB/s chip)\n", wgs, rk, D, NT, \
             2.0 * rk * 1024 * reps / med(h, wgs), 2.0 * rk * 1024 * reps / (ms * 1e-3) / 1e9, 2.0 * rk * 1024 * reps * wgs / (ms * 1e-3) / 1e12); }
      RUN_ADAM_IL(1, false) RUN_ADAM_IL(2, false) RUN_ADAM_IL(2, true) RUN_ADAM_IL(4, true)
    }
  }
  return 0;
}
