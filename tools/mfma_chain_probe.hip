// Does the order in which a wave visits its accumulator tiles matter?  16 v_mfma_f32_32x32x2_f32 per iteration on four
// accumulator tiles; CH consecutive MFMAs go to the same tile (CH = 1: round robin, 4: the Winograd kernel's old step order,
// 16: one tile only).  Optionally one ds_read_b32 (immediate offset) per MFMA.  1 / 2 / 3 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_chain_probe.hip -o tools/mfma_chain_probe && tools/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH, int NL>
__global__ void __launch_bounds__(256) chain(float* out, const float* rnd, int iters) {
  __shared__ float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = rnd[i];
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float a = 0.25f + threadIdx.x * 1e-6f;
  float b = 0.5f;
  unsigned laddr = (unsigned)(size_t)sm + (threadIdx.x & 63) * 4;
  float r[2] = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      constexpr int dummy = 0; (void)dummy;
      const int t = (m / CH) & 3;
      if (NL) {
        // fragment for MFMA m+1 requested before MFMA m issues; wait leaves that one request in flight
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[(m + 1) & 1]) : "v"(laddr), "n"((m * 256) & 0x3fff));
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r[m & 1]));
        b = r[m & 1];
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
}

template <int CH, int NL>
void run(int wps, int iters) {
  float *out, *rnd;
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&rnd, 4096 * 4);
  (void)hipMemset(rnd, 0, 4096 * 4);
  const int blocks = 256 * wps;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  chain<CH, NL><<<blocks, 256>>>(out, rnd, iters);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    chain<CH, NL><<<blocks, 256>>>(out, rnd, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double flops = (double)blocks * 4 * iters * 16.0 * 4096.0;
  const double cyc = best * 1e-3 * 2.4e9 / ((double)iters * 16.0 * wps);
  printf("chain %2d  ds_read/MFMA %d  %d waves/SIMD: %7.1f TFLOP/s  %5.1f cycles per MFMA per SIMD\n", CH, NL, wps, flops / best * 1e-9, cyc);
  (void)hipFree(out); (void)hipFree(rnd);
}

int main() {
  const int it = 20000;
  for (int w = 1; w <= 3; ++w) {
    run<1, 0>(w, it); run<2, 0>(w, it); run<4, 0>(w, it); run<16, 0>(w, it);
    run<1, 1>(w, it); run<2, 1>(w, it); run<4, 1>(w, it); run<16, 1>(w, it);
  }
  return 0;
}
