// Do fp32 MFMAs and other work overlap on one SIMD?  v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD),
// so the question for the wave-specialised kernels is whether a producer wave's VALU / LDS instructions make progress while a
// consumer wave on the same SIMD keeps the matrix pipe saturated.  Workgroup = 8 waves, one per CU: waves 0-3 issue NM dependent
// MFMAs (4 per accumulator, as the Winograd consumers do), waves 4-7 issue NV instructions of one kind.  Modes: MFMA waves
// alone, other waves alone, both.  Reported in shader cycles of wave 0 / wave 4 (s_memtime), so DVFS does not matter.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_probe.hip -o tools/mfma_valu_probe && tools/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND: 0 = v_fma_f32 (independent), 1 = v_max_f32, 2 = ds_write_b32, 3 = ds_read_b32, 4 = v_mov_b32, 5 = v_pk_mul_f32
template <int KIND>
__global__ void __launch_bounds__(512) probe(long long* out, int do_mfma, int do_other, int iters, int self_valu) {
  __shared__ float sm[8192];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 8192; i += 512) sm[i] = 0.001f * i;
  __syncthreads();
  const long long c0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (do_mfma) {
      f32x16 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
      const float a = 0.25f + tid * 1e-6f, b = 0.5f;
      float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          acc[(m >> 2) & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(m >> 2) & 3], 0, 0, 0);
          if (self_valu > 0) {                     // VALU of the SAME wave between its MFMAs
#pragma unroll 1
            for (int q = 0; q < 1; ++q) {
              if (self_valu >= 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(a), "v"(b));
              if (self_valu >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(a), "v"(b));
              if (self_valu >= 4) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(a), "v"(b)); }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      float s = v0 + v1 + v2 + v3;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
      if (s == 12345.678f) out[1000] = (long long)s;
    }
  } else if (do_other) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.5f + j + tid * 1e-6f;
    const float a = 1.0001f, b = 1e-7f;
    unsigned laddr = (unsigned)(size_t)sm + (tid & 63) * 4 + (wave - 4) * 4096;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[m & 7]) : "v"(a), "v"(b));
        if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[m & 7]) : "v"(a));
        if (KIND == 2) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(laddr), "v"(v[m & 7]), "n"((m * 256) & 0xfff) : "memory");
        if (KIND == 3) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[m & 7]) : "v"(laddr), "n"((m * 256) & 0xfff) : "memory");
        if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(v[m & 7]) : "v"(a));
        if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[(2 * m) & 6])) : "v"(*reinterpret_cast<const double*>(&v[0])));
      }
      if (KIND == 2 || KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 12345.678f) out[1001] = (long long)s;
  }
  const long long c1 = __builtin_readcyclecounter();
  if ((tid & 63) == 0 && blockIdx.x == 0) out[wave] = c1 - c0;
}

template <int KIND>
void run(const char* name, int iters_m, int iters_o) {
  long long* out;
  (void)hipMalloc(&out, 2048 * 8);
  long long h[8];
  auto go = [&](int m, int o, int im, int sv) {
    (void)hipMemset(out, 0, 2048 * 8);
    probe<KIND><<<256, 512>>>(out, m, o, im, sv);
    probe<KIND><<<256, 512>>>(out, m, o, im, sv);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
  };
  (void)iters_o;
  go(1, 0, iters_m, 0); const double tm = (double)h[0];
  go(0, 1, iters_m, 0); const double to = (double)h[4];
  go(1, 1, iters_m, 0); const double tbm = (double)h[0], tbo = (double)h[4];
  printf("%-12s MFMA waves alone %8.0f cyc (%.1f / MFMA) | other waves alone %8.0f cyc (%.2f / instr) | together: MFMA waves %8.0f (%.1f / MFMA), other waves %8.0f  -> sum %.0f, max %.0f\n",
         name, tm, tm / (16.0 * iters_m), to, to / (16.0 * iters_m), tbm, tbm / (16.0 * iters_m), tbo, tm + to, tm > to ? tm : to);
  (void)hipFree(out);
}

int main() {
  const int it = 4000;
  run<0>("v_fma_f32", it, it);
  run<1>("v_max_f32", it, it);
  run<4>("v_mov_b32", it, it);
  run<5>("v_pk_mul_f32", it, it);
  run<2>("ds_write_b32", it, it);
  run<3>("ds_read_b32", it, it);
  // VALU of the same wave between its own MFMAs
  long long* out; (void)hipMalloc(&out, 2048 * 8);
  for (int sv : {0, 1, 2, 4}) {
    long long h[8];
    probe<0><<<256, 512>>>(out, 1, 0, it, sv);
    probe<0><<<256, 512>>>(out, 1, 0, it, sv);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("same wave: %d v_fma_f32 per MFMA: %.1f cycles per MFMA\n", sv, (double)h[0] / (16.0 * it));
  }
  return 0;
}
