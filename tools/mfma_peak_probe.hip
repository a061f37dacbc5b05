// Sustained rate of v_mfma_f32_32x32x2_f32 with no memory traffic at all: the practical ceiling for the fp32
// implicit-GEMM kernels (the data-sheet number assumes the peak engine clock).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak_probe.hip -o tools/mfma_peak_probe && tools/mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
}

// same loop, operands are per-lane random values (8 a's and 8 b's cycled): realistic switching activity
template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop_rand(float* out, const float* rnd, int iters) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a[8], b[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { a[r] = rnd[(threadIdx.x * 16 + r) & 4095]; b[r] = rnd[(threadIdx.x * 16 + 8 + r + blockIdx.x) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[(r + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
void run_rand(int blocks, int threads, int iters, const char* what) {
  float *out, *rnd;
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&rnd, 4096 * 4);
  float h[4096];
  unsigned x = 12345;
  for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) / 8388608.0f - 1.0f) * 0.9f; }
  (void)hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  mfma_loop_rand<NACC><<<blocks, threads>>>(out, rnd, iters / 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  mfma_loop_rand<NACC><<<blocks, threads>>>(out, rnd, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64;
  const double flops = waves * iters * 8.0 * NACC * (2.0 * 32 * 32 * 2);
  printf("%-44s blocks %5d x %4d thr, %d acc tiles: %8.3f ms  %7.1f TFLOP/s\n", what, blocks, threads, NACC, ms, flops / ms * 1e-9);
  (void)hipFree(out); (void)hipFree(rnd);
}

// MFMA stream with the operand traffic of the convolution kernels added: per 16 MFMAs (4 k-steps x 4 accumulator
// tiles) LDSR ds_read_b32 (activation fragments) and GLD 16-byte global loads that hit in L2 (weight fragments).
// Reports the rate and the shader clock (s_memtime ticks per 100 MHz s_memrealtime tick).
template <int LDSR, int GLD, int HB = 0>
__global__ void __launch_bounds__(256) mfma_traffic(float* out, const float* rnd, int iters, long long* clk, float4* big = nullptr, long long bigN = 0) {
  __shared__ float sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = rnd[i & 4095];
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float4* g4 = reinterpret_cast<const float4*>(rnd);
  float bv[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
  float4 av[4] = {make_float4(.1f, .2f, .3f, .4f), make_float4(.1f, .2f, .3f, .4f), make_float4(.1f, .2f, .3f, .4f), make_float4(.1f, .2f, .3f, .4f)};
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  int off = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    // issue the next group's operand requests, then 16 MFMAs on the current ones
    float bn[8]; float4 an[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) bn[i] = bv[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) an[i] = av[i];
#pragma unroll
    for (int i = 0; i < LDSR; ++i) bn[i & 7] = sm[(off + i * 184) & 8191];
#pragma unroll
    for (int i = 0; i < GLD; ++i) an[i & 3] = g4[(off + i * 64) & 1023];
    off += 33;
    if constexpr (HB > 0) if ((it % HB) == 0) {   // streaming HBM traffic: 16 B read + 16 B written per lane
      const long long gi = (((long long)blockIdx.x * (iters / HB + 1) + it / HB) * 256 + threadIdx.x) % (bigN / 2);
      float4 v = big[gi];
      v.x += 1.0f;
      big[bigN / 2 + gi] = v;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(s == 0 ? av[i >> 1].x : s == 1 ? av[i >> 1].y : s == 2 ? av[i >> 1].z : av[i >> 1].w, bv[2 * s + (i & 1)], acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = bn[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = an[i];
  }
  const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int LDSR, int GLD, int HB = 0>
void run_traffic(int blocks, int iters) {
  float *out, *rnd; long long* clk;
  float4* big = nullptr;
  const long long bigN = 1ll << 27;      // 2 GiB of float4
  if (HB > 0) { (void)hipMalloc(&big, bigN * 16); (void)hipMemset(big, 0, bigN * 16); }
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&rnd, 4096 * 4);
  (void)hipMalloc(&clk, 16);
  float h[4096];
  unsigned x = 12345;
  for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) / 8388608.0f - 1.0f) * 0.9f; }
  (void)hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  mfma_traffic<LDSR, GLD, HB><<<blocks, 256>>>(out, rnd, iters / 10, clk, big, bigN);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  mfma_traffic<LDSR, GLD, HB><<<blocks, 256>>>(out, rnd, iters, clk, big, bigN);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long hc[2];
  (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
  const double waves = (double)blocks * 4;
  const double flops = waves * iters * 16.0 * (2.0 * 32 * 32 * 2);
  const double hbm = (double)blocks * 256 * (HB > 0 ? iters / (HB > 0 ? HB : 1) : 0) * 32.0 / ms * 1e-9;
  printf("3 waves/SIMD, per 16 MFMAs: %d ds_read_b32 + %d global 16B loads, HBM stream %.2f TB/s: %8.3f ms %7.1f TFLOP/s  shader clock %.0f MHz\n",
         LDSR, GLD, hbm, ms, flops / ms * 1e-9, 100.0 * hc[0] / hc[1]);
  (void)hipFree(out); (void)hipFree(rnd); (void)hipFree(clk); if (big) (void)hipFree(big);
}

// Instruction-class costs: 16 MFMAs per iteration plus NV plain VALU ops, NL ds_read_b32 (immediate offsets, no
// address math), NG global_load_dwordx4 (scalar base + fixed lane offset, L2 hits), NS SALU ops.
template <int NV, int NL, int NG, int NS>
__global__ void __launch_bounds__(256) mfma_mix(float* out, const float* rnd, int iters, long long* clk) {
  __shared__ float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = rnd[i];
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float a = 0.25f + threadIdx.x * 1e-6f, b = 0.5f;
  unsigned laddr = (threadIdx.x & 63) * 4;
  unsigned voff = threadIdx.x * 16;
  int x = threadIdx.x, sx = iters;
  float r0; float4 g0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NL; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r0) : "v"(laddr), "n"(i * 256));
#pragma unroll
    for (int i = 0; i < NG; ++i) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(g0) : "v"(voff), "s"(rnd), "n"(i * 1024));
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x));
#pragma unroll
    for (int i = 0; i < NS; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (NG > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float s = (float)x + (float)sx;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
}

// Same instruction counts, but spread: one VALU / ds_read / SALU is placed after each of the first N MFMAs instead of
// in one block ahead of them.
template <int NV, int NL, int NS>
__global__ void __launch_bounds__(256) mfma_interleaved(float* out, const float* rnd, int iters) {
  __shared__ float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = rnd[i];
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float a = 0.25f + threadIdx.x * 1e-6f, b = 0.5f;
  unsigned laddr = (threadIdx.x & 63) * 4;
  int x = threadIdx.x, sx = iters;
  float r0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (m < NV) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x));
      if (m + 16 < NV) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x));
      if (m < NL) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r0) : "v"(laddr), "n"(m * 256));
      if (m < NS) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = (float)x + (float)sx;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
}

template <int NV, int NL, int NS>
void run_interleaved(int blocks, int iters) {
  float *out, *rnd;
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&rnd, 65536 * 4);
  (void)hipMemset(rnd, 0, 65536 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  mfma_interleaved<NV, NL, NS><<<blocks, 256>>>(out, rnd, iters / 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  mfma_interleaved<NV, NL, NS><<<blocks, 256>>>(out, rnd, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 16.0 * (2.0 * 32 * 32 * 2);
  printf("interleaved per 16 MFMAs: %2d VALU %2d ds_read %2d SALU (%d waves/SIMD): %7.1f TFLOP/s\n", NV, NL, NS, blocks / 256, flops / ms * 1e-9);
  (void)hipFree(out); (void)hipFree(rnd);
}

template <int NV, int NL, int NG, int NS>
void run_mix(int blocks, int iters) {
  float *out, *rnd; long long* clk;
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&rnd, 65536 * 4);
  (void)hipMalloc(&clk, 16);
  (void)hipMemset(rnd, 0, 65536 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  mfma_mix<NV, NL, NG, NS><<<blocks, 256>>>(out, rnd, iters / 10, clk);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  mfma_mix<NV, NL, NG, NS><<<blocks, 256>>>(out, rnd, iters, clk);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 16.0 * (2.0 * 32 * 32 * 2);
  const double cyc_per_iter = ms * 1e-3 * 2.4e9 / iters / 3.0;     // per wave-iteration at 3 waves per SIMD
  printf("mix per 16 MFMAs: %2d VALU %2d ds_read %2d global_x4 %2d SALU: %7.1f TFLOP/s  (%.0f cycles per 16 MFMAs, ideal 1024)\n", NV, NL, NG, NS,
         flops / ms * 1e-9, cyc_per_iter);
  (void)hipFree(out); (void)hipFree(rnd); (void)hipFree(clk);
}

// Do the matrix pipe and the vector ALU of one SIMD run side by side?  Workgroups of 8 waves: waves 0-3 (one per SIMD)
// issue only MFMAs, waves 4-7 (their SIMD partners) only v_fma_f32 / v_pk_fma_f32 on independent accumulators.
// mode 0: MFMA waves only work, 1: VALU waves only, 2: both.  Rates are reported per kind.
template <bool PACKED>
__global__ void __launch_bounds__(512) mfma_valu_coexec(float* out, int iters, int mode, float a0) {
  const int wave = threadIdx.x >> 6;
  float s = 0.f;
  if (wave < 4) {
    if (mode == 1) return;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const float a = a0 + threadIdx.x * 1e-6f, b = 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) s += acc[i][j];
  } else {
    if (mode == 0) return;
    typedef float f2 __attribute__((ext_vector_type(2)));
    // per iteration 256 v_fma_f32 (or 128 v_pk_fma_f32) on 16 independent accumulators (pairs)
    f2 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f2){0.f, 0.f};
    const f2 x = (f2){a0 + threadIdx.x * 1e-6f, a0 * 0.5f};
    const f2 w = (f2){0.999f, 1.001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (PACKED) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(w));
          } else {
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(x.x), "v"(w.x));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(x.y), "v"(w.y));
          }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
  }
  if (s == 12345.678f) out[0] = s;
}

template <bool PACKED>
void run_coexec(int cus, int iters) {
  float* out;
  (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    mfma_valu_coexec<PACKED><<<cus, 512>>>(out, iters / 10, mode, 1e-3f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mfma_valu_coexec<PACKED><<<cus, 512>>>(out, iters, mode, 1e-3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = (mode != 1) ? (double)cus * 4 * iters * 16.0 * (2.0 * 32 * 32 * 2) / ms * 1e-9 : 0.0;
    const double vf = (mode != 0) ? (double)cus * 4 * iters * 256.0 * 64 * 2.0 / ms * 1e-9 : 0.0;
    printf("coexec (%s, 1 MFMA wave + 1 VALU wave per SIMD) mode %s: %8.3f ms  MFMA %6.1f TFLOP/s  VALU %6.1f TFLOP/s\n",
           PACKED ? "v_pk_fma_f32" : "v_fma_f32", mode == 0 ? "MFMA only" : mode == 1 ? "VALU only" : "both     ", ms, mf, vf);
  }
  (void)hipFree(out);
}

template <int NACC>
void run(int blocks, int threads, int iters, const char* what) {
  float* out;
  (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, threads>>>(out, iters / 10, 1e-3f, 1e-3f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, threads>>>(out, iters, 1e-3f, 1e-3f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64;
  const double flops = waves * iters * 8.0 * NACC * (2.0 * 32 * 32 * 2);
  printf("%-44s blocks %5d x %4d thr, %d acc tiles: %8.3f ms  %7.1f TFLOP/s\n", what, blocks, threads, NACC, ms, flops / ms * 1e-9);
  (void)hipFree(out);
}

int main() {
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  printf("# %s, %d CUs, clockRate %d kHz\n", pr.name, cus, pr.clockRate);
  run<4>(cus, 256, 20000, "1 wave/SIMD, short (0.1 s class)");
  run<4>(cus * 2, 256, 20000, "2 waves/SIMD");
  run<4>(cus * 3, 256, 20000, "3 waves/SIMD");
  run<1>(cus * 4, 256, 40000, "4 waves/SIMD, 1 acc tile (dependent chain)");
  run<4>(cus * 3, 256, 200000, "3 waves/SIMD, long (1 s class)");
  run<4>(cus, 256, 2000, "1 wave/SIMD, very short (10 ms class)");
  run_rand<4>(cus * 3, 256, 20000, "3 waves/SIMD, random operands");
  run_rand<4>(cus * 3, 256, 200000, "3 waves/SIMD, random operands, long");
  run_rand<4>(cus, 256, 2000, "1 wave/SIMD, random operands, 10 ms class");
  run_traffic<0, 0>(cus * 3, 100000);
  run_traffic<8, 0>(cus * 3, 100000);
  run_traffic<16, 0>(cus * 3, 100000);
  run_traffic<0, 2>(cus * 3, 100000);
  run_traffic<0, 4>(cus * 3, 100000);
  run_traffic<8, 2>(cus * 3, 100000);
  run_traffic<4, 4>(cus * 3, 100000);
  run_traffic<8, 2, 8>(cus * 3, 100000);
  run_traffic<8, 2, 2>(cus * 3, 100000);
  run_traffic<8, 2, 1>(cus * 3, 100000);
  run_traffic<0, 0, 1>(cus * 3, 100000);
  run_mix<0, 0, 0, 0>(cus * 3, 50000);
  run_mix<8, 0, 0, 0>(cus * 3, 50000);
  run_mix<16, 0, 0, 0>(cus * 3, 50000);
  run_mix<32, 0, 0, 0>(cus * 3, 50000);
  run_mix<0, 4, 0, 0>(cus * 3, 50000);
  run_mix<0, 8, 0, 0>(cus * 3, 50000);
  run_mix<0, 16, 0, 0>(cus * 3, 50000);
  run_mix<0, 0, 2, 0>(cus * 3, 50000);
  run_mix<0, 0, 4, 0>(cus * 3, 50000);
  run_mix<0, 0, 0, 16>(cus * 3, 50000);
  run_mix<8, 8, 2, 8>(cus * 3, 50000);
  run_interleaved<8, 0, 0>(cus * 3, 50000);
  run_interleaved<16, 0, 0>(cus * 3, 50000);
  run_interleaved<32, 0, 0>(cus * 3, 50000);
  run_interleaved<0, 8, 0>(cus * 3, 50000);
  run_interleaved<0, 0, 16>(cus * 3, 50000);
  run_interleaved<8, 8, 8>(cus * 3, 50000);
  run_interleaved<16, 8, 16>(cus * 3, 50000);
  run_interleaved<16, 8, 16>(cus * 1, 50000);
  run_mix<16, 8, 0, 16>(cus * 3, 50000);
  run_mix<16, 8, 0, 16>(cus * 1, 50000);
  run_coexec<false>(cus, 20000);
  run_coexec<true>(cus, 20000);
  return 0;
}
