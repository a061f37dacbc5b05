// What would a three-way bf16 split of the fp32 operands cost on the matrix pipe?  v_mfma_f32_32x32x16_bf16 (gfx950) against
// v_mfma_f32_32x32x2_f32: cycles per instruction for one wave per SIMD with four accumulators in rotation, operands that change
// every instruction (live data, as in the convolution streams), and the shader clock both sustain (s_memtime against the 100 MHz
// s_memrealtime).  Six bf16 MFMAs (k = 16 each) replace eight fp32 MFMAs (k = 2 each) per 16 k-steps of a 32 x 32 tile.
//   hipcc -O3 --offload-arch=gfx950 tools/bf16x3_probe.hip -o tools/bf16x3_probe && tools/bf16x3_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: fp32 32x32x2, 1: bf16 32x32x16
__global__ void __launch_bounds__(256) probe(long long* out, int iters, const float* src) {
  const int tid = threadIdx.x;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float fa[8], fb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { fa[j] = src[(tid * 8 + j) & 4095]; fb[j] = src[(tid * 8 + j + 1777) & 4095]; }
  bf16x8 ba[4], bb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) { ba[q][j] = (__bf16)(fa[j] * (1.0f + 0.25f * q)); bb[q][j] = (__bf16)(fb[j] * (1.0f - 0.125f * q)); }
  const long long w0 = (long long)wall_clock64();
  const long long c0 = (long long)__builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (MODE == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m & 7], fb[(m + 3) & 7], acc[m & 3], 0, 0, 0);
      else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba[m & 3], bb[(m + 1) & 3], acc[m & 3], 0, 0, 0);
    }
  }
  const long long c1 = (long long)__builtin_readcyclecounter();
  const long long w1 = (long long)wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[1000] = (long long)s;
  if (tid == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

int main() {
  long long* out; float* src;
  (void)hipMalloc(&out, 2048 * 8); (void)hipMalloc(&src, 4096 * 4);
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = 0.001f * ((i * 2654435761u) % 2000) - 1.0f;
  (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  const int it = 20000;
  for (int mode = 0; mode < 2; ++mode) {
    long long r[2];
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) probe<0><<<256, 256>>>(out, it, src); else probe<1><<<256, 256>>>(out, it, src);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
    const double cyc = (double)r[0] / (16.0 * it), us = (double)r[1] / 100.0, mhz = (double)r[0] / us;
    const double flop = mode == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    printf("%s: %.1f cycles per MFMA, shader clock %.0f MHz under load, %.1f TFLOP/s on 256 CUs (one wave per SIMD)\n",
           mode == 0 ? "v_mfma_f32_32x32x2_f32  " : "v_mfma_f32_32x32x16_bf16", cyc, mhz, flop / cyc * mhz * 1e6 * 1024 / 1e12);
  }
  printf("six bf16 MFMAs (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; k = 16) against eight fp32 MFMAs (k = 2) for 16 k-steps\n");
  return 0;
}
