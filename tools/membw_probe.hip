// Tile-pattern copy bandwidth probe: src,dst are [ROWS][L] fp32; each workgroup copies one [R][W] tile
// (R rows x W contiguous floats), float4 per lane, all loads of a thread issued before its stores.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membw_probe.hip -o /tmp/membw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int U>
__global__ void __launch_bounds__(256) tile_copy(const float* __restrict__ src, float* __restrict__ dst, int L, int R, int W, int ntx, int mode) {
  const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx;
  const int w4 = W / 4, total = R * w4;
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(dst);
  for (int base = threadIdx.x; base < total; base += 256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * 256;
      if (i < total) { const int r = i / w4, c = i - r * w4; v[u] = s4[((size_t)(ty * R + r) * L + (size_t)tx * W) / 4 + c]; }
    }
    if (mode == 1) continue;            // read only
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * 256;
      if (i < total) { const int r = i / w4, c = i - r * w4; float4 q = v[u]; if (mode == 2) q = make_float4(1.f, 2.f, 3.f, 4.f); d4[((size_t)(ty * R + r) * L + (size_t)tx * W) / 4 + c] = q; }
    }
  }
}
__global__ void sink(float* p) { if (p[0] == 123.f) p[1] = 0; }
int main() {
  const size_t bytes = 268435456;   // one decoder stage tensor at B=16, T=512
  float *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { int rows_total, L, R, W; } cfgs[] = {
    {512, 131072, 32, 256}, {512, 131072, 32, 512}, {512, 131072, 32, 1024}, {512, 131072, 32, 4096}, {512, 131072, 8, 1024}, {512, 131072, 1, 8192},
    {1024, 65536, 64, 256}, {2048, 32768, 128, 128}, {2048, 32768, 32, 128}, {4096, 16384 /*8192*2*/, 256, 128}};
  printf("%-28s %10s %10s %10s\n", "tile [R x W] of [rows x L]", "copy GB/s", "read GB/s", "write GB/s");
  for (auto c : cfgs) {
    const int ntx = c.L / c.W, nty = c.rows_total / c.R;
    double res[3];
    for (int mode = 0; mode < 3; ++mode) {
      for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(tile_copy<8>, dim3(ntx * nty), dim3(256), 0, 0, a, b, c.L, c.R, c.W, ntx, mode);
      hipEventRecord(e0);
      for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(tile_copy<8>, dim3(ntx * nty), dim3(256), 0, 0, a, b, c.L, c.R, c.W, ntx, mode);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double moved = (mode == 0 ? 2.0 : 1.0) * (double)c.rows_total * c.L * 4 * 5;
      res[mode] = moved / (ms * 1e-3) / 1e9;
    }
    char nm[64]; snprintf(nm, sizeof nm, "[%d x %d] of [%d x %d]", c.R, c.W, c.rows_total, c.L);
    printf("%-28s %10.0f %10.0f %10.0f\n", nm, res[0], res[1], res[2]);
  }
  return 0;
}
