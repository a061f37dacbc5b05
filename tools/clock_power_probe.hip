// What pulls the shader clock from 2.4 GHz to ~2.05 GHz under the Winograd kernels (the stamps' effective clock)?
// One wave per SIMD on every CU issues dependent-by-four v_mfma_f32_32x32x2_f32 streams for ~100 ms per mode; the effective clock is
// shader cycles (s_memtime) over the 100 MHz wall counter, measured inside the kernel by wave 0 of every workgroup.
//   mode 0  constant operands in two registers (r01's probe: holds 2.39 GHz)
//   mode 1  CHANGING operands, all in registers (16 A and 16 B values per lane, no memory instruction in the stream)
//   mode 2  B fragments from LDS, one ds_read_b32 per MFMA (the kernels' stream), A in registers
//   mode 3  B fragments from LDS, one ds_read_b128 per FOUR MFMAs (0.25 LDS instructions per MFMA), A in registers
//   mode 4  mode 2 + A through one buffer_load_b128 per four MFMAs from an L2-resident image (the kernels' weight stream)
//   mode 5  mode 3 + the same A stream
// If mode 1 already sits at the low clock, the matrix datapath on toggling operands is what draws the power and no instruction diet helps;
// if only modes 2 / 4 do, the LDS / L2 traffic is, and mode 3 / 5 say what wide fragment reads would buy.
//   modes 6-9: the same streams with four more waves per workgroup doing what the kernels' producers do (see PROD below)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/clock_power_probe.hip -o tools/clock_power_probe && tools/clock_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <class F>
__device__ __forceinline__ void static_for_16(F& f) { static_for_impl(f, std::make_integer_sequence<int, 16>{}); }

__device__ __forceinline__ long long wall100() { return (long long)__builtin_amdgcn_s_memrealtime(); }

// PROD: 0 = four waves (one consumer per SIMD); otherwise four more waves beside them, as the kernels' producers:
//   bit 0: packed-fp32 VALU work + LDS stores at about the producers' duty (32 VALU + 8 ds_write_b64 per ~2000 cycles),
//   bit 1: HBM reads: one 16-byte load per lane per ~800 cycles, every wave walking its own contiguous 4 MB of a 4 GB buffer (~2.5 TB/s over the chip),
//   bit 2: HBM reads + writes (what was read is written to the other half of the chunk)
template <int MODE, int PROD = 0>
__global__ void __launch_bounds__(PROD ? 512 : 256) probe(long long* out, const float* __restrict__ wimg, int iters, const float4* __restrict__ big = nullptr,
                                                       long long big_n = 0) {
  extern __shared__ __attribute__((aligned(16))) float sm[];            // 64 KB of "planes" (+ 16 KB of producer area + flag)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned seed = 0x9e3779b9u * (unsigned)(blockIdx.x * 256 + tid + 1);
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((int)(seed >> 9) - (1 << 22)) * (1.0f / (1 << 22)); };
  for (int i = tid; i < 16384; i += (int)blockDim.x) sm[i] = rnd();
  if (tid == 0) *reinterpret_cast<volatile int*>(sm + 16384 + 4096) = 0;
  __syncthreads();                                           // the ONLY workgroup barrier: the producers leave the common path right behind it
  if constexpr (PROD != 0) {
    volatile int* flag = reinterpret_cast<volatile int*>(sm + 16384 + 4096);
    if (wave >= 4) {
      __builtin_amdgcn_s_setprio(3);
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (f32x2){0.5f + j + tid * 1e-6f, 0.25f + j};
      const f32x2 ka = {1.0001f, 0.9999f}, kb = {1e-7f, -1e-7f};
      float* pw = sm + 16384 + (wave - 4) * 1024 + lane * 2;
      // HBM streaming: every producer wave walks its own contiguous chunk of the buffer (sequential 1 KB wave-loads: no TLB thrash), eight loads
      // in flight; bit 2 of PROD: it also WRITES 1 KB per iteration to the second half of its chunk
      const long long chunk = big_n / (256LL * 4);                       // float4 per wave
      const float4* gsrc = big + ((long long)blockIdx.x * 4 + (wave - 4)) * chunk;
      float4* gdst = const_cast<float4*>(gsrc) + chunk / 2;
      const long long span = (PROD & 4) ? chunk / 2 : chunk;
      long long gi = lane;
      float4 ld[8] = {};
      long long n = 0;
      while (*flag == 0) {
        if constexpr (PROD & 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_elementwise_fma(v[j], ka, kb);
            *reinterpret_cast<f32x2*>(pw + 128 * r) = v[r];
            *reinterpret_cast<f32x2*>(pw + 128 * r + 512) = v[r + 4];
          }
        }
        if constexpr (PROD & 6) {
          ld[n & 7] = gsrc[gi];
          const float4 q = ld[(n + 1) & 7];                 // consumed seven iterations later
          if constexpr (PROD & 4) gdst[gi] = q;
          else v[0].x += q.x + q.y + q.z + q.w;
          gi += 64; if (gi >= span) gi = lane;
          __builtin_amdgcn_s_sleep(PROD & 1 ? 8 : 11);
        } else {
          __builtin_amdgcn_s_sleep(28);
        }
        ++n;
      }
      float s_ = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s_ += v[j].x + v[j].y;
      if (s_ == 12345.678f) out[4097] = (long long)s_;
      if (lane == 0 && wave == 4) out[1024 + blockIdx.x] = n;
      return;
    }
  }
  float ar[16], br[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { ar[i] = rnd(); br[i] = rnd(); }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const unsigned lbase = (unsigned)(size_t)sm + (unsigned)wave * 16384u;
  unsigned la32 = lbase + (unsigned)lane * 4u;                     // ds_read_b32: 64 consecutive floats per read (conflict-free)
  unsigned la128 = lbase + (unsigned)lane * 16u;                   // ds_read_b128: 64 consecutive float4
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wimg), 0, 0x7fffffff, 0x00020000);
  const unsigned wl = (unsigned)lane * 16u;
  float bb[2][4];
  float4 av[4];
  auto rdB = [&](auto gc, float (&d)[4]) {                               // B fragments of group G: four ds_read_b32 or one ds_read_b128
    constexpr int G = decltype(gc)::value & 15;
    const unsigned la32_ = la32, la128_ = la128;         // (inline-asm operands alone do not capture in a generic lambda)
    if constexpr (MODE == 2 || MODE == 4) {
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d[0]) : "v"(la32_), "n"((4 * G + 0) * 256) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d[1]) : "v"(la32_), "n"((4 * G + 1) * 256) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d[2]) : "v"(la32_), "n"((4 * G + 2) * 256) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d[3]) : "v"(la32_), "n"((4 * G + 3) * 256) : "memory");
    }
    if constexpr (MODE == 3 || MODE == 5) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(*reinterpret_cast<float4*>(&d[0])) : "v"(la128_), "n"(G * 1024) : "memory");
    }
  };
  auto ldA = [&](auto gc, int soff) {
    constexpr int G = decltype(gc)::value & 15;
    if constexpr (MODE >= 4) {
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)wl + (G & 3) * 1024, soff + (G >> 2) * 4096, 0);      // immediate < 4096, the rest in the SGPR
      av[decltype(gc)::value & 3] = *reinterpret_cast<const float4*>(&t);
    }
  };
  const long long w0 = wall100();
  const long long c0 = (long long)__builtin_readcyclecounter();
  rdB(std::integral_constant<int, 0>{}, bb[0]);
  ldA(std::integral_constant<int, 0>{}, 0); ldA(std::integral_constant<int, 1>{}, 0);
  for (int it = 0; it < iters; ++it) {
    // 64 MFMAs per iteration: 16 groups of four (one accumulator each, as the consumers' k-groups); fragments one group ahead, A two groups ahead
    const int soff = __builtin_amdgcn_readfirstlane(((it * 7 + wave * 3 + (int)blockIdx.x) & 63) * 16384);      // 1 MB image: stays in the L2
    auto group = [&](auto gc) {
      constexpr int G = decltype(gc)::value;
      rdB(std::integral_constant<int, G + 1>{}, bb[(G + 1) & 1]);
      ldA(std::integral_constant<int, G + 2>{}, soff);
      float(&b)[4] = bb[G & 1];
      if constexpr (MODE == 2 || MODE == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      if constexpr (MODE == 3 || MODE == 5) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      float a4[4], b4[4];
      if constexpr (MODE == 0) { a4[0] = a4[1] = a4[2] = a4[3] = ar[0]; b4[0] = b4[1] = b4[2] = b4[3] = br[0]; }
      else {
#pragma unroll
        for (int s = 0; s < 4; ++s) { a4[s] = ar[(4 * G + s) & 15]; b4[s] = br[(4 * G + s + (G >> 2)) & 15]; }
      }
      if constexpr (MODE >= 2) { b4[0] = b[0]; b4[1] = b[1]; b4[2] = b[2]; b4[3] = b[3]; }
      if constexpr (MODE >= 4) { const float4 f = av[G & 3]; a4[0] = f.x; a4[1] = f.y; a4[2] = f.z; a4[3] = f.w; }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[G & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[s], acc[G & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    static_for_16(group);
    if ((it & 63) == 63) {                 // keep the accumulators finite without touching the stream often
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] *= 1e-3f;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long c1 = (long long)__builtin_readcyclecounter();
  const long long w1 = wall100();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[4096] = (long long)s;
  if (tid == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
  if constexpr (PROD != 0) { if (tid == 0) *reinterpret_cast<volatile int*>(sm + 16384 + 4096) = 1; }
}

static float4* g_big = nullptr;
static long long g_big_n = 0;
template <int MODE, int PROD = 0>
static void run(const char* name, long long* out, const float* wimg, int iters) {
  std::vector<long long> h(1536);
  auto kern = probe<MODE, PROD>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 16384 + 64);
  for (int rep = 0; rep < 2; ++rep) {      // the second launch is the one reported (clocks settled)
    (void)hipMemset(out, 0, 8192 * 8);
    hipLaunchKernelGGL(kern, dim3(256), dim3(PROD ? 512 : 256), 65536 + 16384 + 64, 0, out, wimg, iters, (const float4*)g_big, g_big_n);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpy(h.data(), out, 1536 * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0, cmin = 1e30, cmax = 0;
  for (int b = 0; b < 256; ++b) {
    cyc += (double)h[2 * b]; wall += (double)h[2 * b + 1];
    const double f = (double)h[2 * b] / ((double)h[2 * b + 1] * 10.0);      // cycles per ns = GHz (wall counter: 100 MHz = 10 ns)
    cmin = f < cmin ? f : cmin; cmax = f > cmax ? f : cmax;
  }
  const double ghz = cyc / (wall * 10.0);
  const double per = cyc / 256.0 / (64.0 * iters);
  double pit = 0;
  for (int b = 0; b < 256; ++b) pit += (double)h[1024 + b];
  const double secs = wall / 256.0 * 10e-9;
  printf("%-76s %6.1f cycles / MFMA   clock %.3f GHz (workgroups %.3f .. %.3f)   %6.1f TFLOP/s   %.1f ms", name, per, ghz, cmin, cmax,
         256.0 * 4 * 64.0 * iters * 4096.0 / secs / 1e12, secs * 1e3);
  if (PROD) printf("   producers: %.0f iterations per wave%s", pit / 256.0, (PROD & 2) ? "" : "");
  if (PROD & 6) printf(", %.2f TB/s read%s", pit * 4 * 1024.0 / secs / 1e12, (PROD & 4) ? " + as much written" : "");
  printf("\n");
}

int main() {
  long long* out; (void)hipMalloc(&out, 8192 * 8);
  float* wimg; (void)hipMalloc(&wimg, 2 << 20);
  std::vector<float> hw((2 << 20) / 4);
  unsigned s = 12345u;
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
  (void)hipMemcpy(wimg, hw.data(), 2 << 20, hipMemcpyHostToDevice);
  const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 60000;      // 60000 x 64 MFMAs x 64 cycles = 246 M cycles ~ 100 ms
  const int only = getenv("ONLY") ? atoi(getenv("ONLY")) : -1;
  if (only == 6) {
    g_big_n = (64LL << 20) / 16; (void)hipMalloc(&g_big, 64LL << 20);
    run<4, 1>("6 = 4 + four producer waves: packed VALU + LDS stores", out, wimg, iters);
    return 0;
  }
  printf("one wave per SIMD on 256 CUs, %d x 64 dependent-by-four v_mfma_f32_32x32x2_f32 per wave\n", iters);
  run<0>("0 constant operands (registers)", out, wimg, iters);
  run<1>("1 changing operands, registers only", out, wimg, iters);
  run<2>("2 B: ds_read_b32 per MFMA", out, wimg, iters);
  run<3>("3 B: ds_read_b128 per four MFMAs", out, wimg, iters);
  run<4>("4 B: ds_read_b32 per MFMA, A: buffer_load_b128 per four MFMAs (L2)", out, wimg, iters);
  run<5>("5 B: ds_read_b128 per four MFMAs, A: buffer_load_b128 per four MFMAs (L2)", out, wimg, iters);
  run<0>("0 again (drift check)", out, wimg, iters);
  g_big_n = (4LL << 30) / 16;
  if (hipMalloc(&g_big, 4LL << 30) == hipSuccess) {
    (void)hipMemset(g_big, 0, 4LL << 30);
    run<4, 1>("6 = 4 + four producer waves: packed VALU + LDS stores", out, wimg, iters);
    run<4, 2>("7 = 4 + four producer waves: HBM reads", out, wimg, iters);
    run<4, 4>("8 = 4 + four producer waves: HBM reads + writes", out, wimg, iters);
    run<4, 5>("9 = 4 + four producer waves: VALU + LDS stores + HBM reads + writes", out, wimg, iters);
    run<1, 4>("10 = 1 (registers only) + four producer waves: HBM reads + writes", out, wimg, iters);
    run<4>("4 again", out, wimg, iters);
  }
  return 0;
}
