// Is hipLaunchCooperativeKernel usable for the persistent WN launches (csrc/wn_stack.hip, csrc/wn_mesh.hip)?  They are part of captured plans
// (hipGraph replay), so the question is (1) what a cooperative launch does on a CAPTURING stream, (2) what it checks at launch time - the same
// occupancy x CU bound the library asks hipOccupancyMaxActiveBlocksPerMultiprocessor for - and (3) what that query returns for a kernel with the
// stack launch's footprint (768 threads, 150 KB of dynamic LDS).  Round 6, VERDICT r5 item 1c.  Build: hipcc --offload-arch=gfx950 -O2 -o coop_capture_probe coop_capture_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>

__global__ void __launch_bounds__(768) probe_kernel(int* out) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (int)lds[1]);
}

#define CK(x) do { hipError_t e_ = (x); printf("%-78s -> %s\n", #x, hipGetErrorName(e_)); } while (0)

int main() {
  int dev = 0, cus = 0, coop = 0;
  hipSetDevice(dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
  printf("CUs %d, hipDeviceAttributeCooperativeLaunch %d\n", cus, coop);
  const size_t lds = 153616;
  CK(hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int per_cu = -1;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)probe_kernel, 768, lds));
  printf("occupancy query: %d workgroup(s) per CU of 768 threads + %zu B LDS -> capacity %d\n", per_cu, lds, per_cu * cus);
  int* out = nullptr;
  hipMalloc(&out, 4); hipMemset(out, 0, 4);
  hipStream_t st; hipStreamCreate(&st);
  void* args[] = {&out};
  printf("--- plain stream\n");
  CK(hipLaunchCooperativeKernel((const void*)probe_kernel, dim3(cus), dim3(768), args, (unsigned)lds, st));
  CK(hipStreamSynchronize(st));
  CK(hipLaunchCooperativeKernel((const void*)probe_kernel, dim3(cus + 1), dim3(768), args, (unsigned)lds, st));      // one more than fits
  CK(hipStreamSynchronize(st));
  (void)hipGetLastError();
  printf("--- capturing stream\n");
  hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  CK(hipLaunchCooperativeKernel((const void*)probe_kernel, dim3(cus), dim3(768), args, (unsigned)lds, st));
  CK(hipStreamEndCapture(st, &g));
  (void)hipGetLastError();
  if (g) {
    size_t n = 0; hipGraphGetNodes(g, nullptr, &n); printf("captured graph holds %zu node(s)\n", n);
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    if (ge) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
  }
  int h = 0; hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
  printf("kernel executions counted: %d\n", h);
  return 0;
}
