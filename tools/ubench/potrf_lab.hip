// Where the 53 us of k_potrf_diag go (round 6): the product kernel (zigp_kernels.h) on one 128 x 128 SPD block, timed with HIP events over
// back-to-back launches (one workgroup each, so launch overhead ~2 us is in every number), for 0..4 real 32-column panels, with and
// without the triangular inverse.  npan = 0 is load + store only.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I../../zero-inflated-gp_amd/csrc -I../../include potrf_lab.hip -o potrf_lab
#include "zigp_gemm.h"
#include "zigp_kernels.h"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace zigp;
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e_),__LINE__); return 1;} }while(0)
int main() {
  const int n = 128, ld = 1024;
  std::vector<double> h((size_t)n * ld, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) h[(size_t)i * ld + j] = std::exp(-0.5 * (i - j) * (i - j) / 9.0) + (i == j ? 1e-3 : 0.0);
  double *dA, *dL, *dW; int* dinfo;
  CK(hipMalloc(&dA, sizeof(double) * h.size())); CK(hipMalloc(&dL, sizeof(double) * h.size())); CK(hipMalloc(&dW, sizeof(double) * h.size()));
  CK(hipMalloc(&dinfo, sizeof(int))); CK(hipMemset(dinfo, 0, sizeof(int)));
  CK(hipMemcpy(dA, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
  const size_t shm = sizeof(double) * PB * PBLD;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_potrf_diag), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 200;
  for (int wantw = 1; wantw >= 0; --wantw)
    for (int npan = 4; npan >= 0; --npan) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(1024), shm, 0, dA, dL, wantw ? dW : nullptr, (int64_t)ld, 0, dinfo, npan, 0.0);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(1024), shm, 0, dA, dL, wantw ? dW : nullptr, (int64_t)ld, 0, dinfo, npan, 0.0);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      int info; CK(hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost));
      printf("npan %d  W %d : %7.2f us per launch (info %d)\n", npan, wantw, ms / reps * 1e3, info);
    }
  return 0;
}
