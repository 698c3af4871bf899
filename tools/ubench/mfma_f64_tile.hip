// Microbenchmark (round 4): v_mfma_f64_16x16x4_f64 against 4 x v_mfma_f64_4x4x4_4b_f64 in the register pattern of a real 64 x 64 wave
// tile (4 A fragments x 4 B fragments, 16 / 64 independent accumulator groups), chip-wide, 1..4 waves per SIMD -- and the lane layout
// of the 16x16x4 operands.  Why: rocBLAS' gfx950 DGEMM kernels (MT128x128x16, MI16x16x4x1) reach 74-77 TFLOP/s on this chip
// (tools/dgemm_probe.py), which the round-1 rate test (every MFMA reading the SAME operand registers) did not predict.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_tile.hip -o mfma_f64_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__global__ void k_layout(const double* A, const double* B, double* D) {   // A [16][4], B [4][16] row-major; D out [64 lanes][4]
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16];      // assumed: lane -> (row l % 16, k l / 16)
  const double b = B[(l / 16) * 16 + l % 16];     // assumed: lane -> (k l / 16, col l % 16)
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[l * 4 + i] = acc[i];
}

template <int TM, int TN>
__global__ void __launch_bounds__(256, 2) k_tile16(double* out, int iters, double a0, double b0) {
  d4 acc[TM][TN];
  double a[TM], b[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a[i] = a0 + (threadIdx.x + 17 * i) * 1e-3;
#pragma unroll
  for (int j = 0; j < TN; ++j) b[j] = b0 + (threadIdx.x * 3 + j) * 1e-3;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (d4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int TM, int TN>
__global__ void __launch_bounds__(256, 2) k_tile4(double* out, int iters, double a0, double b0) {
  double acc[TM][TN][4];
  double a[TM][4], b[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) a[i][r] = a0 + (threadIdx.x + 17 * i + 5 * r) * 1e-3;
#pragma unroll
  for (int j = 0; j < TN; ++j) b[j] = b0 + (threadIdx.x * 3 + j) * 1e-3;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i][r], b[j], acc[i][j][r], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> double timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e-3;
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  {   // layout
    std::vector<double> A(64), B(64), D(256), R(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i + 0.01 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + 0.1 * k + 0.001 * j;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
    double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, 1, 64, 0, 0, dA, dB, dD); CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0;   // candidate D layouts: (a) element i of lane l = row 4 (l / 16) + i   (b) row 4 i + l / 16
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
      e1 = fmax(e1, fabs(D[l * 4 + i] - R[(4 * (l / 16) + i) * 16 + l % 16]));
      e2 = fmax(e2, fabs(D[l * 4 + i] - R[(4 * i + l / 16) * 16 + l % 16]));
    }
    printf("16x16x4 layout with A lane->(row l%%16, k l/16), B lane->(k l/16, col l%%16): D[i] = row 4*(l/16)+i: err %.2e ; D[i] = row 4*i+l/16: err %.2e\n", e1, e2);
  }
  double* out; CK(hipMalloc(&out, sizeof(double) * ncu * 8 * 256 * 4));
  const int iters = 20000;
  for (int wpc : {1, 2, 4}) {
    const int grid = ncu * wpc;
    { double t = timeit([&] { hipLaunchKernelGGL((k_tile16<4, 4>), grid, 256, 0, 0, out, iters, 1.0, 1e-3); });
      printf("16x16x4 tile 4x4 (16 acc d4)  wg/CU=%d: %.3f ms %.2f TF/s\n", wpc, t * 1e3, (double)grid * 4 * iters * 16 * 2048.0 / t * 1e-12); }
    { double t = timeit([&] { hipLaunchKernelGGL((k_tile4<4, 4>), grid, 256, 0, 0, out, iters, 1.0, 1e-3); });
      printf("4x4x4   tile 4x4 (64 acc)     wg/CU=%d: %.3f ms %.2f TF/s\n", wpc, t * 1e3, (double)grid * 4 * iters * 64 * 512.0 / t * 1e-12); }
    { double t = timeit([&] { hipLaunchKernelGGL((k_tile16<2, 4>), grid, 256, 0, 0, out, iters, 1.0, 1e-3); });
      printf("16x16x4 tile 2x4 (8 acc d4)   wg/CU=%d: %.3f ms %.2f TF/s\n", wpc, t * 1e3, (double)grid * 4 * iters * 8 * 2048.0 / t * 1e-12); }
    { double t = timeit([&] { hipLaunchKernelGGL((k_tile4<2, 4>), grid, 256, 0, 0, out, iters, 1.0, 1e-3); });
      printf("4x4x4   tile 2x4 (32 acc)     wg/CU=%d: %.3f ms %.2f TF/s\n", wpc, t * 1e3, (double)grid * 4 * iters * 32 * 512.0 / t * 1e-12); }
  }
  return 0;
}
