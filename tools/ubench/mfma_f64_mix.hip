// Microbenchmark (round 4): what does an instruction issued between v_mfma_f64_16x16x4_f64 cost?  A 2 x 4 register tile (8 MFMAs per
// k-step, as the 8-wave GEMM kernels) with NR LDS reads (ds_read_b64, results consumed by the MFMAs of the next round) and NV v_mul_f64
// per round; chip-wide, 4 waves per SIMD (2 workgroups of 8 waves per CU).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_mix.hip -o mfma_f64_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

template <int NR, int NV, int WIDE>
__global__ void __launch_bounds__(512, 2) k_mix(double* out, int iters, double a0) {
  __shared__ double lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = a0 + i * 1e-6;
  __syncthreads();
  d4 acc[2][4];
  double a[2], b[4], v[8];
#pragma unroll
  for (int i = 0; i < 2; ++i) a[i] = a0 + (threadIdx.x + 17 * i) * 1e-3;
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = a0 + (threadIdx.x * 3 + j) * 1e-3;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 1.0 + j * 1e-9;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0, 0, 0, 0};
  const int base = (threadIdx.x & 63) * (WIDE ? 2 : 1);
  for (int it = 0; it < iters; ++it) {
    double r[8];
    const int o = base + (it & 7) * 128;
    if (WIDE) {
#pragma unroll
      for (int q = 0; q < NR / 2; ++q) { const double2 t = *reinterpret_cast<const double2*>(&lds[o + q * 1024]); r[2 * q] = t.x; r[2 * q + 1] = t.y; }
    } else {
#pragma unroll
      for (int q = 0; q < NR; ++q) r[q] = lds[o + q * 1024];
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] *= 1.0000001;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    // the reads feed the NEXT round's operands (a real loop's dependence)
    if (NR >= 1) a[0] = r[0];
    if (NR >= 2) a[1] = r[1];
#pragma unroll
    for (int q = 2; q < NR && q < 6; ++q) b[q - 2] = r[q];
    if (NR > 6) b[0] += r[6] * 0.0;
    if (NR > 7) b[1] += r[7] * 0.0;
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> double timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e-3;
}
template <int NR, int NV, int WIDE> void run(double* out, int grid, int iters) {
  double t = timeit([&] { hipLaunchKernelGGL((k_mix<NR, NV, WIDE>), grid, 512, 0, 0, out, iters, 1.0); });
  const double tf = (double)grid * 8 * iters * 8 * 2048.0 / t * 1e-12;
  printf("8 MFMA16 + %d ds_read_%s + %d v_mul_f64 per round: %.2f TF/s  (%.1f cycles per round per wave-slot at 2.3 GHz)\n", NR, WIDE ? "b128(as pairs)" : "b64", NV, tf,
         t * 2.3e9 / iters / 4.0);
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int grid = p.multiProcessorCount * 2, iters = 20000;
  double* out; CK(hipMalloc(&out, sizeof(double) * grid * 512));
  run<0, 0, 0>(out, grid, iters);
  run<2, 0, 0>(out, grid, iters);
  run<6, 0, 0>(out, grid, iters);
  run<8, 0, 0>(out, grid, iters);
  run<6, 0, 1>(out, grid, iters);
  run<8, 0, 1>(out, grid, iters);
  run<0, 4, 0>(out, grid, iters);
  run<0, 8, 0>(out, grid, iters);
  run<6, 4, 0>(out, grid, iters);
  return 0;
}
