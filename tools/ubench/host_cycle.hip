// Host-side cost of one "upload - kernels - download - synchronise" cycle, the skeleton of a Kronecker minibatch step:
//   a) empty kernel + hipStreamSynchronize          b) a) + hipMemcpyAsync D2H of 2 KB into pinned memory
//   c) kernel stores 2 KB straight into mapped pinned memory + synchronise      d) b) + hipMemcpyAsync H2D of 32 KB before the kernel
//   e) c) + the kernel reads its 32 KB input from mapped pinned memory           f) 10 empty kernels + synchronise
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_empty() {}
__global__ void k_store(double* out, const double* in, int nin) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nin; i += blockDim.x) s += in[i];
  out[threadIdx.x] = s + threadIdx.x;
}
template <class F> double timeit(F f, int n = 2000) {
  for (int i = 0; i < 50; ++i) f();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f();
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double *d_out, *d_in, *h_out, *h_in, *hd_out, *hd_in;
  CK(hipMalloc(&d_out, 2048)); CK(hipMalloc(&d_in, 32768));
  CK(hipHostMalloc(&h_out, 2048, hipHostMallocMapped)); CK(hipHostMalloc(&h_in, 32768, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&hd_out, h_out, 0)); CK(hipHostGetDevicePointer((void**)&hd_in, h_in, 0));
  for (int i = 0; i < 4096; ++i) h_in[i] = i;
  CK(hipMemset(d_in, 0, 32768));
  printf("a) kernel + sync                         %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, st); hipStreamSynchronize(st); }));
  printf("b) kernel + D2H 2 KB + sync              %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_store, 1, 256, 0, st, d_out, d_in, 0); hipMemcpyAsync(h_out, d_out, 2048, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }));
  printf("c) kernel stores to pinned + sync        %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_store, 1, 256, 0, st, hd_out, d_in, 0); hipStreamSynchronize(st); }));
  printf("d) H2D 32 KB + kernel + D2H 2 KB + sync  %.1f us\n", timeit([&] { hipMemcpyAsync(d_in, h_in, 32768, hipMemcpyHostToDevice, st); hipLaunchKernelGGL(k_store, 1, 256, 0, st, d_out, d_in, 4096); hipMemcpyAsync(h_out, d_out, 2048, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }));
  printf("e) kernel reads pinned, stores pinned    %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_store, 1, 256, 0, st, hd_out, hd_in, 4096); hipStreamSynchronize(st); }));
  printf("f) 10 kernels + sync                     %.1f us\n", timeit([&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, 1, 64, 0, st); hipStreamSynchronize(st); }));
  printf("g) memset 4 B + kernel + sync            %.1f us\n", timeit([&] { hipMemsetAsync(d_out, 0, 4, st); hipLaunchKernelGGL(k_empty, 1, 64, 0, st); hipStreamSynchronize(st); }));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  printf("h) kernel + event record + event sync    %.1f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, st); hipEventRecord(ev, st); hipEventSynchronize(ev); }));
  volatile double* flag = h_out;
  printf("i) kernel stores pinned, host spins on it %.1f us\n", timeit([&] { static double tag = 1.0; tag += 1.0; h_in[0] = tag; flag[0] = 0.0; hipLaunchKernelGGL(k_store, 1, 64, 0, st, hd_out, hd_in, 1); while (flag[0] != tag) {} }));
  hipStreamSynchronize(st);
  return 0;
}
