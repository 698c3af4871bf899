// Microbenchmark: fp64 MFMA and VALU FMA issue rates on gfx950 (MI355X).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_rate.hip -o mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

template<int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i=0;i<NACC;i++) acc[i]=(d4){0,0,0,0};
  double a=a0+threadIdx.x*1e-9, b=b0;
  for (int it=0; it<iters; ++it) {
#pragma unroll
    for (int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int NACC>
__global__ void __launch_bounds__(256) k_mfma4(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i=0;i<NACC;i++) acc[i]=0;
  double a=a0+threadIdx.x*1e-9, b=b0;
  for (int it=0; it<iters; ++it) {
#pragma unroll
    for (int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int NACC>
__global__ void __launch_bounds__(256) k_fma(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i=0;i<NACC;i++) acc[i]=threadIdx.x*1e-3+i;
  double a=a0, b=b0;
  for (int it=0; it<iters; ++it) {
#pragma unroll
    for (int i=0;i<NACC;i++) acc[i]=__builtin_fma(acc[i],a,b);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
// mixed: NACC mfma + NF fma per iteration
template<int NACC,int NF>
__global__ void __launch_bounds__(256) k_mix(double* out, int iters, double a0, double b0) {
  d4 acc[NACC]; double f[NF];
  for (int i=0;i<NACC;i++) acc[i]=(d4){0,0,0,0};
  for (int i=0;i<NF;i++) f[i]=threadIdx.x*1e-3+i;
  double a=a0+threadIdx.x*1e-9, b=b0;
  for (int it=0; it<iters; ++it) {
#pragma unroll
    for (int i=0;i<NACC;i++) {
      acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
#pragma unroll
      for (int j=0;j<NF/NACC;j++) f[i*(NF/NACC)+j]=__builtin_fma(f[i*(NF/NACC)+j],a0,b0);
    }
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  for (int i=0;i<NF;i++) s+=f[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<typename F> double timeit(F f){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms*1e-3;
}
int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs %d clock %d kHz\n",p.name,p.multiProcessorCount,p.clockRate);
  int ncu=p.multiProcessorCount; double* out; CK(hipMalloc(&out, sizeof(double)*ncu*8*256*4));
  int iters=20000;
  for (int wpc : {1,2,4}) { // workgroups(256 thr = 4 waves) per CU => waves per SIMD
    int grid=ncu*wpc;
    { double t=timeit([&]{ hipLaunchKernelGGL(k_mfma<8>,grid,256,0,0,out,iters,1.0,1e-9);});
      double fl=(double)grid*4*iters*8*2048.0; printf("mfma16x16x4 f64 acc8  wg/CU=%d: %.3f ms %.2f TF/s, cyc/mfma/SIMD@2.4GHz=%.1f\n",wpc,t*1e3,fl/t*1e-12, t*2.4e9/(iters*8.0*wpc)); }
    { double t=timeit([&]{ hipLaunchKernelGGL(k_mfma<1>,grid,256,0,0,out,iters,1.0,1e-9);});
      double fl=(double)grid*4*iters*1*2048.0; printf("mfma16x16x4 f64 acc1 (dependent) wg/CU=%d: %.3f ms %.2f TF/s cyc/mfma=%.1f\n",wpc,t*1e3,fl/t*1e-12, t*2.4e9/(iters*1.0*wpc)); }
    { double t=timeit([&]{ hipLaunchKernelGGL(k_mfma<2>,grid,256,0,0,out,iters,1.0,1e-9);});
      double fl=(double)grid*4*iters*2*2048.0; printf("mfma16x16x4 f64 acc2 wg/CU=%d: %.3f ms %.2f TF/s\n",wpc,t*1e3,fl/t*1e-12); }
    { double t=timeit([&]{ hipLaunchKernelGGL(k_mfma4<8>,grid,256,0,0,out,iters,1.0,1e-9);});
      double fl=(double)grid*4*iters*8*(4*4*4*4*2.0); printf("mfma4x4x4 f64 (4 blocks) acc8 wg/CU=%d: %.3f ms %.2f TF/s\n",wpc,t*1e3,fl/t*1e-12); }
    { double t=timeit([&]{ hipLaunchKernelGGL(k_fma<16>,grid,256,0,0,out,iters,1.0000001,1e-9);});
      double fl=(double)grid*256*iters*16*2.0; printf("v_fma_f64 acc16 wg/CU=%d: %.3f ms %.2f TF/s\n",wpc,t*1e3,fl/t*1e-12); }
    { double t=timeit([&]{ hipLaunchKernelGGL((k_mix<8,8>),grid,256,0,0,out,iters,1.0000001,1e-9);});
      double fl=(double)grid*4*iters*8*2048.0, fv=(double)grid*256*iters*8*2.0; printf("mix 8mfma+8fma wg/CU=%d: %.3f ms mfma %.2f TF/s + valu %.2f TF/s\n",wpc,t*1e3,fl/t*1e-12,fv/t*1e-12); }
    { double t=timeit([&]{ hipLaunchKernelGGL((k_mix<8,32>),grid,256,0,0,out,iters,1.0000001,1e-9);});
      double fl=(double)grid*4*iters*8*2048.0, fv=(double)grid*256*iters*32*2.0; printf("mix 8mfma+32fma wg/CU=%d: %.3f ms mfma %.2f TF/s + valu %.2f TF/s\n",wpc,t*1e3,fl/t*1e-12,fv/t*1e-12); }
    { double t=timeit([&]{ hipLaunchKernelGGL((k_mix<8,64>),grid,256,0,0,out,iters,1.0000001,1e-9);});
      double fl=(double)grid*4*iters*8*2048.0, fv=(double)grid*256*iters*64*2.0; printf("mix 8mfma+64fma wg/CU=%d: %.3f ms mfma %.2f TF/s + valu %.2f TF/s\n",wpc,t*1e3,fl/t*1e-12,fv/t*1e-12); }
  }
  return 0;
}
