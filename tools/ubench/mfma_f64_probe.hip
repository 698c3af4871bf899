// Probe: (1) lane layout of v_mfma_f64_4x4x4_4b_f64 incl. cbsz/abid broadcast, (2) in-kernel clock under
// fp64 MFMA load (s_memtime / s_memrealtime), random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

template<int CBSZ,int ABID,int BLGP>
__global__ void k_layout(const double* a, const double* b, double* d){
  int l=threadIdx.x; double acc=0;
  acc=__builtin_amdgcn_mfma_f64_4x4x4f64(a[l],b[l],acc,CBSZ,ABID,BLGP);
  d[l]=acc;
}
__global__ void k_layout16(const double* a, const double* b, double* d){
  int l=threadIdx.x; d4 acc={0,0,0,0};
  acc=__builtin_amdgcn_mfma_f64_16x16x4f64(a[l],b[l],acc,0,0,0);
  for(int r=0;r<4;r++) d[r*64+l]=acc[r];
}

// clock probes
template<int MODE>
__global__ void __launch_bounds__(256) k_clock(const double* in, double* out, unsigned long long* stamps, int iters){
  double a=in[threadIdx.x], b=in[256+threadIdx.x];
  d4 acc[8]; double acc1[32];
  for(int i=0;i<8;i++) acc[i]=(d4){0,0,0,0};
  for(int i=0;i<32;i++) acc1[i]=in[(i*7+threadIdx.x)&511];
  unsigned long long t0=__builtin_amdgcn_s_memtime(), r0=__builtin_amdgcn_s_memrealtime();
  for(int it=0;it<iters;++it){
    if(MODE==0){
#pragma unroll
      for(int i=0;i<8;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
    } else if(MODE==1){
#pragma unroll
      for(int i=0;i<8;i++){
        acc1[4*i+0]=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc1[4*i+0],2,0,0);
        acc1[4*i+1]=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc1[4*i+1],2,1,0);
        acc1[4*i+2]=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc1[4*i+2],2,2,0);
        acc1[4*i+3]=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc1[4*i+3],2,3,0);
      }
    } else if(MODE==2){
#pragma unroll
      for(int i=0;i<32;i++) acc1[i]=__builtin_fma(acc1[i],a,b);
    }
  }
  unsigned long long t1=__builtin_amdgcn_s_memtime(), r1=__builtin_amdgcn_s_memrealtime();
  double s=0; for(int i=0;i<8;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  for(int i=0;i<32;i++) s+=acc1[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  if(threadIdx.x==0){ stamps[2*blockIdx.x]=t1-t0; stamps[2*blockIdx.x+1]=r1-r0; }
}

int main(){
  std::vector<double> ha(64),hb(64),hd(256);
  double *a,*b,*d; CK(hipMalloc(&a,512)); CK(hipMalloc(&b,512)); CK(hipMalloc(&d,2048));
  // layout probe: A value encodes lane: 1000+lane ; B one-hot
  // Strategy: set a[l]=l+1, b = delta at lane q -> d[lane] = sum over matching (a*1)
  auto run=[&](int which,int q,std::vector<double>&out){
    for(int l=0;l<64;l++){ha[l]=l+1; hb[l]=(l==q)?1.0:0.0;}
    CK(hipMemcpy(a,ha.data(),512,hipMemcpyHostToDevice)); CK(hipMemcpy(b,hb.data(),512,hipMemcpyHostToDevice));
    if(which==0) hipLaunchKernelGGL((k_layout<0,0,0>),1,64,0,0,a,b,d);
    if(which==1) hipLaunchKernelGGL((k_layout<2,0,0>),1,64,0,0,a,b,d);
    if(which==2) hipLaunchKernelGGL((k_layout<2,1,0>),1,64,0,0,a,b,d);
    if(which==3) hipLaunchKernelGGL((k_layout<2,3,0>),1,64,0,0,a,b,d);
    if(which==4) hipLaunchKernelGGL(k_layout16,1,64,0,0,a,b,d);
    CK(hipDeviceSynchronize()); out.resize(256); CK(hipMemcpy(out.data(),d,2048,hipMemcpyDeviceToHost));
  };
  const char* names[]={"4x4x4 cbsz0","4x4x4 cbsz2 abid0","4x4x4 cbsz2 abid1","4x4x4 cbsz2 abid3","16x16x4"};
  for(int which=0;which<5;which++){
    printf("== %s: for B one-hot at lane q, list (dlane[:reg]<-Alane)\n",names[which]);
    for(int q=0;q<64;q+= (which==4?1:1)){
      std::vector<double> o; run(which,q,o);
      printf("q=%2d:",q);
      int n=(which==4)?256:64;
      for(int i=0;i<n;i++) if(o[i]!=0) { if(which==4) printf(" %d:%d<-%d",i%64,i/64,(int)o[i]-1); else printf(" %d<-%d",i,(int)o[i]-1);} 
      printf("\n");
      if(q>=20 && q<60) q+=7;
    }
  }
  // clock
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0)); int ncu=p.multiProcessorCount;
  std::vector<double> hin(512); srand(1); for(auto&x:hin) x=(rand()/(double)RAND_MAX)*2-1;
  double* din; CK(hipMalloc(&din,4096)); CK(hipMemcpy(din,hin.data(),4096,hipMemcpyHostToDevice));
  double* out; CK(hipMalloc(&out,sizeof(double)*ncu*4*256)); unsigned long long* st; CK(hipMalloc(&st,16*ncu*4));
  std::vector<unsigned long long> hs(2*ncu*4);
  for(int mode=0;mode<3;mode++) for(int wpc: {1,2,4}){
    int grid=ncu*wpc; int iters=200000;
    hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto L=[&]{ if(mode==0) hipLaunchKernelGGL(k_clock<0>,grid,256,0,0,din,out,st,iters);
                if(mode==1) hipLaunchKernelGGL(k_clock<1>,grid,256,0,0,din,out,st,iters);
                if(mode==2) hipLaunchKernelGGL(k_clock<2>,grid,256,0,0,din,out,st,iters);};
    L(); L(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); L(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    CK(hipMemcpy(hs.data(),st,16*grid,hipMemcpyDeviceToHost));
    double cyc=0,rt=0; for(int i=0;i<grid;i++){cyc+=hs[2*i]; rt+=hs[2*i+1];} cyc/=grid; rt/=grid;
    double clk=cyc/rt*100e6;
    double flops = (mode==2)? (double)grid*256*iters*32*2.0 : (double)grid*4*iters*8*2048.0;
    double per = (mode==2)? cyc/(iters*32.0) : cyc/(iters*8.0);
    printf("mode %d (%s) wg/CU=%d: %.2f ms %.2f TF/s clock %.3f GHz, shader cycles per %s per wave = %.1f\n",mode,
      mode==0?"mfma16x16x4":mode==1?"4x mfma4x4x4 bcast":"v_fma_f64 x32", wpc, ms, flops/(ms*1e-3)*1e-12, clk*1e-9, mode==2?"fma":"16x16x4-equiv", per);
  }
  return 0;
}
