// qsad issue/latency microbenchmark: CHAINS independent accumulator chains per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef unsigned long long u64; typedef unsigned u32;
constexpr int ITERS = 2048;

template<int CHAINS>
__global__ void __launch_bounds__(256) k(const u32* in, u64* out){
  u32 t = blockIdx.x*blockDim.x+threadIdx.x;
  u64 r[4]; u32 c[4];
  for(int i=0;i<4;i++){ r[i]=((u64)in[(t+i)&1023]<<32)|in[(t+7*i)&1023]; c[i]=in[(t+3+i)&1023]; }
  u64 a[CHAINS];
  for(int i=0;i<CHAINS;i++) a[i]=i;
  for(int it=0;it<ITERS;it++){
#pragma unroll
    for(int rep=0; rep<16/CHAINS; rep++)
#pragma unroll
      for(int i=0;i<CHAINS;i++) a[i]=__builtin_amdgcn_qsad_pk_u16_u8(r[(i+rep)&3],c[(i*3+rep)&3],a[i]);
  }
  u64 s=0; for(int i=0;i<CHAINS;i++) s^=a[i];
  out[t]=s;
}
template<int CHAINS>
__global__ void __launch_bounds__(256) ksad(const u32* in, u64* out){
  u32 t = blockIdx.x*blockDim.x+threadIdx.x;
  u32 r[4]; u32 c[4];
  for(int i=0;i<4;i++){ r[i]=in[(t+7*i)&1023]; c[i]=in[(t+3+i)&1023]; }
  u32 a[CHAINS];
  for(int i=0;i<CHAINS;i++) a[i]=i;
  for(int it=0;it<ITERS;it++){
#pragma unroll
    for(int rep=0; rep<16/CHAINS; rep++)
#pragma unroll
      for(int i=0;i<CHAINS;i++) a[i]=__builtin_amdgcn_sad_u8(r[(i+rep)&3],c[(i*3+rep)&3],a[i]);
  }
  u32 s=0; for(int i=0;i<CHAINS;i++) s^=a[i];
  out[t]=s;
}
template<typename K> double run(K kern, const u32* in, u64* out, int blocks, int threads){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks,threads>>>(in,out); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for(int r=0;r<3;r++) kern<<<blocks,threads>>>(in,out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms/3.0;
}
int main(){
  u32* in; u64* out; CK(hipMalloc(&in,4096)); CK(hipMalloc(&out,(size_t)256*8*256*8));
  u32 h[1024]; for(int i=0;i<1024;i++) h[i]=rand()*2654435761u; CK(hipMemcpy(in,h,4096,hipMemcpyHostToDevice));
  // waves per SIMD = wps: one block of 256 threads = 1 wave per SIMD on one CU; grid = 256 CUs * wps
  for(int wps=1; wps<=8; wps*=2){
    int blocks=256*wps;
    double t1=run(k<1>,in,out,blocks,256), t2=run(k<2>,in,out,blocks,256), t4=run(k<4>,in,out,blocks,256), t8=run(k<8>,in,out,blocks,256), t16=run(k<16>,in,out,blocks,256);
    double n=(double)ITERS*16;  // qsads per wave
    // cycles per qsad per SIMD at 2.4 GHz = t * 2.4e9 / (n * wps)
    printf("qsad wps=%d  cyc/inst/SIMD: chains1 %.1f  chains2 %.1f  chains4 %.1f  chains8 %.1f chains16 %.1f\n", wps,
      t1*1e-3*2.4e9/(n*wps), t2*1e-3*2.4e9/(n*wps), t4*1e-3*2.4e9/(n*wps), t8*1e-3*2.4e9/(n*wps), t16*1e-3*2.4e9/(n*wps));
  }
  for(int wps=1; wps<=8; wps*=2){
    int blocks=256*wps;
    double t1=run(ksad<1>,in,out,blocks,256), t2=run(ksad<2>,in,out,blocks,256), t4=run(ksad<4>,in,out,blocks,256), t8=run(ksad<8>,in,out,blocks,256);
    double n=(double)ITERS*16;
    printf("sad_u8 wps=%d  cyc/inst/SIMD: chains1 %.1f  chains2 %.1f  chains4 %.1f  chains8 %.1f\n", wps,
      t1*1e-3*2.4e9/(n*wps), t2*1e-3*2.4e9/(n*wps), t4*1e-3*2.4e9/(n*wps), t8*1e-3*2.4e9/(n*wps));
  }
  return 0;
}
