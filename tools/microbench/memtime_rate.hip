// s_memtime calibration: ticks per microsecond of wall time, and ticks per v_qsad_pk_u16_u8 issue, under a
// chip-wide qsad load (4 waves per SIMD on every CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef unsigned long long u64; typedef unsigned u32;
constexpr int ITERS = 8192;
__global__ void __launch_bounds__(256) k(const u32* in, u64* out, u64* ticks){
  u32 t = blockIdx.x*blockDim.x+threadIdx.x;
  u64 r[4]; u32 c[4];
  for(int i=0;i<4;i++){ r[i]=((u64)in[(t+i)&1023]<<32)|in[(t+7*i)&1023]; c[i]=in[(t+3+i)&1023]; }
  u64 a[8];
  for(int i=0;i<8;i++) a[i]=i;
  const u64 t0 = __builtin_amdgcn_s_memtime();
  for(int it=0;it<ITERS;it++){
#pragma unroll
    for(int rep=0; rep<2; rep++)
#pragma unroll
      for(int i=0;i<8;i++) a[i]=__builtin_amdgcn_qsad_pk_u16_u8(r[(i+rep)&3],c[(i*3+rep)&3],a[i]);
  }
  u64 s=0; for(int i=0;i<8;i++) s^=a[i];
  const u64 t1 = __builtin_amdgcn_s_memtime();
  out[t]=s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0 + (s == 12345 ? 1 : 0);
}
int main(){
  u32* in; u64* out; u64* ticks; const int blocks = 256*4;
  CK(hipMalloc(&in,4096)); CK(hipMalloc(&out,(size_t)blocks*256*8)); CK(hipMalloc(&ticks, blocks*8));
  u32 h[1024]; for(int i=0;i<1024;i++) h[i]=rand()*2654435761u; CK(hipMemcpy(in,h,4096,hipMemcpyHostToDevice));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<<<blocks,256>>>(in,out,ticks); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); k<<<blocks,256>>>(in,out,ticks); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  u64 ht[blocks]; CK(hipMemcpy(ht,ticks,blocks*8,hipMemcpyDeviceToHost));
  double mean=0, mx=0, mn=1e30; for(int i=0;i<blocks;i++){ mean+=ht[i]; if(ht[i]>mx) mx=ht[i]; if(ht[i]<mn) mn=ht[i]; } mean/=blocks; printf("ticks min %.0f max %.0f -> max/us %.1f\n", mn, mx, mx/(ms*1e3));
  const double qs = (double)ITERS*16;   // qsads per wave; 4 waves per SIMD
  printf("kernel %.3f ms; mean ticks per workgroup %.0f -> %.1f ticks/us (if a workgroup spans the kernel)\n", ms, mean, mean/(ms*1e3));
  printf("ticks per qsad issue per SIMD: %.2f ; ns per qsad per SIMD: %.2f\n", mean/(qs*4), ms*1e6/(qs*4));
  return 0;
}
