// Store-flavour microbenchmark: plain vs nontemporal, 8/16 B per lane, chunk per block vs row-interleaved blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
template<int W, bool NT>
__global__ void __launch_bounds__(256) fill(u32* out, size_t ndw_per_block, u32 v){
  u32* base = out + (size_t)blockIdx.x*ndw_per_block;
  for(size_t i=(size_t)threadIdx.x*(W/4); i+W/4<=ndw_per_block; i+=(size_t)blockDim.x*(W/4)){
    if constexpr(W==16){ u32x4 x={v,v+1,v+2,v+3}; if(NT) __builtin_nontemporal_store(x,reinterpret_cast<u32x4*>(base+i)); else *reinterpret_cast<u32x4*>(base+i)=x; }
    else { u32x2 x={v,v+1}; if(NT) __builtin_nontemporal_store(x,reinterpret_cast<u32x2*>(base+i)); else *reinterpret_cast<u32x2*>(base+i)=x; }
  }
}
template<typename F> double run(F f){ const int N=60; hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int i=0;i<10;i++) f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); for(int i=0;i<N;i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/N; }
int main(){
  size_t bytes=(size_t)1536<<20; u32* d; CK(hipMalloc(&d,bytes)); size_t ndw=bytes/4;
  for(int blocks : {512, 1024, 2048, 8192, 65536}){
    size_t per=ndw/blocks;
    double a=run([&]{fill<16,false><<<blocks,256>>>(d,per,1);}), b=run([&]{fill<16,true><<<blocks,256>>>(d,per,1);});
    double c=run([&]{fill<8,false><<<blocks,256>>>(d,per,1);}), e=run([&]{fill<8,true><<<blocks,256>>>(d,per,1);});
    printf("blocks %6d: 16B plain %.2f  16B nt %.2f  8B plain %.2f  8B nt %.2f TB/s\n", blocks, bytes/a/1e9, bytes/b/1e9, bytes/c/1e9, bytes/e/1e9);
  }
  return 0;
}
