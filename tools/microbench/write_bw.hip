// Pure-store bandwidth microbenchmark: how fast can gfx950 write a 1.5 GB buffer?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef unsigned u32;
template<int W>  // bytes per lane per store: 4, 8, 16
__global__ void __launch_bounds__(256) fill(u32* out, size_t ndw, u32 v){
  size_t i = ((size_t)blockIdx.x*blockDim.x+threadIdx.x)*(W/4), stride=(size_t)gridDim.x*blockDim.x*(W/4);
  for(; i+W/4<=ndw; i+=stride){
    if constexpr(W==4) out[i]=v;
    else if constexpr(W==8) *reinterpret_cast<uint2*>(out+i)=make_uint2(v,v+1);
    else *reinterpret_cast<uint4*>(out+i)=make_uint4(v,v+1,v+2,v+3);
  }
}
// each block streams its own contiguous chunk (like one workgroup writing its CTU's rows)
template<int W>
__global__ void __launch_bounds__(256) fill_chunk(u32* out, size_t ndw_per_block, u32 v){
  u32* base = out + (size_t)blockIdx.x*ndw_per_block;
  for(size_t i=(size_t)threadIdx.x*(W/4); i+W/4<=ndw_per_block; i+=(size_t)blockDim.x*(W/4)){
    if constexpr(W==8) *reinterpret_cast<uint2*>(base+i)=make_uint2(v,v+1);
    else *reinterpret_cast<uint4*>(base+i)=make_uint4(v,v+1,v+2,v+3);
  }
}
template<typename F> double timeit(F f){ hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for(int i=0;i<5;i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/5; }
int main(){
  size_t bytes = (size_t)1536<<20; u32* d; CK(hipMalloc(&d,bytes)); size_t ndw=bytes/4;
  for(int blocks : {1024, 2048, 4096, 16384}){
    double t4=timeit([&]{fill<4><<<blocks,256>>>(d,ndw,1);}), t8=timeit([&]{fill<8><<<blocks,256>>>(d,ndw,1);}), t16=timeit([&]{fill<16><<<blocks,256>>>(d,ndw,1);});
    printf("grid-stride blocks=%5d: 4B %.2f TB/s  8B %.2f TB/s  16B %.2f TB/s\n", blocks, bytes/t4/1e9, bytes/t8/1e9, bytes/t16/1e9);
  }
  for(int blocks : {512, 1024, 2048, 4096}){
    size_t per=ndw/blocks;
    double t8=timeit([&]{fill_chunk<8><<<blocks,256>>>(d,per,1);}), t16=timeit([&]{fill_chunk<16><<<blocks,256>>>(d,per,1);});
    printf("chunk-per-block blocks=%5d: 8B %.2f TB/s  16B %.2f TB/s\n", blocks, bytes/t8/1e9, bytes/t16/1e9);
  }
  double tm=timeit([&]{CK(hipMemsetAsync(d,0,bytes,0));}); printf("hipMemset: %.2f TB/s\n", bytes/tm/1e9);
  return 0;
}
