// Streaming ceilings of one gfx950 for the traffic mixes of the batch kernels: read-only (block metrics: two input
// streams, a 4-byte result per block), copy, and 2 reads : 3 writes (transform-unit chain: source + prediction in,
// 16-bit levels + reconstruction out). 16 bytes per lane per access, grid-stride, 1 GiB of traffic per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef unsigned u32;
__global__ void __launch_bounds__(256) read2(const uint4* a, const uint4* b, u32* out, size_t n){
  size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x, st=(size_t)gridDim.x*blockDim.x; u32 s=0;
  for(; i<n; i+=st){ uint4 x=a[i], y=b[i]; s+=x.x^y.x; s+=x.y^y.y; s+=x.z^y.z; s+=x.w^y.w; }
  if(s==0x12345678u) out[0]=s;
}
__global__ void __launch_bounds__(256) copy1(const uint4* a, uint4* o, size_t n){
  size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x, st=(size_t)gridDim.x*blockDim.x;
  for(; i<n; i+=st) o[i]=a[i];
}
// per 16 input bytes of each of two streams: 32 bytes of "levels" + 16 bytes of "reconstruction"
__global__ void __launch_bounds__(256) mix23(const uint4* a, const uint4* b, uint4* lv, uint4* rec, size_t n){
  size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x, st=(size_t)gridDim.x*blockDim.x;
  for(; i<n; i+=st){ uint4 x=a[i], y=b[i]; lv[2*i]=make_uint4(x.x-y.x,x.y-y.y,x.z-y.z,x.w-y.w); lv[2*i+1]=make_uint4(x.x+y.x,x.y+y.y,x.z+y.z,x.w+y.w);
    rec[i]=make_uint4(x.x^y.x,x.y^y.y,x.z^y.z,x.w^y.w); }
}
template<typename F> double timeit(F f){ hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for(int i=0;i<10;i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/10*1e-3; }
int main(){
  const size_t unit=(size_t)256<<20;            // bytes per input stream
  uint4 *a,*b,*lv,*rec; u32* out; CK(hipMalloc(&a,unit)); CK(hipMalloc(&b,unit)); CK(hipMalloc(&lv,2*unit)); CK(hipMalloc(&rec,unit)); CK(hipMalloc(&out,64));
  CK(hipMemset(a,1,unit)); CK(hipMemset(b,2,unit));
  const size_t n=unit/16;
  for(int blocks : {2048, 8192, 32768}){
    double tr=timeit([&]{read2<<<blocks,256>>>(a,b,out,n);}), tc=timeit([&]{copy1<<<blocks,256>>>(a,rec,n);}), tm=timeit([&]{mix23<<<blocks,256>>>(a,b,lv,rec,n);});
    printf("blocks=%6d: read-only %.2f TB/s   copy (1:1) %.2f TB/s   2 reads : 3 writes %.2f TB/s\n", blocks, 2*unit/tr/1e12, 2*unit/tc/1e12, 5*unit/tm/1e12);
  }
  return 0;
}
