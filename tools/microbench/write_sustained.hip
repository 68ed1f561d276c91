// Sustained pure-store bandwidth: 400 back-to-back fills of a 1.5 GB buffer, per-launch times from events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef unsigned u32;
__global__ void __launch_bounds__(256) fill_chunk16(u32* out, size_t ndw_per_block, u32 v){
  u32* base = out + (size_t)blockIdx.x*ndw_per_block;
  for(size_t i=(size_t)threadIdx.x*4; i+4<=ndw_per_block; i+=(size_t)blockDim.x*4) *reinterpret_cast<uint4*>(base+i)=make_uint4(v,v+1,v+2,v+3);
}
int main(){
  size_t bytes=(size_t)1536<<20; u32* d; CK(hipMalloc(&d,bytes)); size_t ndw=bytes/4;
  const int N=400; std::vector<hipEvent_t> ev(N+1); for(size_t q=0;q<ev.size();q++) CK(hipEventCreate(&ev[q]));
  for(int mode=0; mode<2; mode++){
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(ev[0]));
    for(int i=0;i<N;i++){ if(mode==0) fill_chunk16<<<2048,256>>>(d,ndw/2048,i); else CK(hipMemsetAsync(d,i,bytes,0)); CK(hipEventRecord(ev[i+1])); }
    CK(hipDeviceSynchronize());
    std::vector<float> t(N); for(int i=0;i<N;i++) CK(hipEventElapsedTime(&t[i],ev[i],ev[i+1]));
    auto avg=[&](int a,int b){double s=0; for(int i=a;i<b;i++) s+=t[i]; return s/(b-a);};
    printf("%s: first10 %.3f ms (%.2f TB/s) | 10-50 %.3f | 50-200 %.3f | 200-400 %.3f ms (%.2f TB/s) | min %.3f\n", mode? "hipMemset":"fill_chunk16",
      avg(0,10), bytes/avg(0,10)/1e9, avg(10,50), avg(50,200), avg(200,400), bytes/avg(200,400)/1e9, *std::min_element(t.begin(),t.end()));
  }
  return 0;
}
