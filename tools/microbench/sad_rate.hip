// Issue-rate microbenchmark for the gfx950 SAD instructions used by the ME kernels.
// Each kernel runs a long dependent-free chain per lane; rate = lane-ops / time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

typedef unsigned long long u64; typedef unsigned u32;
constexpr int ITERS = 4096;

__global__ void __launch_bounds__(256) k_qsad(const u32* in, u64* out){
  u32 t = blockIdx.x*blockDim.x+threadIdx.x;
  u64 r0 = ((u64)in[t&1023]<<32)|in[(t+7)&1023], r1 = r0*3+1; u32 c0=in[(t+3)&1023], c1=in[(t+5)&1023];
  u64 a0=0,a1=0,a2=0,a3=0,a4=0,a5=0,a6=0,a7=0;
  for(int i=0;i<ITERS;i++){
    a0=__builtin_amdgcn_qsad_pk_u16_u8(r0,c0,a0); a1=__builtin_amdgcn_qsad_pk_u16_u8(r1,c1,a1);
    a2=__builtin_amdgcn_qsad_pk_u16_u8(r0,c1,a2); a3=__builtin_amdgcn_qsad_pk_u16_u8(r1,c0,a3);
    a4=__builtin_amdgcn_qsad_pk_u16_u8(r0,c0,a4); a5=__builtin_amdgcn_qsad_pk_u16_u8(r1,c1,a5);
    a6=__builtin_amdgcn_qsad_pk_u16_u8(r0,c1,a6); a7=__builtin_amdgcn_qsad_pk_u16_u8(r1,c0,a7);
  }
  out[t]=a0^a1^a2^a3^a4^a5^a6^a7;
}
__global__ void __launch_bounds__(256) k_sad(const u32* in, u64* out){
  u32 t = blockIdx.x*blockDim.x+threadIdx.x;
  u32 r0=in[t&1023], r1=in[(t+7)&1023], c0=in[(t+3)&1023], c1=in[(t+5)&1023];
  u32 a0=0,a1=0,a2=0,a3=0,a4=0,a5=0,a6=0,a7=0;
  for(int i=0;i<ITERS;i++){
    a0=__builtin_amdgcn_sad_u8(r0,c0,a0); a1=__builtin_amdgcn_sad_u8(r1,c1,a1);
    a2=__builtin_amdgcn_sad_u8(r0,c1,a2); a3=__builtin_amdgcn_sad_u8(r1,c0,a3);
    a4=__builtin_amdgcn_sad_u8(r0,c0,a4); a5=__builtin_amdgcn_sad_u8(r1,c1,a5);
    a6=__builtin_amdgcn_sad_u8(r0,c1,a6); a7=__builtin_amdgcn_sad_u8(r1,c0,a7);
  }
  out[t]=(u64)(a0^a1^a2^a3^a4^a5^a6^a7);
}
__global__ void __launch_bounds__(256) k_align_sad(const u32* in, u64* out){
  u32 t = blockIdx.x*blockDim.x+threadIdx.x;
  u32 r0=in[t&1023], r1=in[(t+7)&1023], c0=in[(t+3)&1023], c1=in[(t+5)&1023], sh=in[(t+9)&1023]&3;
  u32 a0=0,a1=0,a2=0,a3=0;
  for(int i=0;i<ITERS;i++){
    u32 x0=__builtin_amdgcn_alignbyte(r1,r0,sh+ (a0&1)); a0=__builtin_amdgcn_sad_u8(x0,c0,a0);
    u32 x1=__builtin_amdgcn_alignbyte(r0,r1,sh+ (a1&1)); a1=__builtin_amdgcn_sad_u8(x1,c1,a1);
    u32 x2=__builtin_amdgcn_alignbyte(r1,r0,sh+ (a2&1)); a2=__builtin_amdgcn_sad_u8(x2,c1,a2);
    u32 x3=__builtin_amdgcn_alignbyte(r0,r1,sh+ (a3&1)); a3=__builtin_amdgcn_sad_u8(x3,c0,a3);
  }
  out[t]=(u64)(a0^a1^a2^a3);
}
__global__ void __launch_bounds__(256) k_mqsad32(const u32* in, u64* out){
  u32 t = blockIdx.x*blockDim.x+threadIdx.x;
  u64 r0 = ((u64)in[t&1023]<<32)|in[(t+7)&1023], r1 = r0*3+1; u32 c0=in[(t+3)&1023]|0x01010101u, c1=in[(t+5)&1023]|0x01010101u;
  typedef u32 v4 __attribute__((ext_vector_type(4)));
  v4 a0={0,0,0,0},a1=a0,a2=a0,a3=a0;
  for(int i=0;i<ITERS;i++){
    a0=__builtin_amdgcn_mqsad_u32_u8(r0,c0,a0); a1=__builtin_amdgcn_mqsad_u32_u8(r1,c1,a1);
    a2=__builtin_amdgcn_mqsad_u32_u8(r0,c1,a2); a3=__builtin_amdgcn_mqsad_u32_u8(r1,c0,a3);
  }
  v4 s=a0^a1^a2^a3; out[t]=(u64)(s.x^s.y^s.z^s.w);
}
__global__ void __launch_bounds__(256) k_pkadd(const u32* in, u64* out){
  u32 t = blockIdx.x*blockDim.x+threadIdx.x;
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  us2 r0={(unsigned short)in[t&1023],(unsigned short)in[(t+1)&1023]}, r1={(unsigned short)in[(t+2)&1023],(unsigned short)in[(t+3)&1023]};
  us2 a0={0,0},a1=a0,a2=a0,a3=a0,a4=a0,a5=a0,a6=a0,a7=a0;
  for(int i=0;i<ITERS;i++){ a0+=r0; a1+=r1; a2+=r0; a3+=r1; a4+=r0; a5+=r1; a6+=r0; a7+=r1;
    asm volatile("" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7)); }
  us2 s=a0^a1^a2^a3^a4^a5^a6^a7; out[t]=(u64)s.x|((u64)s.y<<16);
}

template<typename K> double run(K kern, const u32* in, u64* out, int blocks){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<<<blocks,256>>>(in,out); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for(int r=0;r<5;r++) kern<<<blocks,256>>>(in,out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms/5.0;
}
int main(){
  int blocks = 256*8; // 8 blocks (32 waves) per CU
  u32* in; u64* out; CK(hipMalloc(&in,4096)); CK(hipMalloc(&out,(size_t)blocks*256*8));
  u32 h[1024]; for(int i=0;i<1024;i++) h[i]=rand()*2654435761u; CK(hipMemcpy(in,h,4096,hipMemcpyHostToDevice));
  double lanes=(double)blocks*256;
  double t;
  t=run(k_qsad,in,out,blocks);      printf("qsad_pk_u16_u8 : %.3f ms  %.2f Tlane-op/s (x16 absdiff = %.1f T absdiff/s)\n",t,lanes*ITERS*8/t/1e9, lanes*ITERS*8*16/t/1e9);
  t=run(k_sad,in,out,blocks);       printf("sad_u8         : %.3f ms  %.2f Tlane-op/s (x4 absdiff = %.1f T absdiff/s)\n",t,lanes*ITERS*8/t/1e9, lanes*ITERS*8*4/t/1e9);
  t=run(k_align_sad,in,out,blocks); printf("alignbyte+sad  : %.3f ms  %.2f Tpair/s\n",t,lanes*ITERS*4/t/1e9);
  t=run(k_mqsad32,in,out,blocks);   printf("mqsad_u32_u8   : %.3f ms  %.2f Tlane-op/s\n",t,lanes*ITERS*4/t/1e9);
  t=run(k_pkadd,in,out,blocks);     printf("pk_add_u16     : %.3f ms  %.2f Tlane-op/s\n",t,lanes*ITERS*8/t/1e9);
  return 0;
}
