// Checks the assumed lane maps of v_mfma_i32_32x32x32_i8 with exact integer data:
//   A: lane l (r=l&31,h=l>>5) byte j of the 128-bit operand = A[r][16h+j];  B: = B[16h+j][r];
//   D: reg g of lane l = D[(g&3)+8(g>>2)+4h][r].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const int8_t* A, const int8_t* B, int* D){
  int l=threadIdx.x, r=l&31, h=l>>5;
  v4i a, b;
  a = *reinterpret_cast<const v4i*>(A + r*32 + 16*h);
  int8_t tmp[16]; for(int j=0;j<16;j++) tmp[j]=B[(16*h+j)*32 + r];
  b = *reinterpret_cast<v4i*>(tmp);
  v16i c={0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a,b,c,0,0,0);
  for(int g=0;g<16;g++) D[((g&3)+8*(g>>2)+4*h)*32 + r] = c[g];
}
int main(){
  int8_t hA[1024], hB[1024]; int hD[1024];
  srand(3); for(int i=0;i<1024;i++){ hA[i]=(int8_t)(rand()%256-128); hB[i]=(int8_t)(rand()%256-128); }
  int8_t *dA,*dB; int* dD; CK(hipMalloc(&dA,1024)); CK(hipMalloc(&dB,1024)); CK(hipMalloc(&dD,4096));
  CK(hipMemcpy(dA,hA,1024,hipMemcpyHostToDevice)); CK(hipMemcpy(dB,hB,1024,hipMemcpyHostToDevice));
  k<<<1,64>>>(dA,dB,dD); CK(hipMemcpy(hD,dD,4096,hipMemcpyDeviceToHost));
  int bad=0; for(int m=0;m<32;m++) for(int n=0;n<32;n++){ int s=0; for(int kk=0;kk<32;kk++) s+=hA[m*32+kk]*hB[kk*32+n]; if(s!=hD[m*32+n]) bad++; }
  printf("mfma_i32_32x32x32_i8 lane-map check: %d mismatches of 1024\n", bad);
  return bad!=0;
}
