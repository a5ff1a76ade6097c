// Hardware-fact probe for gfx950: MFMA fragment layouts + ds_read_b64_tr_b16 semantics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
static inline unsigned short f2bf(float f){ unsigned u; memcpy(&u,&f,4); u += 0x7fff + ((u>>16)&1); return u>>16; }
static inline float bf2f(unsigned short h){ unsigned u = ((unsigned)h)<<16; float f; memcpy(&f,&u,4); return f; }

// D[32x32] = A[32x16] * B[16x32]; A row-major [32][16], Bt row-major [32(n)][16(k)]
__global__ void k_mfma32(const unsigned short* A, const unsigned short* Bt, float* D){
  int l = threadIdx.x;
  bf16x8 a = *(const bf16x8*)(A + (l&31)*16 + 8*(l>>5));
  bf16x8 b = *(const bf16x8*)(Bt + (l&31)*16 + 8*(l>>5));
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0,0,0);
  for(int r=0;r<16;r++){ int row=(r&3)+8*(r>>2)+4*(l>>5); int col=l&31; D[row*32+col]=c[r]; }
}
// D[16x16] = A[16x32]*B[32x16]
__global__ void k_mfma16(const unsigned short* A, const unsigned short* Bt, float* D){
  int l = threadIdx.x;
  bf16x8 a = *(const bf16x8*)(A + (l&15)*32 + 8*(l>>4));
  bf16x8 b = *(const bf16x8*)(Bt + (l&15)*32 + 8*(l>>4));
  f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0,0,0);
  for(int r=0;r<4;r++){ int row=(l>>4)*4+r; int col=l&15; D[row*16+col]=c[r]; }
}
// tr read: LDS holds u16 value = index (0..1023). lane supplies byte address addr[l]; returns 4 u16.
__global__ void k_tr(const int* addr, unsigned short* out){
  __shared__ __attribute__((aligned(16))) unsigned short lds[2048];
  int l = threadIdx.x;
  for(int i=l;i<2048;i+=64) lds[i]=(unsigned short)i;
  __syncthreads();
  unsigned a = (unsigned)(size_t)(&lds[0]) + addr[l];
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for(int j=0;j<4;j++) out[l*4+j]=(unsigned short)v[j];
}
int main(){
  // --- mfma32
  { std::vector<unsigned short> A(32*16),B(32*16); std::vector<float> Af(32*16),Bf(32*16);
    srand(1); for(int i=0;i<32*16;i++){ float x=(rand()%17-8)/4.f; A[i]=f2bf(x); Af[i]=bf2f(A[i]); float y=(rand()%13-6)/3.f; B[i]=f2bf(y); Bf[i]=bf2f(B[i]); }
    unsigned short *dA,*dB; float* dD; hipMalloc(&dA,A.size()*2); hipMalloc(&dB,B.size()*2); hipMalloc(&dD,32*32*4);
    hipMemcpy(dA,A.data(),A.size()*2,hipMemcpyHostToDevice); hipMemcpy(dB,B.data(),B.size()*2,hipMemcpyHostToDevice);
    k_mfma32<<<1,64>>>(dA,dB,dD); std::vector<float> D(32*32); hipMemcpy(D.data(),dD,32*32*4,hipMemcpyDeviceToHost);
    double me=0; for(int i=0;i<32;i++)for(int j=0;j<32;j++){ double s=0; for(int k=0;k<16;k++) s+=Af[i*16+k]*Bf[j*16+k]; me=fmax(me,fabs(s-D[i*32+j])); }
    printf("MFMA32x32x16 maxerr %g %s\n",me, me<1e-3?"PASS":"FAIL"); }
  { std::vector<unsigned short> A(16*32),B(16*32); std::vector<float> Af(16*32),Bf(16*32);
    srand(2); for(int i=0;i<16*32;i++){ float x=(rand()%17-8)/4.f; A[i]=f2bf(x); Af[i]=bf2f(A[i]); float y=(rand()%13-6)/3.f; B[i]=f2bf(y); Bf[i]=bf2f(B[i]); }
    unsigned short *dA,*dB; float* dD; hipMalloc(&dA,A.size()*2); hipMalloc(&dB,B.size()*2); hipMalloc(&dD,16*16*4);
    hipMemcpy(dA,A.data(),A.size()*2,hipMemcpyHostToDevice); hipMemcpy(dB,B.data(),B.size()*2,hipMemcpyHostToDevice);
    k_mfma16<<<1,64>>>(dA,dB,dD); std::vector<float> D(16*16); hipMemcpy(D.data(),dD,16*16*4,hipMemcpyDeviceToHost);
    double me=0; for(int i=0;i<16;i++)for(int j=0;j<16;j++){ double s=0; for(int k=0;k<32;k++) s+=Af[i*32+k]*Bf[j*32+k]; me=fmax(me,fabs(s-D[i*16+j])); }
    printf("MFMA16x16x32 maxerr %g %s\n",me, me<1e-3?"PASS":"FAIL"); }
  // --- tr: experiment 1: lane l supplies addr of element 4*l (8B chunk l): contiguous 64 chunks
  for(int exp=0; exp<3; exp++){
    std::vector<int> addr(64);
    for(int l=0;l<64;l++){
      if(exp==0) addr[l]=8*l;                         // chunk l
      if(exp==1) addr[l]=((l&15)/4)*128 + (l&3)*8 + (l>>4)*32;   // [4 rows stride 64 elems][16 cols] per group, groups offset 16 cols
      if(exp==2) addr[l]=8*(63-l);
    }
    int* dAd; unsigned short* dO; hipMalloc(&dAd,256); hipMalloc(&dO,64*4*2); hipMemcpy(dAd,addr.data(),256,hipMemcpyHostToDevice);
    k_tr<<<1,64>>>(dAd,dO); std::vector<unsigned short> o(256); hipMemcpy(o.data(),dO,512,hipMemcpyDeviceToHost);
    printf("TR exp %d\n",exp);
    for(int l=0;l<64;l++){ printf(" l%02d a=%4d(el %4d): %4d %4d %4d %4d\n",l,addr[l],addr[l]/2,o[l*4],o[l*4+1],o[l*4+2],o[l*4+3]); }
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p,0); printf("dev %s CUs %d clock %d MHz mem %zu GB\n",p.gcnArchName,p.multiProcessorCount,p.clockRate/1000,p.totalGlobalMem>>30);
  return 0;
}
