// probes/mfma_mix.hip — what MFMA rate does a CU sustain next to the GEMM's LDS and LDS-DMA traffic when NOTHING synchronises
// the waves?  (Evidence for DESIGN §2.1: separates "the schedule leaves the matrix pipes idle" from "the instruction mix itself
// cannot run faster".)  One 512-thread workgroup per CU, every wave loops over
//     R x ds_read_b128 (fragments: random bf16 in LDS)  ->  F x v_mfma_f32_32x32x16_bf16  (+ D x buffer_load_dwordx4 ... lds)
// with the per-k16-step mix of the 256x256 tile (8 MFMA, 6 fragment reads, 1 DMA piece per wave), no barrier, no vmcnt wait
// inside the loop (the DMA pieces land in a scratch ring nobody reads).
//   hipcc --offload-arch=gfx950 -O3 -o probes/mfma_mix probes/mfma_mix.hip && probes/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int FM, int FN, bool READS, int DMA, bool PREFETCH>
__global__ __launch_bounds__(512) void mix_kernel(const unsigned char* src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // fill 64 KB of LDS with the (random) source bytes
  for (int i = tid; i < 4096; i += 512) ((uint4*)smem)[i] = ((const uint4*)src)[i];
  __syncthreads();
  __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x80000000u, 0x00020000);
  f32x16 acc[FM][FN];
  for (int i = 0; i < FM; ++i)
    for (int j = 0; j < FN; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // conflict-free fragment addressing of the GEMM image: row*128 + ((chunk ^ swz) << 4)
  const unsigned qsw = (unsigned)(((lane >> 5) ^ ((lane >> 1) & 7)) << 4);
  const unsigned rowb = (unsigned)((wave & 3) * 64 + (lane & 31)) * 128u;
  bf16x8 af[2][FM], bfr[2][FN];
  unsigned off = (unsigned)wave * 1024u;
  unsigned char* ring = smem + 65536 + wave * 4096;
  auto load = [&](int set, int kk) {
    const unsigned ko = qsw ^ ((unsigned)(kk & 3) << 5);
#pragma unroll
    for (int i = 0; i < FM; ++i) af[set][i] = *(const bf16x8*)(smem + ((rowb + i * 4096 + ko) & 0xffff));
#pragma unroll
    for (int j = 0; j < FN; ++j) bfr[set][j] = *(const bf16x8*)(smem + ((rowb + 8192 + j * 4096 + ko) & 0xffff));
  };
  if (!READS || PREFETCH) load(0, 0);
  if (!READS) load(1, 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cur = kk & 1;
      if (READS) {
        if (PREFETCH) load(cur ^ 1, kk + 1);
        else load(cur, kk);
      }
      if (DMA > 0 && (kk % (4 / DMA)) == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(ring + (kk & 3) * 1024 + lane * 16), 16,
                                                 lane * 16, (int)off, 0, 0);
        off += 8192u;
        if (off >= (2u << 20)) off -= (2u << 20);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < FM; ++i)
    for (int j = 0; j < FN; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;
}

template <int FM, int FN, bool READS, int DMA, bool PREFETCH>
void run(const char* name, const unsigned char* src, float* out) {
  auto kern = mix_kernel<FM, FN, READS, DMA, PREFETCH>;
  const int smem = 65536 + 8 * 4096;
  CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  const int iters = 2000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), smem, 0, src, out, 50);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), smem, 0, src, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = 256.0 * 8 * iters * 4.0 * FM * FN * 2.0 * 32 * 32 * 16;
  const double mfma_per_simd = 2.0 * iters * 4.0 * FM * FN;       // two waves per SIMD
  printf("%-58s %8.3f ms  %7.1f TF/s  %5.1f %% of 2.5 PF   %5.1f ns per MFMA and SIMD\n", name, ms, flops / ms / 1e9,
         flops / ms / 1e9 / 25.0, ms * 1e6 / mfma_per_simd);
  fflush(stdout);
}

int main() {
  unsigned char* src;
  float* out;
  CHECK(hipMalloc(&src, 4 << 20));
  CHECK(hipMalloc(&out, 64));
  std::vector<unsigned short> h(2 << 20);
  srand(1);
  for (auto& v : h) {                                  // random bf16 in about [-2, 2): full-range mantissas and signs
    unsigned short m = rand() & 0x7f, s = (rand() & 1) << 15, e = 124 + (rand() & 3);
    v = s | (e << 7) | m;
  }
  CHECK(hipMemcpy(src, h.data(), 4 << 20, hipMemcpyHostToDevice));
  run<2, 4, false, 0, false>("wave 64x128: MFMA only (operands in registers)", src, out);
  run<2, 4, true, 0, false>("wave 64x128: + 6 fragment reads per k16, read-then-multiply", src, out);
  run<2, 4, true, 0, true>("wave 64x128: + fragment reads prefetched one k16 ahead", src, out);
  run<2, 4, true, 1, true>("wave 64x128: + prefetch + 1 LDS-DMA piece per 4 k16 (x0.25)", src, out);
  run<2, 4, true, 4, true>("wave 64x128: + prefetch + 1 LDS-DMA piece per k16 (256x256 mix)", src, out);
  run<2, 4, true, 4, false>("wave 64x128: read-then-multiply + 1 LDS-DMA piece per k16", src, out);
  run<2, 6, true, 4, false>("wave 64x192: read-then-multiply + 1 LDS-DMA piece per k16", src, out);
  run<1, 6, true, 4, true>("wave 32x192: prefetch + 1 LDS-DMA piece per k16", src, out);
  run<2, 2, true, 4, true>("wave 64x64: prefetch + 1 LDS-DMA piece per k16", src, out);
  return 0;
}
