// probes/dma_bw.hip — how fast can the CUs pull operand tiles into LDS?  (evidence for DESIGN §2.1: the ceiling of the
// L2 -> LDS operand path that the GEMM family runs against.)
//   hipcc --offload-arch=gfx950 -O3 -o probes/dma_bw probes/dma_bw.hip && probes/dma_bw
// Every workgroup (512 threads) streams 16-byte chunks with `buffer_load_dwordx4 ... lds` (the GEMM loader's instruction) into
// a ring of LDS slots, `depth` 1-KiB wave-instructions in flight per wave, nothing else in the loop.  Patterns:
//   shared  : all workgroups read the SAME `span` bytes (weight-tile like; L2 hits after the first touch)
//   private : workgroup b reads its own `span` bytes over and over (activation-panel like)
//   rows    : like private, but each wave-instruction gathers 8 rows of 128 B at a row pitch (the A-tile access shape)
// Reported: aggregate TB/s and bytes/clk/CU at the measured time (2.1 GHz assumed for the per-clk figure).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const unsigned char* base, long long wg_stride, unsigned span, int iters,
                                                  int pitch, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* p = base + (long long)blockIdx.x * wg_stride;
  __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x80000000u, 0x00020000);
  // per-lane offset inside one 1-KiB piece: contiguous (mode 0/1) or 8 rows x 128 B at `pitch` (mode 2)
  unsigned loff = mode == 2 ? (unsigned)((lane >> 3) * pitch + (lane & 7) * 16) : (unsigned)lane * 16u;
  const unsigned piece = mode == 2 ? (unsigned)(8 * pitch) : 1024u;     // address advance per wave-instruction
  unsigned off = (unsigned)wave * piece;                                 // scalar walk through the span
  const unsigned step = 8u * piece;
  unsigned char* slot = smem + wave * (DEPTH * 1024);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(slot + d * 1024 + lane * 16), 16,
                                               (int)loff, (int)off, 0, 0);
      off += step;
      if (off >= span) off -= span;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (smem[tid] == 123 && iters < 0) ((unsigned char*)base)[0] = 1;     // keep the LDS image live
}

template <int DEPTH>
float run(const unsigned char* buf, int wgs, long long wg_stride, unsigned span, int iters, int pitch, int mode) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int smem = 8 * DEPTH * 1024;
  hipLaunchKernelGGL(dma_kernel<DEPTH>, dim3(wgs), dim3(512), smem, 0, buf, wg_stride, span, 8, pitch, mode);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(dma_kernel<DEPTH>, dim3(wgs), dim3(512), smem, 0, buf, wg_stride, span, iters, pitch, mode);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  const size_t total = (size_t)1 << 30;
  unsigned char* buf;
  CHECK(hipMalloc(&buf, total));
  CHECK(hipMemset(buf, 1, total));
  struct Case { const char* name; int wgs; long long wg_stride; unsigned span; int pitch; int mode; };
  std::vector<Case> cases = {
      {"shared 2 MB, 256 WGs", 256, 0, 2u << 20, 0, 0},
      {"shared 2 MB, 64 WGs", 64, 0, 2u << 20, 0, 0},
      {"shared 256 KB, 256 WGs", 256, 0, 256u << 10, 0, 0},
      {"private 64 KB each, 256 WGs (16 MB total)", 256, 64 << 10, 64u << 10, 0, 0},
      {"private 256 KB each, 256 WGs (64 MB total: beyond L2, in MALL)", 256, 256 << 10, 256u << 10, 0, 0},
      {"private 2 MB each, 256 WGs (512 MB: HBM)", 256, 2 << 20, 2u << 20, 0, 0},
      {"rows pitch 640 B, private 80 KB, 256 WGs", 256, 80 << 10, 80u << 10, 640, 2},
      {"rows pitch 1024 B, private 128 KB, 256 WGs", 256, 128 << 10, 128u << 10, 1024, 2},
      {"rows pitch 5760 B (weights K=2880), shared 1.9 MB, 256 WGs", 256, 0, 336u * 5760u, 5760, 2},
      {"rows pitch 640 B, private 80 KB, 64 WGs", 64, 80 << 10, 80u << 10, 640, 2},
      // row-pitch sweep, L2-resident and too big for the 32 KB L1s: every workgroup walks the SAME 2048-row panel (first 128 B
      // of each row = one K slice), 8 rows per wave-instruction — do power-of-two-ish pitches pile the rows onto few L2 channels?
      {"panel 2048 rows, pitch  640 B (C=320)", 256, 0, 2048u * 640u, 640, 2},
      {"panel 2048 rows, pitch 1280 B (C=640)", 256, 0, 2048u * 1280u, 1280, 2},
      {"panel 2048 rows, pitch 2560 B (C=1280)", 256, 0, 2048u * 2560u, 2560, 2},
      {"panel 2048 rows, pitch 2688 B (C=1280 + 64 pad)", 256, 0, 2048u * 2688u, 2688, 2},
      {"panel 2048 rows, pitch 1024 B (C=512)", 256, 0, 2048u * 1024u, 1024, 2},
      {"panel 2048 rows, pitch 1152 B (C=512 + 64 pad)", 256, 0, 2048u * 1152u, 1152, 2},
      {"panel 1296 rows, pitch 23040 B (weights K=11520)", 256, 0, 1296u * 23040u, 23040, 2},
      {"panel 1296 rows, pitch 23168 B (K=11520 + 64 pad)", 256, 0, 1296u * 23168u, 23168, 2},
      {"panel 1296 rows, pitch 20480 B (weights K=10240)", 256, 0, 1296u * 20480u, 20480, 2},
      {"panel 1296 rows, pitch 20608 B (K=10240 + 64 pad)", 256, 0, 1296u * 20608u, 20608, 2},
  };
  for (const Case& c : cases) {
    for (int depth : {4, 16}) {
      const int iters = depth == 4 ? 4096 : 1024;
      float ms = depth == 4 ? run<4>(buf, c.wgs, c.wg_stride, c.span, iters, c.pitch, c.mode)
                            : run<16>(buf, c.wgs, c.wg_stride, c.span, iters, c.pitch, c.mode);
      double bytes = (double)c.wgs * 8.0 * iters * depth * 1024.0;
      double tbs = bytes / (ms * 1e-3) / 1e12;
      printf("%-64s depth %2d: %8.3f ms  %6.2f TB/s  %5.1f B/clk/CU(active)\n", c.name, depth, ms, tbs,
             bytes / c.wgs / (ms * 1e-3) / 2.1e9);
      fflush(stdout);
    }
  }
  return 0;
}
