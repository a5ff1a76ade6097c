// common.h — shared device helpers for the gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "t2v_abi.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;  // 16-byte chunk = 8 bf16
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(2))) short bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// fp32 -> bf16, round-to-nearest-even, through the hardware conversion (v_cvt_pk_bf16_f32 on gfx950: one instruction per
// pair instead of ~5 integer VALU ops per element — the epilogues and streaming kernels convert every output element)
typedef __attribute__((ext_vector_type(2))) float t2v_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 t2v_bf16x2;
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const t2v_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, t2v_bf16x2));
}
__device__ __forceinline__ unsigned short f2bf(float f) { return (unsigned short)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ bf16x8 pack8bf(const float (&v)[8]) {
  const unsigned a = pack2bf(v[0], v[1]), b = pack2bf(v[2], v[3]), c = pack2bf(v[4], v[5]), d = pack2bf(v[6], v[7]);
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 q = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, q);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

// Dropout epoch (t2v_set_dropout_epoch, include/t2v_abi.h): an optional DEVICE counter folded into every dropout seed.  Seeds
// reach the kernels by value, so a captured HIP graph would replay the same masks every step; with the epoch the captured
// launches carry the counter's ADDRESS and the step bumps its value (forward and backward of one step read the same value).
extern const unsigned long long* t2v_drop_epoch;      // host-side: what the next launches pass to their kernels (may be null)
// The epoch goes THROUGH the hash before it meets the seed: drop_keep() walks the element index along the lattice
// seed + (idx+1)*G, so adding epoch*G (round 2) made the mask of step e+1 the mask of step e shifted by one element.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned long long eff_seed(unsigned long long seed, const unsigned long long* epoch) {
  return epoch ? seed ^ mix64(*epoch + 0x9E3779B97F4A7C15ull) : seed;
}

// Counter-based dropout keep decision (protocol v2, round 4; restated in oracle/dropout.py).
// Elements are decided FOUR AT A TIME: quad q = idx >> 2 of the token matrix (idx = row * width + column; width % 8 == 0, so
// a quad never straddles rows) gets two 32-bit hashes built from 32-bit multiplies only, element e = idx & 3 takes one
// 16-bit field of them and is kept iff field >= round(p * 65536).  The first version hashed every element with splitmix64
// (two 64-bit multiplies = ~100 VALU cycles per element on a part whose 32-bit integer multiply is quarter rate): the
// streaming kernels that carry a mask were ALU-bound by it and a GEMM epilogue could not afford it at all.  One quad costs
// 4 multiplies + ~12 simple operations.
//   key   : z = mix64(seed) (seed = eff_seed(...), uniform per launch); s0 = low word, s1 = high word
//   a     = fmix32((uint32)q ^ s0 ^ (uint32)(q >> 32) * 0x9E3779B1)            (murmur3 finaliser: a bijection of q)
//   b     = ((a ^ s1) * 0x9E3779B1; b ^= b >> 15; b *= 0x85EBCA77; b ^= b >> 13)
//   field : e = 0: a & 0xffff, 1: a >> 16, 2: b & 0xffff, 3: b >> 16
struct DropKey {
  unsigned s0, s1, thr;
};
__device__ __forceinline__ DropKey drop_key(unsigned long long seed, float p) {
  const unsigned long long z = mix64(seed);
  DropKey k;
  k.s0 = (unsigned)z;
  k.s1 = (unsigned)(z >> 32);
  k.thr = (unsigned)(p * 65536.f + 0.5f);
  return k;
}
struct DropQuad {
  unsigned a, b;
};
__device__ __forceinline__ DropQuad drop_quad(const DropKey& k, unsigned long long q) {
  unsigned a = (unsigned)q ^ k.s0 ^ ((unsigned)(q >> 32) * 0x9E3779B1u);
  a ^= a >> 16;
  a *= 0x85EBCA6Bu;
  a ^= a >> 13;
  a *= 0xC2B2AE35u;
  a ^= a >> 16;
  unsigned b = (a ^ k.s1) * 0x9E3779B1u;
  b ^= b >> 15;
  b *= 0x85EBCA77u;
  b ^= b >> 13;
  return DropQuad{a, b};
}
// keep decision of element e (0..3) of a quad
__device__ __forceinline__ bool quad_keep(const DropKey& k, const DropQuad& h, int e) {
  const unsigned w = (e & 2) ? h.b : h.a;
  const unsigned f = (e & 1) ? (w >> 16) : (w & 0xffffu);
  return f >= k.thr;
}
// v[0..3] = keep ? v * ks : 0 for the four elements of quad q
__device__ __forceinline__ void drop_apply4(const DropKey& k, unsigned long long q, float ks, float& v0, float& v1, float& v2, float& v3) {
  const DropQuad h = drop_quad(k, q);
  v0 = (h.a & 0xffffu) >= k.thr ? v0 * ks : 0.f;
  v1 = (h.a >> 16) >= k.thr ? v1 * ks : 0.f;
  v2 = (h.b & 0xffffu) >= k.thr ? v2 * ks : 0.f;
  v3 = (h.b >> 16) >= k.thr ? v3 * ks : 0.f;
}
// eight consecutive elements starting at idx (idx % 4 == 0): two quads
__device__ __forceinline__ void drop_apply8(const DropKey& k, unsigned long long idx, float ks, float (&v)[8]) {
  drop_apply4(k, idx >> 2, ks, v[0], v[1], v[2], v[3]);
  drop_apply4(k, (idx >> 2) + 1, ks, v[4], v[5], v[6], v[7]);
}
// keep bits of eight consecutive elements starting at idx (idx % 4 == 0): bit e set = element idx + e is kept
__device__ __forceinline__ unsigned drop_bits8(const DropKey& k, unsigned long long idx) {
  const DropQuad h0 = drop_quad(k, idx >> 2), h1 = drop_quad(k, (idx >> 2) + 1);
  unsigned m = 0;
  m |= (h0.a & 0xffffu) >= k.thr ? 1u : 0u;
  m |= (h0.a >> 16) >= k.thr ? 2u : 0u;
  m |= (h0.b & 0xffffu) >= k.thr ? 4u : 0u;
  m |= (h0.b >> 16) >= k.thr ? 8u : 0u;
  m |= (h1.a & 0xffffu) >= k.thr ? 16u : 0u;
  m |= (h1.a >> 16) >= k.thr ? 32u : 0u;
  m |= (h1.b & 0xffffu) >= k.thr ? 64u : 0u;
  m |= (h1.b >> 16) >= k.thr ? 128u : 0u;
  return m;
}
// single element (slow path: tails, per-element kernels)
__device__ __forceinline__ bool drop_keep(const DropKey& k, unsigned long long idx) {
  return quad_keep(k, drop_quad(k, idx >> 2), (int)(idx & 3));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}


// ---- measurement hook (t2v_launch_timing_events, include/t2v_abi.h): a pair of events that the NEXT instrumented entry point of
// this thread fills with its kernel's own begin / end timestamps (hipExtLaunchKernelGGL) — what a kernel trace reports, without
// the dispatch gap an event pair recorded around the call also contains.  Entry points that issue two kernels give the start
// event to the first and the stop event to the last (T2V_LAUNCH_FIRST / T2V_LAUNCH_LAST).
extern thread_local hipEvent_t t2v_time_start, t2v_time_stop;
extern thread_local int t2v_time_used;
template <typename F, typename... Args>
inline void t2v_launch_timed(int which, F kern, dim3 grid, dim3 block, unsigned smem, hipStream_t s, Args... args) {
  hipEvent_t e0 = (which & 1) ? t2v_time_start : nullptr, e1 = (which & 2) ? t2v_time_stop : nullptr;
  if (e0 || e1) {
    hipExtLaunchKernelGGL(kern, grid, block, smem, s, e0, e1, 0, args...);
    if (which & 1) t2v_time_start = nullptr;
    if (which & 2) t2v_time_stop = nullptr;
    t2v_time_used = 1;
  } else {
    hipLaunchKernelGGL(kern, grid, block, smem, s, args...);
  }
}
#define T2V_LAUNCH(kern, grid, block, smem, stream, ...) t2v_launch_timed(3, kern, grid, block, smem, stream, __VA_ARGS__)
#define T2V_LAUNCH_FIRST(kern, grid, block, smem, stream, ...) t2v_launch_timed(1, kern, grid, block, smem, stream, __VA_ARGS__)
#define T2V_LAUNCH_LAST(kern, grid, block, smem, stream, ...) t2v_launch_timed(2, kern, grid, block, smem, stream, __VA_ARGS__)

void t2v_set_error(const char* fmt, ...);
// gemm_w8.hip: 8-wave one-workgroup-per-CU GEMM configurations (descriptor already validated by gemm.hip)
int t2v_gemm_w8_launch(const T2VGemm& p, int cfg, int nstep, int splits, hipStream_t s);
extern "C" int t2v_gemm_w8_configs(void);
int t2v_gemm_w8_bm(int cfg);
#define T2V_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      t2v_set_error(__VA_ARGS__);         \
      return T2V_EINVAL;                  \
    }                                     \
  } while (0)
#define T2V_CHECK_LAUNCH()                                              \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      t2v_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
      return T2V_ELAUNCH;                                               \
    }                                                                   \
  } while (0)

// Request every cache line of the kernel-argument segment in ONE batch of scalar loads.  The compiler otherwise loads each
// argument (field of a by-value descriptor) next to its first use; with descriptors of several cache lines the set-up of a
// kernel then pays one memory round trip after the other — ~3000 cycles of a 20 us GEMM launch before this was added
// (profiles/r03_w8_timeline.txt).  The later loads hit the scalar cache.
template <int NBYTES>
__device__ __forceinline__ void warm_kernargs() {
  const __attribute__((address_space(4))) unsigned* ka =
      (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned warm = 0;
#pragma unroll
  for (int q = 0; q < (NBYTES + 63) / 64; ++q) warm |= ka[q * 16];
  asm volatile("" ::"s"(warm));
}
