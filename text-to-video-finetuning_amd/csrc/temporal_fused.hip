// temporal_fused.hip — the temporal self-attention unit of TransformerTemporalModel as ONE launch (round 6):
//   out = x + softmax_F( (LN(x) Wq^T) (LN(x) Wk^T)^T / 8 ) (LN(x) Wv^T) Wo^T + bo
// i.e. `norm1 -> attn1 -> +residual` (and `norm2 -> attn2`, double_self_attention) of the BasicTransformerBlock inside
// TransformerTemporalModel — /root/reference/models/unet_3d_blocks.py:331-340,491-500,726-735 (temp_attentions),
// /root/reference/models/unet_3d_condition.py:147-152,407-411 (transformer_in); SURVEY.md 8(d)'s "fused temporal unit"
// (8 C^2 T + 4 T F C flops for T token rows of width C = heads * 64).
//
// Forward only (no tensors kept for a backward): the sampling path (train.py:908-958, inference.py:153-267), validation, and
// every other no-grad UNet call.  The training forward keeps its separate launches: it has to leave LN(x), q, k, v and the
// attention output in HBM for the backward anyway.
//
// Work decomposition.  The attention of a temporal layer couples only the F rows of ONE pixel, so the whole unit is local to a
// row tile: a wave owns a BLOCK of 32 row slots = PP = floor(32 / F) pixels x F frames (F = 16: two pixels, F = 24: one pixel
// and 8 idle slots), a workgroup four blocks.  Per wave and block, everything that depends on the rows lives in REGISTERS:
//   * LN(x) as the C/16 A/B fragments of v_mfma_f32_32x32x16_bf16 (lane = row slot, 8 consecutive channels per k-step),
//   * per head: q^T and k^T from the "swapped" product W n^T (lane = row slot, registers = head dims), v from n W^T
//     (lane = head dim, registers = row slots): S^T = k q^T then needs no data movement at all — the accumulators of k^T / q^T
//     ARE the A / B operands (the head-dim permutation the accumulator layout imposes is the same on both sides), the softmax
//     runs over registers + one exchange between the half-waves, P^T is the B operand and v the A operand of O^T = v^T P^T,
//   * O^T (lane = row slot, registers = head dims) is an operand of the output projection as it stands; the permutation of its
//     head dims is undone on the HOST: Wo arrives with its input index permuted inside every group of 16 (t2v_abi.h), so that
//     the matching Wo fragment is one 16-byte LDS read.  The projection is computed transposed (out^T = Wo O^T: lane = row
//     slot) and leaves through a per-wave fp32 staging tile as row-contiguous 16-byte stores, bias and residual added in fp32.
// Only WEIGHTS go through LDS: [192 x 64] chunks (q | k | v rows of one head, 64 input channels) and then [NBO*32 x 64] chunks of
// Wo stream through a four-stage ring by LDS-DMA (buffer_load ... lds, 16 B per lane, lane-linear image with the XOR swizzle
// of csrc/gemm.hip), counted s_waitcnt vmcnt + one s_barrier per chunk.  One workgroup per CU (up to ~460 registers per lane).
//
// Rooflines: MFMA-bound, 8 C^2 + 4 F C flops per row against 2.5 PFLOP/s; every workgroup reads all 4 C^2 weights once from
// its XCD's L2 (3.3 MB at C = 640) and its rows of x twice.  Measured: 0.22 - 0.24 of the peak at C = 320 / 512 (DESIGN.md 2.4 has
// the break-down by ablation: T2VTemporalFused.ablate).
//
// NOTE on the hand-issued LDS reads below: a fragment register is written by an inline-asm ds_read and becomes valid at the
// explicit s_waitcnt of tf_lds_wait — the compiler must not copy or spill it in between.  Every instantiation is checked
// against an fp32 reference by tests/test_kernels_gpu.py::test_temporal_unit_fused_forward; re-run it after ANY edit here.
#include "common.h"

namespace {

constexpr int TF_BK = 64;
constexpr int TF_STAGE_ROWS = 192;
constexpr int TF_STAGE = TF_STAGE_ROWS * TF_BK * 2;   // 24 KB
constexpr int TF_NSTAGE = 4;
constexpr int TF_NT = 256;
constexpr int TF_STG_PITCH = 36;                      // floats per row of the per-wave output staging tile (32 columns + 16 bytes)
constexpr int TF_SMEM = TF_NSTAGE * TF_STAGE + 4 * 32 * TF_STG_PITCH * 4;

template <int N>
__device__ __forceinline__ void tf_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ bf16x8 tf_pack8(const f32x16& a, int r0) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 q = {pack2bf(a[r0], a[r0 + 1]), pack2bf(a[r0 + 2], a[r0 + 3]), pack2bf(a[r0 + 4], a[r0 + 5]), pack2bf(a[r0 + 6], a[r0 + 7])};
  return __builtin_bit_cast(bf16x8, q);
}

// Fragment reads are hand-issued (ds_read_b128 through inline asm, explicit s_waitcnt lgkmcnt): the compiler's scheduler sinks
// every LDS read to just in front of the MFMA that consumes it (read -> wait -> multiply, one at a time: measured 0.11 of peak),
// and with one wave per SIMD nothing else hides that latency.  Here the reads of k-step i+1 are issued, then the MFMAs of k-step i.
// (fragment registers are 4 x u32 vectors while they are asm operands: as short8 operands the compiler may re-pack them element by
//  element — v_lshrrev / v_perm on a register whose read is still in flight; seen in csrc/gemm.hip's K-major loads, round 6)
typedef __attribute__((ext_vector_type(4))) unsigned tfreg_t;
__device__ __forceinline__ bf16x8 tf_bf(const tfreg_t& w) { return __builtin_bit_cast(bf16x8, w); }
template <int OFF>
__device__ __forceinline__ void tf_lds_read(tfreg_t& w, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void tf_lds_read_n(tfreg_t (&w)[N], unsigned addr) {
  tf_lds_read<0>(w[0], addr);
  if constexpr (N > 1) tf_lds_read<4096>(w[1], addr);
  if constexpr (N > 2) tf_lds_read<8192>(w[2], addr);
  if constexpr (N > 3) tf_lds_read<12288>(w[3], addr);
  if constexpr (N > 4) tf_lds_read<16384>(w[4], addr);
  if constexpr (N > 5) tf_lds_read<20480>(w[5], addr);
}
template <int N>
__device__ __forceinline__ void tf_lds_wait(tfreg_t (&w)[N]) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < N; ++j) asm volatile("" : "+v"(w[j]));      // the fragments are defined from here on
}

// C = heads * 64; NBO = 32-column tiles per output-projection chunk (C % (NBO * 32) == 0, NBO <= 6)
template <int C, int NBO>
__global__ __launch_bounds__(TF_NT) void temporal_fused_fwd_kernel(const T2VTemporalFused p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  warm_kernargs<(int)sizeof(T2VTemporalFused)>();
  constexpr int H = C / 64, KC = C / 64, KS = C / 16;
  constexpr int NOUT = C / (NBO * 32);
  constexpr int NA = H * KC, NTOT = NA + NOUT * KC;
  static_assert(C % (NBO * 32) == 0 && NBO <= 6 && NBO >= 1, "output chunking");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hl = lane >> 5, l32 = lane & 31;
  const int F = p.F, HW = p.HW;
  const int PP = 32 / F;
  const long long npix = (long long)p.B * HW;
  const long long blk = (long long)blockIdx.x * 4 + wave;          // this wave's block of PP pixels

  // ---- weight ring (all four waves issue; chunk g of the flat stream: g < NA -> head g / KC, k-chunk g % KC of [Wq; Wk; Wv],
  //      then output chunk (g - NA) / KC, k-chunk (g - NA) % KC of Wo)
  const int kcs = (tid & 7) ^ ((tid >> 4) & 7);                       // source 16-byte chunk of this lane's LDS slot (swizzle)
  const unsigned vbase = ((unsigned)(tid >> 3) * (unsigned)C + (unsigned)kcs * 8u) * 2u;
  const int abl = p.ablate;
  auto issue = [&](int g) {
    if (abl & 4) return;
    unsigned char* st = smem + (g & (TF_NSTAGE - 1)) * TF_STAGE;
    if (g < NA) {
      const int h = g / KC, kc = g - h * KC;
      const int so = (h * 64 * C + kc * 64) * 2;
      // (the descriptor is rebuilt from the kernel arguments at every issue: kept live across the kernel it was spilled to
      //  scratch as a VECTOR value, and every LDS-DMA became a readfirstlane waterfall loop behind an s_waitcnt vmcnt(0))
      const __amdgpu_buffer_rsrc_t srdQ = __builtin_amdgcn_make_buffer_rsrc((void*)p.wqkv, 0, 0x80000000u, 0x00020000);
#pragma unroll
      for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdQ, (__attribute__((address_space(3))) void*)(st + (tid + TF_NT * i) * 16), 16,
                                                 (int)(vbase + (unsigned)(((i >> 1) * C + (i & 1) * 32) * C * 2)), so, 0, 0);
    } else {
      const int nb = (g - NA) / KC, kc = (g - NA) - nb * KC;
      const int so = (nb * NBO * 32 * C + kc * 64) * 2;
      const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void*)p.wo, 0, 0x80000000u, 0x00020000);
#pragma unroll
      for (int i = 0; i < NBO; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdO, (__attribute__((address_space(3))) void*)(st + (tid + TF_NT * i) * 16), 16,
                                                 (int)(vbase + (unsigned)(32 * i * C * 2)), so, 0, 0);
    }
  };
  // Four stages.  sync_chunk(g), called before the first fragment of chunk g is multiplied: chunk g + 1 has landed for every wave
  // (so the LAST k-step of chunk g can already read the first fragments of chunk g + 1, ahead of the next barrier), every wave is
  // past chunk g - 1, whose stage takes chunk g + 3.  Only the loads of chunk g + 2 may still be in flight at the wait.
  auto sync_chunk = [&](int g) {
    if (g + 2 < NTOT) {
      if (g + 2 < NA) tf_wait_vmcnt<6>();
      else tf_wait_vmcnt<NBO>();
    } else {
      tf_wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (g + 3 < NTOT) issue(g + 3);
  };
  issue(0);
  if (NTOT > 1) issue(1);
  if (NTOT > 2) issue(2);

  // ---- row slots of this wave: slot s = pixel (s / F) of the block, frame s % F; global row (b, f, pixel) = (b F + f) HW + pixel
  int arow;                                                         // the row of slot l32 (-1: idle slot / beyond the last pixel)
  {
    const int pi = l32 / F, f = l32 - pi * F;
    const long long gp = blk * PP + pi;
    if (pi >= PP || gp >= npix) {
      arow = -1;
    } else {
      const long long b = gp / HW, pos = gp - b * HW;
      arow = (int)((b * F + f) * (long long)HW + pos);             // (B F HW < 2^31: checked by the entry point)
    }
  }
  unsigned kmask = 0;                                               // bit r: key slot 8(r/4)+4hl+r%4 belongs to the query slot l32's pixel
  {
    const int qp = l32 / F;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int s = 8 * (r >> 2) + 4 * hl + (r & 3);
      if (s / F == qp && qp < PP) kmask |= 1u << r;
    }
  }

  // ---- phase 0: LayerNorm of the wave's 32 rows into MFMA fragments (two-pass statistics over the register copy)
  bf16x8 nf[KS];
  {
    const bf16_t* xr = (const bf16_t*)p.x + (long long)(arow >= 0 ? arow : 0) * p.ldx + 8 * hl;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) nf[ks] = (arow >= 0 && !(abl & 2)) ? *(const bf16x8*)(xr + 16 * ks) : zero;
    float s1 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) s1 += bf2f((unsigned short)nf[ks][e]);
    s1 += __shfl_xor(s1, 32);
    const float mean = s1 * (1.f / C);
    float s2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = bf2f((unsigned short)nf[ks][e]) - mean;
        s2 += d * d;
      }
    s2 += __shfl_xor(s2, 32);
    const float rstd = rsqrtf(s2 * (1.f / C) + p.eps);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 g0 = *(const f32x4*)(p.gamma + 16 * ks + 8 * hl), g1 = *(const f32x4*)(p.gamma + 16 * ks + 8 * hl + 4);
      const f32x4 b0 = *(const f32x4*)(p.beta + 16 * ks + 8 * hl), b1 = *(const f32x4*)(p.beta + 16 * ks + 8 * hl + 4);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gm = e < 4 ? g0[e & 3] : g1[e & 3], bt = e < 4 ? b0[e & 3] : b1[e & 3];
        v[e] = (bf2f((unsigned short)nf[ks][e]) - mean) * rstd * gm + bt;
      }
      nf[ks] = pack8bf(v);
    }
  }
  // (the x / gamma / beta loads above are younger than the ring's first three chunks and have been waited for: those chunks landed)
  tf_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // fragment reads: a [192 x 64] (or [NBO*32 x 64]) stage is rows of 128 bytes, 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7);
  // tile j's fragment of this lane is 4096 j bytes further (the swizzle term does not depend on j).  The Wo image is the same: its
  // K index comes permuted from the host (see t2v_abi.h), so that the eight head dims an O^T accumulator quad-pair holds are one chunk.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (unsigned)l32 * 128u;
  const unsigned swz = (unsigned)(l32 >> 1) & 7u;
  auto frag_addr = [&](int g, int kk) -> unsigned {
    return lds0 + (unsigned)(g & (TF_NSTAGE - 1)) * (unsigned)TF_STAGE + ((((unsigned)kk * 2u + (unsigned)hl) ^ swz) << 4);
  };

  // ---- phase A: per head q^T, k^T, v -> softmax -> O^T, kept as the fragments of the output projection
  bf16x8 of[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) of[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  const float sc = p.scale;
  tfreg_t wc[6], wn[6];                                              // fragments of the k-step being multiplied / of the next one
  tf_lds_read_n<6>(wc, frag_addr(0, 0));
  tf_lds_wait<6>(wc);
  for (int h = 0; h < H; ++h) {
    f32x16 aq[2], ak[2], av[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) aq[j][r] = ak[j][r] = av[j][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int g = h * KC + kc;
      sync_chunk(g);
      if (abl & 8) continue;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bool more = kk < 3 || g + 1 < NA;
        __builtin_amdgcn_sched_barrier(0);
        if (more) tf_lds_read_n<6>(wn, kk < 3 ? frag_addr(g, kk + 1) : frag_addr(g + 1, 0));
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 a = nf[kc * 4 + kk];
#pragma unroll
        for (int j = 0; j < 2; ++j) aq[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf_bf(wc[j]), a, aq[j], 0, 0, 0);          // q^T[d][row]
#pragma unroll
        for (int j = 0; j < 2; ++j) ak[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf_bf(wc[2 + j]), a, ak[j], 0, 0, 0);      // k^T[d][row]
#pragma unroll
        for (int j = 0; j < 2; ++j) av[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tf_bf(wc[4 + j]), av[j], 0, 0, 0);      // v[row][d]
        __builtin_amdgcn_sched_barrier(0);
        if (more) tf_lds_wait<6>(wn);
#pragma unroll
        for (int j = 0; j < 6; ++j) wc[j] = wn[j];
      }
    }
    // S^T[key][query] = sum_d k[key][d] q[query][d]: lane = query slot, registers = key slots
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf_pack8(ak[dt], 8 * kb), tf_pack8(aq[dt], 8 * kb), s, 0, 0, 0);
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] *= sc;
      if ((kmask >> r) & 1u) m = fmaxf(m, s[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = ((kmask >> r) & 1u) ? __expf(s[r] - m) : 0.f;
      l += s[r];
    }
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] *= inv;
    // O^T[d][query] = sum_key v[key][d] P[query][key]
    f32x16 ot[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[dt][r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf_pack8(av[dt], 8 * kb), tf_pack8(s, 8 * kb), ot[dt], 0, 0, 0);
    }
#pragma unroll
    for (int hh = 0; hh < H; ++hh)
      if (hh == h) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) of[hh * 4 + dt * 2 + kb] = tf_pack8(ot[dt], 8 * kb);
      }
  }

  // ---- phase B: out^T[chunk][row] = Wo[chunk, :] O^T: lane = row slot, registers = four runs of four consecutive columns per tile (+ bo + x)
  float* stg = (float*)(smem + TF_NSTAGE * TF_STAGE) + wave * (32 * TF_STG_PITCH);
  int orow[2];                                                      // rows this lane stores in the output pass: slots lane/4 and lane/4 + 16
#pragma unroll
  for (int i = 0; i < 2; ++i) orow[i] = __shfl(arow, (lane >> 2) + 16 * i);
  tfreg_t vc[NBO], vn[NBO];
  tf_lds_read_n<NBO>(vc, frag_addr(NA, 0));
  tf_lds_wait<NBO>(vc);
  for (int nb = 0; nb < NOUT; ++nb) {
    f32x16 acc[NBO];
#pragma unroll
    for (int j = 0; j < NBO; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // the residual rows of this chunk's output pass are requested BEFORE its k-loop where registers allow (C <= 320): issued
    // inside the pass, every workgroup of the launch waited out their latency at the same time, with its MFMA pipes idle
    constexpr bool PREX = C <= 320;
    bf16x8 xpre[PREX ? NBO : 1][2];
    if constexpr (PREX) {
      if (!(abl & 1)) {
#pragma unroll
        for (int j = 0; j < NBO; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            if (orow[i] >= 0) xpre[j][i] = *(const bf16x8*)((const bf16_t*)p.x + (long long)orow[i] * p.ldx + nb * NBO * 32 + j * 32 + 8 * (lane & 3));
      }
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int g = NA + nb * KC + kc;
      sync_chunk(g);
      if (abl & 8) continue;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bool more = kk < 3 || g + 1 < NTOT;
        __builtin_amdgcn_sched_barrier(0);
        if (more) tf_lds_read_n<NBO>(vn, kk < 3 ? frag_addr(g, kk + 1) : frag_addr(g + 1, 0));
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 a = of[kc * 4 + kk];
#pragma unroll
        for (int j = 0; j < NBO; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf_bf(vc[j]), a, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) tf_lds_wait<NBO>(vn);
#pragma unroll
        for (int j = 0; j < NBO; ++j) vc[j] = vn[j];
      }
    }
    // Output pass through a per-wave fp32 staging tile (32 rows x 32 columns): the accumulators hold 4-column runs of 32 DIFFERENT
    // rows per lane — stored from there, every instruction wrote 16-byte pieces of 32 cache lines (17 of 52 us at C = 320).  Read
    // back row-wise, a lane owns 8 consecutive columns of a row: 16-byte residual load, 16-byte store, 64 contiguous bytes per row
    // and 16 rows per instruction; bias, residual and the product meet in fp32, one rounding.
    if (!(abl & 1)) {
#pragma unroll
      for (int j = 0; j < NBO; ++j) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          *(f32x4*)(stg + l32 * TF_STG_PITCH + 8 * q4 + 4 * hl) = f32x4{acc[j][4 * q4], acc[j][4 * q4 + 1], acc[j][4 * q4 + 2], acc[j][4 * q4 + 3]};
        const int c = nb * NBO * 32 + j * 32 + 8 * (lane & 3);
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (p.bo) {
          b0 = *(const f32x4*)(p.bo + c);
          b1 = *(const f32x4*)(p.bo + c + 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = (lane >> 2) + 16 * i;
          const f32x4 a0 = *(const f32x4*)(stg + row * TF_STG_PITCH + 8 * (lane & 3));
          const f32x4 a1 = *(const f32x4*)(stg + row * TF_STG_PITCH + 8 * (lane & 3) + 4);
          if (orow[i] >= 0) {
            bf16x8 xr8;
            if constexpr (PREX) xr8 = xpre[j][i];
            else xr8 = *(const bf16x8*)((const bf16_t*)p.x + (long long)orow[i] * p.ldx + c);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = a0[e] + b0[e] + bf2f((unsigned short)xr8[e]);
              v[4 + e] = a1[e] + b1[e] + bf2f((unsigned short)xr8[4 + e]);
            }
            *(bf16x8*)((bf16_t*)p.out + (long long)orow[i] * p.ldo + c) = pack8bf(v);
          }
        }
      }
    }
  }
}

template <int C, int NBO>
int tf_launch(const T2VTemporalFused& p, hipStream_t s) {
  constexpr int SMEM = TF_SMEM;
  auto kern = temporal_fused_fwd_kernel<C, NBO>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr = true;
  }
  const int PP = 32 / p.F;
  const long long nblk = ((long long)p.B * p.HW + PP - 1) / PP;
  const long long grid = (nblk + 3) / 4;
  T2V_LAUNCH(kern, dim3((unsigned)grid), dim3(TF_NT), SMEM, s, p);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

}  // namespace

extern "C" int t2v_temporal_fused_ok(int C, int F) {
  return (C == 64 || C == 128 || C == 320 || C == 512 || C == 640) && F >= 1 && F <= 32;
}

extern "C" int t2v_temporal_fused_fwd(const T2VTemporalFused* pp, t2v_stream_t stream) {
  const T2VTemporalFused& p = *pp;
  hipStream_t s = (hipStream_t)stream;
  T2V_CHECK_ARG(t2v_temporal_fused_ok(p.C, p.F), "t2v_temporal_fused_fwd: unsupported width %d / clip length %d", p.C, p.F);
  T2V_CHECK_ARG(p.x && p.out && p.wqkv && p.wo && p.gamma && p.beta, "t2v_temporal_fused_fwd: null operand");
  T2V_CHECK_ARG(p.B >= 1 && p.HW >= 1 && p.ldx >= p.C && p.ldo >= p.C && (p.ldx & 7) == 0 && (p.ldo & 7) == 0, "t2v_temporal_fused_fwd: bad geometry");
  T2V_CHECK_ARG((((uintptr_t)p.x | (uintptr_t)p.out | (uintptr_t)p.bo | (uintptr_t)p.wqkv | (uintptr_t)p.wo | (uintptr_t)p.gamma | (uintptr_t)p.beta) & 15) == 0,
                "t2v_temporal_fused_fwd: operands must be 16-byte aligned");
  T2V_CHECK_ARG((long long)p.B * p.F * p.HW < (1ll << 31), "t2v_temporal_fused_fwd: too many rows");
  if ((long long)p.B * p.HW == 0) return T2V_OK;
  switch (p.C) {
    case 64: return tf_launch<64, 2>(p, s);
    case 128: return tf_launch<128, 4>(p, s);
    case 320: return tf_launch<320, 5>(p, s);
    case 512: return tf_launch<512, 4>(p, s);
    case 640: return tf_launch<640, 5>(p, s);
  }
  return T2V_EINVAL;
}
