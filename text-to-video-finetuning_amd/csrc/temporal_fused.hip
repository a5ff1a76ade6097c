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
//   * O^T (lane = row slot, registers = head dims) is the A operand of the output projection; the permutation of its head dims
//     is undone by reading the Wo fragment as two 8-byte halves instead of one 16-byte chunk.
// Only WEIGHTS go through LDS: [192 x 64] chunks (q | k | v rows of one head, 64 input channels) and then [NBO*32 x 64] chunks of
// Wo stream through a three-stage ring by LDS-DMA (buffer_load ... lds, 16 B per lane, lane-linear image with the XOR swizzle
// of csrc/gemm.hip), counted s_waitcnt vmcnt + one s_barrier per chunk.  One workgroup per CU (up to ~460 registers per lane).
//
// Rooflines: MFMA-bound, 8 C^2 + 4 F C flops per row against 2.5 PFLOP/s; every workgroup reads all 4 C^2 weights once from
// its XCD's L2 (3.3 MB at C = 640) and its rows of x twice.
#include "common.h"

namespace {

constexpr int TF_BK = 64;
constexpr int TF_STAGE_ROWS = 192;
constexpr int TF_STAGE = TF_STAGE_ROWS * TF_BK * 2;   // 24 KB
constexpr int TF_NSTAGE = 3;
constexpr int TF_NT = 256;

template <int N>
__device__ __forceinline__ void tf_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ bf16x8 tf_pack8(const f32x16& a, int r0) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 q = {pack2bf(a[r0], a[r0 + 1]), pack2bf(a[r0 + 2], a[r0 + 3]), pack2bf(a[r0 + 4], a[r0 + 5]), pack2bf(a[r0 + 6], a[r0 + 7])};
  return __builtin_bit_cast(bf16x8, q);
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
  const __amdgpu_buffer_rsrc_t srdQ = __builtin_amdgcn_make_buffer_rsrc((void*)p.wqkv, 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc((void*)p.wo, 0, 0x80000000u, 0x00020000);
  const int kcs = (tid & 7) ^ ((tid >> 4) & 7);                       // source 16-byte chunk of this lane's LDS slot (swizzle)
  const unsigned vbase = ((unsigned)(tid >> 3) * (unsigned)C + (unsigned)kcs * 8u) * 2u;
  auto issue = [&](int g) {
    unsigned char* st = smem + (g % TF_NSTAGE) * TF_STAGE;
    if (g < NA) {
      const int h = g / KC, kc = g - h * KC;
      const int so = (h * 64 * C + kc * 64) * 2;
#pragma unroll
      for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdQ, (__attribute__((address_space(3))) void*)(st + (tid + TF_NT * i) * 16), 16,
                                                 (int)(vbase + (unsigned)(((i >> 1) * C + (i & 1) * 32) * C * 2)), so, 0, 0);
    } else {
      const int nb = (g - NA) / KC, kc = (g - NA) - nb * KC;
      const int so = (nb * NBO * 32 * C + kc * 64) * 2;
#pragma unroll
      for (int i = 0; i < NBO; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdO, (__attribute__((address_space(3))) void*)(st + (tid + TF_NT * i) * 16), 16,
                                                 (int)(vbase + (unsigned)(32 * i * C * 2)), so, 0, 0);
    }
  };
  // chunk g has landed for every wave, the stage of chunk g - 1 is free: refill it with chunk g + 2
  auto sync_chunk = [&](int g) {
    if (g + 1 < NTOT) {
      if (g + 1 < NA) tf_wait_vmcnt<6>();
      else tf_wait_vmcnt<NBO>();
    } else {
      tf_wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (g + 2 < NTOT) issue(g + 2);
  };
  issue(0);
  if (NTOT > 1) issue(1);

  // ---- row slots of this wave: slot s = pixel (s / F) of the block, frame s % F; global row (b, f, pixel) = (b F + f) HW + pixel
  auto slot_row = [&](int s) -> int {                               // (B F HW < 2^31: checked by the entry point)
    const int pi = s / F, f = s - pi * F;
    const long long gp = blk * PP + pi;
    if (pi >= PP || gp >= npix) return -1;
    const long long b = gp / HW, pos = gp - b * HW;
    return (int)((b * F + f) * (long long)HW + pos);
  };
  const int arow = slot_row(l32);                                   // the row this lane feeds as an operand row
  int drow[16];                                                     // the rows this lane holds in an accumulator: slot 8(r/4)+4hl+r%4
  unsigned kmask = 0;                                               // bit r: key slot of register r belongs to the query slot l32's pixel
  {
    const int qp = l32 / F;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int s = 8 * (r >> 2) + 4 * hl + (r & 3);
      drow[r] = slot_row(s);
      if (s / F == qp && qp < PP) kmask |= 1u << r;
    }
  }

  // ---- phase 0: LayerNorm of the wave's 32 rows into MFMA fragments (two-pass statistics over the register copy)
  bf16x8 nf[KS];
  {
    const bf16_t* xr = (const bf16_t*)p.x + (long long)(arow >= 0 ? arow : 0) * p.ldx + 8 * hl;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) nf[ks] = arow >= 0 ? *(const bf16x8*)(xr + 16 * ks) : zero;
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

  // ---- phase A: per head q^T, k^T, v -> softmax -> O^T, kept as the A fragments of the output projection
  bf16x8 of[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) of[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  const float sc = p.scale;
  for (int h = 0; h < H; ++h) {
    f32x16 aq[2], ak[2], av[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) aq[j][r] = ak[j][r] = av[j][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      sync_chunk(h * KC + kc);
      const unsigned char* st = smem + ((h * KC + kc) % TF_NSTAGE) * TF_STAGE;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 a = nf[kc * 4 + kk];
        const int kch = kk * 2 + hl;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int row = j * 32 + l32;
          const bf16x8 w = *(const bf16x8*)(st + row * 128 + ((kch ^ ((row >> 1) & 7)) << 4));
          if (j < 2) aq[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, aq[j], 0, 0, 0);                // q^T[d][row]
          else if (j < 4) ak[j - 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, ak[j - 2], 0, 0, 0);   // k^T[d][row]
          else av[j - 4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w, av[j - 4], 0, 0, 0);              // v[row][d]
        }
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

  // ---- phase B: out[:, chunk] = O Wo[chunk, :]^T + bo + x
  const bf16_t* X = (const bf16_t*)p.x;
  bf16_t* OUT = (bf16_t*)p.out;
  for (int nb = 0; nb < NOUT; ++nb) {
    f32x16 acc[NBO];
#pragma unroll
    for (int j = 0; j < NBO; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int g = NA + nb * KC + kc;
      sync_chunk(g);
      const unsigned char* st = smem + (g % TF_NSTAGE) * TF_STAGE;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 a = of[kc * 4 + kk];
#pragma unroll
        for (int j = 0; j < NBO; ++j) {
          const int row = j * 32 + l32, sw = (row >> 1) & 7;
          // k slots 0..3 <-> head dims 16 kk + 4 hl + (0..3), slots 4..7 <-> 16 kk + 8 + 4 hl + (0..3) (the accumulator layout of O^T)
          const bf16x4 w0 = *(const bf16x4*)(st + row * 128 + (((2 * kk) ^ sw) << 4) + 8 * hl);
          const bf16x4 w1 = *(const bf16x4*)(st + row * 128 + (((2 * kk + 1) ^ sw) << 4) + 8 * hl);
          const bf16x8 w = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w, acc[j], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NBO; ++j) {
      const int c = nb * NBO * 32 + j * 32 + l32;
      const float bias = p.bo ? p.bo[c] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (drow[r] >= 0) {
          const float v = acc[j][r] + bias + bf2f(X[(long long)drow[r] * p.ldx + c]);
          OUT[(long long)drow[r] * p.ldo + c] = f2bf(v);
        }
      }
    }
  }
}

template <int C, int NBO>
int tf_launch(const T2VTemporalFused& p, hipStream_t s) {
  constexpr int SMEM = TF_NSTAGE * TF_STAGE;
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
  T2V_CHECK_ARG(p.B >= 1 && p.HW >= 1 && p.ldx >= p.C && p.ldo >= p.C && (p.ldx & 7) == 0, "t2v_temporal_fused_fwd: bad geometry");
  T2V_CHECK_ARG((((uintptr_t)p.x | (uintptr_t)p.wqkv | (uintptr_t)p.wo | (uintptr_t)p.gamma | (uintptr_t)p.beta) & 15) == 0,
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
