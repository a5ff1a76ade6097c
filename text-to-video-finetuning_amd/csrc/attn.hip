// attn.hip — scaled-dot-product attention core (head_dim 64, no mask), forward + backward, for gfx950.
//
// One wave64 owns a 32-row query block (fwd, dQ) or a 32-row key block (dK/dV) and streams the other
// sequence in 32-row tiles; everything stays in registers (flash-style online softmax), MFMA 32x32x16 bf16:
//   fwd : S^T = K Q^T (lane = query -> row max/sum are lane-local + one cross-half shuffle),
//         O^T += V^T P^T  where P^T is fed straight from the S^T accumulator registers as the B operand and
//         V^T fragments come from a wave-private LDS tile through ds_read_b64_tr_b16 (hardware transpose) — each tile
//         row is loaded ONCE, as MFMA operand fragments, and the same registers are written to the LDS tile; the next
//         tile's rows are in flight while the current tile is processed;
//         the MFMA k-slot <-> key permutation implied by the accumulator layout is applied to both operands.
//   dQ  : same layout, dS^T from registers, K^T via transpose reads.        (also produces delta = rowsum(dO*O))
//   dKdV: S = Q K^T (lane = key), dV^T += dO^T P, dK^T += Q^T dS with Q^T/dO^T via transpose reads.
// Strided operands (t2v_abi.h, T2VAttnOperand) let the same kernel serve temporal attention over frames
// (sequence stride = H*W*C), per-frame spatial attention and text cross-attention without any permute copy.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int LDT = 104;  // LDS tile row stride (elements): 64 + 40 pad — row-per-lane 16-byte writes and the transposed reads
                          // both spread over the banks (52-dword row pitch)

__device__ __forceinline__ long long op_off(const T2VAttnOperand& o, int b, int h) {
  return (long long)(b / o.bdiv) * o.bstride_hi + (long long)(b % o.bdiv) * o.bstride_lo + h * 64;
}
__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ bf16x8 ldg8(const bf16_t* p, bool ok) {
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return ok ? *(const bf16x8*)p : z;
}
// the lane's four 16-byte pieces of tile row (r0 + lane&31): pieces 2*kd + (lane>>5) — exactly the MFMA operand fragments
// of X (rows = tile rows); the same registers are written to LDS for the transposed reads (no second global load)
__device__ __forceinline__ void load_row_frags(bf16x8 (&f)[4], const bf16_t* base, long long ss, int r0, int rmax, int lane) {
  const int row = r0 + (lane & 31), hi = lane >> 5;
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) f[kd] = ldg8(base + (long long)row * ss + 16 * kd + 8 * hi, row < rmax);
}
__device__ __forceinline__ void frags_to_lds(bf16_t* s, const bf16x8 (&f)[4], int lane) {
  const int rr = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) *(bf16x8*)(s + rr * LDT + 16 * kd + 8 * hi) = f[kd];
}
// A-operand fragment of X^T (rows = 32 feature columns [32*fm, 32*fm+32), k = the 16 tile rows of k-step kk,
// permuted exactly like the accumulator-register order) from an LDS tile X[row][feature]
__device__ __forceinline__ bf16x8 trfrag(const bf16_t* s, int fm, int kk, int lane) {
  int gq = lane >> 4, li = lane & 15;
  int k0 = 16 * kk + 4 * (gq >> 1) + (li >> 2);
  int col = 32 * fm + 16 * (gq & 1) + 4 * (li & 3);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(s + k0 * LDT + col));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(s + (k0 + 8) * LDT + col));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int kk) {      // four v_cvt_pk_bf16_f32, no per-element assembly
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 q = {pack2bf(v[8 * kk], v[8 * kk + 1]), pack2bf(v[8 * kk + 2], v[8 * kk + 3]), pack2bf(v[8 * kk + 4], v[8 * kk + 5]),
                   pack2bf(v[8 * kk + 6], v[8 * kk + 7])};
  return __builtin_bit_cast(bf16x8, q);
}
// store the lane's 32 values of a transposed accumulator pair (lane = row `row`, regs = features) as bf16
__device__ __forceinline__ void store_rowT(bf16_t* base, long long ss, int row, const f32x16& a0, const f32x16& a1, float mul,
                                           int hi) {
#pragma unroll
  for (int fm = 0; fm < 2; ++fm) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x16& a = fm ? a1 : a0;
      uint2 w;
      w.x = pack2bf(a[4 * rq] * mul, a[4 * rq + 1] * mul);
      w.y = pack2bf(a[4 * rq + 2] * mul, a[4 * rq + 3] * mul);
      *(uint2*)(base + (long long)row * ss + 32 * fm + 8 * rq + 4 * hi) = w;
    }
  }
}

__global__ __launch_bounds__(64) void attn_fwd_kernel(const T2VAttn p) {
  warm_kernargs<(int)sizeof(T2VAttn)>();
  __shared__ __attribute__((aligned(16))) bf16_t sV[32 * LDT];
  const int lane = threadIdx.x, hi = lane >> 5, l31 = lane & 31;
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const bf16_t* Q = (const bf16_t*)p.q.ptr + op_off(p.q, b, h);
  const bf16_t* K = (const bf16_t*)p.k.ptr + op_off(p.k, b, h);
  const bf16_t* V = (const bf16_t*)p.v.ptr + op_off(p.v, b, h);
  bf16_t* O = (bf16_t*)p.o.ptr + op_off(p.o, b, h);
  const int Sq = p.Sq, Sk = p.Sk;
  const int q = qb * 32 + l31;
  const bool qok = q < Sq;
  bf16x8 qf[4];
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) qf[kd] = ldg8(Q + (long long)q * p.q.sstride + 16 * kd + 8 * hi, qok);
  float m = -INFINITY, l = 0.f;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
  bf16x8 kn[4], vn[4];                                   // next tile's K / V rows, in flight during this tile's math
  load_row_frags(kn, K, p.k.sstride, 0, Sk, lane);
  load_row_frags(vn, V, p.v.sstride, 0, Sk, lane);
  for (int kt = 0; kt < Sk; kt += 32) {
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
      kf[kd] = kn[kd];
      vf[kd] = vn[kd];
    }
    if (kt + 32 < Sk) {
      load_row_frags(kn, K, p.k.sstride, kt + 32, Sk, lane);
      load_row_frags(vn, V, p.v.sstride, kt + 32, Sk, lane);
    }
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kd], qf[kd], s, 0, 0, 0);
    frags_to_lds(sV, vf, lane);
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // (causal: CLIP's text tower — a query sees the keys up to its own position; key 0 is always visible, so m stays finite)
      float sv = (kt + crow(r, hi) < Sk && (!p.causal || kt + crow(r, hi) <= q)) ? s[r] * p.scale : -INFINITY;
      s[r] = sv;
      mx = fmaxf(mx, sv);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pv = __expf(s[r] - mn);
      s[r] = pv;
      rs += pv;
    }
    rs += __shfl_xor(rs, 32);
    l = l * alpha + rs;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0[r] *= alpha;
      o1[r] *= alpha;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 pb = pack8(s, kk);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sV, 0, kk, lane), pb, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sV, 1, kk, lane), pb, o1, 0, 0, 0);
    }
  }
  if (qok) {
    store_rowT(O, p.o.sstride, q, o0, o1, 1.f / l, hi);
    if (hi == 0 && p.lse) p.lse[((long long)b * p.heads + h) * Sq + q] = m + __logf(l);
  }
}

// dQ (and delta = rowsum(dO * O)); one wave per 32-query block
__global__ __launch_bounds__(64) void attn_bwd_dq_kernel(const T2VAttn p) {
  warm_kernargs<(int)sizeof(T2VAttn)>();
  __shared__ __attribute__((aligned(16))) bf16_t sK[32 * LDT];
  const int lane = threadIdx.x, hi = lane >> 5, l31 = lane & 31;
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const bf16_t* Q = (const bf16_t*)p.q.ptr + op_off(p.q, b, h);
  const bf16_t* K = (const bf16_t*)p.k.ptr + op_off(p.k, b, h);
  const bf16_t* V = (const bf16_t*)p.v.ptr + op_off(p.v, b, h);
  const bf16_t* O = (const bf16_t*)p.o.ptr + op_off(p.o, b, h);
  const bf16_t* dO = (const bf16_t*)p.d_o.ptr + op_off(p.d_o, b, h);
  bf16_t* dQ = (bf16_t*)p.dq.ptr + op_off(p.dq, b, h);
  const int Sq = p.Sq, Sk = p.Sk;
  const int q = qb * 32 + l31;
  const bool qok = q < Sq;
  bf16x8 qf[4], dof[4];
  float dl = 0.f;
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) {
    qf[kd] = ldg8(Q + (long long)q * p.q.sstride + 16 * kd + 8 * hi, qok);
    dof[kd] = ldg8(dO + (long long)q * p.d_o.sstride + 16 * kd + 8 * hi, qok);
    bf16x8 of = ldg8(O + (long long)q * p.o.sstride + 16 * kd + 8 * hi, qok);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += bf2f((unsigned short)of[e]) * bf2f((unsigned short)dof[kd][e]);
  }
  dl += __shfl_xor(dl, 32);
  const long long sidx = ((long long)b * p.heads + h) * Sq + q;
  if (qok && hi == 0) p.delta[sidx] = dl;
  const float c = p.scale * 1.44269504088896341f;                        // p = exp2(s*c - lse*log2e)
  const float lse2 = qok ? p.lse[sidx] * 1.44269504088896341f : 0.f;
  f32x16 a0, a1;
#pragma unroll
  for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
  bf16x8 kn[4], vn[4];
  load_row_frags(kn, K, p.k.sstride, 0, Sk, lane);
  load_row_frags(vn, V, p.v.sstride, 0, Sk, lane);
  for (int kt = 0; kt < Sk; kt += 32) {
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
      kf[kd] = kn[kd];
      vf[kd] = vn[kd];
    }
    if (kt + 32 < Sk) {
      load_row_frags(kn, K, p.k.sstride, kt + 32, Sk, lane);
      load_row_frags(vn, V, p.v.sstride, kt + 32, Sk, lane);
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kd], qf[kd], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kd], dof[kd], dp, 0, 0, 0);
    }
    frags_to_lds(sK, kf, lane);
    const bool ragged = kt + 32 > Sk;                                     // wave-uniform: only the last tile masks keys
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pv = __builtin_amdgcn_exp2f(fmaf(s[r], c, -lse2));
      if (ragged && kt + crow(r, hi) >= Sk) pv = 0.f;
      if (p.causal && kt + crow(r, hi) > q) pv = 0.f;
      s[r] = pv * (dp[r] - dl) * p.scale;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 db = pack8(s, kk);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sK, 0, kk, lane), db, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sK, 1, kk, lane), db, a1, 0, 0, 0);
    }
  }
  if (qok) store_rowT(dQ, p.dq.sstride, q, a0, a1, 1.f, hi);
}

// dK, dV; one wave per 32-key block
__global__ __launch_bounds__(64) void attn_bwd_dkdv_kernel(const T2VAttn p) {
  warm_kernargs<(int)sizeof(T2VAttn)>();
  __shared__ __attribute__((aligned(16))) bf16_t sQ[32 * LDT];
  __shared__ __attribute__((aligned(16))) bf16_t sD[32 * LDT];
  const int lane = threadIdx.x, hi = lane >> 5, l31 = lane & 31;
  const int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const bf16_t* Q = (const bf16_t*)p.q.ptr + op_off(p.q, b, h);
  const bf16_t* K = (const bf16_t*)p.k.ptr + op_off(p.k, b, h);
  const bf16_t* V = (const bf16_t*)p.v.ptr + op_off(p.v, b, h);
  const bf16_t* dO = (const bf16_t*)p.d_o.ptr + op_off(p.d_o, b, h);
  bf16_t* dK = (bf16_t*)p.dk.ptr + op_off(p.dk, b, h);
  bf16_t* dV = (bf16_t*)p.dv.ptr + op_off(p.dv, b, h);
  const int Sq = p.Sq, Sk = p.Sk;
  const int key = kb * 32 + l31;
  const bool kok = key < Sk;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) {
    kf[kd] = ldg8(K + (long long)key * p.k.sstride + 16 * kd + 8 * hi, kok);
    vf[kd] = ldg8(V + (long long)key * p.v.sstride + 16 * kd + 8 * hi, kok);
  }
  const float* lsep = p.lse + ((long long)b * p.heads + h) * Sq;
  const float* dlp = p.delta + ((long long)b * p.heads + h) * Sq;
  const float c = p.scale * 1.44269504088896341f;
  f32x16 k0, k1, v0, v1;
#pragma unroll
  for (int r = 0; r < 16; ++r) k0[r] = k1[r] = v0[r] = v1[r] = 0.f;
  bf16x8 qn[4], dn[4];
  load_row_frags(qn, Q, p.q.sstride, 0, Sq, lane);
  load_row_frags(dn, dO, p.d_o.sstride, 0, Sq, lane);
  for (int qt = 0; qt < Sq; qt += 32) {
    bf16x8 qa[4], da[4];
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
      qa[kd] = qn[kd];
      da[kd] = dn[kd];
    }
    if (qt + 32 < Sq) {
      load_row_frags(qn, Q, p.q.sstride, qt + 32, Sq, lane);
      load_row_frags(dn, dO, p.d_o.sstride, qt + 32, Sq, lane);
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[kd], kf[kd], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[kd], vf[kd], dp, 0, 0, 0);
    }
    frags_to_lds(sQ, qa, lane);
    frags_to_lds(sD, da, lane);
    f32x16 pr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = qt + crow(r, hi);
      // (round 6: clamped loads + a data select instead of `ok ? load : 0` / `ok ? exp2 : 0`, which the compiler turned into
      //  exec-mask branches per score element — see attn_bwd_dkdv_wg2_kernel)
      const bool ok = (qr < Sq) & kok & (!p.causal | (key <= qr));
      const int qc = min(qr, Sq - 1);
      const float L = lsep[qc] * 1.44269504088896341f;
      const float D = dlp[qc];
      float pv = __builtin_amdgcn_exp2f(fmaf(s[r], c, -L));
      pv = ok ? pv : 0.f;
      pr[r] = pv;
      s[r] = pv * (dp[r] - D) * p.scale;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 pb = pack8(pr, kk), db = pack8(s, kk);
      v0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sD, 0, kk, lane), pb, v0, 0, 0, 0);
      v1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sD, 1, kk, lane), pb, v1, 0, 0, 0);
      k0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sQ, 0, kk, lane), db, k0, 0, 0, 0);
      k1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sQ, 1, kk, lane), db, k1, 0, 0, 0);
    }
  }
  if (kok) {
    store_rowT(dK, p.dk.sstride, key, k0, k1, 1.f, hi);
    store_rowT(dV, p.dv.sstride, key, v0, v1, 1.f, hi);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Workgroup variants (long sequences: spatial self-attention S = H*W >= 128, its text cross-attention):  4 wave64 share the
// streamed tiles through LDS.  A workgroup owns 128 consecutive queries (fwd, dQ) or keys (dK/dV) of one (batch, head) —
// one 32-row block per wave, register layouts exactly as in the one-wave kernels above — and the OTHER sequence is staged
// in 64-row tile pairs (K|V, or Q|dO) by all 256 threads: each row is fetched from memory once per WORKGROUP instead of
// once per wave (4x less L2 traffic, no per-wave load latency), double-buffered with the fetch of tile t+1 issued before the
// math of tile t and written to the other LDS stage after it (one barrier per tile).
constexpr int WG_ROWS = 64;                      // staged rows per tile
struct __attribute__((aligned(16))) StagePair {
  bf16_t a[WG_ROWS * LDT];
  bf16_t b[WG_ROWS * LDT];
};
struct PairRegs {
  bf16x8 a[2], b[2];
};
// thread t: rows (t >> 3) and (t >> 3) + 32, 16-byte chunk (t & 7)
__device__ __forceinline__ void pair_fetch(PairRegs& r, const bf16_t* A, long long sa, const bf16_t* B, long long sb, int row0,
                                           int rmax, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = row0 + (tid >> 3) + 32 * i;
    const bool ok = row < rmax;
    r.a[i] = ldg8(A + (long long)row * sa + (tid & 7) * 8, ok);
    r.b[i] = ldg8(B + (long long)row * sb + (tid & 7) * 8, ok);
  }
}
__device__ __forceinline__ void pair_store(StagePair& st, const PairRegs& r, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid >> 3) + 32 * i;
    *(bf16x8*)(st.a + row * LDT + (tid & 7) * 8) = r.a[i];
    *(bf16x8*)(st.b + row * LDT + (tid & 7) * 8) = r.b[i];
  }
}
// MFMA operand fragments of tile rows [32*blk, 32*blk+32) (rows = lanes, k = features): what load_row_frags gives from memory
__device__ __forceinline__ void lds_row_frags(bf16x8 (&f)[4], const bf16_t* s, int blk, int lane) {
  const bf16_t* row = s + (32 * blk + (lane & 31)) * LDT + 8 * (lane >> 5);
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) f[kd] = *(const bf16x8*)(row + 16 * kd);
}

// --- fragment reads with the per-lane part of the address computed ONCE (kernel entry) and everything else a compile-time
// element offset that folds into the ds_read immediate: the counters of the first workgroup forward showed ~20 VALU per MFMA,
// a third of it LDS address arithmetic re-derived per read
__device__ __forceinline__ int lane_row_off(int lane) { return (lane & 31) * LDT + 8 * (lane >> 5); }
__device__ __forceinline__ int lane_tr_off(int lane) {
  const int gq = lane >> 4, li = lane & 15;
  return (4 * (gq >> 1) + (li >> 2)) * LDT + 16 * (gq & 1) + 4 * (li & 3);
}
template <int OFF>
__device__ __forceinline__ void lds_row_frags_c(bf16x8 (&f)[4], const bf16_t* lane_base) {
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) f[kd] = *(const bf16x8*)(lane_base + OFF + 16 * kd);
}
template <int OFF>   // OFF: tile origin (+ 32*fm columns + 16*kk rows) in elements
__device__ __forceinline__ bf16x8 trfrag_c(const bf16_t* lane_base) {
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(lane_base + OFF));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(lane_base + OFF + 8 * LDT));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256) void attn_fwd_wg_kernel(const T2VAttn p) {
  warm_kernargs<(int)sizeof(T2VAttn)>();
  __shared__ StagePair st[2];
  const bf16_t* const lds0 = (const bf16_t*)&st[0];
  constexpr int STG = sizeof(StagePair) / 2, VOFF = WG_ROWS * LDT;     // element offsets: next stage, V inside a stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const bf16_t* Q = (const bf16_t*)p.q.ptr + op_off(p.q, b, h);
  const bf16_t* K = (const bf16_t*)p.k.ptr + op_off(p.k, b, h);
  const bf16_t* V = (const bf16_t*)p.v.ptr + op_off(p.v, b, h);
  bf16_t* O = (bf16_t*)p.o.ptr + op_off(p.o, b, h);
  const int Sq = p.Sq, Sk = p.Sk;
  const int q = blockIdx.x * 128 + wave * 32 + l31;
  const bool qok = q < Sq;
  bf16x8 qf[4];
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) qf[kd] = ldg8(Q + (long long)q * p.q.sstride + 16 * kd + 8 * hi, qok);
  const float c = p.scale * 1.44269504088896341f;          // softmax in base 2: exp(s*scale - m) = exp2(s*c - m*c)
  float m = -INFINITY, l = 0.f;                             // m: running max of the RAW scores
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
  // Tiles are fetched TWO iterations ahead into alternating register sets (a K|V tile's loads have a whole iteration of math
  // plus a barrier to land before they are written to LDS) — the math of one 64-key tile is shorter than a load round trip.
  const int ntile = (Sk + WG_ROWS - 1) / WG_ROWS;
  PairRegs nxa, nxb;
  pair_fetch(nxa, K, p.k.sstride, V, p.v.sstride, 0, Sk, tid);
  if (ntile > 1) pair_fetch(nxb, K, p.k.sstride, V, p.v.sstride, WG_ROWS, Sk, tid);
  pair_store(st[0], nxa, tid);
  __syncthreads();
  const bf16_t* const rowb = lds0 + lane_row_off(lane);
  const bf16_t* const trb = lds0 + lane_tr_off(lane);
  auto tile_math = [&](auto stage_c, int t) {
    constexpr int S0 = decltype(stage_c)::value * STG;
    const int kt0 = t * WG_ROWS;
    const bool ragged = kt0 + WG_ROWS > Sk;                 // workgroup-uniform: only the last tile masks keys
    auto block = [&](auto kb_c) {
      constexpr int kb = decltype(kb_c)::value;
      bf16x8 kf[4];
      lds_row_frags_c<S0 + 32 * kb * LDT>(kf, rowb);
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int kd = 0; kd < 4; ++kd) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kd], qf[kd], s, 0, 0, 0);
      if (ragged) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt0 + 32 * kb + crow(r, hi) >= Sk) s[r] = -INFINITY;
      }
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      // deferred rescale: as long as no row's maximum grew by more than DEFER (in exponent units) the running maximum is
      // kept — p <= 2^DEFER instead of <= 1, exact in fp32/bf16 relative terms — and the 32 multiplies of O are skipped.
      // The decision is taken BEFORE this block's P is exponentiated and covers everything accumulated so far (O and l).
      constexpr float DEFER = 6.f;
      if (!__all((mx - m) * c <= DEFER)) {
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);            // m = -inf on the first block: alpha = 0, O and l are still 0
        m = mn;
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o0[r] *= alpha;
          o1[r] *= alpha;
        }
      }
      const float mc = m * c;
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], c, -mc));
        s[r] = pv;
        rs += pv;
      }
      rs += __shfl_xor(rs, 32);
      l += rs;
      constexpr int V0 = S0 + VOFF + 32 * kb * LDT;
      {
        const bf16x8 pb = pack8(s, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_c<V0>(trb), pb, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_c<V0 + 32>(trb), pb, o1, 0, 0, 0);
      }
      {
        const bf16x8 pb = pack8(s, 1);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_c<V0 + 16 * LDT>(trb), pb, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_c<V0 + 16 * LDT + 32>(trb), pb, o1, 0, 0, 0);
      }
    };
    block(std::integral_constant<int, 0>{});
    if (kt0 + 32 < Sk) block(std::integral_constant<int, 1>{});
  };
  for (int t = 0; t < ntile; t += 2) {
    // even tile t: nxb holds tile t+1 (in flight or landed); fetch t+2 into nxa
    if (t + 2 < ntile) pair_fetch(nxa, K, p.k.sstride, V, p.v.sstride, (t + 2) * WG_ROWS, Sk, tid);
    tile_math(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntile) pair_store(st[1], nxb, tid);
    __syncthreads();
    if (t + 1 >= ntile) break;
    // odd tile t+1: nxa holds tile t+2; fetch t+3 into nxb
    if (t + 3 < ntile) pair_fetch(nxb, K, p.k.sstride, V, p.v.sstride, (t + 3) * WG_ROWS, Sk, tid);
    tile_math(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 < ntile) pair_store(st[0], nxa, tid);
    __syncthreads();
  }
  if (qok) {
    store_rowT(O, p.o.sstride, q, o0, o1, 1.f / l, hi);
    if (hi == 0 && p.lse) p.lse[((long long)b * p.heads + h) * Sq + q] = m * p.scale + __logf(l);
  }
}

// dQ (and delta = rowsum(dO * O)); workgroup = 128 queries, K|V tiles shared
__global__ __launch_bounds__(256) void attn_bwd_dq_wg_kernel(const T2VAttn p) {
  warm_kernargs<(int)sizeof(T2VAttn)>();
  __shared__ StagePair st[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const bf16_t* Q = (const bf16_t*)p.q.ptr + op_off(p.q, b, h);
  const bf16_t* K = (const bf16_t*)p.k.ptr + op_off(p.k, b, h);
  const bf16_t* V = (const bf16_t*)p.v.ptr + op_off(p.v, b, h);
  const bf16_t* O = (const bf16_t*)p.o.ptr + op_off(p.o, b, h);
  const bf16_t* dO = (const bf16_t*)p.d_o.ptr + op_off(p.d_o, b, h);
  bf16_t* dQ = (bf16_t*)p.dq.ptr + op_off(p.dq, b, h);
  const int Sq = p.Sq, Sk = p.Sk;
  const int q = blockIdx.x * 128 + wave * 32 + l31;
  const bool qok = q < Sq;
  bf16x8 qf[4], dof[4];
  float dl = 0.f;
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) {
    qf[kd] = ldg8(Q + (long long)q * p.q.sstride + 16 * kd + 8 * hi, qok);
    dof[kd] = ldg8(dO + (long long)q * p.d_o.sstride + 16 * kd + 8 * hi, qok);
    const bf16x8 of = ldg8(O + (long long)q * p.o.sstride + 16 * kd + 8 * hi, qok);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += bf2f((unsigned short)of[e]) * bf2f((unsigned short)dof[kd][e]);
  }
  dl += __shfl_xor(dl, 32);
  const long long sidx = ((long long)b * p.heads + h) * Sq + q;
  if (qok && hi == 0) p.delta[sidx] = dl;
  const float c = p.scale * 1.44269504088896341f;
  const float lse2 = qok ? p.lse[sidx] * 1.44269504088896341f : 0.f;      // p = exp2(s*c - lse*log2e)
  f32x16 a0, a1;
#pragma unroll
  for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
  PairRegs nx;
  pair_fetch(nx, K, p.k.sstride, V, p.v.sstride, 0, Sk, tid);
  pair_store(st[0], nx, tid);
  __syncthreads();
  const int ntile = (Sk + WG_ROWS - 1) / WG_ROWS;
  for (int t = 0; t < ntile; ++t) {
    const StagePair& cur = st[t & 1];
    if (t + 1 < ntile) pair_fetch(nx, K, p.k.sstride, V, p.v.sstride, (t + 1) * WG_ROWS, Sk, tid);
    const int kt0 = t * WG_ROWS;
    const bool ragged = kt0 + WG_ROWS > Sk;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kt0 + 32 * kb >= Sk) break;
      bf16x8 kf[4], vf[4];
      lds_row_frags(kf, cur.a, kb, lane);
      lds_row_frags(vf, cur.b, kb, lane);
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
      for (int kd = 0; kd < 4; ++kd) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kd], qf[kd], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kd], dof[kd], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pv = __builtin_amdgcn_exp2f(fmaf(s[r], c, -lse2));
        if (ragged && kt0 + 32 * kb + crow(r, hi) >= Sk) pv = 0.f;
        s[r] = pv * (dp[r] - dl) * p.scale;
      }
      const bf16_t* sk = cur.a + 32 * kb * LDT;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 db = pack8(s, kk);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sk, 0, kk, lane), db, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sk, 1, kk, lane), db, a1, 0, 0, 0);
      }
    }
    if (t + 1 < ntile) pair_store(st[(t + 1) & 1], nx, tid);
    __syncthreads();
  }
  if (qok) store_rowT(dQ, p.dq.sstride, q, a0, a1, 1.f, hi);
}

// dK, dV; workgroup = 128 keys, Q|dO tiles (+ their lse / delta rows) shared
__global__ __launch_bounds__(256) void attn_bwd_dkdv_wg_kernel(const T2VAttn p) {
  warm_kernargs<(int)sizeof(T2VAttn)>();
  __shared__ StagePair st[2];
  __shared__ float sL[2][WG_ROWS], sDl[2][WG_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const bf16_t* Q = (const bf16_t*)p.q.ptr + op_off(p.q, b, h);
  const bf16_t* K = (const bf16_t*)p.k.ptr + op_off(p.k, b, h);
  const bf16_t* V = (const bf16_t*)p.v.ptr + op_off(p.v, b, h);
  const bf16_t* dO = (const bf16_t*)p.d_o.ptr + op_off(p.d_o, b, h);
  bf16_t* dK = (bf16_t*)p.dk.ptr + op_off(p.dk, b, h);
  bf16_t* dV = (bf16_t*)p.dv.ptr + op_off(p.dv, b, h);
  const int Sq = p.Sq, Sk = p.Sk;
  const int key = blockIdx.x * 128 + wave * 32 + l31;
  const bool kok = key < Sk;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) {
    kf[kd] = ldg8(K + (long long)key * p.k.sstride + 16 * kd + 8 * hi, kok);
    vf[kd] = ldg8(V + (long long)key * p.v.sstride + 16 * kd + 8 * hi, kok);
  }
  const float* lsep = p.lse + ((long long)b * p.heads + h) * Sq;
  const float* dlp = p.delta + ((long long)b * p.heads + h) * Sq;
  const float c = p.scale * 1.44269504088896341f;
  f32x16 k0, k1, v0, v1;
#pragma unroll
  for (int r = 0; r < 16; ++r) k0[r] = k1[r] = v0[r] = v1[r] = 0.f;
  PairRegs nx;
  float nl = 0.f, nd = 0.f;                                 // threads 0..63 carry the tile's lse (base 2) / delta rows
  auto fetch_rows = [&](int row0) {
    if (tid < WG_ROWS) {
      const int qr = row0 + tid;
      nl = qr < Sq ? lsep[qr] * 1.44269504088896341f : 0.f;
      nd = qr < Sq ? dlp[qr] : 0.f;
    }
  };
  pair_fetch(nx, Q, p.q.sstride, dO, p.d_o.sstride, 0, Sq, tid);
  fetch_rows(0);
  pair_store(st[0], nx, tid);
  if (tid < WG_ROWS) {
    sL[0][tid] = nl;
    sDl[0][tid] = nd;
  }
  __syncthreads();
  const int ntile = (Sq + WG_ROWS - 1) / WG_ROWS;
  for (int t = 0; t < ntile; ++t) {
    const StagePair& cur = st[t & 1];
    if (t + 1 < ntile) {
      pair_fetch(nx, Q, p.q.sstride, dO, p.d_o.sstride, (t + 1) * WG_ROWS, Sq, tid);
      fetch_rows((t + 1) * WG_ROWS);
    }
    const int qt0 = t * WG_ROWS;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      if (qt0 + 32 * qb >= Sq) break;
      bf16x8 qa[4], da[4];
      lds_row_frags(qa, cur.a, qb, lane);
      lds_row_frags(da, cur.b, qb, lane);
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
      for (int kd = 0; kd < 4; ++kd) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[kd], kf[kd], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[kd], vf[kd], dp, 0, 0, 0);
      }
      f32x16 pr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = 32 * qb + crow(r, hi);
        const bool ok = (qt0 + ql < Sq) && kok;
        const float pv = ok ? __builtin_amdgcn_exp2f(fmaf(s[r], c, -sL[t & 1][ql])) : 0.f;
        pr[r] = pv;
        s[r] = pv * (dp[r] - sDl[t & 1][ql]) * p.scale;
      }
      const bf16_t* sq = cur.a + 32 * qb * LDT;
      const bf16_t* sd = cur.b + 32 * qb * LDT;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 pb = pack8(pr, kk), db = pack8(s, kk);
        v0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sd, 0, kk, lane), pb, v0, 0, 0, 0);
        v1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sd, 1, kk, lane), pb, v1, 0, 0, 0);
        k0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sq, 0, kk, lane), db, k0, 0, 0, 0);
        k1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sq, 1, kk, lane), db, k1, 0, 0, 0);
      }
    }
    if (t + 1 < ntile) {
      pair_store(st[(t + 1) & 1], nx, tid);
      if (tid < WG_ROWS) {
        sL[(t + 1) & 1][tid] = nl;
        sDl[(t + 1) & 1][tid] = nd;
      }
    }
    __syncthreads();
  }
  if (kok) {
    store_rowT(dK, p.dk.sstride, key, k0, k1, 1.f, hi);
    store_rowT(dV, p.dv.sstride, key, v0, v1, 1.f, hi);
  }
}

// ---- backward with TWO 32-row blocks per wave (workgroup = 256 queries / keys).  The one-block workgroup kernels above lose to
// the one-wave kernels: every 32x32 score block needs the staged tile in two orientations (row fragments for the score
// products, transposed fragments for the gradient products) — 16 KB of LDS reads per 12-16 MFMAs per wave — and one block per
// wave is a single dependent chain (scores -> exp -> gradient products) with nothing to overlap at one wave per SIMD.  Here
// every fragment read from LDS feeds two blocks (half the LDS traffic per MFMA, half the global traffic again), and the two
// blocks' chains are independent, so one block's MFMAs run under the other's exponentials.
template <int V>
using ic = std::integral_constant<int, V>;

__global__ __launch_bounds__(256) void attn_bwd_dq_wg2_kernel(const T2VAttn p) {
  warm_kernargs<(int)sizeof(T2VAttn)>();
  __shared__ StagePair st[2];
  constexpr int VOFF = WG_ROWS * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const bf16_t* Q = (const bf16_t*)p.q.ptr + op_off(p.q, b, h);
  const bf16_t* K = (const bf16_t*)p.k.ptr + op_off(p.k, b, h);
  const bf16_t* V = (const bf16_t*)p.v.ptr + op_off(p.v, b, h);
  const bf16_t* O = (const bf16_t*)p.o.ptr + op_off(p.o, b, h);
  const bf16_t* dO = (const bf16_t*)p.d_o.ptr + op_off(p.d_o, b, h);
  bf16_t* dQ = (bf16_t*)p.dq.ptr + op_off(p.dq, b, h);
  const int Sq = p.Sq, Sk = p.Sk;
  const float c = p.scale * 1.44269504088896341f;
  int q[2];
  bool qok[2];
  bf16x8 qf[2][4], dof[2][4];
  float dl[2], lse2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    q[j] = blockIdx.x * 256 + wave * 64 + 32 * j + l31;
    qok[j] = q[j] < Sq;
    float d = 0.f;
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
      qf[j][kd] = ldg8(Q + (long long)q[j] * p.q.sstride + 16 * kd + 8 * hi, qok[j]);
      dof[j][kd] = ldg8(dO + (long long)q[j] * p.d_o.sstride + 16 * kd + 8 * hi, qok[j]);
      const bf16x8 of = ldg8(O + (long long)q[j] * p.o.sstride + 16 * kd + 8 * hi, qok[j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) d += bf2f((unsigned short)of[e]) * bf2f((unsigned short)dof[j][kd][e]);
    }
    d += __shfl_xor(d, 32);
    dl[j] = d;
    const long long sidx = ((long long)b * p.heads + h) * Sq + q[j];
    if (qok[j] && hi == 0) p.delta[sidx] = d;
    lse2[j] = qok[j] ? p.lse[sidx] * 1.44269504088896341f : 0.f;      // p = exp2(s*c - lse*log2e)
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][0][r] = acc[j][1][r] = 0.f;
  const int roff = lane_row_off(lane), toff = lane_tr_off(lane);
  PairRegs nx;
  pair_fetch(nx, K, p.k.sstride, V, p.v.sstride, 0, Sk, tid);
  pair_store(st[0], nx, tid);
  __syncthreads();
  const int ntile = (Sk + WG_ROWS - 1) / WG_ROWS;
  for (int t = 0; t < ntile; ++t) {
    const bf16_t* cur = st[t & 1].a;
    const bf16_t* rb = cur + roff;
    const bf16_t* tb = cur + toff;
    if (t + 1 < ntile) pair_fetch(nx, K, p.k.sstride, V, p.v.sstride, (t + 1) * WG_ROWS, Sk, tid);
    const int kt0 = t * WG_ROWS;
    const bool ragged = kt0 + WG_ROWS > Sk;
    auto block = [&](auto KB) {
      constexpr int kb = decltype(KB)::value, K0 = 32 * kb * LDT;
      bf16x8 kf[4], vf[4], tk[2][2];
      lds_row_frags_c<K0>(kf, rb);
      lds_row_frags_c<VOFF + K0>(vf, rb);
      tk[0][0] = trfrag_c<K0>(tb);
      tk[0][1] = trfrag_c<K0 + 32>(tb);
      tk[1][0] = trfrag_c<K0 + 16 * LDT>(tb);
      tk[1][1] = trfrag_c<K0 + 16 * LDT + 32>(tb);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
        for (int kd = 0; kd < 4; ++kd) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kd], qf[j][kd], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kd], dof[j][kd], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pv = __builtin_amdgcn_exp2f(fmaf(s[r], c, -lse2[j]));
          if (ragged && kt0 + 32 * kb + crow(r, hi) >= Sk) pv = 0.f;
          s[r] = pv * (dp[r] - dl[j]) * p.scale;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const bf16x8 db = pack8(s, kk);
          acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tk[kk][0], db, acc[j][0], 0, 0, 0);
          acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tk[kk][1], db, acc[j][1], 0, 0, 0);
        }
      }
    };
    block(ic<0>{});
    if (kt0 + 32 < Sk) block(ic<1>{});
    if (t + 1 < ntile) pair_store(st[(t + 1) & 1], nx, tid);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (qok[j]) store_rowT(dQ, p.dq.sstride, q[j], acc[j][0], acc[j][1], 1.f, hi);
}

__global__ __launch_bounds__(256) void attn_bwd_dkdv_wg2_kernel(const T2VAttn p) {
  warm_kernargs<(int)sizeof(T2VAttn)>();
  __shared__ StagePair st[2];
  __shared__ __attribute__((aligned(16))) float sL[2][WG_ROWS], sDl[2][WG_ROWS];
  constexpr int VOFF = WG_ROWS * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const bf16_t* Q = (const bf16_t*)p.q.ptr + op_off(p.q, b, h);
  const bf16_t* K = (const bf16_t*)p.k.ptr + op_off(p.k, b, h);
  const bf16_t* V = (const bf16_t*)p.v.ptr + op_off(p.v, b, h);
  const bf16_t* dO = (const bf16_t*)p.d_o.ptr + op_off(p.d_o, b, h);
  bf16_t* dK = (bf16_t*)p.dk.ptr + op_off(p.dk, b, h);
  bf16_t* dV = (bf16_t*)p.dv.ptr + op_off(p.dv, b, h);
  const int Sq = p.Sq, Sk = p.Sk;
  int key[2];
  bool kok[2];
  bf16x8 kf[2][4], vf[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    key[j] = blockIdx.x * 256 + wave * 64 + 32 * j + l31;
    kok[j] = key[j] < Sk;
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
      kf[j][kd] = ldg8(K + (long long)key[j] * p.k.sstride + 16 * kd + 8 * hi, kok[j]);
      vf[j][kd] = ldg8(V + (long long)key[j] * p.v.sstride + 16 * kd + 8 * hi, kok[j]);
    }
  }
  const float* lsep = p.lse + ((long long)b * p.heads + h) * Sq;
  const float* dlp = p.delta + ((long long)b * p.heads + h) * Sq;
  const float c = p.scale * 1.44269504088896341f;
  f32x16 ka[2][2], va[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) ka[j][0][r] = ka[j][1][r] = va[j][0][r] = va[j][1][r] = 0.f;
  const int roff = lane_row_off(lane), toff = lane_tr_off(lane);
  PairRegs nx;
  float nl = 0.f, nd = 0.f;                                 // threads 0..63 carry the tile's lse (base 2) / delta rows
  auto fetch_rows = [&](int row0) {
    if (tid < WG_ROWS) {
      const int qr = row0 + tid;
      nl = qr < Sq ? lsep[qr] * 1.44269504088896341f : 0.f;
      nd = qr < Sq ? dlp[qr] : 0.f;
    }
  };
  pair_fetch(nx, Q, p.q.sstride, dO, p.d_o.sstride, 0, Sq, tid);
  fetch_rows(0);
  pair_store(st[0], nx, tid);
  if (tid < WG_ROWS) {
    sL[0][tid] = nl;
    sDl[0][tid] = nd;
  }
  __syncthreads();
  const int ntile = (Sq + WG_ROWS - 1) / WG_ROWS;
  for (int t = 0; t < ntile; ++t) {
    const bf16_t* cur = st[t & 1].a;
    const bf16_t* rb = cur + roff;
    const bf16_t* tb = cur + toff;
    const float* cl = sL[t & 1];
    const float* cd = sDl[t & 1];
    if (t + 1 < ntile) {
      pair_fetch(nx, Q, p.q.sstride, dO, p.d_o.sstride, (t + 1) * WG_ROWS, Sq, tid);
      fetch_rows((t + 1) * WG_ROWS);
    }
    const int qt0 = t * WG_ROWS;
    auto block = [&](auto QB) {
      constexpr int qb = decltype(QB)::value, Q0 = 32 * qb * LDT;
      const int qlim = Sq - qt0 - 32 * qb;                   // rows of this 32-query block that exist (>= 32 except in the last tile)
      bf16x8 qa[4], da[4], tq[2][2], td[2][2];
      lds_row_frags_c<Q0>(qa, rb);
      lds_row_frags_c<VOFF + Q0>(da, rb);
      tq[0][0] = trfrag_c<Q0>(tb);
      tq[0][1] = trfrag_c<Q0 + 32>(tb);
      tq[1][0] = trfrag_c<Q0 + 16 * LDT>(tb);
      tq[1][1] = trfrag_c<Q0 + 16 * LDT + 32>(tb);
      td[0][0] = trfrag_c<VOFF + Q0>(tb);
      td[0][1] = trfrag_c<VOFF + Q0 + 32>(tb);
      td[1][0] = trfrag_c<VOFF + Q0 + 16 * LDT>(tb);
      td[1][1] = trfrag_c<VOFF + Q0 + 16 * LDT + 32>(tb);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
        for (int kd = 0; kd < 4; ++kd) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[kd], kf[j][kd], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[kd], vf[j][kd], dp, 0, 0, 0);
        }
        f32x16 pr;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {                       // rows crow(4*g4 + i, hi) = 8*g4 + 4*hi + i: four consecutive floats
          const float4 l4 = *(const float4*)(cl + 32 * qb + 8 * g4 + 4 * hi);
          const float4 d4 = *(const float4*)(cd + 32 * qb + 8 * g4 + 4 * hi);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g4 + i;
            float pv = __builtin_amdgcn_exp2f(fmaf(s[r], c, -lv[i]));
            // (round 6: a data select, no short-circuit — `!kok[j] || (ragged && ...)` was compiled into an exec-mask branch PER
            //  ELEMENT with the exponential sunk into it: ~12 scalar instructions per score element, 879 k SALU against 1 311 k VALU
            //  instructions per SIMD in the counter passes of profiles/r06_attention_counters.txt)
            const bool live = kok[j] & (crow(r, hi) < qlim);
            pv = live ? pv : 0.f;
            pr[r] = pv;
            s[r] = pv * (dp[r] - dv[i]) * p.scale;
          }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const bf16x8 pb = pack8(pr, kk), db = pack8(s, kk);
          va[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[kk][0], pb, va[j][0], 0, 0, 0);
          va[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[kk][1], pb, va[j][1], 0, 0, 0);
          ka[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[kk][0], db, ka[j][0], 0, 0, 0);
          ka[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[kk][1], db, ka[j][1], 0, 0, 0);
        }
      }
    };
    block(ic<0>{});
    if (qt0 + 32 < Sq) block(ic<1>{});
    if (t + 1 < ntile) {
      pair_store(st[(t + 1) & 1], nx, tid);
      if (tid < WG_ROWS) {
        sL[(t + 1) & 1][tid] = nl;
        sDl[(t + 1) & 1][tid] = nd;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (kok[j]) {
      store_rowT(dK, p.dk.sstride, key[j], ka[j][0], ka[j][1], 1.f, hi);
      store_rowT(dV, p.dv.sstride, key[j], va[j][0], va[j][1], 1.f, hi);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Short self-attention sequences (S = Sq = Sk <= 16, S | 32: the temporal attention over F frames): P = 32/S sequences
// of consecutive batches are packed into ONE 32-row tile — lane&31 = packed row = (sequence lane/S, position lane%S) —
// and the 32x32 score tile is block-diagonal-masked.  Every lane does useful work (a lone S=16 sequence would idle half
// the wave) and the whole problem of a tile is resident in one wave: forward is one pass, backward produces dQ, dK and dV
// in ONE kernel from a single load of Q, K, V, dO (the two score orientations cost 8 extra MFMAs, nothing is re-read).
struct PackedRow {
  int b, pos, sub;
  bool ok;
};
__device__ __forceinline__ PackedRow packed_row(const T2VAttn& p, int S, int P, int l31) {
  PackedRow r;
  r.sub = l31 / S;
  r.pos = l31 - r.sub * S;
  r.b = blockIdx.x * P + r.sub;
  r.ok = r.b < p.nbatch;
  if (!r.ok) r.b = 0;
  return r;
}
__device__ __forceinline__ const bf16_t* prow(const T2VAttnOperand& o, const PackedRow& r, int h) {
  return (const bf16_t*)o.ptr + op_off(o, r.b, h) + (long long)r.pos * o.sstride;
}
__device__ __forceinline__ void load_frags(bf16x8 (&f)[4], const bf16_t* row, bool ok, int hi) {
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) f[kd] = ldg8(row + 16 * kd + 8 * hi, ok);
}

__global__ __launch_bounds__(64) void attn_fwd_packed_kernel(const T2VAttn p, int S, int P) {
  warm_kernargs<(int)sizeof(T2VAttn)>();
  __shared__ __attribute__((aligned(16))) bf16_t sV[32 * LDT];
  const int lane = threadIdx.x, hi = lane >> 5, l31 = lane & 31, h = blockIdx.y;
  const PackedRow R = packed_row(p, S, P, l31);
  bf16x8 qf[4], kf[4], vf[4];
  load_frags(qf, prow(p.q, R, h), R.ok, hi);
  load_frags(kf, prow(p.k, R, h), R.ok, hi);
  load_frags(vf, prow(p.v, R, h), R.ok, hi);
  f32x16 s;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
  for (int kd = 0; kd < 4; ++kd) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kd], qf[kd], s, 0, 0, 0);
  frags_to_lds(sV, vf, lane);
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float sv = (R.ok && crow(r, hi) / S == R.sub) ? s[r] * p.scale : -INFINITY;
    s[r] = sv;
    mx = fmaxf(mx, sv);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const float m = R.ok ? mx : 0.f;
  float l = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float pv = __expf(s[r] - m);
    s[r] = pv;
    l += pv;
  }
  l += __shfl_xor(l, 32);
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const bf16x8 pb = pack8(s, kk);
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sV, 0, kk, lane), pb, o0, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sV, 1, kk, lane), pb, o1, 0, 0, 0);
  }
  if (R.ok) {
    store_rowT((bf16_t*)prow(p.o, R, h), 0, 0, o0, o1, 1.f / l, hi);
    if (hi == 0 && p.lse) p.lse[((long long)R.b * p.heads + h) * S + R.pos] = m + __logf(l);
  }
}

__global__ __launch_bounds__(64) void attn_bwd_packed_kernel(const T2VAttn p, int S, int P) {
  warm_kernargs<(int)sizeof(T2VAttn)>();
  __shared__ __attribute__((aligned(16))) bf16_t sK[32 * LDT];
  __shared__ __attribute__((aligned(16))) bf16_t sQ[32 * LDT];
  __shared__ __attribute__((aligned(16))) bf16_t sD[32 * LDT];
  __shared__ float sL[32], sDl[32];
  const int lane = threadIdx.x, hi = lane >> 5, l31 = lane & 31, h = blockIdx.y;
  const PackedRow R = packed_row(p, S, P, l31);
  bf16x8 qf[4], kf[4], vf[4], dof[4];
  load_frags(qf, prow(p.q, R, h), R.ok, hi);
  load_frags(kf, prow(p.k, R, h), R.ok, hi);
  load_frags(vf, prow(p.v, R, h), R.ok, hi);
  load_frags(dof, prow(p.d_o, R, h), R.ok, hi);
  float dl = 0.f;
  {
    const bf16_t* orow = prow(p.o, R, h);
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
      const bf16x8 of = ldg8(orow + 16 * kd + 8 * hi, R.ok);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += bf2f((unsigned short)of[e]) * bf2f((unsigned short)dof[kd][e]);
    }
  }
  dl += __shfl_xor(dl, 32);
  const long long sidx = ((long long)R.b * p.heads + h) * S + R.pos;
  const float lse = R.ok ? p.lse[sidx] : 0.f;
  if (hi == 0) {
    if (R.ok) p.delta[sidx] = dl;
    sL[l31] = lse;
    sDl[l31] = dl;
  }
  frags_to_lds(sK, kf, lane);
  frags_to_lds(sQ, qf, lane);
  frags_to_lds(sD, dof, lane);
  __syncthreads();
  // orientation 1: lane = query.  S^T = K Q^T, dP^T = V dO^T  ->  dQ^T = K^T dS^T
  {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kd], qf[kd], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kd], dof[kd], dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool live = R.ok & (crow(r, hi) / S == R.sub);
      const float e = __expf(s[r] * p.scale - lse);
      const float pv = live ? e : 0.f;
      s[r] = pv * (dp[r] - dl) * p.scale;
    }
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 db = pack8(s, kk);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sK, 0, kk, lane), db, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sK, 1, kk, lane), db, a1, 0, 0, 0);
    }
    if (R.ok) store_rowT((bf16_t*)prow(p.dq, R, h), 0, 0, a0, a1, 1.f, hi);
  }
  // orientation 2: lane = key.  S = Q K^T, dP = dO V^T  ->  dV^T = dO^T P, dK^T = Q^T dS
  {
    f32x16 s, dp, pr;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[kd], kf[kd], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[kd], vf[kd], dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = crow(r, hi);
      const bool ok = R.ok & (rr / S == R.sub);
      const float e = __expf(s[r] * p.scale - sL[rr]);
      const float pv = ok ? e : 0.f;
      pr[r] = pv;
      s[r] = pv * (dp[r] - sDl[rr]) * p.scale;
    }
    f32x16 k0, k1, v0, v1;
#pragma unroll
    for (int r = 0; r < 16; ++r) k0[r] = k1[r] = v0[r] = v1[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 pb = pack8(pr, kk), db = pack8(s, kk);
      v0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sD, 0, kk, lane), pb, v0, 0, 0, 0);
      v1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sD, 1, kk, lane), pb, v1, 0, 0, 0);
      k0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sQ, 0, kk, lane), db, k0, 0, 0, 0);
      k1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(sQ, 1, kk, lane), db, k1, 0, 0, 0);
    }
    if (R.ok) {
      store_rowT((bf16_t*)prow(p.dk, R, h), 0, 0, k0, k1, 1.f, hi);
      store_rowT((bf16_t*)prow(p.dv, R, h), 0, 0, v0, v1, 1.f, hi);
    }
  }
}

// packing applies to self-attention-shaped problems with a short power-of-two sequence
__host__ inline int packed_seqs(const T2VAttn& p) {
  if (p.Sq != p.Sk || p.Sq > 16 || (32 % p.Sq) != 0 || p.nbatch < 2) return 0;
  return 32 / p.Sq;
}

// workgroup variants pay when the workgroup's 128 rows are (mostly) real: sequences of at least 128
bool use_wg(int rows) {
  static const int off = [] { const char* e = getenv("T2V_ATTN_WG"); return e && e[0] == '0'; }();
  return !off && rows >= 128;
}

// One-block-per-wave workgroup backward (128 rows per workgroup): reads BOTH MFMA operands of the streamed tile from LDS and
// measured 15-20 % slower than the one-wave kernels at S = 2880 / 9216 (LDS-read-bound, one dependent chain per wave).
// Selectable for the A/B with T2V_ATTN_WG_BWD=1; superseded by the two-block kernels below.
bool use_wg_bwd(int rows) {
  static const int on = [] { const char* e = getenv("T2V_ATTN_WG_BWD"); return e && e[0] == '1'; }();
  return on && rows >= 128;
}

// two 32-row blocks per wave (256 rows per workgroup): T2V_ATTN_WG_BWD=2 (the default) selects them for runs of >= 256 rows,
// T2V_ATTN_WG_BWD=0 keeps the one-wave pair everywhere
bool use_wg2_bwd(int rows) {
  static const int mode = [] { const char* e = getenv("T2V_ATTN_WG_BWD"); return e ? atoi(e) : 2; }();
  return mode == 2 && rows >= 256;
}

int check_op(const char* fn, const char* name, const T2VAttnOperand& o) {
  if (!o.ptr || o.bdiv <= 0 || ((uintptr_t)o.ptr & 15) || (o.sstride % 8) || (o.bstride_hi % 8) || (o.bstride_lo % 8)) {
    t2v_set_error("%s: operand %s invalid (ptr %p, bdiv %d, strides must be multiples of 8 elements)", fn, name, o.ptr,
                  o.bdiv);
    return T2V_EINVAL;
  }
  return T2V_OK;
}
}  // namespace

extern "C" int t2v_attn_fwd(const T2VAttn* p, t2v_stream_t stream) {
  T2V_CHECK_ARG(p && p->nbatch > 0 && p->heads > 0 && p->Sq > 0 && p->Sk > 0, "t2v_attn_fwd: bad dims");
  if (int e = check_op("t2v_attn_fwd", "q", p->q)) return e;
  if (int e = check_op("t2v_attn_fwd", "k", p->k)) return e;
  if (int e = check_op("t2v_attn_fwd", "v", p->v)) return e;
  if (int e = check_op("t2v_attn_fwd", "o", p->o)) return e;
  T2V_CHECK_ARG(!p->causal || p->Sq == p->Sk, "t2v_attn_fwd: the causal mask is defined for self-attention (Sq == Sk)");
  if (const int P = p->causal ? 0 : packed_seqs(*p)) {
    T2V_CHECK_ARG(p->heads <= 65535, "t2v_attn_fwd: heads exceed grid limits (%d)", p->heads);
    T2V_LAUNCH(attn_fwd_packed_kernel, dim3((p->nbatch + P - 1) / P, p->heads), dim3(64), 0, (hipStream_t)stream, *p,
                       p->Sq, P);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
  }
  dim3 grid((p->Sq + 31) / 32, p->heads, p->nbatch);
  T2V_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "t2v_attn_fwd: heads/nbatch exceed grid limits (%d, %d)", p->heads,
                p->nbatch);
  if (!p->causal && use_wg(p->Sq)) { // long query sequences: 4 waves share the K|V tiles through LDS (causal: one-wave kernels only)
    T2V_LAUNCH(attn_fwd_wg_kernel, dim3((p->Sq + 127) / 128, p->heads, p->nbatch), dim3(256), 0, (hipStream_t)stream, *p);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
  }
  T2V_LAUNCH(attn_fwd_kernel, grid, dim3(64), 0, (hipStream_t)stream, *p);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

extern "C" int t2v_attn_bwd(const T2VAttn* p, t2v_stream_t stream) {
  T2V_CHECK_ARG(p && p->nbatch > 0 && p->heads > 0 && p->Sq > 0 && p->Sk > 0, "t2v_attn_bwd: bad dims");
  T2V_CHECK_ARG(p->lse && p->delta, "t2v_attn_bwd: lse and delta are required");
  const T2VAttnOperand* ops[] = {&p->q, &p->k, &p->v, &p->o, &p->d_o, &p->dq, &p->dk, &p->dv};
  const char* names[] = {"q", "k", "v", "o", "d_o", "dq", "dk", "dv"};
  for (int i = 0; i < 8; ++i)
    if (int e = check_op("t2v_attn_bwd", names[i], *ops[i])) return e;
  T2V_CHECK_ARG(!p->causal || p->Sq == p->Sk, "t2v_attn_bwd: the causal mask is defined for self-attention (Sq == Sk)");
  if (const int P = p->causal ? 0 : packed_seqs(*p)) {
    T2V_CHECK_ARG(p->heads <= 65535, "t2v_attn_bwd: heads exceed grid limits (%d)", p->heads);
    T2V_LAUNCH(attn_bwd_packed_kernel, dim3((p->nbatch + P - 1) / P, p->heads), dim3(64), 0, (hipStream_t)stream, *p,
                       p->Sq, P);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
  }
  T2V_CHECK_ARG(p->heads <= 65535 && p->nbatch <= 65535, "t2v_attn_bwd: heads/nbatch exceed grid limits");
  dim3 gq((p->Sq + 31) / 32, p->heads, p->nbatch);
  if (!p->causal && use_wg2_bwd(p->Sq))
    T2V_LAUNCH_FIRST(attn_bwd_dq_wg2_kernel, dim3((p->Sq + 255) / 256, p->heads, p->nbatch), dim3(256), 0, (hipStream_t)stream, *p);
  else if (!p->causal && use_wg_bwd(p->Sq))
    T2V_LAUNCH_FIRST(attn_bwd_dq_wg_kernel, dim3((p->Sq + 127) / 128, p->heads, p->nbatch), dim3(256), 0, (hipStream_t)stream, *p);
  else
    T2V_LAUNCH_FIRST(attn_bwd_dq_kernel, gq, dim3(64), 0, (hipStream_t)stream, *p);
  T2V_CHECK_LAUNCH();
  dim3 gk((p->Sk + 31) / 32, p->heads, p->nbatch);
  if (!p->causal && use_wg2_bwd(p->Sk))
    T2V_LAUNCH_LAST(attn_bwd_dkdv_wg2_kernel, dim3((p->Sk + 255) / 256, p->heads, p->nbatch), dim3(256), 0, (hipStream_t)stream, *p);
  else if (!p->causal && use_wg_bwd(p->Sk))
    T2V_LAUNCH_LAST(attn_bwd_dkdv_wg_kernel, dim3((p->Sk + 127) / 128, p->heads, p->nbatch), dim3(256), 0, (hipStream_t)stream, *p);
  else
    T2V_LAUNCH_LAST(attn_bwd_dkdv_kernel, gk, dim3(64), 0, (hipStream_t)stream, *p);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}
