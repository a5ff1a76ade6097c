// colsum.h — GroupNorm statistics in a GEMM epilogue (T2VGemm.colsum, include/t2v_abi.h): shared by the 4-wave kernels of
// gemm.hip and the 8-wave kernels of gemm_w8.hip.  A thread of the row-writing pass owns ONE 8-column chunk for all rows it
// writes; it accumulates the two sums over its rows, the threads of a chunk are combined in fixed order through LDS, and the
// tile's column sums go to colsum[4 + (tile_row * Nb + column) * 2].
#pragma once
#include "common.h"

struct CsState {
  float s1[8], s2[8];               // running sums of this thread's columns
  float mu[8], rs[8], g[8], b[8];   // mode 2: forward statistics and affine terms of the norm, per column
  DropKey dk;                       // mode 2 with cs_drop_p > 0: the dropout the norm applies behind its SiLU (TemporalConvLayer)
  float ks;
  int width;
};

// `active`: this thread writes base-output columns col..col+7 of a tile whose rows start at m0 (inside ONE domain)
__device__ __forceinline__ void cs_init(CsState& c, const T2VGemm& p, int mode, bool active, long long m0, int col, int Nb) {
#pragma unroll
  for (int e = 0; e < 8; ++e) c.s1[e] = c.s2[e] = c.mu[e] = c.rs[e] = c.g[e] = c.b[e] = 0.f;
  c.ks = 1.f;
  c.width = Nb;
  if (mode == 2 && p.cs_drop_p > 0.f) {
    c.dk = drop_key(eff_seed(p.cs_drop_seed, p.drop_epoch), p.cs_drop_p);
    c.ks = 1.f / (1.f - p.cs_drop_p);
  }
  if (mode == 2 && active) {
    const int cpg = Nb / p.cs_G;
    const int dom = (int)(m0 / p.cs_domain_rows);
    const float icnt = 1.f / ((float)p.cs_domain_rows * (float)cpg);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cc = col + e;
      const float* sp = p.cs_sums + ((long long)dom * p.cs_G + cc / cpg) * 2;
      const float m = sp[0] * icnt;
      c.mu[e] = m;
      c.rs[e] = rsqrtf(fmaxf(sp[1] * icnt - m * m, 0.f) + p.cs_eps);
      c.g[e] = p.cs_gamma[cc];
      c.b[e] = p.cs_beta[cc];
    }
  }
}

// one stored row chunk: ov = the bf16 values just written; xrow (mode 2) = the norm's input at the same place; (row, col) = the
// chunk's position in the norm's [rows, C] matrix (the dropout mask index, t2v_gn_bwd_stats' protocol); drop: cs_drop_p > 0
__device__ __forceinline__ void cs_add(CsState& c, int mode, const bf16x8& ov, const bf16x8& xrow, int silu, unsigned row = 0,
                                       int col = 0, bool drop = false) {
  if (mode == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float q = bf2f((unsigned short)ov[e]);
      c.s1[e] += q;
      c.s2[e] += q * q;
    }
  } else if (mode == 2) {                          // the two sums of gn_stats_kernel<true> (norm.hip), same arithmetic
    // (the uniform flags are tested once per chunk, not per element — round 6, see gn_apply_kernel in norm.hip; without a mask
    //  kb = all ones and ks = 1, so the select needs no flag at all)
    const unsigned kb = drop ? drop_bits8(c.dk, (unsigned long long)row * (unsigned)c.width + (unsigned)col) : 0xffu;
    float xh[8], dz[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xh[e] = (bf2f((unsigned short)xrow[e]) - c.mu[e]) * c.rs[e];
      dz[e] = ((kb >> e) & 1u) ? bf2f((unsigned short)ov[e]) * c.ks : 0.f;
    }
    if (silu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float zz = xh[e] * c.g[e] + c.b[e];
        const float sg = sigmoid_f(zz);
        dz[e] *= sg * (1.f + zz * (1.f - sg));
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      c.s1[e] += dz[e] * c.g[e];
      c.s2[e] += dz[e] * c.g[e] * xh[e];
    }
  }
}

// combine the RG threads (index rg) of every column chunk (index cc, CPR chunks of a BNT-wide tile) in fixed order and store the
// tile's sums; `red`: LDS scratch of RG*BNT*2 floats, free on entry (the caller has synchronised); all NT threads take part
__device__ __forceinline__ void cs_flush(const CsState& c, const T2VGemm& p, float* red, bool active, int rg, int RG, int cc, int BNT,
                                         int tid, int NT, int n0, int ncols, int Nb, int tile_m, int BM) {
  if (active) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[((rg * BNT) + cc * 8 + e) * 2] = c.s1[e];
      red[((rg * BNT) + cc * 8 + e) * 2 + 1] = c.s2[e];
    }
  }
  __syncthreads();
  for (int c2 = tid; c2 < BNT; c2 += NT) {
    const int gc = n0 + c2;
    if (c2 >= ncols || gc >= Nb) continue;
    float a0 = 0.f, a1 = 0.f;
    for (int q = 0; q < RG; ++q) {
      a0 += red[((q * BNT) + c2) * 2];
      a1 += red[((q * BNT) + c2) * 2 + 1];
    }
    float* o = p.colsum + 4 + ((long long)tile_m * Nb + gc) * 2;
    o[0] = a0;
    o[1] = a1;
  }
  if (tile_m == 0 && n0 == 0 && tid == 0) {
    ((int*)p.colsum)[0] = BM;
    ((int*)p.colsum)[1] = Nb;
  }
}
