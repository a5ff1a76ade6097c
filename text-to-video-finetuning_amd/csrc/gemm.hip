// gemm.hip — bf16 MFMA GEMM family for gfx950 (CDNA4): one tile core, pluggable operand loaders.
//
//   D[M,N] = act(alpha * sum_k A[m,k] B[n,k] + bias[n] + rowbias[m/rpr, n]) + beta * R[m,n]
//
// Operand loaders (t2v_abi.h, T2VGemm):
//   A dense rows | A sliding-window gather (implicit-GEMM Conv2d 3x3 / stride 2 / nearest-upsampled source /
//   bwd-data of stride 2 / (3,1,1) temporal Conv3d) | A stored K-major (weight gradients)
//   B [N,K] weights | B stored K-major | B K-major with gathered rows (conv weight gradients)
//
// Structure (per 256-thread workgroup = 4 wave64): BMxBNx64 tile, v_mfma_f32_32x32x16_bf16, fp32 accumulators,
// register-staged global->LDS double buffer (global loads for tile t+1 are issued before the MFMAs of tile t and
// written to the other LDS stage after them: one barrier per K step), XOR-swizzled K-contiguous LDS image for
// row-major operands (conflict-free ds_read_b128 fragments), padded K-major LDS image + ds_read_b64_tr_b16
// hardware-transpose fragment reads for K-major operands, XCD-aware tile order (consecutive N tiles of one
// M tile share an XCD/L2).
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "colsum.h"

namespace {

constexpr int BK = 64;

struct Pos {
  int n, oy, ox, ok;
};

__device__ __forceinline__ Pos decompose(long long m, long long M, const T2VConvGeom& g) {
  Pos p;
  int hw = g.Ho * g.Wo;
  p.ok = m < M;
  int mm = p.ok ? (int)m : 0;
  p.n = mm / hw;
  int r = mm - p.n * hw;
  p.oy = r / g.Wo;
  p.ox = r - p.oy * g.Wo;
  return p;
}

// source row of output position `p` under tap `tap` (see T2VConvGeom); valid=false -> zero padding
__device__ __forceinline__ long long src_row(const Pos& p, int tap, const T2VConvGeom& g, bool& valid) {
  int ky = tap / g.KW, kx = tap - ky * g.KW;
  int vy = p.oy * g.sy + ky - g.py, vx = p.ox * g.sx + kx - g.px;
  bool v = p.ok != 0;
  if (g.tdiv == 2) {
    v = v && (((vy | vx) & 1) == 0);
    vy >>= 1;
    vx >>= 1;
  }
  v = v && ((unsigned)vy < (unsigned)g.Hv) && ((unsigned)vx < (unsigned)g.Wv);
  int Hr = g.Hv >> g.up, Wr = g.Wv >> g.up;
  valid = v;
  return ((long long)p.n * Hr + (vy >> g.up)) * Wr + (vx >> g.up);
}

// one 8-column output chunk: alpha, dropout, bias, row-bias, activation, residual, store (or second output block)
__device__ __forceinline__ void finish_chunk(const T2VGemm& p, float (&v)[8], long long row, int col, int z, long long zoffD,
                                             long long zoffR) {
  const int M = p.M, N = p.N;
  const float* bias = (const float*)p.bias;
  const bf16_t* rowbias = (const bf16_t*)p.rowbias;
  const bf16_t* R = p.R ? (const bf16_t*)p.R + zoffR : nullptr;
  const float keep_scale = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  do {
    const int nv = min(8, N - col);
    const bool full = nv == 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
    if (p.n_split > 0 && col >= p.n_split) {     // second output block (LoRA down projection): alpha only, bf16
      bf16_t* dp = (bf16_t*)p.D2 + row * p.ldd2 + (col - p.n_split);
      if (full) {
        *(bf16x8*)dp = pack8bf(v);
      } else {
        for (int e = 0; e < nv; ++e) dp[e] = f2bf(v[e]);
      }
      break;
    }
    if (p.drop_p > 0.f) {
      const DropKey dkey = drop_key(eff_seed(p.drop_seed, p.drop_epoch), p.drop_p);
      const unsigned kb = drop_bits8(dkey, ((unsigned long long)z * M + row) * N + col);     // (col % 8 == 0, N % 8 == 0: check_gemm)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ((kb >> e) & 1u) ? v[e] * keep_scale : 0.f;
    }
    if (bias) {
      if (full) {
        const float4 b0 = *(const float4*)(bias + col), b1 = *(const float4*)(bias + col + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < nv) v[e] += bias[col + e];
      }
    }
    if (rowbias) {
      const bf16_t* rb = rowbias + (row / p.rows_per_rb) * p.ldrb + col;
      if (full) {
        bf16x8 t = *(const bf16x8*)rb;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf2f((unsigned short)t[e]);
      } else {
        for (int e = 0; e < nv; ++e) v[e] += bf2f(rb[e]);
      }
    }
    if (p.act == T2V_ACT_SILU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
    }
    if (R) {
      const bf16_t* rp = R + row * p.ldr + col;
      if (full) {
        bf16x8 t = *(const bf16x8*)rp;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.beta * bf2f((unsigned short)t[e]);
      } else {
        for (int e = 0; e < nv; ++e) v[e] += p.beta * bf2f(rp[e]);
      }
    }
    const long long di = zoffD + row * p.ldd + col;
    if (p.out_mode == T2V_OUT_BF16) {
      bf16_t* dp = (bf16_t*)p.D + di;
      if (full) {
        *(bf16x8*)dp = pack8bf(v);
      } else {
        for (int e = 0; e < nv; ++e) dp[e] = f2bf(v[e]);
      }
    } else if (p.out_mode == T2V_OUT_F32) {
      float* dp = (float*)p.D + di;
      if (full) {
        *(float4*)dp = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(dp + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        for (int e = 0; e < nv; ++e) dp[e] = v[e];
      }
    } else {
      float* dp = (float*)p.D + di;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < nv) atomicAdd(dp + e, v[e]);
    }
  } while (0);
}

// ---- shared epilogue: accumulators -> LDS (fp32) -> 16-byte coalesced rows (bias / rowbias / act / residual fused)
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void epilogue(const T2VGemm& p, f32x16 (&acc)[BM / (WM * 32)][BN / (WN * 32)], unsigned char* smem,
                                         long long m0, int n0, int z, long long zoffD, long long zoffR) {
  constexpr int NT = WM * WN * 64;
  constexpr int FM = BM / (WM * 32), FN = BN / (WN * 32);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int M = p.M, N = p.N;
  float* sC = (float*)smem;   // (WM*32) x BN fp32 staging, one 32-row fragment band per pass (keeps LDS/WG small)
  constexpr int CPR = BN / 8;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    if (i > 0) __syncthreads();
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int rl = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int cl = wc * (FN * 32) + j * 32 + (lane & 31);
        sC[rl * BN + cl] = acc[i][j][r];
      }
    __syncthreads();
    // Accumulating fp32 output with no other epilogue term (the weight-gradient launches, dW += x^T dy): consecutive lanes take
    // consecutive columns of a row, so a wave's atomics cover two cache lines instead of sixteen, and a launch without K
    // splits owns every element and adds with plain loads / stores.  The 8-columns-per-thread form below ran at ~40 G atomic
    // floats/s: half of a 90 us weight-gradient launch at config C3 was its epilogue (scripts/kmajor_probe.py).
    if (p.out_mode == T2V_OUT_F32_ATOMIC && !p.bias && !p.rowbias && !p.R && p.act == T2V_ACT_NONE && p.drop_p == 0.f &&
        p.n_split <= 0) {
      const bool owned = p.split_k <= 1;
      float* Dp = (float*)p.D + zoffD;
      if (owned && (p.ldd & 3) == 0 && (N & 3) == 0) {
#pragma unroll 1
        for (int c = tid; c < WM * 32 * (BN / 4); c += NT) {
          const int rl = c / (BN / 4), c4 = c - rl * (BN / 4);
          const long long row = m0 + (rl >> 5) * (FM * 32) + i * 32 + (rl & 31);
          const int col = n0 + c4 * 4;
          if (row >= M || col >= N) continue;
          const float4 a = *(const float4*)(sC + rl * BN + c4 * 4);
          float4* dp = (float4*)(Dp + row * p.ldd + col);
          float4 d = *dp;
          d.x += p.alpha * a.x; d.y += p.alpha * a.y; d.z += p.alpha * a.z; d.w += p.alpha * a.w;
          *dp = d;
        }
      } else {
#pragma unroll 1
        for (int c = tid; c < WM * 32 * BN; c += NT) {
          const int rl = c / BN, cl = c - rl * BN;
          const long long row = m0 + (rl >> 5) * (FM * 32) + i * 32 + (rl & 31);
          const int col = n0 + cl;
          if (row >= M || col >= N) continue;
          const float v = p.alpha * sC[rl * BN + cl];
          if (owned) Dp[row * p.ldd + col] += v;
          else atomicAdd(Dp + row * p.ldd + col, v);
        }
      }
      continue;
    }
#pragma unroll 1
    for (int c = tid; c < WM * 32 * CPR; c += NT) {
      const int rl = c / CPR, cc = c - rl * CPR;
      const long long row = m0 + (rl >> 5) * (FM * 32) + i * 32 + (rl & 31);
      const int col = n0 + cc * 8;
      if (row >= M || col >= N) continue;
      float v[8];
      {
        const float4 a = *(const float4*)(sC + rl * BN + cc * 8);
        const float4 b = *(const float4*)(sC + rl * BN + cc * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      }
      if (p.ws_split > 1) {      // split-K partial: plain store into this split's slice (summed in fixed order by the finalize pass)
        float* wp = (float*)p.workspace + ((long long)z * M + row) * (long long)N + col;
        const int nvv = min(8, N - col);
        if (nvv == 8 && (N & 3) == 0) {
          *(float4*)wp = make_float4(v[0], v[1], v[2], v[3]);
          *(float4*)(wp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          for (int e = 0; e < nvv; ++e) wp[e] = v[e];
        }
        continue;
      }
      finish_chunk(p, v, row, col, z, zoffD, zoffR);
    }
  }
}

// ---- lean epilogue (bf16 output, no dropout / split-K, N % 8 == 0, every offset below 2^31 — what lean_ok() admits):
// 32-bit offsets, vector bias, hardware bf16 conversion, and the residual tile PREFETCHED into registers before the K loop
// (the load latency of `+ R` otherwise sits, un-overlapped, between the barrier-separated passes of the epilogue).
template <int BM, int BN, int WM, int WN>
struct EpiPre {
  static constexpr int NT = WM * WN * 64, FM = BM / (WM * 32), CPR = BN / 8;
  static constexpr int ITERS = (WM * 32 * CPR + NT - 1) / NT;
  static constexpr bool PRE = FM * ITERS <= 8;          // register budget: up to 8 x 16 bytes per thread
  bf16x8 r[PRE ? FM : 1][PRE ? ITERS : 1];
};

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void epi_prefetch(const T2VGemm& p, EpiPre<BM, BN, WM, WN>& pre, long long m0, int n0) {
  using E = EpiPre<BM, BN, WM, WN>;
  if constexpr (E::PRE) {
    const int tid = threadIdx.x;
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < E::FM; ++i)
#pragma unroll
      for (int it = 0; it < E::ITERS; ++it) {
        const int c = tid + E::NT * it;
        const int rl = c / E::CPR, cc = c - rl * E::CPR;
        const unsigned row = (unsigned)m0 + (rl >> 5) * (E::FM * 32) + i * 32 + (rl & 31);
        const int col = n0 + cc * 8;
        const bool ok = c < WM * 32 * E::CPR && row < (unsigned)p.M && col < p.N && (p.n_split <= 0 || col < p.n_split);
        pre.r[i][it] = (ok && p.R) ? *(const bf16x8*)((const bf16_t*)p.R + row * (unsigned)p.ldr + col) : z8;
      }
  }
}

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void epilogue_lean(const T2VGemm& p, f32x16 (&acc)[BM / (WM * 32)][BN / (WN * 32)],
                                              unsigned char* smem, long long m0, int n0, const EpiPre<BM, BN, WM, WN>& pre, int tm) {
  using E = EpiPre<BM, BN, WM, WN>;
  constexpr int NT = E::NT, FM = E::FM, FN = BN / (WN * 32), CPR = E::CPR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  float* sC = (float*)smem;
  const float* bias = (const float*)p.bias;
  const bf16_t* rowbias = (const bf16_t*)p.rowbias;
  const bf16_t* R = (const bf16_t*)p.R;
  // GroupNorm statistics of the stored tile (colsum.h): a thread keeps its column chunk over the passes when CPR divides NT
  constexpr bool CS_OK = NT % CPR == 0;
  const int cs_mode = (CS_OK && p.colsum) ? p.cs_mode : 0;
  const int Nb = p.n_split > 0 ? p.n_split : p.N;
  const int ccf = tid % CPR, rgf = tid / CPR, colf = n0 + ccf * 8;
  const bool cs_act = cs_mode != 0 && colf < Nb;
  CsState cst;
  if (cs_mode != 0) cs_init(cst, p, cs_mode, cs_act, m0, colf, Nb);
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    if (i > 0) __syncthreads();
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int rl = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int cl = wc * (FN * 32) + j * 32 + (lane & 31);
        sC[rl * BN + cl] = acc[i][j][r];
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < E::ITERS; ++it) {
      const int c = tid + NT * it;
      if (c >= WM * 32 * CPR) break;
      const int rl = c / CPR, cc = c - rl * CPR;
      const unsigned row = (unsigned)m0 + (rl >> 5) * (FM * 32) + i * 32 + (rl & 31);
      const int col = n0 + cc * 8;
      if (row >= (unsigned)p.M || col >= p.N) continue;
      float v[8];
      {
        const float4 a = *(const float4*)(sC + rl * BN + cc * 8);
        const float4 b = *(const float4*)(sC + rl * BN + cc * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      }
      if (p.alpha != 1.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
      }
      if (p.n_split > 0 && col >= p.n_split) {          // rank columns: second output block, alpha only
        *(bf16x8*)((bf16_t*)p.D2 + row * (unsigned)p.ldd2 + (col - p.n_split)) = pack8bf(v);
        continue;
      }
      if (bias) {
        const float4 b0 = *(const float4*)(bias + col), b1 = *(const float4*)(bias + col + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (rowbias) {
        const bf16x8 t = *(const bf16x8*)(rowbias + (row / (unsigned)p.rows_per_rb) * (unsigned)p.ldrb + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf2f((unsigned short)t[e]);
      }
      if (p.act == T2V_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
      }
      if (R) {
        bf16x8 t;
        if constexpr (E::PRE) t = pre.r[i][it];
        else t = *(const bf16x8*)(R + row * (unsigned)p.ldr + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.beta * bf2f((unsigned short)t[e]);
      }
      const bf16x8 ov = pack8bf(v);
      *(bf16x8*)((bf16_t*)p.D + row * (unsigned)p.ldd + col) = ov;
      if (cs_mode != 0) {
        bf16x8 xrow = ov;
        if (cs_mode == 2) xrow = *(const bf16x8*)((const bf16_t*)p.cs_x + row * (unsigned)p.cs_ldx + col);
        cs_add(cst, cs_mode, ov, xrow, p.cs_silu, row, col, cs_mode == 2 && p.cs_drop_p > 0.f);
      }
    }
  }
  if (cs_mode != 0) {
    __syncthreads();
    cs_flush(cst, p, sC, cs_act, rgf, NT / CPR, ccf, BN, tid, NT, n0, min(BN, p.N - n0), Nb, tm, BM);
  }
}

__device__ __attribute__((aligned(16))) unsigned g_zero_page[64];   // source for predicated-off LDS-DMA lanes (zero padding)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// NN kernel, LDS-DMA pipeline: operands stream HBM/L2 -> LDS with global_load_lds (16 B per lane, no VGPR staging)
// through an NSTAGE-deep ring; counted s_waitcnt vmcnt(N) + one raw s_barrier per K step keep NSTAGE-2 tiles in
// flight across the barrier, so small, latency-bound shapes (most of this UNet at batch 1) no longer serialise on a
// single prefetch.  LDS image is lane-linear (what the DMA writes); the XOR swizzle is applied to the per-lane
// SOURCE chunk and to the fragment reads (same involution).  Conv zero padding / edge rows read a zero page.
// LEAN = true: the loader of the common case (K % 64 == 0; windowed A: C % 64 == 0, so a K step never straddles a tap; every
// operand below 2 GiB).  Counters of the first build (SQ_INSTS_VALU : MFMA = 19-24 : 1, SQ_INSTS_SALU 11-16 : 1 on the 128x64
// tile, profiles/r02_gemm_counters.txt) showed the K loop bound by address arithmetic, not by MFMA or memory: every 16-byte
// chunk re-derived a 64-bit pointer, its validity and a zero-page select each K step.  Here operands go through buffer
// descriptors: a row's byte offset is a 32-bit VGPR that changes only when the window moves to the next tap, the K advance is
// a SCALAR offset shared by the wave, and rows that are padding / out of range carry an out-of-range offset so the hardware
// bounds check writes the zeros (no zero page, no select).
template <int BM, int BN, int WM, int WN, int NSTAGE, bool LEAN>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel_dma(const T2VGemm p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  warm_kernargs<(int)sizeof(T2VGemm)>();
  constexpr int FM = BM / (WM * 32), FN = BN / (WN * 32);
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int NT = WM * WN * 64, RPP = NT / 8;      // threads, tile rows covered per 16-byte-chunk pass
  constexpr int NCA = BM / RPP, NCB = BN / RPP, LPT = NCA + NCB;
  static_assert(BM % RPP == 0 && BN % RPP == 0 && RPP % 16 == 0, "tile/threads mismatch");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int M = p.M, N = p.N;
  const T2VConvGeom g = p.geom;
  const int ntn = (N + BN - 1) / BN;
  const int ntiles = gridDim.x;
  int t;
  {
    int q = ntiles >> 3, r = ntiles & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  if (ntn == 1) {              // (an integer division is ~40 instructions here)
    tm = t;
    tn = 0;
  } else if (p.raster_n & 1) { // an XCD's run covers a few N-tiles x all M-tiles: its weight columns are fetched once, by it alone
    const int ntm = (M + BM - 1) / BM;
    tn = t / ntm;
    tm = t - tn * ntm;
  } else {                     // an XCD's run covers a few M-tiles x all N-tiles: its activation rows stay in its L2
    tm = t / ntn;
    tn = t - tm * ntn;
  }
  const long long m0 = (long long)tm * BM;
  const int n0 = tn * BN;
  const int z = blockIdx.z;
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;
  long long zoffD = 0, zoffR = 0;
  int kbeg = 0, kend = p.K;
  if (p.ws_split > 1) {
    int per = ((p.K + p.ws_split - 1) / p.ws_split + BK - 1) / BK * BK;
    kbeg = z * per;
    kend = min(p.K, kbeg + per);
    if (kbeg >= kend) return;
  } else {
    A += (long long)z * p.strideA;
    B += (long long)z * p.strideB;
    zoffD = (long long)z * p.strideD;
    zoffR = (long long)z * p.strideR;
  }
  const bf16_t* zp = (const bf16_t*)g_zero_page;

  // chunk c = tid + 256*i of a tile sits at LDS byte c*16 (row c>>3, slot c&7) and holds source chunk slot^swz(row)
  const int kc = (tid & 7) ^ ((tid >> 4) & 7);          // (row>>1)&7 is the same for all i: rows differ by multiples of 16
  // conv gather state: per row the window origin (n, vy0, vx0); per thread the running (k index, channel, ky, kx) of its
  // chunk — advanced incrementally every K step, so the steady-state loop has no integer division
  int rn[NCA], rvy[NCA], rvx[NCA];
  const bf16_t* arow[NCA];
  bool aok[NCA];
  const bool is_conv = p.a_mode == T2V_A_CONV;
  const int Hr = g.Hv >> g.up, Wr = g.Wv >> g.up;
#pragma unroll
  for (int i = 0; i < NCA; ++i) {
    long long m = m0 + (tid >> 3) + RPP * i;
    aok[i] = m < M;
    if (is_conv) {
      Pos ps = decompose(m, M, g);
      rn[i] = ps.n * Hr;
      rvy[i] = ps.oy * g.sy - g.py;
      rvx[i] = ps.ox * g.sx - g.px;
      arow[i] = A;
    } else {
      rn[i] = rvy[i] = rvx[i] = 0;
      arow[i] = A + (aok[i] ? m : 0) * p.lda;
    }
  }
  int ck = 0, cky = 0, ckx = 0;                          // channel offset / tap row / tap col of this thread's chunk
  if (is_conv) {
    const int kidx0 = kbeg + kc * 8;
    const int tap0 = kidx0 / g.C;
    ck = kidx0 - tap0 * g.C;
    cky = tap0 / g.KW;
    ckx = tap0 - cky * g.KW;
  }
  const bf16_t* brow[NCB];
  bool bok[NCB];
  unsigned b2mask = 0;                                   // rows of the second weight block (valid only inside its K window)
  const int wlo = p.b2_klen > 0 ? p.b2_k0 : 0, whi = p.b2_klen > 0 ? p.b2_k0 + p.b2_klen : 0x7fffffff;
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    int n = n0 + (tid >> 3) + RPP * i;
    bok[i] = n < N;
    if (p.n_split > 0 && n >= p.n_split) {
      brow[i] = (const bf16_t*)p.B2 + (long long)(bok[i] ? n - p.n_split : 0) * p.ldb2 - wlo;
      b2mask |= 1u << i;
    } else {
      brow[i] = B + (long long)(bok[i] ? n : 0) * p.ldb;
    }
  }

  // ---- lean loader state (LEAN only)
  constexpr unsigned OOB = 0x80000000u;                 // >= num_records of every descriptor: the load returns zeros
  unsigned va[NCA], vb[NCB];
  int sc = 0, sky = 0, skx = 0;                          // scalar position of the NEXT K step inside the window: channel, tap row/col
  __amdgpu_buffer_rsrc_t srdA, srdB, srdB2;
  unsigned b2lane = 0;                                   // load i of this WAVE reads rows of the second weight block
  auto conv_rows = [&]() {                               // byte offsets of this thread's A rows under tap (sky, skx)
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      int vy = rvy[i] + sky, vx = rvx[i] + skx;
      bool v = aok[i];
      if (g.tdiv == 2) {
        v = v && (((vy | vx) & 1) == 0);
        vy >>= 1;
        vx >>= 1;
      }
      v = v && ((unsigned)vy < (unsigned)g.Hv) && ((unsigned)vx < (unsigned)g.Wv);
      const unsigned sr = (unsigned)((rn[i] + (vy >> g.up)) * Wr + (vx >> g.up));
      va[i] = v ? (sr * (unsigned)p.lda + (unsigned)kc * 8u) * 2u : OOB;
    }
  };
  if constexpr (LEAN) {
    srdA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x80000000u, 0x00020000);
    srdB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x80000000u, 0x00020000);
    srdB2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.n_split > 0 ? p.B2 : p.B), 0, 0x80000000u, 0x00020000);
    if (is_conv) {
      const int tap0 = kbeg / g.C;
      sc = kbeg - tap0 * g.C;
      sky = tap0 / g.KW;
      skx = tap0 - sky * g.KW;
      conv_rows();
    } else {
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        const long long m = m0 + (tid >> 3) + RPP * i;
        va[i] = aok[i] ? ((unsigned)m * (unsigned)p.lda + (unsigned)kc * 8u) * 2u : OOB;
      }
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int n = n0 + (tid >> 3) + RPP * i;
      // rows of one wave and one i are 8 consecutive n starting at a multiple of 8, and n_split % 8 == 0: wave-uniform side
      const bool second = p.n_split > 0 && (n0 + (__builtin_amdgcn_readfirstlane(wave) << 3) + RPP * i) >= p.n_split;
      if (second) b2lane |= 1u << i;
      const unsigned row = (unsigned)(second ? n - p.n_split : n);
      vb[i] = bok[i] ? (row * (unsigned)(second ? p.ldb2 : p.ldb) + (unsigned)kc * 8u) * 2u : OOB;   // K window origin: scalar offset below
    }
  }

  auto issue = [&](int k0, int stage) {
    unsigned char* sA = smem + stage * STAGE;
    unsigned char* sB = sA + A_BYTES;
    if constexpr (LEAN) {
      const int soa = (is_conv ? sc : k0) * 2;           // scalar byte offset of this K step inside the A rows
#pragma unroll
      for (int i = 0; i < NCA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (__attribute__((address_space(3))) void*)(sA + (tid + NT * i) * 16), 16,
                                                 (int)va[i], soa, 0, 0);
      const bool win = k0 >= wlo && k0 < whi;            // K steps never straddle the window either (b2_k0, b2_klen % 64 == 0)
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        const bool second = (b2lane >> i) & 1u;          // wave-uniform
        const unsigned vo = (second && !win) ? OOB : vb[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? srdB2 : srdB,
                                                 (__attribute__((address_space(3))) void*)(sB + (tid + NT * i) * 16), 16, (int)vo,
                                                 (second ? k0 - wlo : k0) * 2, 0, 0);   // offsets add zero-extended: never negative
      }
      if (is_conv) {                                     // advance the window position of the next K step
        sc += BK;
        if (sc >= g.C) {
          sc = 0;
          if (++skx == g.KW) {
            skx = 0;
            ++sky;
          }
          conv_rows();
        }
      }
      return;
    }
    const int kidx = k0 + kc * 8;
    const bool kok = kidx < kend;
    if (is_conv) {
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        int vy = rvy[i] + cky, vx = rvx[i] + ckx;
        bool v = aok[i] && kok;
        if (g.tdiv == 2) {
          v = v && (((vy | vx) & 1) == 0);
          vy >>= 1;
          vx >>= 1;
        }
        v = v && ((unsigned)vy < (unsigned)g.Hv) && ((unsigned)vx < (unsigned)g.Wv);
        const long long sr = (long long)(rn[i] + (vy >> g.up)) * Wr + (vx >> g.up);
        const bf16_t* src = v ? A + sr * p.lda + ck : zp;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sA + (tid + NT * i) * 16), 16, 0, 0);
      }
      ck += BK;                                           // next K step of this thread's chunk
      while (ck >= g.C) {
        ck -= g.C;
        if (++ckx == g.KW) {
          ckx = 0;
          ++cky;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        const bf16_t* src = (aok[i] && kok) ? arow[i] + kidx : zp;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sA + (tid + NT * i) * 16), 16, 0, 0);
      }
    }
    const bool win = kidx >= wlo && kidx < whi;
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const bf16_t* src = (bok[i] && kok && (win || !((b2mask >> i) & 1u))) ? brow[i] + kidx : zp;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sB + (tid + NT * i) * 16), 16, 0, 0);
    }
  };

  // lean epilogue conditions beyond lean_ok(): bf16 output, no dropout, no split-K, whole 8-column chunks
  const bool epi_fast = LEAN && p.out_mode == T2V_OUT_BF16 && p.ws_split <= 1 && p.drop_p == 0.f && (N & 7) == 0 && !(p.raster_n & 2);
  EpiPre<BM, BN, WM, WN> pre;
  if constexpr (LEAN) {
    if (epi_fast) epi_prefetch<BM, BN, WM, WN>(p, pre, m0, n0);      // issued BEFORE the first LDS-DMA: older than every counted load
  }
  // ... and the ring's first stages go out before the rest of the set-up (accumulators): their latency is the longest item
  // in front of the first MFMA
  const int nt = (kend - kbeg + BK - 1) / BK;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nt) issue(kbeg + s * BK, s);

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage) {
    const unsigned char* sA = smem + stage * STAGE;
    const unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        int row = wr * (FM * 32) + i * 32 + (lane & 31);
        int kch = kk * 2 + (lane >> 5);
        af[i] = *(const bf16x8*)(sA + row * 128 + ((kch ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        int row = wc * (FN * 32) + j * 32 + (lane & 31);
        int kch = kk * 2 + (lane >> 5);
        bfr[j] = *(const bf16x8*)(sB + row * 128 + ((kch ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  int stage = 0;
  for (int it = 0; it < nt; ++it) {
    const int ahead = min(nt, it + NSTAGE - 1) - (it + 1);      // tiles allowed to stay in flight
    if (ahead >= 2) wait_vmcnt<2 * LPT>();
    else if (ahead == 1) wait_vmcnt<LPT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                                // tile `it` landed for every wave; stage (it-1)%S is free
    if (it + NSTAGE - 1 < nt) {
      int ws = stage + NSTAGE - 1;
      if (ws >= NSTAGE) ws -= NSTAGE;
      issue(kbeg + (it + NSTAGE - 1) * BK, ws);
    }
    compute(stage);
    if (++stage == NSTAGE) stage = 0;
  }
  wait_vmcnt<0>();
  __syncthreads();                                               // ring is idle: reuse it for the epilogue staging
  if constexpr (LEAN) {
    if (epi_fast) {
      epilogue_lean<BM, BN, WM, WN>(p, acc, smem, m0, n0, pre, tm);
      return;
    }
  }
  epilogue<BM, BN, WM, WN>(p, acc, smem, m0, n0, z, zoffD, zoffR);
}

template <int BM, int BN, int WM, int WN, bool AT, bool BT>
__device__ __forceinline__ void gemm_body(const T2VGemm& p, unsigned char* smem, const int bid, const int ntiles, const int z) {
  constexpr int FM = BM / (WM * 32), FN = BN / (WN * 32);
  constexpr int LDA_T = BM + 32, LDB_T = BN + 32;  // K-major LDS row strides (elements): +64 B keeps tr reads conflict-free
  constexpr int A_BYTES = AT ? BK * LDA_T * 2 : BM * BK * 2;
  constexpr int B_BYTES = BT ? BK * LDB_T * 2 : BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NCA = BM / 32, NCB = BN / 32;  // 16-byte chunks per thread per K step

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int M = p.M, N = p.N;
  const T2VConvGeom g = p.geom;

  // ---- XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous run of tiles
  const int ntn = (N + BN - 1) / BN;
  int t;
  {
    int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = t / ntn, tn = t - tm * ntn;
  const long long m0 = (long long)tm * BM;
  const int n0 = tn * BN;

  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;
  int kbeg = 0, kend = p.K;
  long long zoffD = 0, zoffR = 0;
  if (p.split_k > 1) {
    int per = ((p.K + p.split_k - 1) / p.split_k + BK - 1) / BK * BK;
    kbeg = z * per;
    kend = min(p.K, kbeg + per);
    if (kbeg >= kend) return;
  } else {
    A += (long long)z * p.strideA;
    B += (long long)z * p.strideB;
    zoffD = (long long)z * p.strideD;
    zoffR = (long long)z * p.strideR;
  }

  // ---- per-thread loader state
  // row-major operand: thread owns rows (tid>>3)+32*i, 16-B column chunk (tid&7)
  // K-major operand:  thread owns k rows tid/(R/8) + (256/(R/8))*i, 8-row chunk tid%(R/8)
  Pos posA[NCA];
  const bf16_t* arow[NCA];
  bool aok[NCA];
  if constexpr (!AT) {
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      long long m = m0 + (tid >> 3) + 32 * i;
      aok[i] = m < M;
      if (p.a_mode == T2V_A_CONV) {
        posA[i] = decompose(m, M, g);
        arow[i] = A;
      } else {
        posA[i] = Pos{0, 0, 0, 0};
        arow[i] = A + (aok[i] ? m : 0) * p.lda;
      }
    }
  }
  const bf16_t* brow[NCB];
  bool bok[NCB];
  int btap = 0, bc = 0;
  unsigned b2mask = 0;
  const int wlo = p.b2_klen > 0 ? p.b2_k0 : 0, whi = p.b2_klen > 0 ? p.b2_k0 + p.b2_klen : 0x7fffffff;
  if constexpr (!BT) {
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      int n = n0 + (tid >> 3) + 32 * i;
      bok[i] = n < N;
      if (p.n_split > 0 && n >= p.n_split) {
        brow[i] = (const bf16_t*)p.B2 + (long long)(bok[i] ? n - p.n_split : 0) * p.ldb2 - wlo;
        b2mask |= 1u << i;
      } else {
        brow[i] = B + (long long)(bok[i] ? n : 0) * p.ldb;
      }
    }
  } else if (p.b_conv) {
    int nn = n0 + (tid % (BN / 8)) * 8;
    btap = nn / g.C;
    bc = nn - btap * g.C;
  }

  // three register sets: the global loads of K step i+3 are issued while step i computes (the operands of the weight-gradient
  // launches stream from HBM exactly once: with ONE step of prefetch every K step paid a full memory round trip, 3.6 us per
  // 64-deep step at config C3 — profiles/r03_c3_kernel_stats.txt)
  // (the sets are 4 x u32 vectors: as asm operands of type short8 the compiler re-packed them element-wise — v_lshrrev / v_perm on
  //  registers whose loads were still in flight, found in the 64x64 instantiation)
  typedef __attribute__((ext_vector_type(4))) unsigned kreg_t;
  kreg_t rA[3][NCA], rB[3][NCB];

  // Round 6, what the ISA of the weight-gradient launches showed (config C3: 575 launches, 40 ms, 0.093 of the MFMA peak; 63 % of
  // the wave cycles parked, scripts/kmajor_shape_run.py under the counter passes):
  //  (1) with the load inside the three-way branch on the (kernel-uniform) operand mode the compiler emitted one global_load per
  //      branch and an `s_waitcnt vmcnt(0)` at every merge: the loads of a step went out one at a time, each behind the previous
  //      one's round trip.  Addresses are formed first, the loads follow back to back (invalid chunks read the zero page instead
  //      of being predicated: no exec-mask branches either).
  //  (2) the three register sets bought nothing: the wait in front of the LDS store was vmcnt(0), i.e. it also waited for the loads
  //      issued a moment earlier for step i + 3 — every merge of the `if (i + 3 < nt)` / `if (i + 1 < nt)` conditionals inside
  //      the unrolled body makes the wait-count analysis conservative.  The steady state below is branch-free (see the main loop).
  //  (Hand-issued asm loads with counted waits were tried and REJECTED: the register sets are loop-carried, and the compiler
  //   copies / re-packs loop-carried values at the loop entry — i.e. while their loads are still in flight; wrong results in
  //   the 64x64 instantiation.)
  const bf16_t* zpage = (const bf16_t*)g_zero_page;
  auto ld16 = [](kreg_t& r, const bf16_t* ptr) { r = *(const kreg_t*)ptr; };
  auto load_tiles = [&](int k0, kreg_t(&ra)[NCA], kreg_t(&rb)[NCB]) {
    const bf16_t* pa[NCA];
    const bf16_t* pb[NCB];
    if constexpr (!AT) {
      const int kidx = k0 + (tid & 7) * 8;
      const bool kok = kidx < kend;
      if (p.a_mode == T2V_A_CONV) {
        const int tap = kidx / g.C;
        const int c = kidx - tap * g.C;
#pragma unroll
        for (int i = 0; i < NCA; ++i) {
          bool v;
          long long sr = src_row(posA[i], tap, g, v);
          v = v && kok;
          pa[i] = v ? A + sr * p.lda + c : zpage;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NCA; ++i) pa[i] = (aok[i] && kok) ? arow[i] + kidx : zpage;
      }
    } else {
      constexpr int CPR = BM / 8, RPP = 256 / CPR;
      const long long mm = m0 + (tid % CPR) * 8;
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        int kk = k0 + tid / CPR + RPP * i;
        bool v = (kk < kend) && (mm < M);
        pa[i] = v ? A + (long long)kk * p.lda + mm : zpage;
      }
    }
    if constexpr (!BT) {
      const int kidx = k0 + (tid & 7) * 8;
      const bool kok = kidx < kend;
      const bool win = kidx >= wlo && kidx < whi;
#pragma unroll
      for (int i = 0; i < NCB; ++i) pb[i] = (bok[i] && kok && (win || !((b2mask >> i) & 1u))) ? brow[i] + kidx : zpage;
    } else {
      constexpr int CPR = BN / 8, RPP = 256 / CPR;
      const int nn = n0 + (tid % CPR) * 8;
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        int kk = k0 + tid / CPR + RPP * i;
        bool v = (kk < kend) && (nn < N);
        const bf16_t* src;
        if (p.b_conv) {
          Pos ps = decompose(kk, kend, g);
          bool v2;
          long long sr = src_row(ps, btap, g, v2);
          v = v && v2;
          src = B + sr * p.ldb + bc;
        } else if (p.b_tapflip) {
          int tapi = kk / g.C, jr = kk - tapi * g.C;
          src = B + (long long)jr * p.ldb + (long long)(g.KH * g.KW - 1 - tapi) * N + nn;
        } else {
          src = B + (long long)kk * p.ldb + nn;
        }
        pb[i] = v ? src : zpage;
      }
    }
#pragma unroll
    for (int i = 0; i < NCA; ++i) ld16(ra[i], pa[i]);
#pragma unroll
    for (int i = 0; i < NCB; ++i) ld16(rb[i], pb[i]);
  };
  auto store_tiles = [&](int stage, const kreg_t(&ra)[NCA], const kreg_t(&rb)[NCB]) {
    unsigned char* sA = smem + stage * STAGE;
    unsigned char* sB = sA + A_BYTES;
    if constexpr (!AT) {
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        int row = (tid >> 3) + 32 * i;
        *(kreg_t*)(sA + row * 128 + ((((tid & 7) ^ ((row >> 1) & 7))) << 4)) = ra[i];
      }
    } else {
      constexpr int CPR = BM / 8, RPP = 256 / CPR;
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        int kr = tid / CPR + RPP * i;
        *(kreg_t*)(sA + (kr * LDA_T + (tid % CPR) * 8) * 2) = ra[i];
      }
    }
    if constexpr (!BT) {
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        int row = (tid >> 3) + 32 * i;
        *(kreg_t*)(sB + row * 128 + ((((tid & 7) ^ ((row >> 1) & 7))) << 4)) = rb[i];
      }
    } else {
      constexpr int CPR = BN / 8, RPP = 256 / CPR;
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        int kr = tid / CPR + RPP * i;
        *(kreg_t*)(sB + (kr * LDB_T + (tid % CPR) * 8) * 2) = rb[i];
      }
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int stage) {
    const unsigned char* sA = smem + stage * STAGE;
    const unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if constexpr (!AT) {
          int row = wr * (FM * 32) + i * 32 + (lane & 31);
          int kch = kk * 2 + (lane >> 5);
          af[i] = *(const bf16x8*)(sA + row * 128 + ((kch ^ ((row >> 1) & 7)) << 4));
        } else {
          int gq = lane >> 4, li = lane & 15;
          int mrow = wr * (FM * 32) + i * 32 + 16 * (gq & 1) + 4 * (li & 3);
          int kb = kk * 16 + 8 * (gq >> 1) + (li >> 2);
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sA + (kb * LDA_T + mrow) * 2));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sA + ((kb + 4) * LDA_T + mrow) * 2));
          af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if constexpr (!BT) {
          int row = wc * (FN * 32) + j * 32 + (lane & 31);
          int kch = kk * 2 + (lane >> 5);
          bfr[j] = *(const bf16x8*)(sB + row * 128 + ((kch ^ ((row >> 1) & 7)) << 4));
        } else {
          int gq = lane >> 4, li = lane & 15;
          int nrow = wc * (FN * 32) + j * 32 + 16 * (gq & 1) + 4 * (li & 3);
          int kb = kk * 16 + 8 * (gq >> 1) + (li >> 2);
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sB + (kb * LDB_T + nrow) * 2));
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(sB + ((kb + 4) * LDB_T + nrow) * 2));
          bfr[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- main loop: one barrier per K step
  const int nt = (kend - kbeg + BK - 1) / BK;
  load_tiles(kbeg, rA[0], rB[0]);
  if (nt > 1) load_tiles(kbeg + BK, rA[1], rB[1]);
  if (nt > 2) load_tiles(kbeg + 2 * BK, rA[2], rB[2]);
  store_tiles(0, rA[0], rB[0]);
  __syncthreads();
  int it = 0;
  // steady state: K steps it, it+1, it+2 with NO conditional inside — step i sits in register set i % 3 (until stored) and LDS
  // stage i & 1; every sub-step requests step i + 3, multiplies step i and stores step i + 1
  for (; it + 5 < nt; it += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = it + u, cur = i & 1;
      load_tiles(kbeg + (i + 3) * BK, rA[u], rB[u]);
      compute(cur);
      store_tiles(cur ^ 1, rA[(u + 1) % 3], rB[(u + 1) % 3]);
      __syncthreads();
    }
  }
  for (; it < nt; it += 3) {                         // the last (up to five) steps
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = it + u;
      if (i < nt) {
        const int cur = i & 1;
        if (i + 3 < nt) load_tiles(kbeg + (i + 3) * BK, rA[u], rB[u]);
        compute(cur);
        if (i + 1 < nt) store_tiles(cur ^ 1, rA[(u + 1) % 3], rB[(u + 1) % 3]);
        __syncthreads();
      }
    }
  }

  epilogue<BM, BN, WM, WN>(p, acc, smem, m0, n0, z, zoffD, zoffR);
}

// (two workgroups per CU: the branch-free steady-state copy of the K loop would otherwise take the 128x128 instantiation past 256
//  registers — one wave per SIMD, and the launches with many tiles lost more than the precise waits won)
template <int BM, int BN, int WM, int WN, bool AT, bool BT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const T2VGemm p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  warm_kernargs<(int)sizeof(T2VGemm)>();
  gemm_body<BM, BN, WM, WN, AT, BT>(p, smem, blockIdx.x, gridDim.x, blockIdx.z);
}

// two independent problems in ONE launch (the two factor gradients dU, dD of a LoRA layer): flat grid, problem 0 owns
// the first n0 = tiles0*z0 workgroups
template <int BM, int BN, int WM, int WN, bool AT, bool BT>
__global__ __launch_bounds__(256) void gemm_pair_kernel(const T2VGemm p0, const T2VGemm p1, int tiles0, int z0, int tiles1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  warm_kernargs<2 * (int)sizeof(T2VGemm) + 12>();
  const int b = blockIdx.x, n0 = tiles0 * z0;
  if (b < n0)
    gemm_body<BM, BN, WM, WN, AT, BT>(p0, smem, b % tiles0, tiles0, b / tiles0);
  else
    gemm_body<BM, BN, WM, WN, AT, BT>(p1, smem, (b - n0) % tiles1, tiles1, (b - n0) / tiles1);
}

template <int BM, int BN, int WM, int WN, bool AT, bool BT>
int launch(const T2VGemm& p, hipStream_t s) {
  constexpr int A_BYTES = AT ? BK * (BM + 32) * 2 : BM * BK * 2;
  constexpr int B_BYTES = BT ? BK * (BN + 32) * 2 : BN * BK * 2;
  constexpr int SMEM = 2 * (A_BYTES + B_BYTES);
  auto kern = gemm_kernel<BM, BN, WM, WN, AT, BT>;
  static bool attr_set = false;
  if (!attr_set) {
    if (SMEM > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, 1, p.split_k > 1 ? p.split_k : (p.batch > 1 ? p.batch : 1));
  hipLaunchKernelGGL(kern, grid, dim3(256), SMEM, s, p);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

bool g_force_regstage = false;   // T2V_GEMM_REGSTAGE=1: A/B the register-staged mainloop against the LDS-DMA pipeline

bool lean_ok(const T2VGemm& p) {
  static const int off = [] { const char* e = getenv("T2V_GEMM_LEAN"); return e && e[0] == '0'; }();
  if (off || p.K % 64 != 0 || p.batch > 1) return false;
  const long long lim = 0x7ff00000ll;                      // every byte offset (incl. a K step) stays below the 2 GiB record limit
  if (p.a_mode == T2V_A_CONV) {
    const T2VConvGeom& g = p.geom;
    if (g.C % 64 != 0) return false;
    const long long nimg = ((long long)p.M + (long long)g.Ho * g.Wo - 1) / ((long long)g.Ho * g.Wo);
    if (nimg * (g.Hv >> g.up) * (g.Wv >> g.up) * p.lda * 2 > lim) return false;
  } else if ((long long)p.M * p.lda * 2 > lim) {
    return false;
  }
  if ((long long)p.N * p.ldb * 2 > lim) return false;
  if (p.n_split > 0 && ((long long)(p.N - p.n_split) * p.ldb2 * 2 > lim || p.n_split % 8 != 0)) return false;
  if (p.b2_klen > 0 && (p.b2_k0 % 64 != 0 || p.b2_klen % 64 != 0)) return false;
  // the lean epilogue indexes D, R, D2 and the row-bias with 32-bit element offsets
  if ((long long)p.M * p.ldd > lim || (p.R && (long long)p.M * p.ldr > lim) || (p.n_split > 0 && (long long)p.M * p.ldd2 > lim)) return false;
  if (p.rowbias && (long long)(p.M / (p.rows_per_rb > 0 ? p.rows_per_rb : 1) + 1) * p.ldrb > lim) return false;
  return true;
}

template <int BM, int BN, int WM, int WN, int NSTAGE, bool LEAN>
int launch_dma_v(const T2VGemm& p, hipStream_t s) {
  constexpr int RING = NSTAGE * (BM + BN) * BK * 2;
  constexpr int EPI = WM * 32 * BN * 4;
  constexpr int SMEM = RING > EPI ? RING : EPI;
  auto kern = gemm_kernel_dma<BM, BN, WM, WN, NSTAGE, LEAN>;
  static bool attr_set = false;
  if (!attr_set) {
    if (SMEM > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, 1, p.ws_split > 1 ? p.ws_split : (p.batch > 1 ? p.batch : 1));
  if (p.ws_split <= 1)      // (a split-K launch is two kernels: it leaves the measurement hook's events untouched)
    T2V_LAUNCH(kern, grid, dim3(WM * WN * 64), SMEM, s, p);
  else
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), SMEM, s, p);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
int launch_dma(const T2VGemm& p, hipStream_t s) {
  return lean_ok(p) ? launch_dma_v<BM, BN, WM, WN, NSTAGE, true>(p, s) : launch_dma_v<BM, BN, WM, WN, NSTAGE, false>(p, s);
}

__global__ __launch_bounds__(256) void zero_f32_kernel(float4* p, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
    p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// finalize pass of a workspace split-K launch: full epilogue over the fp32 partial sums
__global__ __launch_bounds__(256) void gemm_finalize_kernel(const T2VGemm p, int nsplit) {
  const int cpr = (p.N + 7) / 8;
  const long long n = (long long)p.M * cpr;
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < n; c += (long long)gridDim.x * 256) {
    const long long row = c / cpr;
    const int col = (int)(c - row * cpr) * 8;
    float v[8];
    const int nv = min(8, p.N - col);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int zz = 0; zz < nsplit; ++zz) {
      const float* wp = (const float*)p.workspace + ((long long)zz * p.M + row) * (long long)p.N + col;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < nv) v[e] += wp[e];
    }
    finish_chunk(p, v, row, col, 0, 0, 0);
  }
}

// pick the tile that minimises (waves of workgroups) x (tile cost); ~2 workgroups resident per CU
// ---- NN (LDS-DMA) launch configurations: tile x ring depth x workspace split-K -------------------------------------
// conditions of the 8-wave kernels (gemm_w8.hip): the lean loader plus the bf16 epilogue without dropout / split-K
bool w8_ok(const T2VGemm& p) {
  return !p.a_trans && !p.b_trans && p.split_k <= 1 && p.batch <= 1 && p.out_mode == T2V_OUT_BF16 && p.drop_p == 0.f &&
         (p.N & 7) == 0 && lean_ok(p);
}

constexpr int W8_BASE = 100;   // DmaCfg.tile >= W8_BASE: 8-wave configuration tile - W8_BASE, stages = column step / 32 (0 = BN), split = K splits
struct DmaCfg {
  int tile;    // 0: 128x128, 1: 128x64, 2: 64x64, 3: 128x32, 4: 256x128, 5: 128x320, 6: 256x320, 7: 256x256, 8: 128x256 (4-7: 8 waves)
  int stages;  // 2 = occupancy variant, 0 = deep ring (3 for 128x128, 4 otherwise)
  int split;   // 1 = none, >1 = split K through the fp32 workspace + finalize pass
};

DmaCfg heuristic_cfg(const T2VGemm& p);

// tile rows of configuration `c` if its epilogue can emit T2VGemm.colsum for descriptor p, else 0
int colsum_bm(const T2VGemm& p, const DmaCfg& c) {
  if (!w8_ok(p)) return 0;
  if (c.tile >= W8_BASE) {
    const int w = c.tile - W8_BASE;
    // (mode 2 multiplies COMPLETE accumulators: the staged epilogue of a K-group configuration sums the groups too late)
    if ((p.lr_mode == 2 || p.lr_mode == 3) && (w == 14 || w == 16 || w == 19 || w == 20 || c.split > 1)) return 0;
    return t2v_gemm_w8_bm(w);
  }
  static const int no_epi = [] { const char* e = getenv("T2V_GEMM_EPI"); return e && e[0] == '0'; }();
  if (no_epi || c.split > 1) return 0;
  switch (c.tile) {                       // the lean epilogue keeps a thread on one column chunk when BN/8 divides the threads
    case 0: case 1: case 3: case 8: return 128;
    case 2: return 64;
    case 4: case 7: return 256;
    default: return 0;
  }
}

int launch_dma_cfg(const T2VGemm& p, const DmaCfg& c, hipStream_t s) {
  static const bool trace = getenv("T2V_GEMM_TRACE") != nullptr;      // one stderr line per launch: signature -> configuration
  if (trace)
    fprintf(stderr, "[t2v_gemm] M=%d N=%d K=%d a_mode=%d n_split=%d res=%d conv=%dx%d C=%d colsum=%d -> tile %d stages %d split %d\n", p.M, p.N, p.K,
            p.a_mode, p.n_split, p.R != nullptr, p.geom.KH, p.geom.KW, p.geom.C, p.colsum != nullptr, c.tile, c.stages, c.split);
  T2V_CHECK_ARG(!(p.colsum && p.cs_mode != 0) || colsum_bm(p, c) > 0,
                "t2v_gemm: colsum requested but the kernel selected for this descriptor cannot emit it (ask t2v_gemm_colsum_rows first)");
  if (c.tile >= W8_BASE) {
    if (w8_ok(p)) return t2v_gemm_w8_launch(p, c.tile - W8_BASE, c.stages * 32, c.split, s);
    T2V_CHECK_ARG(p.lr_mode == 0, "t2v_gemm: a rank-wide epilogue term (lr_mode) needs a descriptor the 8-wave kernels take");
    return launch_dma_cfg(p, heuristic_cfg(p), s);     // (a table entry met a descriptor outside the 8-wave domain)
  }
  T2V_CHECK_ARG(p.lr_mode == 0, "t2v_gemm: a rank-wide epilogue term (lr_mode) is only built into the 8-wave kernels");
  T2VGemm q = p;
  int split = c.split;
  // the first 64 KB of the caller's scratch belong to the arrival counters of the 8-wave kernels' in-launch split-K (they must
  // stay zero between launches): this path's partial slices live behind them
  constexpr size_t WS_RESERVED = 65536;
  // a cached / pinned configuration may ask for split-K while THIS launch brings no (or too small a) scratch buffer
  if (split > 1 && (!p.workspace || p.batch > 1 || p.out_mode != T2V_OUT_BF16 || p.workspace_bytes <= WS_RESERVED ||
                    (size_t)p.M * p.N * 4 * (size_t)split + 16 > p.workspace_bytes - WS_RESERVED))
    split = 1;
  if (split > 1) {
    q.workspace = (unsigned char*)p.workspace + WS_RESERVED;
    q.workspace_bytes = p.workspace_bytes - WS_RESERVED;
  }
  if (split > 1) {
    const int per = ((p.K + split - 1) / split + BK - 1) / BK * BK;
    split = (p.K + per - 1) / per;              // every split slice is written (no empty K ranges)
  }
  q.ws_split = split > 1 ? split : 0;
  {
    // Rasterisation by HBM-side traffic.  Each of the 8 XCDs has its own L2: with M-major runs every XCD streams the whole
    // weight matrix (8 |B| + |A|), with N-major runs every XCD streams all activations (|B| + 8 |A|).  |A| = unique source
    // bytes (a 3x3 window re-gathers rows that are already in the L2), |B| = N x K weights.
    // Measured on the C2 step (A/B in one gpurun call, 98.9 vs 99.1 ms): no difference — the Infinity Cache absorbs the
    // re-streamed weights — so the M-major order stays the default; T2V_GEMM_RASTER=2 selects by the traffic model, 1 forces N-major.
    static const int force = [] { const char* e = getenv("T2V_GEMM_RASTER"); return e ? atoi(e) : 0; }();
    const double bytesA = (double)p.M * (p.a_mode == T2V_A_CONV ? p.geom.C : p.K) * 2.0;
    const double bytesB = (double)p.N * p.K * 2.0;
    q.raster_n = force == 2 ? (bytesB + 8.0 * bytesA < bytesA + 8.0 * bytesB ? 1 : 0) : (force == 1 ? 1 : 0);
    if (p.batch > 1) q.raster_n = 0;
    static const int no_epi = [] { const char* e = getenv("T2V_GEMM_EPI"); return e && e[0] == '0'; }();   // A/B switch
    if (no_epi) q.raster_n |= 2;
  }
  int rc;
  const bool s2 = c.stages == 2;
  switch (c.tile) {
    case 0: rc = s2 ? launch_dma<128, 128, 2, 2, 2>(q, s) : launch_dma<128, 128, 2, 2, 3>(q, s); break;
    case 1: rc = s2 ? launch_dma<128, 64, 2, 2, 2>(q, s) : launch_dma<128, 64, 2, 2, 4>(q, s); break;
    case 2: rc = s2 ? launch_dma<64, 64, 2, 2, 2>(q, s) : launch_dma<64, 64, 2, 2, 4>(q, s); break;
    case 4: rc = s2 ? launch_dma<256, 128, 4, 2, 2>(q, s) : launch_dma<256, 128, 4, 2, 3>(q, s); break;
    // wide-N tiles: operand traffic per flop ~ (BM+BN)/(BM*BN); the L2->LDS DMA path (~11-12 TB/s) is what binds these
    // kernels, so N = 320/640/1280 layers want the whole (or half the) N extent in one tile
    case 5: rc = launch_dma<128, 320, 2, 2, 2>(q, s); break;
    case 6: rc = launch_dma<256, 320, 4, 2, 2>(q, s); break;
    case 7: rc = launch_dma<256, 256, 4, 2, 2>(q, s); break;
    case 8: rc = s2 ? launch_dma<128, 256, 2, 2, 2>(q, s) : launch_dma<128, 256, 2, 2, 3>(q, s); break;
    // N = 320 + 16 rank columns (LoRA down-projection riding in the base launch) in ONE tile instead of six 64-wide ones
    case 9: rc = launch_dma<128, 384, 2, 2, 2>(q, s); break;
    default: rc = s2 ? launch_dma<128, 32, 4, 1, 2>(q, s) : launch_dma<128, 32, 4, 1, 4>(q, s); break;
  }
  if (rc) return rc;
  if (split > 1) {
    q.ws_split = 1;
    long long nchunk = (long long)p.M * ((p.N + 7) / 8);
    hipLaunchKernelGGL(gemm_finalize_kernel, dim3((int)std::min<long long>((nchunk + 255) / 256, 4096)), dim3(256), 0, s, q, split);
    T2V_CHECK_LAUNCH();
  }
  return T2V_OK;
}

DmaCfg heuristic_cfg(const T2VGemm& p) {
  const bool shortk = p.K <= 640;
  const bool can_split = p.workspace && p.batch <= 1 && p.out_mode == T2V_OUT_BF16;
  if (can_split) {
    int bm = 64, bn = 64;
    if (p.N <= 32) { bm = 128; bn = 32; }
    long long tiles = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
    long long split = std::min<long long>(std::min<long long>(384 / std::max<long long>(1, tiles), p.K / 256), 32);
    split = std::min<long long>(split, (long long)((p.workspace_bytes > 65536 ? p.workspace_bytes - 65536 : 0) / ((size_t)p.M * p.N * 4 + 16)));
    if (tiles <= 96 && split >= 2) return DmaCfg{p.N <= 32 ? 3 : 2, 0, (int)split};
  }
  if (p.N <= 32) return DmaCfg{3, shortk ? 2 : 0, 1};
  const long long zdim = p.batch > 1 ? p.batch : 1;
  auto cost = [&](int bm, int bn, double eff) {
    long long tiles = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * zdim;
    long long waves = (tiles + 511) / 512;
    return (double)waves * bm * bn / eff;
  };
  double c0 = cost(128, 128, 1.0), c1 = cost(128, 64, 0.8), c2 = cost(64, 64, 0.55);
  int tile = (c0 <= c1 && c0 <= c2) ? 0 : (c1 <= c2 ? 1 : 2);
  return DmaCfg{tile, shortk ? 2 : 0, 1};
}

// First-use autotuning (eager launches only; never while a stream capture is active): every candidate is timed
// once with HIP events on the caller's stream and the winner is cached per problem signature.
struct TuneKey {
  int M, N, K, a_mode, n_split, out_mode, has_res, batch, KH, KW, sy, tdiv, up, C;
  bool operator==(const TuneKey& o) const { return memcmp(this, &o, sizeof(TuneKey)) == 0; }
};
struct TuneHash {
  size_t operator()(const TuneKey& k) const {
    const int* w = (const int*)&k;
    size_t h = 1469598103934665603ull;
    for (size_t i = 0; i < sizeof(TuneKey) / sizeof(int); ++i) h = (h ^ (size_t)w[i]) * 1099511628211ull;
    return h;
  }
};
std::unordered_map<TuneKey, DmaCfg, TuneHash> g_tuned;
std::mutex g_tune_mu;
int g_autotune = -1;

TuneKey make_key(const T2VGemm& p) {
  TuneKey key;
  memset(&key, 0, sizeof(key));
  key.M = p.M; key.N = p.N; key.K = p.K; key.a_mode = p.a_mode; key.n_split = (p.n_split > 0 ? 1 : 0) | (p.lr_mode << 1) | (p.lr_group_cols > 0 ? 8 : 0); key.out_mode = p.out_mode;
  key.has_res = p.R != nullptr; key.batch = p.batch > 1 ? p.batch : 1;
  if (p.a_mode == T2V_A_CONV) { key.KH = p.geom.KH; key.KW = p.geom.KW; key.sy = p.geom.sy; key.tdiv = p.geom.tdiv; key.up = p.geom.up; key.C = p.geom.C; }
  return key;
}

// Tile selection.  T2V_GEMM_AUTOTUNE:
//   unset / "table" : the shipped table (t2v_gemm_tune_import, loaded by the host binding from gemm_tune_gfx950.txt), else the
//                     heuristic — NEVER synchronises, every call is asynchronous on the caller's stream (ABI contract);
//   "live"          : unknown signatures are timed on first use (device synchronisation + HIP events; tuning runs only —
//                     scripts/tune_gemm_table.py exports the result as the shipped table);
//   "0"             : heuristic only.
// Configuration of a launch with a rank-wide epilogue term (T2VGemm.lr_mode != 0): one of the KG = 1 8-wave tiles, column step
// and K splits from a small cost model (rounds of 256 workgroups x tile area x (K per split + a fixed part)); table entries of
// the same signature (lr_mode is part of the key) override it.
bool lr_w8_cfg(int w) { return w == 12 || w == 14 || (w >= 16 && w <= 22 && w != 18); }   // (13 / 18: 256x256 LR kernels spill)
int lr_w8_bn(int w) { return (w == 14 || w == 19) ? 192 : ((w == 16 || w == 20) ? 256 : 384); }

DmaCfg lr_heuristic_cfg(const T2VGemm& p) {
  const bool lr2 = p.lr_mode == 2;
  const int rkc = p.lr_mode == 3 ? 32 * ((p.lr_rp + 31) / 32) : (lr2 ? 32 : 0);     // tile columns taken by the rank fragments
  // first choice: what the table holds for the SAME GEMM without the term — the rank-column launch [y | t] (forward) or the plain
  // launch (backward-data) — when that is an 8-wave configuration with an LR instantiation
  for (int pass = p.lr_mode == 3 ? 0 : 1; pass < 2; ++pass) {
    T2VGemm q = p;
    q.lr_mode = 0;
    if (pass == 0) {                               // mode 3: first what the table holds for the mode-1 form of the same launch
      q.lr_mode = 1;
      q.lr_group_cols = 0;
    }
    if (lr2) {
      q.N = p.N + p.lr_rp * (p.lr_group_cols > 0 ? p.N / p.lr_group_cols : 1);
      q.n_split = p.N;
      q.lr_group_cols = 0;
    }
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tuned.find(make_key(q));
    if (it != g_tuned.end() && it->second.tile >= W8_BASE && lr_w8_cfg(it->second.tile - W8_BASE)) {
      DmaCfg c = it->second;
      const int w = c.tile - W8_BASE, bn = lr_w8_bn(w);
      int step = c.stages > 0 ? c.stages * 32 : bn;
      if (p.lr_mode == 3 && step > bn - rkc) {
        // the rank fragments of mode 3 do not fit behind the borrowed column step: keep the step (and with it the number of
        // column tiles) and take the next wider tile of the same family instead of cutting the step
        int w2 = w == 14 ? 16 : (w == 19 ? 20 : (w == 16 ? 12 : (w == 20 ? 17 : -1)));
        while (w2 >= 0 && step > lr_w8_bn(w2) - rkc) w2 = w2 == 16 ? 12 : (w2 == 20 ? 17 : -1);
        if (w2 >= 0) {
          c.tile = W8_BASE + w2;
          c.stages = step / 32;
          return c;
        }
      }
      if (step > bn - rkc) step = bn - rkc;
      if (step >= 32) {
        c.stages = step / 32;
        return c;
      }
    }
  }
  struct Cand { int cfg, bm, bn, kg; };
  static const Cand cands[] = {{17, 128, 384, 1}, {19, 128, 192, 2}, {20, 128, 256, 2}};
  const int nk = p.K / 64;
  double best = 1e300;
  DmaCfg out{W8_BASE + 17, 0, 1};
  for (const Cand& c : cands) {
    const int bn_eff = c.bn - rkc;
    if (bn_eff < 32) continue;
    int ntn = (p.N + bn_eff - 1) / bn_eff;
    int step = ((p.N + ntn - 1) / ntn + 31) / 32 * 32;           // even split of N into whole fragments
    if (step > bn_eff) step = bn_eff;
    ntn = (p.N + step - 1) / step;
    const long long tiles = (long long)((p.M + c.bm - 1) / c.bm) * ntn;
    for (int sp : {1, 2, 3, 4, 6, 8}) {
      if (sp > 1 && (nk / sp < 4 || !p.workspace ||
                     65536ll + tiles * sp * c.bm * c.bn * 4 > (long long)p.workspace_bytes || tiles > 8000))
        continue;
      const long long rounds = (tiles * sp + 255) / 256;
      const double cols = step + rkc;
      const double cost = (double)rounds * c.bm * cols * ((double)p.K / sp + 1536.0 + (sp > 1 ? 768.0 : 0.0));
      if (cost < best) {
        best = cost;
        out = DmaCfg{W8_BASE + c.cfg, step / 32, sp};
      }
    }
  }
  return out;
}

DmaCfg pick_cfg(const T2VGemm& p, hipStream_t s) {
  if (g_autotune < 0) {
    const char* e = getenv("T2V_GEMM_AUTOTUNE");
    g_autotune = !e ? 1 : (e[0] == '0' ? 0 : ((e[0] == 'l' || e[0] == '2') ? 2 : 1));
  }
  const bool lr = p.lr_mode != 0;
  if (g_autotune || lr) {
    if (const char* f = getenv("T2V_GEMM_FORCE_CFG")) {     // "tile,stages,split": pin one configuration (counter passes, A/B runs)
      int t = 0, st = 2, sp = 1;
      if (sscanf(f, "%d,%d,%d", &t, &st, &sp) >= 1 && ((t >= 0 && t <= 9 && !lr) || t >= W8_BASE)) return DmaCfg{t, st, sp < 1 ? 1 : sp};
    }
  }
  if (lr) {
    {
      std::lock_guard<std::mutex> lk(g_tune_mu);
      auto it = g_tuned.find(make_key(p));
      if (it != g_tuned.end() && it->second.tile >= W8_BASE && lr_w8_cfg(it->second.tile - W8_BASE)) return it->second;
    }
    // (the choice never depends on whether column statistics are asked for: colsum_bm() answers 0 when the configuration this
    //  returns cannot emit them — a mode-2 launch on K groups or K splits — and the caller then runs the statistics pass)
    hipStreamCaptureStatus cs0 = hipStreamCaptureStatusNone;
    if (g_autotune != 2 || hipStreamIsCapturing(s, &cs0) != hipSuccess || cs0 != hipStreamCaptureStatusNone)
      return lr_heuristic_cfg(p);
  }
  if (!lr && !g_autotune) return heuristic_cfg(p);
  const TuneKey key = make_key(p);
  if (!lr) {
    {
      std::lock_guard<std::mutex> lk(g_tune_mu);
      auto it = g_tuned.find(key);
      if (it != g_tuned.end()) return it->second;
    }
    if (g_autotune != 2) return heuristic_cfg(p);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return heuristic_cfg(p);
  }
  // candidates
  std::vector<DmaCfg> cand;
  if (lr) {
    // launches with a rank-wide epilogue term: the LR instantiations of the 8-wave family x column steps x K splits
    static const int W8LR[][3] = {{12, 128, 384}, {17, 128, 384}, {21, 128, 384}, {22, 128, 384}, {14, 128, 192}, {19, 128, 192},
                                  {16, 128, 256}, {20, 128, 256}};
    const bool lr2 = p.lr_mode == 2 || p.lr_mode == 3;      // (column tiles of nstep base columns + rank fragments)
    const int rkc = p.lr_mode == 3 ? 32 * ((p.lr_rp + 31) / 32) : (p.lr_mode == 2 ? 32 : 0);
    for (const auto& w : W8LR) {
      const int bm = w[1], bn = w[2], bn_eff = bn - rkc;
      if (bn_eff < 32) continue;
      int steps[4], nsteps = 0;
      auto add = [&](int st) {
        if (st <= 0 || st > bn_eff) return;
        for (int i = 0; i < nsteps; ++i)
          if (steps[i] == st) return;
        steps[nsteps++] = st;
      };
      add(bn_eff);
      {
        const int ntn = (p.N + bn_eff - 1) / bn_eff;
        add(((p.N + ntn - 1) / ntn + 31) / 32 * 32);
      }
      if (p.N > 160) add(160);
      if (p.N > 320) add(320);
      for (int i = 0; i < nsteps; ++i) {
        const int st = steps[i];
        int ntn = 1;
        if (lr2) ntn = (p.N + st - 1) / st;
        else
          while ((long long)(ntn - 1) * st + bn < p.N) ++ntn;
        const long long wgs = (long long)((p.M + bm - 1) / bm) * ntn;
        for (int sp : {1, 2, 3, 4, 6, 8}) {
          if (sp > 1 && (wgs * sp > 320 || p.K / 64 / sp < 4 || !p.workspace)) continue;
          if (wgs * sp < 40 && sp < 8 && p.K / 64 / (sp + 1) >= 4) continue;
          cand.push_back(DmaCfg{W8_BASE + w[0], st / 32, sp});
        }
      }
    }
  }
  const bool can_split = p.workspace && p.batch <= 1 && p.out_mode == T2V_OUT_BF16;
  const int tiles_lo = p.N <= 32 ? 3 : 0, tiles_hi = p.N <= 32 ? 3 : 2;
  std::vector<int> tl;
  if (!lr)
    for (int t = tiles_lo; t <= tiles_hi; ++t) tl.push_back(t);
  static const int BMs[10] = {128, 128, 64, 128, 256, 128, 256, 256, 128, 128}, BNs[10] = {128, 64, 64, 32, 128, 320, 320, 256, 256, 384};
  if (!lr && p.N > 64 && (long long)p.M * p.N >= (long long)256 * 128 * 128) tl.push_back(4);   // big outputs: 8-wave 256x128 tile
  for (int t = 5; t <= 9 && !lr; ++t) {
    if (p.N < 256 || p.M < BMs[t]) continue;
    long long padded = (long long)((p.N + BNs[t] - 1) / BNs[t]) * BNs[t];
    if (padded * 100 > (long long)p.N * 115) continue;                     // <= 15 % padded columns
    if ((long long)((p.M + BMs[t] - 1) / BMs[t]) * (padded / BNs[t]) < 96) continue;   // enough workgroups
    tl.push_back(t);
  }
  for (int t : tl) {
    if (t < 3 && p.N <= 64 && BNs[t] > 64) continue;
    long long tiles = (long long)((p.M + BMs[t] - 1) / BMs[t]) * ((p.N + BNs[t] - 1) / BNs[t]);
    for (int st : {0, 2}) {
      if (st == 0 && (t == 5 || t == 6 || t == 7 || t == 9)) continue;       // these only exist as 2-stage rings (LDS)
      cand.push_back(DmaCfg{t, st, 1});
      if (can_split && st == 0) {
        for (int sp : {2, 4, 8, 16}) {
          if (p.K / sp < 256 || tiles * sp > 2048) continue;
          if ((size_t)p.M * p.N * 4 * sp + 16 + 65536 > p.workspace_bytes) continue;
          cand.push_back(DmaCfg{t, st, sp});
        }
      }
    }
  }
  if (!lr && w8_ok(p) && !getenv("T2V_GEMM_NO_W8")) {
    // 8-wave one-workgroup-per-CU configurations (gemm_w8.hip): tile BM x BN of the production set, column steps that split N
    // evenly into whole fragments (or 160 / 320 = half / whole level-0 width), K splits that bring the launch to about one
    // round of workgroups
    static const int W8[][3] = {{12, 128, 384}, {17, 128, 384}, {13, 256, 256}, {18, 256, 256}, {14, 128, 192}, {19, 128, 192},
                                {16, 128, 256}, {20, 128, 256}, {21, 128, 384}, {22, 128, 384}};
    for (const auto& w : W8) {
      const int bm = w[1], bn = w[2];
      if (p.M < bm / 2) continue;
      int steps[4], nsteps = 0;
      auto add = [&](int st) {
        if (st <= 0 || st > bn) return;
        for (int i = 0; i < nsteps; ++i)
          if (steps[i] == st) return;
        steps[nsteps++] = st;
      };
      add(bn);
      for (int ntn = 1; ntn < 64; ++ntn) {
        const int st = ((p.N + ntn - 1) / ntn + 31) / 32 * 32;
        if (st <= bn) { add(st); break; }
      }
      if (p.N > 160) add(160);
      if (p.N > 320) add(320);
      for (int i = 0; i < nsteps; ++i) {
        const int st = steps[i];
        int ntn = 1;
        while ((long long)(ntn - 1) * st + bn < p.N) ++ntn;
        const long long wgs = (long long)((p.M + bm - 1) / bm) * ntn;
        for (int sp : {1, 2, 3, 4, 6, 8}) {
          if (sp > 1 && (wgs * sp > 320 || p.K / 64 / sp < 4 || !p.workspace)) continue;
          if (wgs * sp < 40) continue;
          cand.push_back(DmaCfg{W8_BASE + w[0], st / 32, sp});
        }
      }
    }
  }
  T2VGemm q = p;
  if (q.R == q.D) q.R = nullptr;       // in-place accumulation must not be applied once per timing run
  q.drop_p = 0.f;
  const char* tl_env = getenv("T2V_GEMM_TUNE_LOG");
  const int tune_log = tl_env ? atoi(tl_env) : 0;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  DmaCfg best = lr ? lr_heuristic_cfg(p) : heuristic_cfg(p);
  float best_ms = 1e30f;
  (void)hipDeviceSynchronize();      // nothing else in flight (side-stream launches of earlier layers would skew the timings)
  // Time every candidate the way the step runs it: operands NOT resident in the XCD L2s (in the step they were just written
  // by another kernel, i.e. they come from the memory side).  The caller's scratch is overwritten before each timed launch
  // to evict them; the best of REPS single-launch timings counts.
  constexpr int REPS = 4;
  const bool evict = p.workspace && p.workspace_bytes >= ((size_t)32 << 20);
  for (const DmaCfg& c : cand) {
    if (launch_dma_cfg(q, c, s) != T2V_OK) continue;      // warm (also sets the LDS attribute)
    float ms = 1e30f;
    bool ok = true;
    for (int rd = 0; rd < REPS && ok; ++rd) {
      if (evict) (void)hipMemsetAsync(p.workspace, 0, p.workspace_bytes, s);
      (void)hipEventRecord(e0, s);
      launch_dma_cfg(q, c, s);
      (void)hipEventRecord(e1, s);
      if (hipEventSynchronize(e1) != hipSuccess) { ok = false; break; }
      float t = 0.f;
      (void)hipEventElapsedTime(&t, e0, e1);
      ms = t < ms ? t : ms;
    }
    if (!ok) continue;
    if (tune_log > 1) fprintf(stderr, "[t2v tune]   M=%d N=%d K=%d conv=%d tile %d stages %d split %d: %.1f us\n", p.M, p.N, p.K, p.a_mode, c.tile, c.stages, c.split, ms * 1000.f);
    if (ms < best_ms) { best_ms = ms; best = c; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipGetLastError();
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tuned[key] = best;
  }
  if (tune_log)
    fprintf(stderr, "[t2v tune] M=%d N=%d K=%d conv=%d -> tile %d stages %d split %d (%.1f us)\n", p.M, p.N, p.K, p.a_mode, best.tile,
            best.stages, best.split, best_ms * 1000.f);
  return best;
}


// ---- skinny dense launches: M <= 96 rows (the CLIP text tower of the step: 77 tokens x 24 layers, train.py:784-790; the
// time-embedding rows).  Such a problem is a weight stream — N*K*2 bytes read once — against <= 96 activation rows that every
// workgroup re-reads from L2; the tiled kernels spend their launch on ring set-up, K-split slabs and a finalize pass (2 x 10-15 us
// per layer).  Here a workgroup owns 32 output columns and its four waves split K in 64-deep chunks: a lane reads 64 contiguous
// bytes of its weight row and of its (up to three) activation rows per chunk straight from memory — the k index is permuted
// identically on both MFMA operands (lane half h holds k = 32h .. 32h+31 of the chunk, MFMA step s takes its s-th eight), so no
// operand passes through LDS and the K loop has no barrier.  Weights are the FIRST MFMA operand: a lane's accumulator then holds
// one token row and 4 x 4 consecutive output columns.  The four partial tiles meet in LDS, are summed in wave order
// (bit-reproducible) and go through the common epilogue (finish_chunk).
template <int MB>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(T2VGemm p) {
  __shared__ float red[4][MB][32][33];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;
  f32x16 acc[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool nok = n0 + r < p.N;
  const bf16_t* wrow = B + (long long)(n0 + r) * p.ldb + 32 * h;
  const int nchunks = p.K >> 6;
  for (int c = w; c < nchunks; c += 4) {
    bf16x8 wf[4], xf[MB][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) wf[s] = nok ? *(const bf16x8*)(wrow + (long long)c * 64 + 8 * s) : zero8;
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      const int row = 32 * b + r;
      const bf16_t* xrow = A + (long long)row * p.lda + (long long)c * 64 + 32 * h;
#pragma unroll
      for (int s = 0; s < 4; ++s) xf[b][s] = row < p.M ? *(const bf16x8*)(xrow + 8 * s) : zero8;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int b = 0; b < MB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], xf[b][s], acc[b], 0, 0, 0);
  }
  // accumulator register v of lane (r, h): output column n0 + 8 (v / 4) + 4 h + (v % 4), token row 32 b + r
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) red[w][b][8 * (v >> 2) + 4 * h + (v & 3)][r] = acc[b][v];
  __syncthreads();
  for (int ch = tid; ch < MB * 32 * 4; ch += 256) {
    const int row = ch >> 2, cq = ch & 3, b = row >> 5, rl = row & 31;
    const int col = n0 + cq * 8;
    if (row >= p.M || col >= p.N) continue;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (red[0][b][cq * 8 + e][rl] + red[1][b][cq * 8 + e][rl]) + (red[2][b][cq * 8 + e][rl] + red[3][b][cq * 8 + e][rl]);
    finish_chunk(p, v, row, col, 0, 0, 0);
  }
}

// Eight-wave variant (T2V_GEMM_SKINNY=2; NOT yet the default — written after the round's GPU time was spent, to be measured):
// the four-wave kernel above is latency-bound as it stands (17.9 us per CLIP layer in the step: one dependent round trip per
// 64-deep chunk and wave, four of them at K = 1024, sixteen at K = 4096).  Here eight waves split K and every wave keeps TWO
// chunks in flight (both chunks' loads are issued before the first MFMA), i.e. one round trip at K = 1024 and four at 4096;
// the eight partial tiles are folded in two stages through the same 50 KB of LDS (waves 4-7 park theirs, waves 0-3 add them to
// their registers in a fixed order and park the sums, the epilogue adds the four) — bit-reproducible like the four-wave form.
template <int MB>
__global__ __launch_bounds__(512) void gemm_skinny8_kernel(T2VGemm p) {
  __shared__ float red[4][MB][32][33];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;
  f32x16 acc[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool nok = n0 + r < p.N;
  const bf16_t* wrow = B + (long long)(n0 + r) * p.ldb + 32 * h;
  const int nchunks = p.K >> 6;
  auto load = [&](int c, bf16x8 (&wf)[4], bf16x8 (&xf)[MB][4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) wf[s] = nok ? *(const bf16x8*)(wrow + (long long)c * 64 + 8 * s) : zero8;
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      const int row = 32 * b + r;
      const bf16_t* xrow = A + (long long)row * p.lda + (long long)c * 64 + 32 * h;
#pragma unroll
      for (int s = 0; s < 4; ++s) xf[b][s] = row < p.M ? *(const bf16x8*)(xrow + 8 * s) : zero8;
    }
  };
  auto mma = [&](const bf16x8 (&wf)[4], const bf16x8 (&xf)[MB][4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int b = 0; b < MB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], xf[b][s], acc[b], 0, 0, 0);
  };
  for (int c = w; c < nchunks; c += 16) {
    bf16x8 wf0[4], xf0[MB][4], wf1[4], xf1[MB][4];
    const bool two = c + 8 < nchunks;              // wave-uniform
    load(c, wf0, xf0);
    if (two) load(c + 8, wf1, xf1);
    mma(wf0, xf0);
    if (two) mma(wf1, xf1);
  }
  // accumulator register v of lane (r, h): output column n0 + 8 (v / 4) + 4 h + (v % 4), token row 32 b + r
  if (w >= 4) {
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) red[w - 4][b][8 * (v >> 2) + 4 * h + (v & 3)][r] = acc[b][v];
  }
  __syncthreads();
  if (w < 4) {
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        float* q = &red[w][b][8 * (v >> 2) + 4 * h + (v & 3)][r];
        *q = acc[b][v] + *q;                         // wave w + (wave w + 4): the same lane wrote and reads this slot
      }
  }
  __syncthreads();
  for (int ch = tid; ch < MB * 32 * 4; ch += 512) {
    const int row = ch >> 2, cq = ch & 3, b = row >> 5, rl = row & 31;
    const int col = n0 + cq * 8;
    if (row >= p.M || col >= p.N) continue;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (red[0][b][cq * 8 + e][rl] + red[1][b][cq * 8 + e][rl]) + (red[2][b][cq * 8 + e][rl] + red[3][b][cq * 8 + e][rl]);
    finish_chunk(p, v, row, col, 0, 0, 0);
  }
}

// K split ACROSS workgroups (T2V_GEMM_SKINNY=3; NOT the default — written after the round's GPU time was spent, to be measured).
// What the probe of the two kernels above showed on hardware (profiles/r04_skinny_probe.txt + the step's trace): both take the same
// time (35 us at N = 1024, K = 4096; ~9 us at K = 1024), i.e. the bound is not the number of dependent round trips per wave but
// the bytes ONE compute unit pulls with four or eight waves (~25 GB/s: 256 KB of weights + 630 KB of re-read activations per
// workgroup at K = 4096) while 224 of the 256 CUs idle.  Here grid.y = KP workgroups share a column block, each owns a contiguous
// K part (two chunks in flight per wave), reduces its four waves through LDS as above and hands its 77 x 32 partial over with the
// in-launch split-K protocol of gemm_w8.hip (ticket; writers: plain stores -> vmcnt(0) -> barrier -> lane-0 agent release ->
// done mark; the LAST arriver waits for the marks, acquires, and sums the slabs IN SPLIT ORDER with its own partial in its
// place: bit-reproducible whoever comes last).  Workspace: counters of column block nb at words 2 nb, 2 nb + 1 of the first
// 64 KB (left zero), slabs behind it.
template <int MB>
__global__ __launch_bounds__(256) void gemm_skinnyk_kernel(T2VGemm p, int KP) {
  __shared__ float red[4][MB][32][33];
  __shared__ int s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nb = blockIdx.x, z = blockIdx.y, n0 = nb * 32;
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;
  f32x16 acc[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool nok = n0 + r < p.N;
  const bf16_t* wrow = B + (long long)(n0 + r) * p.ldb + 32 * h;
  const int per = (p.K >> 6) / KP;                      // chunks of this K part (the launcher makes KP divide the chunk count)
  const int c_end = (z + 1) * per;
  auto load = [&](int c, bf16x8 (&wf)[4], bf16x8 (&xf)[MB][4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) wf[s] = nok ? *(const bf16x8*)(wrow + (long long)c * 64 + 8 * s) : zero8;
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      const int row = 32 * b + r;
      const bf16_t* xrow = A + (long long)row * p.lda + (long long)c * 64 + 32 * h;
#pragma unroll
      for (int s = 0; s < 4; ++s) xf[b][s] = row < p.M ? *(const bf16x8*)(xrow + 8 * s) : zero8;
    }
  };
  auto mma = [&](const bf16x8 (&wf)[4], const bf16x8 (&xf)[MB][4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int b = 0; b < MB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], xf[b][s], acc[b], 0, 0, 0);
  };
  for (int c = z * per + w; c < c_end; c += 8) {
    bf16x8 wf0[4], xf0[MB][4], wf1[4], xf1[MB][4];
    const bool two = c + 4 < c_end;                  // wave-uniform
    load(c, wf0, xf0);
    if (two) load(c + 4, wf1, xf1);
    mma(wf0, xf0);
    if (two) mma(wf1, xf1);
  }
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) red[w][b][8 * (v >> 2) + 4 * h + (v & 3)][r] = acc[b][v];
  __syncthreads();
  // item ch = (token row ch >> 2, column chunk ch & 3); a thread owns items tid and tid + 256
  constexpr int ITEMS = MB * 32 * 4, NIT = (ITEMS + 255) / 256;
  float v[NIT][8];
  bool live[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int ch = tid + 256 * i, row = ch >> 2, cq = ch & 3, b = row >> 5, rl = row & 31;
    live[i] = ch < ITEMS && row < p.M && n0 + cq * 8 < p.N;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[i][e] = live[i] ? (red[0][b][cq * 8 + e][rl] + red[1][b][cq * 8 + e][rl]) + (red[2][b][cq * 8 + e][rl] + red[3][b][cq * 8 + e][rl]) : 0.f;
  }
  if (KP > 1) {
    unsigned* cnt = (unsigned*)p.workspace + 2 * nb;
    float* slab0 = (float*)((unsigned char*)p.workspace + 65536) + (long long)nb * KP * (ITEMS * 8);
    if (tid == 0) s_ticket = (int)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool reducer = s_ticket == KP - 1;
    if (!reducer) {
#pragma unroll
      for (int i = 0; i < NIT; ++i)
        if (live[i]) {
          float4* sp = (float4*)(slab0 + ((long long)z * ITEMS + tid + 256 * i) * 8);
          sp[0] = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
          sp[1] = make_float4(v[i][4], v[i][5], v[i][6], v[i][7]);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's slab rows have left
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the compiler may drop the fence's own wait, guide G16)
        __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(KP - 1)) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 24)) {                   // give up instead of hanging the device; read by check_gemm_workspaces()
          ((unsigned*)p.workspace)[16383] = 0xdeadu;
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // re-armed for the next launch
      __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      if (!live[i]) continue;
      float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int z2 = 0; z2 < KP; ++z2) {              // split order, own partial in its place
        if (z2 == z) {
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] += v[i][e];
        } else {
          const float4* sp = (const float4*)(slab0 + ((long long)z2 * ITEMS + tid + 256 * i) * 8);
          const float4 a = sp[0], c = sp[1];
          t[0] += a.x; t[1] += a.y; t[2] += a.z; t[3] += a.w; t[4] += c.x; t[5] += c.y; t[6] += c.z; t[7] += c.w;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = t[e];
    }
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i)
    if (live[i]) {
      const int ch = tid + 256 * i;
      finish_chunk(p, v[i], ch >> 2, n0 + (ch & 3) * 8, 0, 0, 0);
    }
}

// the descriptors gemm_skinny_kernel takes: plain dense NN, one output block, no statistics / rank-wide term / batch
bool skinny_ok(const T2VGemm& p) {
  static const bool on = [] { const char* e = getenv("T2V_GEMM_SKINNY"); return !(e && e[0] == '0'); }();
  return on && p.M <= 96 && !p.a_trans && !p.b_trans && p.a_mode == T2V_A_DENSE && !p.b_conv && p.K % 64 == 0 && p.n_split <= 0 &&
         p.lr_mode == 0 && !p.colsum && p.batch <= 1 && p.split_k <= 1 && p.out_mode == T2V_OUT_BF16;
}

int launch_skinny(const T2VGemm& p, hipStream_t s) {
  const dim3 grid((unsigned)((p.N + 31) / 32));
  static const int variant = [] { const char* e = getenv("T2V_GEMM_SKINNY"); return e ? atoi(e) : 1; }();
  if (variant == 3) {
    // K parts: enough workgroups for the chip (~256), at least four chunks (one per wave) each, a power of two that divides the
    // chunk count, and slabs that fit the caller's scratch
    const int nchunks = p.K >> 6, nb = (p.N + 31) / 32, mb = p.M <= 32 ? 1 : (p.M <= 64 ? 2 : 3);
    int kp = 1;
    while (kp * 2 * nb <= 256 && nchunks % (kp * 2) == 0 && nchunks / (kp * 2) >= 4) kp *= 2;
    const size_t need = 65536 + (size_t)nb * kp * mb * 32 * 32 * 4;
    if (kp > 1 && (!p.workspace || p.workspace_bytes < need || 2 * nb + 1 >= 16383)) kp = 1;
    const dim3 g3((unsigned)nb, (unsigned)kp);
    if (mb == 1) T2V_LAUNCH(gemm_skinnyk_kernel<1>, g3, dim3(256), 0, s, p, kp);
    else if (mb == 2) T2V_LAUNCH(gemm_skinnyk_kernel<2>, g3, dim3(256), 0, s, p, kp);
    else T2V_LAUNCH(gemm_skinnyk_kernel<3>, g3, dim3(256), 0, s, p, kp);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
  }
  const bool eight = variant == 2;
  if (eight) {
    if (p.M <= 32) T2V_LAUNCH(gemm_skinny8_kernel<1>, grid, dim3(512), 0, s, p);
    else if (p.M <= 64) T2V_LAUNCH(gemm_skinny8_kernel<2>, grid, dim3(512), 0, s, p);
    else T2V_LAUNCH(gemm_skinny8_kernel<3>, grid, dim3(512), 0, s, p);
    T2V_CHECK_LAUNCH();
    return T2V_OK;
  }
  if (p.M <= 32) T2V_LAUNCH(gemm_skinny_kernel<1>, grid, dim3(256), 0, s, p);
  else if (p.M <= 64) T2V_LAUNCH(gemm_skinny_kernel<2>, grid, dim3(256), 0, s, p);
  else T2V_LAUNCH(gemm_skinny_kernel<3>, grid, dim3(256), 0, s, p);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

template <bool AT, bool BT>
int dispatch(const T2VGemm& p, hipStream_t s) {
  constexpr bool DMA = !AT && !BT;
  if constexpr (DMA) {
    if (skinny_ok(p)) return launch_skinny(p, s);
    if (p.split_k <= 1 && !g_force_regstage) return launch_dma_cfg(p, pick_cfg(p, s), s);
  }
  T2V_CHECK_ARG(p.lr_mode == 0, "t2v_gemm: a rank-wide epilogue term (lr_mode) needs the NN LDS-DMA path");
  const long long zdim = p.split_k > 1 ? p.split_k : (p.batch > 1 ? p.batch : 1);
  auto cost = [&](int bm, int bn, double eff) {
    long long tiles = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * zdim;
    long long waves = (tiles + 511) / 512;
    return (double)waves * bm * bn / eff;
  };
  if (p.N <= 32) return launch<128, 32, 4, 1, AT, BT>(p, s);   // skinny outputs (LoRA rank): HBM-bound on A
  static const int force_tile = [] { const char* e = getenv("T2V_GEMM_FORCE_TILE"); return e ? atoi(e) : -1; }();   // probes
  if (force_tile == 0) return launch<128, 128, 2, 2, AT, BT>(p, s);
  if (force_tile == 1) return launch<128, 64, 2, 2, AT, BT>(p, s);
  if (force_tile == 2) return launch<64, 64, 2, 2, AT, BT>(p, s);
  double c0 = cost(128, 128, 1.0), c1 = cost(128, 64, 0.8), c2 = cost(64, 64, 0.55);
  if (c0 <= c1 && c0 <= c2) return launch<128, 128, 2, 2, AT, BT>(p, s);
  if (c1 <= c2) return launch<128, 64, 2, 2, AT, BT>(p, s);
  return launch<64, 64, 2, 2, AT, BT>(p, s);
}

int check_gemm(const T2VGemm& p);

}  // namespace

// Two K-major (a_trans = b_trans = 1) problems with fp32 atomic output in one launch — the LoRA factor gradients
// dU = s t^T dy and dD = s dt^T x_col of one layer (utils/lora.py:57-62 backward).
thread_local hipEvent_t t2v_time_start = nullptr, t2v_time_stop = nullptr;
thread_local int t2v_time_used = 0;
extern "C" int t2v_launch_timing_events(void* start, void* stop) {
  t2v_time_start = (hipEvent_t)start;
  t2v_time_stop = (hipEvent_t)stop;
  t2v_time_used = 0;
  return T2V_OK;
}
// 1 if an entry point called since t2v_launch_timing_events() filled the pair (clears the pair either way)
extern "C" int t2v_launch_timing_consumed(void) {
  const int used = t2v_time_used && !t2v_time_start && !t2v_time_stop;
  t2v_time_start = t2v_time_stop = nullptr;
  return used;
}

extern "C" int t2v_gemm_pair(const T2VGemm* pa, const T2VGemm* pb, t2v_stream_t stream) {
  T2V_CHECK_ARG(pa && pb, "t2v_gemm_pair: null descriptor");
  for (const T2VGemm* q : {pa, pb}) {
    if (int e = check_gemm(*q)) return e;
    T2V_CHECK_ARG(q->a_trans && q->b_trans && q->out_mode == T2V_OUT_F32_ATOMIC && q->batch <= 1,
                  "t2v_gemm_pair: both problems must be K-major with fp32 atomic output");
  }
  constexpr int BM = 64, BN = 64;
  auto kern = gemm_pair_kernel<BM, BN, 2, 2, true, true>;
  constexpr int SMEM = 2 * (BK * (BM + 32) * 2 + BK * (BN + 32) * 2);
  static bool attr_set = false;
  if (!attr_set) {
    if (SMEM > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  const int t0 = ((pa->M + BM - 1) / BM) * ((pa->N + BN - 1) / BN), z0 = pa->split_k > 1 ? pa->split_k : 1;
  const int t1 = ((pb->M + BM - 1) / BM) * ((pb->N + BN - 1) / BN), z1 = pb->split_k > 1 ? pb->split_k : 1;
  hipLaunchKernelGGL(kern, dim3(t0 * z0 + t1 * z1), dim3(256), SMEM, (hipStream_t)stream, *pa, *pb, t0, z0, t1);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

namespace {
int check_gemm(const T2VGemm& p) {
  T2V_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "t2v_gemm: bad dims M=%d N=%d K=%d", p.M, p.N, p.K);
  T2V_CHECK_ARG(p.A && p.B && p.D, "t2v_gemm: null operand");
  T2V_CHECK_ARG((p.a_trans && p.b_trans) || p.K % 8 == 0, "t2v_gemm: K=%d must be a multiple of 8", p.K);
  T2V_CHECK_ARG(p.lda % 8 == 0 && p.ldb % 8 == 0, "t2v_gemm: lda=%lld ldb=%lld must be multiples of 8", p.lda, p.ldb);
  T2V_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0, "t2v_gemm: operands must be 16-byte aligned");
  T2V_CHECK_ARG(p.ldd % 8 == 0 && ((uintptr_t)p.D & 15) == 0, "t2v_gemm: D must be 16-byte aligned with ldd%%8==0 (ldd=%lld)", p.ldd);
  T2V_CHECK_ARG(!p.R || (p.ldr % 8 == 0 && ((uintptr_t)p.R & 15) == 0), "t2v_gemm: R must be 16-byte aligned with ldr%%8==0");
  T2V_CHECK_ARG(!p.rowbias || (p.ldrb % 8 == 0 && ((uintptr_t)p.rowbias & 15) == 0), "t2v_gemm: rowbias must be 16-byte aligned, ldrb%%8==0");
  T2V_CHECK_ARG(p.batch <= 1 || (p.strideD % 8 == 0 && p.strideR % 8 == 0), "t2v_gemm: batch strides must be multiples of 8");
  T2V_CHECK_ARG(!(p.split_k > 1 && p.batch > 1), "t2v_gemm: split_k and batch are exclusive");
  T2V_CHECK_ARG(!(p.split_k > 1) || p.out_mode == T2V_OUT_F32_ATOMIC, "t2v_gemm: split_k needs atomic fp32 output");
  if (p.a_mode == T2V_A_CONV || p.b_conv) {
    const T2VConvGeom& g = p.geom;
    T2V_CHECK_ARG(g.C > 0 && g.C % 8 == 0, "t2v_gemm: conv gather needs C%%8==0 (C=%d)", g.C);
    T2V_CHECK_ARG(g.KH > 0 && g.KW > 0 && g.Ho > 0 && g.Wo > 0 && g.Hv > 0 && g.Wv > 0, "t2v_gemm: bad conv geometry");
    T2V_CHECK_ARG(g.tdiv == 1 || g.tdiv == 2, "t2v_gemm: tdiv must be 1 or 2");
    T2V_CHECK_ARG(g.up == 0 || g.up == 1, "t2v_gemm: up must be 0 or 1");
    if (p.a_mode == T2V_A_CONV) {
      T2V_CHECK_ARG(!p.a_trans, "t2v_gemm: conv gather on A needs a_trans=0");
      T2V_CHECK_ARG(p.K == g.KH * g.KW * g.C, "t2v_gemm: K=%d != KH*KW*C=%d", p.K, g.KH * g.KW * g.C);
    }
    if (p.b_conv) {
      T2V_CHECK_ARG(p.b_trans, "t2v_gemm: b_conv needs b_trans=1");
      T2V_CHECK_ARG(p.N == g.KH * g.KW * g.C, "t2v_gemm: N=%d != KH*KW*C=%d (b_conv)", p.N, g.KH * g.KW * g.C);
    }
  }
  if (p.a_trans) T2V_CHECK_ARG(p.M % 8 == 0, "t2v_gemm: a_trans needs M%%8==0 (M=%d)", p.M);
  if (p.b_trans) T2V_CHECK_ARG(p.N % 8 == 0, "t2v_gemm: b_trans needs N%%8==0 (N=%d)", p.N);
  if (p.rowbias) T2V_CHECK_ARG(p.rows_per_rb > 0, "t2v_gemm: rows_per_rb must be > 0");
  T2V_CHECK_ARG(p.b2_klen <= 0 || (p.n_split > 0 && p.b2_k0 >= 0 && p.b2_k0 % 8 == 0 && p.b2_klen % 8 == 0 &&
                                   p.b2_k0 + p.b2_klen <= p.K),
                "t2v_gemm: bad second-block K window [%d, +%d)", p.b2_k0, p.b2_klen);
  if (p.n_split > 0) {
    T2V_CHECK_ARG(!p.b_trans && p.B2 && p.D2 && p.n_split % 8 == 0 && p.n_split < p.N && p.ldb2 % 8 == 0 && p.ldd2 % 8 == 0 &&
                      p.out_mode == T2V_OUT_BF16 && p.batch <= 1 && p.split_k <= 1,
                  "t2v_gemm: bad split-output arguments (n_split=%d)", p.n_split);
  }
  if (p.b_tapflip)
    T2V_CHECK_ARG(p.b_trans && !p.b_conv && p.a_mode == T2V_A_CONV && p.K == p.geom.KH * p.geom.KW * p.geom.C,
                  "t2v_gemm: b_tapflip needs b_trans=1 and a conv gather on A");
  if (p.lr_mode != 0) {
    T2V_CHECK_ARG(p.lr_mode >= 1 && p.lr_mode <= 3, "t2v_gemm: lr_mode must be 0, 1, 2 or 3");
    T2V_CHECK_ARG(!p.a_trans && !p.b_trans && p.batch <= 1 && p.split_k <= 1 && p.out_mode == T2V_OUT_BF16 && p.drop_p == 0.f && (p.N & 7) == 0,
                  "t2v_gemm: a rank-wide epilogue term needs a plain NN launch with bf16 output");
    T2V_CHECK_ARG(p.lr_b && ((uintptr_t)p.lr_b & 15) == 0 && (p.lr_mode != 1 || (p.lr_a && ((uintptr_t)p.lr_a & 15) == 0)),
                  "t2v_gemm: lr_a / lr_b must be 16-byte aligned");
  }
  return T2V_OK;
}
}  // namespace

// ---- tuned-table transport: one text line per problem signature
//   M N K a_mode n_split out_mode has_res batch KH KW sy tdiv up C  tile stages split
extern "C" long long t2v_gemm_tune_export(char* buf, long long cap) {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  long long need = 0;
  for (const auto& kv : g_tuned) {
    char line[256];
    const TuneKey& k = kv.first;
    int n = snprintf(line, sizeof(line), "%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d\n", k.M, k.N, k.K, k.a_mode, k.n_split,
                     k.out_mode, k.has_res, k.batch, k.KH, k.KW, k.sy, k.tdiv, k.up, k.C, kv.second.tile, kv.second.stages,
                     kv.second.split);
    if (buf && need + n < cap) memcpy(buf + need, line, (size_t)n);
    need += n;
  }
  if (buf && need < cap) buf[need] = 0;
  return need + 1;
}

extern "C" int t2v_gemm_tune_import(const char* text) {
  T2V_CHECK_ARG(text != nullptr, "t2v_gemm_tune_import: null text");
  int count = 0;
  const char* p = text;
  std::lock_guard<std::mutex> lk(g_tune_mu);
  while (*p) {
    TuneKey k;
    memset(&k, 0, sizeof(k));
    DmaCfg c{0, 2, 1};
    int consumed = 0;
    if (sscanf(p, "%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d%n", &k.M, &k.N, &k.K, &k.a_mode, &k.n_split, &k.out_mode,
               &k.has_res, &k.batch, &k.KH, &k.KW, &k.sy, &k.tdiv, &k.up, &k.C, &c.tile, &c.stages, &c.split, &consumed) == 17) {
      const bool w8 = c.tile >= W8_BASE && c.tile < W8_BASE + t2v_gemm_w8_configs() && c.stages >= 0 && c.stages <= 12 && c.split >= 1 &&
                      c.split <= 8;
      if (w8 || (c.tile >= 0 && c.tile <= 9 && (c.stages == 0 || c.stages == 2) && c.split >= 1 && c.split <= 64)) {
        g_tuned[k] = c;
        ++count;
      }
      p += consumed;
    }
    while (*p && *p != '\n') ++p;
    if (*p == '\n') ++p;
  }
  return count;
}

extern "C" int t2v_gemm_w8(const T2VGemm* pp, int cfg, int nstep, int splits, t2v_stream_t stream) {
  T2V_CHECK_ARG(pp != nullptr, "t2v_gemm_w8: null descriptor");
  if (int e = check_gemm(*pp)) return e;
  T2V_CHECK_ARG(w8_ok(*pp), "t2v_gemm_w8: descriptor outside the 8-wave kernels' domain (K%%64, C%%64, bf16 output, no dropout/batch)");
  return t2v_gemm_w8_launch(*pp, cfg, nstep, splits, (hipStream_t)stream);
}

extern "C" int t2v_gemm_lr_ok(const T2VGemm* pp) {
  if (!pp || pp->lr_mode == 0) return 0;
  T2VGemm p = *pp;
  if (check_gemm(p) != T2V_OK || !w8_ok(p) || p.alpha != 1.f) return 0;
  if (p.lr_mode == 3) {
    // dt = (mask dy / (1-p)) U^T computed by the backward-data launch itself: dense A, up to 64 ranks (two fragments), the members
    // of a projection group partition K in whole 64-deep stages, the mask matrix within 2^34 elements (32-bit quad index)
    if (p.lr_rp < 8 || p.lr_rp > 64 || p.lr_rp % 8 != 0 || p.a_mode != T2V_A_DENSE || p.n_split > 0 || p.N < 32 || p.lr_scale != 1.f) return 0;
    if (!p.B2 || !p.D2 || p.ldb2 % 8 != 0 || p.ldd2 % 8 != 0 || p.ldd2 < p.lr_rp || p.lr_taps > 1 || p.b2_klen > 0) return 0;
    if (p.lr_group_cols != 0 && (p.lr_group_cols % 64 != 0 || p.K % p.lr_group_cols != 0 || p.K / p.lr_group_cols > 3)) return 0;
    if ((long long)p.M * (p.lr_group_cols > 0 ? p.lr_group_cols : p.K) >= (1ll << 34)) return 0;
    if ((long long)p.N * p.lr_ldb >= 0x7ff00000ll || (long long)p.lr_rp * p.ldb2 * 2 >= 0x7ff00000ll || (long long)p.M * p.ldd2 >= 0x7ff00000ll) return 0;
    return 1;
  }
  if (p.lr_rp < 8 || p.lr_rp > (p.lr_mode == 1 ? 48 : 32) || p.lr_rp % 8 != 0) return 0;
  if (p.lr_group_cols != 0 && (p.lr_mode != 2 || p.lr_group_cols % 32 != 0 || p.N % p.lr_group_cols != 0 || p.N / p.lr_group_cols > 3)) return 0;
  if (p.lr_mode == 2 && (p.N < 32 || p.n_split > 0)) return 0;
  return 1;
}

extern "C" int t2v_gemm_colsum_rows(const T2VGemm* pp) {
  if (!pp || !w8_ok(*pp) || check_gemm(*pp) != T2V_OK) return 0;
  {
    T2VGemm q = *pp;
    q.colsum = nullptr;
    if (skinny_ok(q)) return 0;                               // (the skinny kernel emits no column statistics)
  }
  if (g_autotune < 0) {
    const char* e = getenv("T2V_GEMM_AUTOTUNE");
    g_autotune = !e ? 1 : (e[0] == '0' ? 0 : ((e[0] == 'l' || e[0] == '2') ? 2 : 1));
  }
  if (g_autotune == 2) return 0;                              // tuning run: the tile is not known before the launch
  return colsum_bm(*pp, pick_cfg(*pp, nullptr));
}

extern "C" int t2v_gemm(const T2VGemm* pp, t2v_stream_t stream) {
  T2V_CHECK_ARG(pp != nullptr, "t2v_gemm: null descriptor");
  T2VGemm p = *pp;
  if (!p.drop_epoch) p.drop_epoch = t2v_drop_epoch;          // the step's dropout epoch (null outside a trainer step)
  if (int e = check_gemm(p)) return e;
  hipStream_t s = (hipStream_t)stream;
  static const bool env_init = [] {
    const char* e = getenv("T2V_GEMM_REGSTAGE");
    g_force_regstage = e && e[0] == '1';
    return true;
  }();
  (void)env_init;
  if (!p.a_trans && !p.b_trans) return dispatch<false, false>(p, s);
  if (p.a_trans && p.b_trans) return dispatch<true, true>(p, s);
  if (!p.a_trans && p.b_trans) return dispatch<false, true>(p, s);
  t2v_set_error("t2v_gemm: a_trans=1 with b_trans=0 is not instantiated");
  return T2V_EINVAL;
}
