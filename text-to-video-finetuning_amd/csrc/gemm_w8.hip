// gemm_w8.hip — the 8-wave, one-workgroup-per-CU member of the bf16 MFMA GEMM family (gfx950).
//
// Same problem statement, operand loaders and epilogue semantics as gemm.hip's gemm_kernel_dma (T2VGemm: dense rows or
// sliding-window gather on A, [N,K] weights, optional second weight/output block = the LoRA rank columns), restricted to what
// lean_ok() admits (K % 64 == 0, C % 64 == 0, bf16 output, no dropout, no split-K workspace, offsets below 2 GiB).
//
// Why a second kernel: at this UNet's problem sizes (M*N <= 32768 x 336 per launch) the 4-wave kernels run 2-3 small
// workgroups per CU and land at ~45 % of the MFMA rate in the K loop (profiles/r02_gemm_kloop_probe.txt) before tile
// quantisation (768 tiles on 512 slots) and column padding (336 -> 384) take their share.  This kernel gives ONE 512-thread
// workgroup the whole CU and sizes the tile so that the launch is one round of ~256 workgroups:
//   * tile BM x BN with BN up to 384 (the whole N = 320 + 16 extent in one tile), waves laid out WM x WN x KG: when the tile is
//     too small for eight 64-row bands, KG = 2 wave groups split every 64-deep K step between them (each keeps a full-size
//     accumulator; the groups are summed through LDS in the epilogue) so that a wave still owns a 64 x 96..192 block and every
//     fragment read from LDS feeds 5-6 MFMAs;
//   * the tile's column origin advances by a run-time step <= BN and fragments that lie wholly beyond the tile's columns are
//     skipped per wave (N = 336 costs 11 column fragments, not 12; their weight rows are never fetched);
//   * SCHED 1: fragment reads are software-pipelined one k16 step ahead in a second register set and the per-stage barrier
//     sits BEFORE the last k16 step of a stage: the wait for the next stage, the barrier, the LDS-DMA issue of the stage after
//     it and the first fragment reads of the next stage all run under the MFMAs of that last step instead of in front of an
//     empty matrix pipe.
// LDS image, swizzle and the buffer-descriptor loader are gemm.hip's (lane-linear DMA image, XOR on source chunk + read).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr int BK = 64;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int BM, int BN, int WM, int WN, int KG, int NSTAGE, int SCHED>
__global__ __launch_bounds__(512) void gemm_w8_kernel(const T2VGemm p, const int nstep, const int ntn, const int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(WM * WN * KG == 8 && (KG == 1 || KG == 2), "eight waves: WM x WN x KG");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  constexpr int KS = 4 / KG;                       // k16 steps of a 64-deep stage that one wave computes
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int NT = 512, RPP = NT / 8;            // 64 tile rows per 16-byte-chunk pass
  constexpr int NCA = BM / RPP, NCB = BN / RPP, LPT = NCA + NCB;
  static_assert(BM % RPP == 0 && BN % RPP == 0 && TM % 32 == 0 && TN % 32 == 0, "tile/wave mismatch");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // waves w and w+4 sit on the same SIMD: they differ in the column block, so a light and a heavy column share a matrix pipe
  const int wr = wave % WM, kg = (wave / WM) % KG, wc = wave / (WM * KG);
  const int M = p.M, N = p.N;
  const T2VConvGeom g = p.geom;
  const int ntiles = gridDim.x;
  int t;
  {
    int q = ntiles >> 3, r = ntiles & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = t / ntn, tn = t - tm * ntn;
  const long long m0 = (long long)tm * BM;
  const int n0 = tn * nstep;
  const int ncols = min(N - n0, tn == ntn - 1 ? BN : nstep);        // columns this tile owns (multiple of 8)
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;

  // ---- loader state (gemm.hip's lean loader at 512 threads: 64 rows per pass)
  const int kc = (tid & 7) ^ ((tid >> 4) & 7);
  constexpr unsigned OOB = 0x80000000u;
  const bool is_conv = p.a_mode == T2V_A_CONV;
  const int Hr = g.Hv >> g.up, Wr = g.Wv >> g.up;
  int rn[NCA], rvy[NCA], rvx[NCA];
  bool aok[NCA];
  unsigned va[NCA], vb[NCB];
#pragma unroll
  for (int i = 0; i < NCA; ++i) {
    const long long m = m0 + (tid >> 3) + RPP * i;
    aok[i] = m < M;
    rn[i] = rvy[i] = rvx[i] = 0;
    if (is_conv) {
      const int hw = g.Ho * g.Wo;
      const int mm = aok[i] ? (int)m : 0;
      const int n = mm / hw, r = mm - n * hw;
      const int oy = r / g.Wo, ox = r - oy * g.Wo;
      rn[i] = n * Hr;
      rvy[i] = oy * g.sy - g.py;
      rvx[i] = ox * g.sx - g.px;
    }
  }
  int sc = 0, sky = 0, skx = 0;                    // scalar window position of the NEXT K step: channel, tap row / col
  auto conv_rows = [&]() {
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
  __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x80000000u, 0x00020000);
  __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x80000000u, 0x00020000);
  __amdgpu_buffer_rsrc_t srdB2 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.n_split > 0 ? p.B2 : p.B), 0, 0x80000000u, 0x00020000);
  if (is_conv) {
    conv_rows();
  } else {
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const long long m = m0 + (tid >> 3) + RPP * i;
      va[i] = aok[i] ? ((unsigned)m * (unsigned)p.lda + (unsigned)kc * 8u) * 2u : OOB;
    }
  }
  unsigned b2lane = 0;                             // pass i of this WAVE reads rows of the second weight block
  const int wlo = p.b2_klen > 0 ? p.b2_k0 : 0, whi = p.b2_klen > 0 ? p.b2_k0 + p.b2_klen : 0x7fffffff;
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    const int nl = (tid >> 3) + RPP * i;           // row of the tile's weight block
    const int n = n0 + nl;
    // a wave's 8 rows of one pass start at a multiple of 8 and n_split % 8 == 0: the side is wave-uniform
    const bool second = p.n_split > 0 && (n0 + (wave << 3) + RPP * i) >= p.n_split;
    if (second) b2lane |= 1u << i;
    const unsigned row = (unsigned)(second ? n - p.n_split : n);
    vb[i] = (nl < ncols) ? (row * (unsigned)(second ? p.ldb2 : p.ldb) + (unsigned)kc * 8u) * 2u : OOB;   // rows past the tile: zeros, no fetch
  }

  auto issue = [&](int k0, int stage) {
    if ((dbg & 1) && k0 >= NSTAGE * BK) return;      // ablation (T2V_W8_DBG=1): steady state without operand traffic
    unsigned char* sA = smem + stage * STAGE;
    unsigned char* sB = sA + A_BYTES;
    const int soa = (is_conv ? sc : k0) * 2;
#pragma unroll
    for (int i = 0; i < NCA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (__attribute__((address_space(3))) void*)(sA + (tid + NT * i) * 16), 16,
                                               (int)va[i], soa, 0, 0);
    const bool win = k0 >= wlo && k0 < whi;
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const bool second = (b2lane >> i) & 1u;
      const unsigned vo = (second && !win) ? OOB : vb[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? srdB2 : srdB,
                                               (__attribute__((address_space(3))) void*)(sB + (tid + NT * i) * 16), 16, (int)vo,
                                               (second ? k0 - wlo : k0) * 2, 0, 0);
    }
    if (is_conv) {
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
  };

  // the same stage issued piece by piece (SCHED 2 spreads a stage's LDS-DMA instructions over several phases): pieces
  // [lo, hi) of the LPT per-thread loads, A passes first; advance_window() after the stage's last A piece
  auto issue_pieces = [&](int k0, int stage, int lo, int hi) {
    if (dbg & 1) return;
    unsigned char* sA = smem + stage * STAGE;
    unsigned char* sB = sA + A_BYTES;
    const int soa = (is_conv ? sc : k0) * 2;
    const bool win = k0 >= wlo && k0 < whi;
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      if (i < lo || i >= hi) continue;
      if (i < NCA) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (__attribute__((address_space(3))) void*)(sA + (tid + NT * i) * 16), 16,
                                                 (int)va[i < NCA ? i : 0], soa, 0, 0);
      } else {
        const int ib = i - NCA;
        const bool second = (b2lane >> ib) & 1u;
        const unsigned vo = (second && !win) ? OOB : vb[ib >= 0 ? ib : 0];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? srdB2 : srdB,
                                                 (__attribute__((address_space(3))) void*)(sB + (tid + NT * ib) * 16), 16, (int)vo,
                                                 (second ? k0 - wlo : k0) * 2, 0, 0);
      }
    }
  };
  auto advance_window = [&]() {
    if (is_conv) {
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
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addressing: row r of a tile sits at r*128 bytes, its 16-byte K chunk q at slot q ^ ((r>>1)&7); TM, TN are
  // multiples of 32, so the swizzle term depends on the lane alone
  const unsigned qsw = (unsigned)(((lane >> 5) ^ ((lane >> 1) & 7)) << 4);
  const unsigned arow = (unsigned)(wr * TM + (lane & 31)) * 128u;
  const unsigned brow = (unsigned)(wc * TN + (lane & 31)) * 128u;
  // column fragments of this wave that hold columns of the tile
  const int nfw = max(0, min(FN, (ncols - wc * TN + 31) >> 5));
  const int nt = p.K / BK;

  auto kloop = [&](auto nf_tag) {
    constexpr int NF = decltype(nf_tag)::value;
    auto load_frags = [&](bf16x8(&af)[FM], bf16x8(&bfr)[FN], int stage, int kk) {
      const unsigned char* sA = smem + stage * STAGE;
      const unsigned char* sB = sA + A_BYTES;
      const unsigned ko = qsw ^ ((unsigned)kk << 5);
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8*)(sA + arow + i * 4096 + ko);
#pragma unroll
      for (int j = 0; j < NF; ++j) bfr[j] = *(const bf16x8*)(sB + brow + j * 4096 + ko);
    };
    auto mfma = [&](const bf16x8(&af)[FM], const bf16x8(&bfr)[FN]) {
      if (dbg & 2) {                                 // ablation (T2V_W8_DBG=2): fragments stay live, no matrix work
#pragma unroll
        for (int i = 0; i < FM; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
        for (int j = 0; j < NF; ++j) asm volatile("" ::"v"(bfr[j]));
        return;
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    };
    if constexpr (SCHED == 0) {
      // classic ring: wait for stage `it`, barrier, refill the stage freed by the barrier, compute
#pragma unroll
      for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nt) issue(s * BK, s);
      int stage = 0;
      for (int it = 0; it < nt; ++it) {
        const int ahead = min(nt, it + NSTAGE - 1) - (it + 1);
        if (ahead >= 2) wait_vmcnt<2 * LPT>();
        else if (ahead == 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (it + NSTAGE - 1 < nt) {
          int ws = stage + NSTAGE - 1;
          if (ws >= NSTAGE) ws -= NSTAGE;
          issue((it + NSTAGE - 1) * BK, ws);
        }
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          bf16x8 af[FM], bfr[FN];
          load_frags(af, bfr, stage, kg * KS + j);
          mfma(af, bfr);
        }
        if (++stage == NSTAGE) stage = 0;
      }
    } else if constexpr (SCHED == 2) {
      // ping-pong: the two waves of a SIMD (w, w+4) run the same phase sequence one phase apart — while one group multiplies
      // (C phase: nothing but MFMAs, raised priority) the other reads its fragments and issues its share of the LDS-DMA
      // (M phase).  With every wave in step the CU alternates between an LDS/DMA burst with idle matrix pipes and an MFMA burst
      // with an idle LDS (measured: skeleton, DMA and MFMA time add up, profiles/r03_w8_ablation.txt); here each phase pairs
      // memory work with matrix work.  One fragment set per wave (read in M, consumed in C).
      //   phase k of group 0 runs beside phase k-1 of group 1 (group 1 passes one extra barrier first, group 0 one at the end)
      //   stage s+1 must have landed before group 0 opens M(s+1, 0): every wave drains its own pieces of it in the phase that
      //   ends at that barrier — C(s, KS-1) for group 0, M(s, KS-1) for group 1
      //   the slot of stage s-1 is refilled during the M phases of stage s (all reads of s-1 ended before M(s, 0) of group 0)
      constexpr int NPH = NSTAGE == 2 ? (KS > 2 ? KS - 2 : 1) : KS;     // M phases of a stage that carry DMA pieces
      constexpr int PP = (LPT + NPH - 1) / NPH;
      const int grp = wave >> 2;
#pragma unroll
      for (int s = 0; s < NSTAGE; ++s)
        if (s < nt) issue(s * BK, s);
      {
        const int ahead = min(nt, NSTAGE) - 1;
        if (ahead >= 2) wait_vmcnt<2 * LPT>();
        else if (ahead == 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (grp == 1) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 af[FM], bfr[FN];
      unsigned long long tacc[5] = {0, 0, 0, 0, 0};
      int stage = 0;
      for (int it = 0; it < nt; ++it) {
        const int rs = it + NSTAGE - 1;                      // stage whose pieces go out during this one
        const bool refill = it >= 1 && rs < nt;
        int rslot = stage - 1;
        if (rslot < 0) rslot += NSTAGE;
        const int ahead = min(nt - 1, it + NSTAGE - 1) - (it + 1);      // stages that may stay in flight behind stage it+1
        auto stage_wait = [&]() {
          if (it + 1 < nt) {
            if (ahead >= 1) wait_vmcnt<LPT>();
            else wait_vmcnt<0>();
          }
        };
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          // ---- M phase
          unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
          if (dbg & 4) t0 = __builtin_readcyclecounter();
          load_frags(af, bfr, stage, kg * KS + j);
          if (j < NPH && refill) {
            issue_pieces(rs * BK, rslot, j * PP, (j + 1) * PP < LPT ? (j + 1) * PP : LPT);
            if (j == NPH - 1) advance_window();
          }
          wait_lgkm0();
          if (j == KS - 1 && grp == 1) stage_wait();
          if (dbg & 4) t1 = __builtin_readcyclecounter();
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          if (dbg & 4) t2 = __builtin_readcyclecounter();
          // ---- C phase
          __builtin_amdgcn_s_setprio(1);
          mfma(af, bfr);
          __builtin_amdgcn_s_setprio(0);
          if (j == KS - 1 && grp == 0) stage_wait();
          if (dbg & 4) t3 = __builtin_readcyclecounter();
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          if (dbg & 4) {
            t4 = __builtin_readcyclecounter();
            tacc[0] += t1 - t0; tacc[1] += t2 - t1; tacc[2] += t3 - t2; tacc[3] += t4 - t3;
            if (j < NPH && refill) tacc[4] += t1 - t0;           // M phases that carried DMA pieces
          }
        }
        if (++stage == NSTAGE) stage = 0;
      }
      if (grp == 0) __builtin_amdgcn_s_barrier();
      if ((dbg & 4) && lane == 0 && p.workspace) {               // phase-cycle probe (T2V_W8_DBG=4): per wave totals
        unsigned long long* o = (unsigned long long*)p.workspace + ((long long)blockIdx.x * 8 + wave) * 8;
        for (int q = 0; q < 5; ++q) o[q] = tacc[q];
        o[5] = (unsigned long long)nt * KS;
      }
    } else {
      // pipelined: all NSTAGE slots are filled up front; inside a stage the fragments of k16 step j+1 are read while step j
      // multiplies; the stage hand-over (wait, barrier, refill, first fragments of the next stage) sits under the last step
      static_assert(KS % 2 == 0, "two fragment sets alternate per k16 step");
#pragma unroll
      for (int s = 0; s < NSTAGE; ++s)
        if (s < nt) issue(s * BK, s);
      {
        const int ahead = min(nt, NSTAGE) - 1;
        if (ahead >= 2) wait_vmcnt<2 * LPT>();
        else if (ahead == 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      bf16x8 af0[FM], bf0[FN], af1[FM], bf1[FN];
      load_frags(af0, bf0, 0, kg * KS);
      int stage = 0;
      for (int it = 0; it < nt; ++it) {
#pragma unroll
        for (int j = 0; j + 2 < KS; j += 2) {               // (KS = 4: steps 0 and 1; KS = 2: none)
          load_frags(af1, bf1, stage, kg * KS + j + 1);
          mfma(af0, bf0);
          load_frags(af0, bf0, stage, kg * KS + j + 2);
          mfma(af1, bf1);
        }
        load_frags(af1, bf1, stage, kg * KS + KS - 1);      // last step's fragments: the last reads of this stage
        mfma(af0, bf0);
        int nstage = stage + 1;
        if (nstage == NSTAGE) nstage = 0;
        if (it + 1 < nt) {
          wait_lgkm0();                                     // this wave is done reading `stage`
          const int ahead = min(nt, it + NSTAGE) - (it + 2);   // stages allowed to stay in flight behind stage it+1
          if (ahead >= 2) wait_vmcnt<2 * LPT>();
          else if (ahead == 1) wait_vmcnt<LPT>();
          else wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();                     // stage it+1 landed for every wave; `stage` is free
          if (it + NSTAGE < nt) issue((it + NSTAGE) * BK, stage);
          load_frags(af0, bf0, nstage, kg * KS);
        }
        mfma(af1, bf1);
        stage = nstage;
      }
    }
  };
  if (nfw >= FN) kloop(std::integral_constant<int, FN>{});
  else if (FN >= 2 && nfw == FN - 1) kloop(std::integral_constant<int, (FN >= 2 ? FN - 1 : FN)>{});
  else kloop(std::integral_constant<int, (FN >= 3 ? FN - 2 : (FN >= 2 ? FN - 1 : FN))>{});

  wait_vmcnt<0>();
  __syncthreads();                                 // ring idle: reuse it as the epilogue staging buffer

  // ---- epilogue: K groups summed through LDS, then accumulators -> LDS (fp32) -> 16-byte coalesced rows
  constexpr int CPR = BN / 8;
  constexpr int WRP = (WM * 32 * BN * 4 > 128 * 1024) ? 2 : 1;      // wave-row groups staged separately (LDS budget)
  constexpr int WPP = WM / WRP;                                     // wave rows per staging pass
  static_assert(WM % WRP == 0, "staging split");
  constexpr int ITERS = (WPP * 32 * CPR + NT - 1) / NT;
  float* sC = (float*)smem;                        // (WPP*32) x BN fp32, one 32-row fragment band per wave row and pass
  const float* bias = (const float*)p.bias;
  const bf16_t* rowbias = (const bf16_t*)p.rowbias;
  const bf16_t* R = (const bf16_t*)p.R;
#pragma unroll
  for (int ps = 0; ps < FM * WRP; ++ps) {
    const int i = ps / WRP, h = ps % WRP;
    if (ps > 0) __syncthreads();
    const bool mine = (wr / WPP) == h;
    const int wrl = wr % WPP;
    if (mine && (KG == 1 || kg == 1)) {
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = wrl * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int cl = wc * TN + j * 32 + (lane & 31);
          sC[rl * BN + cl] = acc[i][j][r];
        }
    }
    __syncthreads();
    if constexpr (KG == 2) {
      if (mine && kg == 0) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = wrl * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int cl = wc * TN + j * 32 + (lane & 31);
            sC[rl * BN + cl] += acc[i][j][r];
          }
      }
      __syncthreads();
    }
#pragma unroll 2
    for (int it = 0; it < ITERS; ++it) {
      const int c = tid + NT * it;
      if (c >= WPP * 32 * CPR) break;
      const int rl = c / CPR, cc = c - rl * CPR;
      const unsigned row = (unsigned)m0 + (h * WPP + (rl >> 5)) * TM + i * 32 + (rl & 31);
      if (row >= (unsigned)M || cc * 8 >= ncols) continue;
      const int col = n0 + cc * 8;
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
      if (p.n_split > 0 && col >= p.n_split) {     // rank columns: second output block, alpha only
        *(bf16x8*)((bf16_t*)p.D2 + row * (unsigned)p.ldd2 + (col - p.n_split)) = pack8bf(v);
        continue;
      }
      if (bias) {
        const float4 b0 = *(const float4*)(bias + col), b1 = *(const float4*)(bias + col + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (rowbias) {
        const bf16x8 tb = *(const bf16x8*)(rowbias + (row / (unsigned)p.rows_per_rb) * (unsigned)p.ldrb + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf2f((unsigned short)tb[e]);
      }
      if (p.act == T2V_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
      }
      if (R) {
        const bf16x8 tr = *(const bf16x8*)(R + row * (unsigned)p.ldr + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.beta * bf2f((unsigned short)tr[e]);
      }
      *(bf16x8*)((bf16_t*)p.D + row * (unsigned)p.ldd + col) = pack8bf(v);
    }
  }
}

template <int BM, int BN, int WM, int WN, int KG, int NSTAGE, int SCHED>
int launch_w8(const T2VGemm& p, int nstep, hipStream_t s) {
  constexpr int RING = NSTAGE * (BM + BN) * BK * 2;
  constexpr int EPI = (WM * 32 * BN * 4 > 128 * 1024 ? WM / 2 : WM) * 32 * BN * 4;
  constexpr int SMEM = RING > EPI ? RING : EPI;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  auto kern = gemm_w8_kernel<BM, BN, WM, WN, KG, NSTAGE, SCHED>;
  static bool attr_set = false;
  if (!attr_set) {
    if (SMEM > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  if (nstep <= 0 || nstep > BN) nstep = BN;
  nstep &= ~31;                                    // whole 32-column fragments per step
  if (nstep <= 0) nstep = BN;
  const int ntm = (p.M + BM - 1) / BM;
  int ntn = 1;
  while ((long long)(ntn - 1) * nstep + BN < p.N) ++ntn;      // the last tile takes up to BN columns
  static const int dbg = [] { const char* e = getenv("T2V_W8_DBG"); return e ? atoi(e) : 0; }();
  T2V_LAUNCH(kern, dim3(ntm * ntn), dim3(512), SMEM, s, p, nstep, ntn, dbg);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

}  // namespace

// Number of W8 configurations and a pinned-configuration launch (tile table / tuning / diagnostics).  The caller (gemm.hip)
// has already validated the descriptor and checked that the lean loader and the bf16 epilogue apply.
int t2v_gemm_w8_configs(void) { return 14; }
int t2v_gemm_w8_launch(const T2VGemm& p, int cfg, int nstep, hipStream_t s) {
  switch (cfg) {
    case 0: return launch_w8<128, 384, 2, 2, 2, 2, 0>(p, nstep, s);      // wave 64x192, K groups, classic ring
    case 1: return launch_w8<128, 384, 4, 2, 1, 2, 1>(p, nstep, s);      // wave 32x192, pipelined
    case 2: return launch_w8<256, 256, 4, 2, 1, 2, 1>(p, nstep, s);      // wave 64x128, pipelined
    case 3: return launch_w8<128, 192, 2, 2, 2, 3, 1>(p, nstep, s);      // wave 64x96, K groups, pipelined, 3 stages
    case 4: return launch_w8<128, 256, 2, 2, 2, 3, 1>(p, nstep, s);      // wave 64x128, K groups, pipelined, 3 stages
    case 5: return launch_w8<256, 256, 4, 2, 1, 2, 0>(p, nstep, s);      // A/B of case 2 with the classic ring
    case 6: return launch_w8<128, 192, 2, 2, 2, 3, 0>(p, nstep, s);      // A/B of case 3 with the classic ring
    case 7: return launch_w8<128, 384, 4, 2, 1, 2, 0>(p, nstep, s);      // A/B of case 1 with the classic ring
    // ping-pong schedule (SCHED 2)
    case 8: return launch_w8<256, 256, 4, 2, 1, 2, 2>(p, nstep, s);      // wave 64x128
    case 9: return launch_w8<128, 384, 2, 2, 2, 2, 2>(p, nstep, s);      // wave 64x192, K groups
    case 10: return launch_w8<128, 256, 2, 2, 2, 3, 2>(p, nstep, s);     // wave 64x128, K groups, 3 stages
    case 11: return launch_w8<128, 192, 2, 2, 2, 3, 2>(p, nstep, s);     // wave 64x96, K groups, 3 stages
    case 12: return launch_w8<256, 384, 4, 2, 1, 2, 2>(p, nstep, s);     // wave 64x192 (one round at M = 32768 needs split N..)
    case 13: return launch_w8<256, 128, 4, 2, 1, 3, 2>(p, nstep, s);     // wave 64x64, 3 stages
    default: t2v_set_error("t2v_gemm_w8: unknown configuration %d", cfg); return T2V_EINVAL;
  }
}
