// gemm_w8.hip — the 8-wave, one-workgroup-per-CU member of the bf16 MFMA GEMM family (gfx950).
//
// Same problem statement, operand loaders and epilogue semantics as gemm.hip's gemm_kernel_dma (T2VGemm: dense rows or
// sliding-window gather on A, [N,K] weights, optional second weight/output block = the LoRA rank columns), restricted to what
// lean_ok() admits (K % 64 == 0, C % 64 == 0, bf16 output, no dropout, offsets below 2 GiB).
//
// Why a second kernel: at this UNet's problem sizes (M*N <= 32768 x 336 per launch) the 4-wave kernels run 2-3 small
// workgroups per CU and land at ~45 % of the MFMA rate in the K loop (profiles/r02_gemm_kloop_probe.txt) before tile
// quantisation (768 tiles on 512 slots) and column padding (336 -> 384) take their share.  This kernel gives ONE 512-thread
// workgroup the whole CU and sizes the tile so that the launch is one round of ~256 workgroups:
//   * tile BM x BN with BN up to 384 (the whole N = 320 + 16 extent in one tile), waves laid out WM x WN x KG: when the tile is
//     too small for eight 64-row bands, KG = 2 wave groups split every K step between them (each keeps a full-size
//     accumulator; the groups are summed through LDS in the epilogue) so that a wave still owns a 64 x 96..192 block and every
//     fragment read from LDS feeds 5-6 MFMAs;
//   * the tile's column origin advances by a run-time step <= BN and fragments that lie wholly beyond the tile's columns are
//     skipped per wave (N = 336 costs 11 column fragments, not 12; their weight rows are never fetched);
//   * schedules (SCHED): 0 = the classic ring of gemm.hip (wait, barrier, refill, compute);
//     2 = ping-pong: the two waves of every SIMD run the same phase sequence ONE PHASE APART — while one multiplies (C phase:
//         MFMAs only, raised priority) the other reads fragments and issues LDS-DMA (M phase);
//     3 = ping-pong over a ring of 32-deep stages with the fragment reads prefetched: the reads of the next k16 step go out at
//         the head of a C phase (second register set) and land under its MFMAs, a stage's LDS-DMA pieces are spread evenly
//         over its M phases (two 1-KiB pieces per wave and phase at 256x256), and four stages give every piece two stages of
//         landing time.
//     4 = classic ring with the refill's LDS-DMA pieces spread over the k16 steps of the stage; 5 = 4 + fragments of the next
//         k16 step read before the current one multiplies + the stage hand-over placed before the last k16 step (the two
//         schedules the tuner may pick; 2 and 3 are kept for the record: they did not pay).
//     Measured motivation (profiles/r03_w8_ablation.txt, r03_w8_phase_probe.txt, r03_dma_bw_probe.txt): with all eight waves
//     in step, LDS-DMA issue (a stage's bytes / 64 B per clk through the CU's address path), fragment-read latency and MFMA
//     time ADD UP per K step; the operand path itself sustains 30-38 TB/s from the L2s, so it is the overlap, not the path.
//   * accumulators are kept TRANSPOSED (weight fragment = first MFMA operand): a lane owns one tile row and four consecutive
//     columns per register quad, which makes the epilogue of launches without column statistics a register affair
//     (v_permlane32_swap -> 8-column chunks -> 16-byte stores; template parameter CS = false) — see the epilogue;
//   * in-launch split-K (ticket + fp32 slabs in the caller's workspace, reduced in split order by the last arriver);
//   * fixed per-launch cost matters as much as the K loop at this UNet's launch sizes (a 20 us + flops/1 PFLOP/s fit over
//     the step's 956 launches): kernel arguments are fetched in one batch, the ring prologue goes out before the rest of
//     the set-up (profiles/r03_w8_timeline.txt).
// LDS image, swizzle and the buffer-descriptor loader are gemm.hip's (lane-linear DMA image, XOR on source chunk + read).
#include <stdlib.h>
#include <type_traits>
#include "colsum.h"
#ifndef W8_PART
#define W8_PART 0
#endif

namespace {

// 64 zero bytes: the target of epilogue operand loads whose operand is absent (the loads themselves are unconditional)
__device__ __attribute__((aligned(64))) unsigned g_w8_zero[16];   // (not const: a constant-address-space object makes the selected pointer flat)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// counted wait with a run-time count drawn from a small set (the immediate must be a literal)
template <int A, int B>
__device__ __forceinline__ void wait_vmcnt_sel(bool first) {
  if (first) wait_vmcnt<A>();
  else wait_vmcnt<B>();
}

// Epilogue staging plan: the fp32 tile goes through LDS in as few passes as the 160 KB hold — every pass costs two or three
// workgroup barriers and a dependent LDS write -> read -> global store chain (~2800 cycles even on an idle memory system,
// profiles/r03_w8_timeline.txt).  A pass stages FG of the FM fragment bands of WPP = WM / WRP wave rows.
template <int BN, int WM, int FM>
struct EpiPlan {
  static constexpr int LDS_MAX = 160 * 1024 - 64;
  static constexpr int SB = BN + 4;                                         // staged row pitch (floats): 16-byte stores of 8 lanes
                                                                            // on 8 rows land on distinct banks
  static constexpr int BAND = 32 * SB * 4;                                  // one 32-row fragment band, fp32
  static constexpr int WRP = (WM * BAND <= LDS_MAX) ? 1 : 2;                // wave-row groups staged separately
  static constexpr int WPP = WM / WRP;
  static constexpr int FG = (FM * WPP * BAND <= LDS_MAX) ? FM : ((FM % 2 == 0 && (FM / 2) * WPP * BAND <= LDS_MAX) ? FM / 2 : 1);
  static constexpr int PROWS = FG * WPP * 32;                               // rows per staging pass
  static constexpr int BYTES = PROWS * SB * 4;
  static constexpr int NPASS = (FM / FG) * WRP;
  static_assert(WM % WRP == 0 && FM % FG == 0 && BYTES <= LDS_MAX, "staging plan");
};

// mode 3 (LR = 2): zero the dropped elements of an A fragment chunk — eight consecutive K elements of one row = two quads of the
// dropout protocol (common.h: quad q of the mask matrix [rows, W], index = row * W + k; q < 2^32, checked by launch_w8).  The
// keep test `field >= thr` of the four 16-bit fields runs two at a time on packed 16-bit lanes: (x -sat (thr - 1)) >= 1.
typedef unsigned short t2v_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned keep_pair(unsigned x, unsigned thrm1_pk) {
  t2v_u16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(t2v_u16x2, x), __builtin_bit_cast(t2v_u16x2, thrm1_pk));
  const t2v_u16x2 one = {1, 1}, all = {0xffff, 0xffff};
  d = __builtin_elementwise_min(d, one) * all;
  return __builtin_bit_cast(unsigned, d);
}
__device__ __forceinline__ bf16x8 mask_chunk(const bf16x8& a, unsigned q, unsigned s0, unsigned s1, unsigned thrm1_pk) {
  const DropKey k{s0, s1, 0u};
  const DropQuad h0 = drop_quad(k, (unsigned long long)q), h1 = drop_quad(k, (unsigned long long)(q + 1u));
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 w = __builtin_bit_cast(u32x4, a);
  w[0] &= keep_pair(h0.a, thrm1_pk);
  w[1] &= keep_pair(h0.b, thrm1_pk);
  w[2] &= keep_pair(h1.a, thrm1_pk);
  w[3] &= keep_pair(h1.b, thrm1_pk);
  return __builtin_bit_cast(bf16x8, w);
}

// the same from eight keep bits (bit e = element e of the chunk), read from the forward launch's keep-bit plane (T2VGemm.lr_plane)
__device__ __forceinline__ bf16x8 mask_chunk_bits(const bf16x8& a, unsigned bits) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 w = __builtin_bit_cast(u32x4, a);
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const unsigned lo = (unsigned)(((int)(bits << (31 - 2 * d))) >> 31), hi = (unsigned)(((int)(bits << (30 - 2 * d))) >> 31);
    w[d] &= (lo & 0xffffu) | (hi & 0xffff0000u);
  }
  return __builtin_bit_cast(bf16x8, w);
}

// CS: the instantiation with the LDS-staged epilogue (column statistics for the GroupNorm fusion need it); the other one
// stores from registers.  Two kernels instead of a run-time branch: with both epilogues in one body the register allocator
// spilled in each of them.
// LR: the instantiation that knows the rank-wide epilogue term (T2VGemm.lr_*: the LoRA branch of a wrapped layer whose dropout
// is active, folded into the launch) — separate kernels so that the plain ones keep their register allocation.  LR = 1: modes
// 1 and 2; LR = 2: mode 3 (round 6: the backward-data launch of a dropped LINEAR wrapper computes dt = (mask dy / (1-p)) U^T
// itself — rank fragments at the END of the tile's columns whose MFMAs take a MASKED copy of the A fragments, the mask hashed in
// the K loop by the one wave column that owns them — and adds s dt D^T in its epilogue: no t2v_lora_drop_dt launch).
template <int BM, int BN, int WM, int WN, int KG, int NSTAGE, int SCHED, int BKT, bool CS, int LR = 0>
__global__ __launch_bounds__(512) void gemm_w8_kernel(const T2VGemm p, const int nstep, const int ntn, const int splits, const int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(WM * WN * KG == 8 && (KG == 1 || KG == 2), "eight waves: WM x WN x KG");
  static_assert(BKT == 64 || BKT == 32, "stage depth");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  constexpr int KS = BKT / 16 / KG;                // k16 steps of a stage that one wave computes
  static_assert(KS >= 1, "stage too shallow for the K groups");
  constexpr int ROWB = BKT * 2;                    // bytes of one tile row in LDS
  constexpr int CPW = BKT / 8;                     // 16-byte chunks per tile row
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int NT = 512, RPP = NT / CPW;          // tile rows per 16-byte-chunk pass (64 / 128)
  constexpr int NCA = BM / RPP, NCB = BN / RPP, LPT = NCA + NCB;
  constexpr int WROWS = 64 / CPW;                  // rows one wave covers in a pass
  static_assert(BM % RPP == 0 && BN % RPP == 0 && TM % 32 == 0 && TN % 32 == 0, "tile/wave mismatch");
  constexpr int RING_BYTES = NSTAGE * STAGE;
  using Epi = EpiPlan<BN, WM, FM>;
  constexpr int EPI_BYTES = Epi::BYTES;
  constexpr int SMEM_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;      // + 16 bytes for the split-K ticket word
  // mode 3 with a keep-bit plane: every rank wave keeps, per ring stage, the plane words of its own TM rows x the stage's two
  // 32-column fragments (TM * 8 bytes) in a slot of its own behind the ring — filled by the wave's OWN LDS-DMA next to the stage's
  // operand pieces and read back only by that wave (its covering vmcnt is the only ordering needed)
  constexpr int NMK = TM / 32;                                      // mask DMA instructions per stage and rank wave (256 B each)
  constexpr int MASK_OFF = SMEM_BYTES + 16, MASK_SLOT = TM * 8;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  warm_kernargs<(int)sizeof(T2VGemm) + 16>();        // (common.h)
  unsigned long long tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // timeline probe (T2V_W8_DBG=8)
  if (dbg & 8) tl[0] = __builtin_readcyclecounter();
  // waves w and w+4 sit on the same SIMD: they differ in the column block, so a light and a heavy column share a matrix pipe
  const int wr = wave % WM, kg = (wave / WM) % KG, wc = wave / (WM * KG);
  const int M = p.M, N = p.N;
  const T2VConvGeom g = p.geom;
  // grid = output tiles x K splits (split index slowest: the splits of a tile sit ntiles apart, on the same XCD when
  // ntiles % 8 == 0); split z owns the K steps [z*nt_all/S, (z+1)*nt_all/S)
  // (integer divisions cost ~40 instructions each on this machine: the common single-split / single-column-block launches
  // skip them)
  const int ntiles = splits == 1 ? (int)gridDim.x : (int)gridDim.x / splits;
  const int bz = splits == 1 ? 0 : (int)blockIdx.x / ntiles, bt = blockIdx.x - bz * ntiles;
  int t;
  {
    int q = ntiles >> 3, r = ntiles & 7, xcd = bt & 7, idx = bt >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int nt_all = p.K / BKT;
  const int ks0 = splits == 1 ? 0 : (int)((long long)bz * nt_all / splits);
  const int ks1 = splits == 1 ? nt_all : (int)((long long)(bz + 1) * nt_all / splits);
  const int nt = ks1 - ks0, kbeg = ks0 * BKT;
  // raster_n & 1: an XCD's contiguous run of tiles covers a few N-tiles x ALL M-tiles (each weight column block is pulled
  // into ONE L2; the activations are streamed by every XCD) instead of a few M-tiles x all N-tiles.  What misses the L2s is
  // served at ~6 TB/s against 30+ TB/s for hits (profiles/r03_dma_bw_probe.txt), so the launcher picks the order that
  // re-streams the SMALLER operand eight times.
  int tm, tn;
  if (ntn == 1) {
    tm = t;
    tn = 0;
  } else if (p.raster_n & 1) {
    const int ntm = ntiles / ntn;
    tn = t / ntm;
    tm = t - tn * ntm;
  } else {
    tm = t / ntn;
    tn = t - tm * ntn;
  }
  const long long m0 = (long long)tm * BM;
  const int n0 = tn * nstep;
  // lr_mode 2: the tile = nbase base columns, padded to whole fragments, + one fragment of rank columns (the lr_rp rows of B2:
  // t = x (*) D^T is computed by EVERY column tile for its own epilogue)
  const bool lr2 = LR == 1 && p.lr_mode == 2;
  constexpr bool lr3 = LR == 2;
  // mode 3: nrf fragments of rank columns (the lr_rp rows of B2 = U, rank-major) occupy the LAST 32 nrf columns of the tile,
  // i.e. the last fragments of the last wave column — a static position, so that the K loop knows at compile time which MFMAs
  // take the masked A fragments
  const int nrf = lr3 ? (p.lr_rp + 31) >> 5 : 0;
  const int rk3 = BN - 32 * nrf;
  const bool rankwave = lr3 && (wave / (WM * KG)) == WN - 1;
  const int nbase = (lr2 || lr3) ? min(N - n0, nstep) : min(N - n0, tn == ntn - 1 ? BN : nstep);
  const int rk0 = lr2 ? ((nbase + 31) & ~31) : 0x40000000;           // tile-local column of the rank fragment
  // projection group (lr_group_cols > 0): the tile belongs to member n0 / lr_group_cols (launch_w8: the column step divides it)
  const int gcols = (lr2 && p.lr_group_cols > 0) ? p.lr_group_cols : 0;
  const int member = gcols ? n0 / gcols : 0;
  const int ncols = lr2 ? rk0 + 32 : ((lr3 && rankwave) ? BN : nbase);     // columns this tile owns (multiple of 8)
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;

  // ---- loader state (gemm.hip's lean loader at 512 threads).  Chunk c = tid + 512*i of a tile sits at LDS byte c*16: row
  // c / CPW, slot c % CPW, and holds source chunk slot ^ swz(row); swz(row) = (row>>1)&7 (64-deep rows) / (row>>2)&3 (32-deep
  // rows) = (tid>>4) & (CPW-1) for every pass (the passes' rows differ by multiples of RPP)
  const int kc = (tid & (CPW - 1)) ^ ((tid >> 4) & (CPW - 1));
  constexpr unsigned OOB = 0x80000000u;
  const bool is_conv = p.a_mode == T2V_A_CONV;
  const int Hr = g.Hv >> g.up, Wr = g.Wv >> g.up;
  int rn[NCA], rvy[NCA], rvx[NCA];
  bool aok[NCA];
  unsigned va[NCA], vb[NCB];
#pragma unroll
  for (int i = 0; i < NCA; ++i) {
    const long long m = m0 + tid / CPW + RPP * i;
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
  int sc = 0, sky = 0, skx = 0;                    // scalar window position of the NEXT stage: channel, tap row / col
  if (is_conv && kbeg > 0) {
    const int tap0 = kbeg / g.C;
    sc = kbeg - tap0 * g.C;
    sky = tap0 / g.KW;
    skx = tap0 - sky * g.KW;
  }
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
      __builtin_amdgcn_make_buffer_rsrc((void*)((p.n_split > 0 || lr2 || lr3) ? p.B2 : p.B), 0, 0x80000000u, 0x00020000);
  if (is_conv) {
    conv_rows();
  } else {
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const long long m = m0 + tid / CPW + RPP * i;
      va[i] = aok[i] ? ((unsigned)m * (unsigned)p.lda + (unsigned)kc * 8u) * 2u : OOB;
    }
  }
  unsigned b2lane = 0;                             // pass i of this WAVE reads rows of the second weight block
  const int wlo = p.b2_klen > 0 ? p.b2_k0 : 0, whi = p.b2_klen > 0 ? p.b2_k0 + p.b2_klen : 0x7fffffff;
#pragma unroll
  for (int i = 0; i < NCB; ++i) {
    const int nl = tid / CPW + RPP * i;            // row of the tile's weight block
    const int n = n0 + nl;
    // a wave's WROWS rows of one pass start at a multiple of WROWS and n_split % WROWS == 0 (launch_w8): wave-uniform side
    bool second = p.n_split > 0 && (n0 + wave * WROWS + RPP * i) >= p.n_split;
    unsigned row = (unsigned)(second ? n - p.n_split : n);
    bool rok = nl < ncols;
    if (lr2) {                                     // (rk0 and nbase are multiples of 8 = WROWS: wave-uniform side)
      second = (wave * WROWS + RPP * i) >= rk0;
      row = (unsigned)(second ? nl - rk0 + member * p.lr_rp : n);
      rok = second ? (nl - rk0 < p.lr_rp) : (nl < nbase);
    }
    if (lr3) {                                     // (rk3 and nbase are multiples of 8 = WROWS: wave-uniform side)
      second = (wave * WROWS + RPP * i) >= rk3;
      row = (unsigned)(second ? nl - rk3 : n);
      rok = second ? (nl - rk3 < p.lr_rp) : (nl < nbase);
    }
    if (second) b2lane |= 1u << i;
    vb[i] = rok ? (row * (unsigned)(second ? p.ldb2 : p.ldb) + (unsigned)kc * 8u) * 2u : OOB;   // rows past the tile: zeros, no fetch
  }

  // ---- mode 3, keep-bit plane: the rank waves fetch their mask words with the stage (see MASK_OFF above)
  const bool m_pl = lr3 && rankwave && p.lr_plane != nullptr && p.lr_drop_p > 0.f;
  __amdgpu_buffer_rsrc_t srdP = __builtin_amdgcn_make_buffer_rsrc((void*)(m_pl ? p.lr_plane : p.B), 0, 0x80000000u, 0x00020000);
  unsigned vpm[NMK];                               // per-lane byte offset of (fragment, row) inside the stage's plane words
  int lk_w = 0x3fffffff;                           // K width of a mask member (the loader runs ahead of the compute)
  unsigned lk_bytes = 0;                           // bytes of one member's plane
  const int rwid = (wave % (WM * KG));             // rank wave id: wr + WM * kg
#pragma unroll
  for (int f = 0; f < NMK; ++f) vpm[f] = OOB;
  if constexpr (lr3) {
    if (m_pl) {
      const int W = p.lr_group_cols > 0 ? p.lr_group_cols : p.K;
      lk_w = W;
      lk_bytes = (unsigned)(((long long)M * W) >> 3);
      // TM = 32: one instruction, lanes 0-31 = fragment 0 / lanes 32-63 = fragment 1 of rows 0..31; TM = 64: instruction f =
      // fragment f, lane = row
#pragma unroll
      for (int f = 0; f < NMK; ++f) {
        const int r = NMK == 1 ? (lane & 31) : lane, fr = NMK == 1 ? (lane >> 5) : f;
        const long long m = m0 + wr * TM + r;
        vpm[f] = m < M ? (unsigned)(((long long)fr * M + m) * 4) : OOB;
      }
    }
  }
  auto issue_mask = [&](int k0, int stage) {
    if constexpr (lr3) {
      const int lmem = (k0 >= lk_w ? 1 : 0) + (k0 >= 2 * lk_w ? 1 : 0);      // (uniform) member of the projection group
      const unsigned so = (unsigned)lmem * lk_bytes + (unsigned)((k0 - lmem * lk_w) >> 5) * (unsigned)M * 4u;
      unsigned char* dst = smem + MASK_OFF + (rwid * NSTAGE + stage) * MASK_SLOT;
#pragma unroll
      for (int f = 0; f < NMK; ++f) {
        if (NMK == 1)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srdP, (__attribute__((address_space(3))) void*)(dst + lane * 4), 4, (int)vpm[0], (int)so, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srdP, (__attribute__((address_space(3))) void*)(dst + f * 256 + lane * 4), 4, (int)vpm[f],
                                                   (int)so, 0, 0);      // (vpm[f] carries the fragment's f * M words)
      }
    }
  };
  // pieces [lo, hi) of the LPT per-thread LDS-DMA loads of one stage (A passes first); the phased schedules spread a stage
  // over several phases.  advance_window() after the stage's last A piece.
  auto issue_pieces = [&](int k0, int stage, int lo, int hi) {
    if ((dbg & 1) && k0 >= kbeg + NSTAGE * BKT) return;     // ablation (T2V_W8_DBG=1): steady state without operand traffic
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
    if constexpr (lr3) {
      if (m_pl && hi >= LPT && lo < LPT) issue_mask(k0, stage);     // (with the stage's last operand piece: LPT + NMK loads per stage)
    }
  };
  auto advance_window = [&]() {
    if (is_conv) {
      sc += BKT;
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
  auto issue = [&](int k0, int stage) {
    issue_pieces(k0, stage, 0, LPT);
    advance_window();
  };

  // the ring's first stages go out BEFORE the rest of the set-up (accumulators, fragment addressing): their memory latency
  // is the longest item in front of the first MFMA
  constexpr int NPRO = (SCHED == 0 || SCHED == 4) ? NSTAGE - 1 : NSTAGE;
#pragma unroll
  for (int s = 0; s < NPRO; ++s)
    if (s < nt) issue(kbeg + s * BKT, s);

  // Accumulators hold the TRANSPOSED fragment (the weight fragment is the MFMA's first operand): lane l of fragment (i, j) has
  // tile row i*32 + (l & 31) and, in register r, column j*32 + 8*(r >> 2) + 4*(l >> 5) + (r & 3) — four consecutive COLUMNS per
  // register quad, which one v_permlane32_swap per register turns into eight consecutive columns per lane: the output chunk of
  // the store (see the epilogue).
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addressing: row r of a tile sits at r*ROWB bytes, its 16-byte K chunk q at slot q ^ swz(r); TM, TN are multiples
  // of 32, so the swizzle term depends on the lane alone; chunk = 2*kk + (lane>>5)
  const unsigned swz = BKT == 64 ? (unsigned)((lane >> 1) & 7) : (unsigned)((lane >> 2) & 3);
  const unsigned qsw = (((unsigned)(lane >> 5)) ^ swz) << 4;
  const unsigned arow = (unsigned)(wr * TM + (lane & 31)) * (unsigned)ROWB;
  const unsigned brow = (unsigned)(wc * TN + (lane & 31)) * (unsigned)ROWB;
  // column fragments of this wave that hold columns of the tile
  const int nfw = max(0, min(FN, (ncols - wc * TN + 31) >> 5));

  // ---- mode 3: mask state of the wave column that owns the rank fragments.  The mask of member i (a projection group's members
  // partition K; a layer of its own is one member spanning K) is the one its forward launch drew: seed i, row width W, column
  // k - i W — the matrix t2v_lora_drop_dt regenerated.  Keys of up to three members live in scalar registers; the K loop walks K
  // monotonically, so the current member advances by comparison (no division per step).
  unsigned mk_s0[3] = {0, 0, 0}, mk_s1[3] = {0, 0, 0}, mk_thr = 0;
  unsigned rowq[FM];
  int m_w = 0x3fffffff;                            // K width of a mask member (>= K: one member)
  bool m_on = false;
#pragma unroll
  for (int i = 0; i < FM; ++i) rowq[i] = 0;
  if constexpr (lr3) {
    if (rankwave && p.lr_drop_p > 0.f) {
      m_on = true;
      const int W = p.lr_group_cols > 0 ? p.lr_group_cols : p.K;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        if (m_pl) continue;                          // (keep bits come from the forward launch's plane: no keys)
        const DropKey dk = drop_key(eff_seed(m == 0 ? p.lr_drop_seed : p.lr_group_seed[m - 1], p.drop_epoch), p.lr_drop_p);
        mk_s0[m] = dk.s0;
        mk_s1[m] = dk.s1;
        mk_thr = (dk.thr - 1u) * 0x00010001u;       // (thr >= 1 for p > 0)
      }
      m_w = W;
#pragma unroll
      for (int i = 0; i < FM; ++i)
        rowq[i] = (unsigned)(((unsigned long long)((unsigned)m0 + wr * TM + i * 32 + (lane & 31)) * (unsigned)W) >> 2);
    }
  }
  if (dbg & 8) tl[1] = __builtin_readcyclecounter();
  auto kloop = [&](auto nf_tag, auto rk_tag) {
    constexpr int NF = decltype(nf_tag)::value;
    constexpr int RK = decltype(rk_tag)::value;      // 1 / 2: this wave owns the rank fragments of a mode-3 tile (NF == FN) and
                                                     // hashes the mask (1) / reads the forward launch's keep-bit plane (2)
    constexpr int LW = LPT + (RK == 2 ? NMK : 0);    // loads this wave issues per stage (counted waits)
    auto load_frags = [&](bf16x8(&af)[FM], bf16x8(&bfr)[FN], int stage, int kk) {
      const unsigned char* sA = smem + stage * STAGE;
      const unsigned char* sB = sA + A_BYTES;
      const unsigned ko = qsw ^ ((unsigned)kk << 5);
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8*)(sA + arow + i * (32 * ROWB) + ko);
#pragma unroll
      for (int j = 0; j < NF; ++j) bfr[j] = *(const bf16x8*)(sB + brow + j * (32 * ROWB) + ko);
    };
    auto mfma = [&](const bf16x8(&af)[FM], const bf16x8(&bfr)[FN], int kq, int cstage) {
      if constexpr (RK != 0) {
        // kq = K index of this k16 step: the lane's chunk holds k = kq + 8 (lane >> 5) .. + 7 of its row.  Straight-line code only:
        // uniform branches around single MFMAs made the compiler copy accumulators and pad with s_nop (round 6, first version) —
        // the member index comes from two compares, the operand of fragment FN - 2 from a select.
        const int mem = (kq >= m_w ? 1 : 0) + (kq >= 2 * m_w ? 1 : 0);      // (m_w >= K for a layer of its own: always 0)
        const unsigned kk = (unsigned)(kq - mem * m_w);                      // member-local k (members span whole 64-deep stages)
        bf16x8 maf[FM];
        if constexpr (RK == 2) {
          const unsigned char* ms = smem + MASK_OFF + (rwid * NSTAGE + cstage) * MASK_SLOT + ((kk >> 5) & 1u) * (TM * 4) + (lane & 31) * 4;
          const unsigned sh = 4u * (((kk & 31u) >> 3) + (unsigned)(lane >> 5));
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            const unsigned w = *(const unsigned*)(ms + i * 128);
            maf[i] = mask_chunk_bits(af[i], ((w >> sh) & 0xfu) | (((w >> (16u + sh)) & 0xfu) << 4));
          }
        } else {
          const unsigned s0 = mem == 0 ? mk_s0[0] : (mem == 1 ? mk_s0[1] : mk_s0[2]);
          const unsigned s1 = mem == 0 ? mk_s1[0] : (mem == 1 ? mk_s1[1] : mk_s1[2]);
          const unsigned kl4 = (kk + 8u * (unsigned)(lane >> 5)) >> 2;
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            const bf16x8 mk = mask_chunk(af[i], rowq[i] + kl4, s0, s1, mk_thr);
            maf[i] = m_on ? mk : af[i];
          }
        }
        // fragment FN - 2 is a rank fragment only with two of them (a projection group's 48 .. 64 ranks); its B rows outside the
        // current member's K range are zero, so both fragments simply multiply at every step
        bf16x8 a2[FM];
        const bool two = nrf == 2;
#pragma unroll
        for (int i = 0; i < FM; ++i) a2[i] = two ? maf[i] : af[i];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < NF; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], j == FN - 1 ? maf[i] : (j == FN - 2 ? a2[i] : af[i]), acc[i][j], 0, 0, 0);
        return;
      }
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
        for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    };
    auto phase_barrier = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (SCHED == 0 || SCHED == 4) {
      // classic ring: wait for stage `it`, barrier, refill the stage freed by the barrier, compute.  SCHED 4 spreads the
      // refill's LDS-DMA pieces over the k16 steps of the stage instead of issuing them in one burst behind the barrier (a
      // burst keeps every wave at its buffer_load for stage-bytes / 64 B per clk with the matrix pipes idle)
      constexpr int NPH = SCHED == 4 ? ((NSTAGE == 2 && KS > 1) ? KS - 1 : KS) : 1;
      constexpr int PP = (LPT + NPH - 1) / NPH;
      int stage = 0;
      for (int it = 0; it < nt; ++it) {
        const int ahead = min(nt, it + NSTAGE - 1) - (it + 1);
        if (ahead >= 2) wait_vmcnt<2 * LW>();
        else if (ahead == 1) wait_vmcnt<LW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if ((dbg & 8) && it == 0) tl[2] = __builtin_readcyclecounter();
        const bool refill = it + NSTAGE - 1 < nt;
        int ws = stage + NSTAGE - 1;
        if (ws >= NSTAGE) ws -= NSTAGE;
        if (SCHED == 0 && refill) issue(kbeg + (it + NSTAGE - 1) * BKT, ws);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          if (SCHED == 4 && refill && j < NPH) {
            issue_pieces(kbeg + (it + NSTAGE - 1) * BKT, ws, j * PP, (j + 1) * PP < LPT ? (j + 1) * PP : LPT);
            if (j == NPH - 1) advance_window();
          }
          bf16x8 af[FM], bfr[FN];
          load_frags(af, bfr, stage, kg * KS + j);
          mfma(af, bfr, kbeg + it * BKT + (kg * KS + j) * 16, stage);
        }
        if (++stage == NSTAGE) stage = 0;
      }
    } else if constexpr (SCHED == 5) {
      // classic ring, software-pipelined: the fragments of k16 step j+1 are read (second register set) before step j multiplies;
      // the per-stage wait + barrier sits BEFORE the last k16 step, so the hand-over and the first fragment reads of the next
      // stage run under that step's MFMAs; the refill of the slot freed by the barrier goes out in three portions (behind the
      // barrier, and in the first two k16 steps of the next stage) instead of one burst.
      static_assert(KS % 2 == 0, "two fragment sets alternate per k16 step");
      constexpr int NP = KS >= 4 ? 3 : 2;                   // portions of a refill
      constexpr int PP = (LPT + NP - 1) / NP;
      {
        const int ahead = min(nt, NSTAGE) - 1;
        if (ahead >= 2) wait_vmcnt<2 * LW>();
        else if (ahead == 1) wait_vmcnt<LW>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (dbg & 8) tl[2] = __builtin_readcyclecounter();
      bf16x8 af0[FM], bf0[FN], af1[FM], bf1[FN];
      load_frags(af0, bf0, 0, kg * KS);
      int stage = 0, pk = 0, pslot = 0;
      bool pend = false;                                    // a refill whose later portions are still to go out
      for (int it = 0; it < nt; ++it) {
        // k16 steps 0 .. KS-2 (pairs: set0 then set1)
#pragma unroll
        for (int j = 0; j + 1 < KS; ++j) {
          if (j & 1) load_frags(af0, bf0, stage, kg * KS + j + 1);
          else load_frags(af1, bf1, stage, kg * KS + j + 1);
          if (pend && j + 1 < NP) {
            issue_pieces(pk, pslot, (j + 1) * PP, (j + 2) * PP < LPT ? (j + 2) * PP : LPT);
            if (j + 2 == NP) {
              advance_window();
              pend = false;
            }
          }
          if (j & 1) mfma(af1, bf1, kbeg + it * BKT + (kg * KS + j) * 16, stage);
          else mfma(af0, bf0, kbeg + it * BKT + (kg * KS + j) * 16, stage);
        }
        int nstage = stage + 1;
        if (nstage == NSTAGE) nstage = 0;
        if (it + 1 < nt) {
          wait_lgkm0();                                     // this wave is done reading `stage`
          // stage it+1 landed; what may stay in flight: the stages behind it (NSTAGE = 3: stage it+2)
          if (NSTAGE >= 3 && it + 2 < nt) wait_vmcnt<LW>();
          else wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();                     // stage it+1 visible to every wave; `stage` is free
          if (it + NSTAGE < nt) {
            pk = kbeg + (it + NSTAGE) * BKT;
            pslot = stage;
            issue_pieces(pk, pslot, 0, PP);
            pend = true;
          }
          load_frags(af0, bf0, nstage, kg * KS);
        }
        mfma(af1, bf1, kbeg + it * BKT + (kg * KS + KS - 1) * 16, stage);     // last k16 step of the stage (KS even: it sits in set 1)
        stage = nstage;
      }
    } else if constexpr (SCHED == 2) {
      // ping-pong, one fragment set (read in M, consumed in C).  Phase k of group 0 runs beside phase k-1 of group 1 (group 1
      // passes one extra barrier first, group 0 one at the end).
      //   stage s+1 must have landed before group 0 opens M(s+1, 0): every wave drains its own pieces of it in the phase that
      //   ends at that barrier — C(s, KS-1) for group 0, M(s, KS-1) for group 1
      //   the slot of stage s-1 is refilled during the M phases of stage s (all reads of s-1 ended before M(s, 0) of group 0)
      constexpr int NPH = NSTAGE == 2 ? (KS > 2 ? KS - 2 : 1) : KS;     // M phases of a stage that carry DMA pieces
      constexpr int PP = (LPT + NPH - 1) / NPH;
      const int grp = wave >> 2;
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
      int stage = 0;
      for (int it = 0; it < nt; ++it) {
        const int rs = it + NSTAGE - 1;                      // stage whose pieces go out during this one
        const bool refill = it >= 1 && rs < nt;
        int rslot = stage - 1;
        if (rslot < 0) rslot += NSTAGE;
        const int ahead = min(nt - 1, it + NSTAGE - 1) - (it + 1);      // stages that may stay in flight behind stage it+1
        auto stage_wait = [&]() {
          if (it + 1 < nt) wait_vmcnt_sel<LPT, 0>(ahead >= 1);
        };
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          // ---- M phase
          load_frags(af, bfr, stage, kg * KS + j);
          if (j < NPH && refill) {
            issue_pieces(kbeg + rs * BKT, rslot, j * PP, (j + 1) * PP < LPT ? (j + 1) * PP : LPT);
            if (j == NPH - 1) advance_window();
          }
          wait_lgkm0();
          if (j == KS - 1 && grp == 1) stage_wait();
          phase_barrier();
          // ---- C phase
          __builtin_amdgcn_s_setprio(1);
          mfma(af, bfr, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          if (j == KS - 1 && grp == 0) stage_wait();
          phase_barrier();
        }
        if (++stage == NSTAGE) stage = 0;
      }
      if (grp == 0) __builtin_amdgcn_s_barrier();
    } else {
      // SCHED 3: ping-pong with prefetched fragments over a deep ring of shallow stages (KS == 2: two k16 steps per stage).
      //   C(s, j): read the fragments of the NEXT k16 step into the other register set, then multiply the current set
      //   M(s, j): this wave's share j of the pieces of stage s+NSTAGE-1 (into the slot of stage s-1), wait for the prefetched
      //            fragments; nothing else
      // Hazards (g0 = group 0, g1 runs one phase behind):
      //   reads of stage s-1 end with C(s-1, 0) (fragments of its step 1), drained in M(s-1, 1); g1's M(s-1, 1) runs beside
      //   g0's C(s-1, 1), i.e. before g0's M(s, 0) issues the first refill piece into that slot;
      //   stage s+1 is first read in C(s, 1): g0 drains its pieces of it at the end of M(s, 1), g1 at the end of C(s, 0) (the
      //   phases that end at the barrier in front of g0's C(s, 1)); with four stages those pieces went out two stages earlier.
      static_assert(KS == 2 && NSTAGE >= 3, "prefetching ping-pong: two k16 steps per stage, ring of >= 3 stages");
      constexpr int PP = (LPT + 1) / 2;
      const int grp = wave >> 2;
      {
        const int ahead = min(nt, NSTAGE) - 1;
        if (ahead >= 3) wait_vmcnt<3 * LPT>();
        else if (ahead == 2) wait_vmcnt<2 * LPT>();
        else if (ahead == 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      bf16x8 af0[FM], bf0[FN], af1[FM], bf1[FN];
      load_frags(af0, bf0, 0, kg * KS);
      if (grp == 1) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      int stage = 0;
      for (int it = 0; it < nt; ++it) {
        const int rs = it + NSTAGE - 1;
        const bool refill = it >= 1 && rs < nt;
        const bool steady = it + NSTAGE - 1 < nt;             // the counted waits below assume the refill pieces went out
        int rslot = stage - 1;
        if (rslot < 0) rslot += NSTAGE;
        int nstage = stage + 1;
        if (nstage == NSTAGE) nstage = 0;
        // ---- M(it, 0)
        if (refill) issue_pieces(kbeg + rs * BKT, rslot, 0, PP);
        wait_lgkm0();
        phase_barrier();
        // ---- C(it, 0)
        load_frags(af1, bf1, stage, kg * KS + 1);
        __builtin_amdgcn_s_setprio(1);
        mfma(af0, bf0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (grp == 1 && it + 1 < nt) {                        // g1: stage it+1 landed (its own pieces), one phase early
          if (steady) wait_vmcnt<(NSTAGE - 3) * LPT + PP>();
          else wait_vmcnt<0>();
        }
        phase_barrier();
        // ---- M(it, 1)
        if (refill) {
          issue_pieces(kbeg + rs * BKT, rslot, PP, LPT);
          advance_window();
        }
        wait_lgkm0();
        if (grp == 0 && it + 1 < nt) {                        // g0: stage it+1 landed
          if (steady) wait_vmcnt<(NSTAGE - 2) * LPT>();
          else wait_vmcnt<0>();
        }
        phase_barrier();
        // ---- C(it, 1)
        if (it + 1 < nt) load_frags(af0, bf0, nstage, kg * KS);
        __builtin_amdgcn_s_setprio(1);
        mfma(af1, bf1, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        phase_barrier();
        stage = nstage;
      }
      if (grp == 0) __builtin_amdgcn_s_barrier();
    }
  };
  using NoRank = std::integral_constant<int, 0>;
  if (lr3 && rankwave && m_pl) kloop(std::integral_constant<int, FN>{}, std::integral_constant<int, lr3 ? 2 : 0>{});
  else if (lr3 && rankwave) kloop(std::integral_constant<int, FN>{}, std::integral_constant<int, lr3 ? 1 : 0>{});
  else if (nfw >= FN) kloop(std::integral_constant<int, FN>{}, NoRank{});
  else if (FN >= 2 && nfw == FN - 1) kloop(std::integral_constant<int, (FN >= 2 ? FN - 1 : FN)>{}, NoRank{});
  else kloop(std::integral_constant<int, (FN >= 3 ? FN - 2 : (FN >= 2 ? FN - 1 : FN))>{}, NoRank{});

  wait_vmcnt<0>();
  __syncthreads();                                 // ring idle: reuse it as the epilogue staging buffer
  if (dbg & 8) tl[3] = __builtin_readcyclecounter();

  // ---- epilogue: K groups summed through LDS, then accumulators -> LDS (fp32) -> 16-byte coalesced rows.
  // A thread keeps ONE 8-column chunk for all rows it writes (its bias / row-bias terms are loaded once, right here, and land
  // under the staging pass), the residual chunks of a pass are requested before the staging barrier, and the row loop is fully
  // unrolled — the first version re-loaded bias and residual inside a rolled loop and spent 8-12 us per launch on exposed
  // load latency (profiles/r03_w8_timeline.txt), more than the HBM time of the tile's stores.
  constexpr int CPR = BN / 8;
  constexpr int WRP = Epi::WRP, WPP = Epi::WPP, FG = Epi::FG, PROWS = Epi::PROWS, NPASS = Epi::NPASS, SB = Epi::SB;
  // staging row rl: band rl >> 5 = fi * WPP + (wave row in the group), fragment row rl & 31; tile row of (pass ib/h, rl):
  auto tile_row = [&](int ib, int h, int rl) {
    const int band = rl >> 5;
    return (h * WPP + band % WPP) * TM + (ib * FG + band / WPP) * 32 + (rl & 31);
  };
  constexpr int RPI = NT / CPR;                                     // rows per output iteration (threads beyond RPI*CPR idle)
  constexpr int ITERS = (PROWS + RPI - 1) / RPI;
  float* sC = (float*)smem;                        // PROWS x BN fp32, one 32-row fragment band per wave row and pass
  const int cc = tid % CPR, r0 = tid / CPR;
  const int col = n0 + cc * 8;
  const bool cact = r0 < RPI && cc * 8 < nbase;
  const bool rankcol = p.n_split > 0 && col >= p.n_split;
  const float alpha = p.alpha, beta = p.beta;
  const bool act_silu = p.act == T2V_ACT_SILU;
  const int Nb = p.n_split > 0 ? p.n_split : N;
  const int cs_mode = p.colsum ? p.cs_mode : 0;
  // ---- split-K hand-off (splits > 1).  The splits of a tile draw a ticket when their K loop is done; the LAST one keeps its
  // partial tile in LDS / registers and becomes the reducer, the others write theirs as fp32 rows to the caller's workspace
  // (slab [tile][z]) and then count themselves done.  The reducer waits for splits-1 done marks (those workgroups finished
  // their K loops BEFORE it did, so they are resident and running: no co-residency assumption), acquires, and adds the slabs
  // IN SPLIT ORDER with its own partial in its place — the sum does not depend on who came last.
  // Guide G16 forms: writers = plain stores -> every wave vmcnt(0) -> barrier -> lane 0 agent release -> counter;
  // reducer = relaxed polls -> one agent acquire -> barrier -> plain loads.
  int role = 0;                                    // 0: no split, 1: writer, 2: reducer
  unsigned* cnt = nullptr;
  float* slab0 = nullptr;                          // slab of (tile, z = 0); slab z at + z * BM*BN floats
  if (splits > 1) {
    cnt = (unsigned*)p.workspace + 2 * (tm * ntn + tn);
    slab0 = (float*)((unsigned char*)p.workspace + 65536) + (long long)(tm * ntn + tn) * splits * (BM * BN);
    int* flag = (int*)(smem + SMEM_BYTES);         // 16 spare bytes behind the ring / staging buffer (same LDS object)
    if (tid == 0) *flag = (int)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(*flag);
    role = ticket == splits - 1 ? 2 : 1;
    if (role == 2) {
      if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(splits - 1)) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1 << 24)) {               // give up instead of hanging the device; flagged for the host
            ((unsigned*)p.workspace)[16383] = 0xdeadu;
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // re-armed for the next launch
        __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
    }
  }
  const bool writer = role == 1;


  // ---- rank-wide epilogue term (T2VGemm.lr_*, LR instantiations): acc[io][j] += lr_scale * mask * sum LA LB^T for the NB
  // fragment bands this wave holds (band io = tile band ib0_ + io of its wave row).  Operands go straight into the MFMA that the
  // K loop uses (weight side = first operand): LB fragment = 16 bytes of row `column` of lr_b, LA fragment = 16 bytes of a rank-
  // wide row (mode 1: from memory, gathered per tap; mode 2: this tile's own rank-column accumulators, rounded to bf16 and
  // passed between the wave columns through LDS — their register order IS a permuted k order, LB is read in the same order).
  // One fragment's operands at a time (the compiler hoists the column fragments' loads of a tap into one batch: ~2 us per tap).
  // Measured alternatives (scripts/lr_probe.py): both operands of two taps buffered in registers — spills, slower; both operands
  // staged in LDS by all 512 threads — three barriers and index arithmetic cost more than the round trips they save.
  // barrier of an LDS hand-over: waits for this wave's LDS operations only.  __syncthreads() also drains vmcnt, i.e. every
  // epilogue operand requested ahead of its use (residual rows, LB rows) would be waited for at the next hand-over
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // `early`: called once, right after the phase's own operand loads have been issued — the register epilogue requests the
  // residual rows of its first chunks there.  vmcnt retires in order: a load that is younger than those requests cannot be waited
  // for without waiting for them, so on the K-group tiles (LBALL) the LB rows of ALL fragments go out first, at the top of the
  // phase; the other kernels read them one fragment ahead (no early requests there), the staged ones — whose whole tile is live
  // in registers during the phase — fragment by fragment.
  auto rank_phase = [&](int ib0_, auto nb_tag, auto&& early) {
    constexpr int NB = decltype(nb_tag)::value;
    constexpr bool LBALL = !CS && KG == 2;
    constexpr int NLB = LBALL ? FN : (CS ? 1 : 2);
    const int half = lane >> 5, l31 = lane & 31;
    const int rp = p.lr_rp, nk = (rp + 15) >> 4, rkp = nk * 16;        // (nk <= 2 in mode 2, <= 3 in mode 1: launch_w8)
    const bool masked = p.lr_drop_p > 0.f;
    const unsigned long long mseed = member == 0 ? p.lr_drop_seed : p.lr_group_seed[member == 1 ? 0 : 1];
    const DropKey dkey = drop_key(masked ? eff_seed(mseed, p.drop_epoch) : 0ull, p.lr_drop_p);
    const unsigned mwidth = gcols ? (unsigned)gcols : (unsigned)N;      // mask row width / column origin of this tile's member
    const int mcol0 = n0 - member * gcols;
    const float sc = masked ? p.lr_scale / (1.f - p.lr_drop_p) : p.lr_scale;
    const bool direct = !masked && sc == 1.f;
    const bf16_t* LB = (const bf16_t*)p.lr_b;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const bf16x4 zero4 = {0, 0, 0, 0};
    bf16x8 la[NB][3];
    unsigned rowg[NB];
#pragma unroll
    for (int io = 0; io < NB; ++io) rowg[io] = (unsigned)m0 + wr * TM + (ib0_ + io) * 32 + l31;
    auto col0_of = [&](int j) { return wc * TN + j * 32; };
    if constexpr (LR == 2) {
      // ---- mode 3: dt = (mask dy / (1-p)) U^T of this tile's rows sits, complete, in the rank fragments of the last wave column
      // (fragments FN - nrf .. FN - 1).  As in mode 2 the transposed accumulator's register order — registers 8s .. 8s+7 of a
      // lane = ranks 32f + 16s + {4h .. 4h+3, 8+4h .. 8+4h+3} of tile row (lane & 31) — is used as the k order of the product
      // with LB = (s D)^T, which is read in the same permuted order: 16-rank chunk c = 2f + s, up to four of them.
      const int nkc = (rp + 15) >> 4;
      const float inv = p.lr_drop_p > 0.f ? 1.f / (1.f - p.lr_drop_p) : 1.f;
      bf16x8* xt = (bf16x8*)smem;                    // [WM * FM bands][4 chunks][64 lanes]
      // LB rows: unconditional loads (columns beyond the base columns read the last one — never stored; a chunk beyond the rank
      // re-reads chunk 0 — never multiplied)
      auto load_lb3 = [&](int j, bf16x8(&lb)[4]) {
        const int cl = min(col0_of(j) + l31, nbase - 1);
        const bf16_t* lbp = LB + (unsigned)(n0 + cl) * (unsigned)p.lr_ldb + 4 * half;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int o = c < nkc ? 16 * c : 0;
          const bf16x4 lo = *(const bf16x4*)(lbp + o);
          const bf16x4 hi = *(const bf16x4*)(lbp + o + 8);
          lb[c] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      };
      bf16x8 lbq[NLB][4];
      if (LBALL) {
#pragma unroll
        for (int j = 0; j < FN; ++j) load_lb3(j, lbq[LBALL ? j : 0]);
      } else if (NLB == 2) {
        load_lb3(0, lbq[0]);
      }
      early();
      if (KG == 2) lds_barrier();                    // (the K-group exchange's reads of this LDS are over)
      if (rankwave) {
#pragma unroll
        for (int jj = FN - 2; jj < FN; ++jj) {
          const int f = jj - (FN - nrf);
          if (f < 0) continue;
#pragma unroll
          for (int io = 0; io < NB; ++io)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const int c = 2 * f + ks;
              if (c >= nkc) continue;
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = acc[io][jj][8 * ks + e] * inv;
              const bf16x8 tv = pack8bf(v);
              xt[((wr * FM + ib0_ + io) * 4 + c) * 64 + lane] = tv;
              if (tn == 0 && rowg[io] < (unsigned)M) {        // (first column tile) dt for the factor gradient dD
                bf16_t* tp = (bf16_t*)p.D2 + rowg[io] * (unsigned)p.ldd2 + 16 * c + 4 * half;
                const bf16x4 lo = {tv[0], tv[1], tv[2], tv[3]}, hi = {tv[4], tv[5], tv[6], tv[7]};
                if (16 * c + 4 * half < rp) *(bf16x4*)tp = lo;
                if (16 * c + 8 + 4 * half < rp) *(bf16x4*)(tp + 8) = hi;
              }
            }
        }
      }
      lds_barrier();
      bf16x8 la3[NB][4];
#pragma unroll
      for (int io = 0; io < NB; ++io)
#pragma unroll
        for (int c = 0; c < 4; ++c) la3[io][c] = c < nkc ? xt[((wr * FM + ib0_ + io) * 4 + c) * 64 + lane] : zero8;
      if constexpr (CS) lds_barrier();               // (the staging passes reuse this LDS)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if (!LBALL && NLB == 2 && j + 1 < FN) load_lb3(j + 1, lbq[(j + 1) & 1]);
        if (NLB == 1) load_lb3(j, lbq[0]);
        if (col0_of(j) >= nbase) continue;           // padding, rank fragments
        const bf16x8(&lb)[4] = lbq[LBALL ? j : (j & (NLB - 1))];
#pragma unroll
        for (int io = 0; io < NB; ++io)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c < nkc) acc[io][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lb[c], la3[io][c], acc[io][j], 0, 0, 0);
      }
      return;
    } else {
    // (mode 2) keep-bit plane for the backward-data launch of the same layer (T2VGemm.lr_plane, fragment-major: the lane's 16 keep
    // bits of a fragment are one 16-bit word; a wave's 64 words of a fragment are one 128-byte line)
    unsigned short* plane = (masked && lr2 && p.lr_plane) ? (unsigned short*)((unsigned char*)p.lr_plane + (size_t)member * (((size_t)M * mwidth) >> 3)) : nullptr;
    // keep bits of fragment (io, j): bit 4q + e = element (band row, fragment column 8q + 4 (lane >> 5) + e) kept.  32-bit quad
    // indices (launch_w8: the mask matrix has < 2^34 elements); straight-line code — thr = 0 without a mask keeps everything.
    // (Hashed in the prologue instead, under the first stages' latency, the 3 000 cycles were not hidden — the set-up already
    // covers that latency — and the words cost registers through the K loop: profiles/r06_w8_epilogue_timeline.txt.)
    unsigned rq[NB];
#pragma unroll
    for (int io = 0; io < NB; ++io) rq[io] = (unsigned)(((unsigned long long)rowg[io] * mwidth) >> 2);
    const unsigned cq0 = (unsigned)(mcol0 + wc * TN + 4 * half) >> 2;
    auto finish = [&](int io, int j, const f32x16& tmp) {       // acc += sc * mask * tmp
      unsigned pw = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const DropQuad h = drop_quad(dkey, (unsigned long long)(rq[io] + cq0 + (unsigned)(8 * j + 2 * q)));
        pw |= (((h.a & 0xffffu) >= dkey.thr ? 1u : 0u) | ((h.a >> 16) >= dkey.thr ? 2u : 0u) | ((h.b & 0xffffu) >= dkey.thr ? 4u : 0u) |
               ((h.b >> 16) >= dkey.thr ? 8u : 0u)) << (4 * q);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned km = (unsigned)__builtin_amdgcn_sbfe((int)pw, r, 1);                         // 0 / ~0
        acc[io][j][r] += __builtin_bit_cast(float, __builtin_bit_cast(unsigned, sc * tmp[r]) & km);
      }
      if (plane && rowg[io] < (unsigned)M)
        plane[(((size_t)((mcol0 + col0_of(j)) >> 5) * (size_t)M + rowg[io]) << 1) + half] = (unsigned short)pw;
    };
    if (lr2) {
      // this tile's t = x (*) D^T sits in the accumulators of the rank fragment (wave column wc_r, fragment j_r): registers
      // 8s .. 8s+7 of a lane = ranks 16s + {4h .. 4h+3, 8+4h .. 8+4h+3} of tile row (lane & 31) — used as k slots 0..7 as they are
      bf16x8* xt = (bf16x8*)smem;                    // [WM * FM bands][2 k16 steps][64 lanes]
      const int wc_r = rk0 / TN, j_r = (rk0 - wc_r * TN) >> 5;
      // LB rows of fragment j, unconditional loads (columns beyond the base columns read the last one — never stored; a k16 step
      // beyond the rank re-reads step 0 — never multiplied), requested at the top of the phase (LBALL) or one fragment ahead
      auto load_lb2 = [&](int j, bf16x8(&lb)[2]) {
        const int cl = min(col0_of(j) + l31, nbase - 1);
        const bf16_t* lbp = LB + (unsigned)(n0 + cl) * (unsigned)p.lr_ldb + 4 * half;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int o = ks < nk ? 16 * ks : 0;
          const bf16x4 lo = *(const bf16x4*)(lbp + o);
          const bf16x4 hi = *(const bf16x4*)(lbp + o + 8);
          lb[ks] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      };
      bf16x8 lbq[NLB][2];
      if (LBALL) {
#pragma unroll
        for (int j = 0; j < FN; ++j) load_lb2(j, lbq[LBALL ? j : 0]);
      } else if (NLB == 2) {
        load_lb2(0, lbq[0]);
      }
      early();
      if (KG == 2) lds_barrier();                    // (the K-group exchange's reads of this LDS are over)
      if (wc == wc_r) {
#pragma unroll
        for (int jj = 0; jj < FN; ++jj) {
          if (jj != j_r) continue;
#pragma unroll
          for (int io = 0; io < NB; ++io)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              if (ks >= nk) continue;
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = acc[io][jj][8 * ks + e];
              const bf16x8 tv = pack8bf(v);
              xt[((wr * FM + ib0_ + io) * 2 + ks) * 64 + lane] = tv;
              if (mcol0 == 0 && rowg[io] < (unsigned)M) {     // (first tile of the layer / member) the saved down-projection
                bf16_t* tp = (bf16_t*)p.D2 + rowg[io] * (unsigned)p.ldd2 + member * rp + 16 * ks + 4 * half;
                const bf16x4 lo = {tv[0], tv[1], tv[2], tv[3]}, hi = {tv[4], tv[5], tv[6], tv[7]};
                if (16 * ks + 4 * half < rp) *(bf16x4*)tp = lo;
                if (16 * ks + 8 + 4 * half < rp) *(bf16x4*)(tp + 8) = hi;
              }
            }
        }
      }
      lds_barrier();
#pragma unroll
      for (int io = 0; io < NB; ++io)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) la[io][ks] = ks < nk ? xt[((wr * FM + ib0_ + io) * 2 + ks) * 64 + lane] : zero8;
      if constexpr (CS) lds_barrier();               // (the staging passes reuse this LDS)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if (!LBALL && NLB == 2 && j + 1 < FN) load_lb2(j + 1, lbq[(j + 1) & 1]);
        if (NLB == 1) load_lb2(j, lbq[0]);
        if (col0_of(j) >= nbase) continue;           // rank fragment, padding
        const bf16x8(&lb)[2] = lbq[LBALL ? j : (j & (NLB - 1))];
#pragma unroll
        for (int io = 0; io < NB; ++io) {
          f32x16 tmp;
#pragma unroll
          for (int r = 0; r < 16; ++r) tmp[r] = 0.f;
          tmp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lb[0], la[io][0], tmp, 0, 0, 0);
          if (nk > 1) tmp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lb[1], la[io][1], tmp, 0, 0, 0);
          finish(io, j, tmp);
        }
      }
      return;
    }
    // ---- mode 1: LA from memory, natural k order (k slot = rank), one k16 step per 16 ranks and tap
    const bf16_t* LA = (const bf16_t*)p.lr_a;
    const int taps = p.lr_taps;
    int gn[NB], gy[NB], gx[NB];
    if (taps > 1) {
      const int hw = g.Ho * g.Wo;
#pragma unroll
      for (int io = 0; io < NB; ++io) {
        const int mm = rowg[io] < (unsigned)M ? (int)rowg[io] : 0;
        gn[io] = mm / hw;
        const int r = mm - gn[io] * hw;
        gy[io] = r / g.Wo;
        gx[io] = r - gy[io] * g.Wo;
      }
    }
    auto load_la = [&](int tap) {
      const int ky = taps > 1 ? tap / g.KW : 0, kx = taps > 1 ? tap - ky * g.KW : 0;
#pragma unroll
      for (int io = 0; io < NB; ++io) {
        bool v = rowg[io] < (unsigned)M;
        unsigned src = rowg[io];
        if (taps > 1) {                              // stride-1 same-size window of the A gather (launch_w8 checks)
          const int vy = gy[io] - g.py + ky, vx = gx[io] - g.px + kx;
          v = v && (unsigned)vy < (unsigned)g.Hv && (unsigned)vx < (unsigned)g.Wv;
          src = (unsigned)((gn[io] * g.Hv + vy) * g.Wv + vx);
        }
        // unconditional loads (a lane without a source row reads row 0, a k16 step beyond the rank re-reads step 0), zeroed by
        // value selects: predicated loads went out one at a time, each behind the exec-mask merge of the one before
        const unsigned srow = (v ? src : 0u) * (unsigned)p.lr_lda;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const bool kv = (ks < nk) & (16 * ks + 8 * half < rp);
          const bf16x8 x = *(const bf16x8*)(LA + srow + (kv ? 16 * ks + 8 * half : 0));
          la[io][ks] = (v & kv) ? x : zero8;
        }
      }
    };
    // LB rows of fragment j: columns beyond the tile's base columns read the last base column (their products are never stored)
    auto load_lb = [&](int j, int tap, bf16x8(&lb)[3]) {
      const int cl = min(col0_of(j) + l31, nbase - 1);
      const bf16_t* lbp = LB + (unsigned)(n0 + cl) * (unsigned)p.lr_ldb + tap * rkp + 8 * half;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) lb[ks] = *(const bf16x8*)(lbp + (ks < nk ? 16 * ks : 0));
    };
    if (direct) {                                    // no mask, unit scale: the products go straight into the accumulators
      for (int tap = 0; tap < taps; ++tap) {
        load_la(tap);
        bf16x8 lbq[NLB][3];
        if (LBALL) {
#pragma unroll
          for (int j = 0; j < FN; ++j) load_lb(j, tap, lbq[LBALL ? j : 0]);
        } else if (NLB == 2) {
          load_lb(0, tap, lbq[0]);
        }
        if (tap == 0) early();
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if (!LBALL && NLB == 2 && j + 1 < FN) load_lb(j + 1, tap, lbq[(j + 1) & 1]);
          if (NLB == 1) load_lb(j, tap, lbq[0]);
          if (col0_of(j) >= nbase) continue;
          const bf16x8(&lb)[3] = lbq[LBALL ? j : (j & (NLB - 1))];
#pragma unroll
          for (int io = 0; io < NB; ++io) {
            acc[io][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lb[0], la[io][0], acc[io][j], 0, 0, 0);
            if (nk > 1) acc[io][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lb[1], la[io][1], acc[io][j], 0, 0, 0);
            if (nk > 2) acc[io][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lb[2], la[io][2], acc[io][j], 0, 0, 0);
          }
        }
      }
    } else {                                         // masked / scaled: one tap (launch_w8 checks), fragment by fragment; the
      load_la(0);                                    // next fragment's LB rows are requested before this one's mask is hashed
      bf16x8 lbq[NLB][3];
      if (LBALL) {
#pragma unroll
        for (int j = 0; j < FN; ++j) load_lb(j, 0, lbq[LBALL ? j : 0]);
      } else if (NLB == 2) {
        load_lb(0, 0, lbq[0]);
      }
      early();
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if (!LBALL && NLB == 2 && j + 1 < FN) load_lb(j + 1, 0, lbq[(j + 1) & 1]);
        if (NLB == 1) load_lb(j, 0, lbq[0]);
        if (col0_of(j) >= nbase) continue;
        const bf16x8(&lb)[3] = lbq[LBALL ? j : (j & (NLB - 1))];
#pragma unroll
        for (int io = 0; io < NB; ++io) {
          f32x16 tmp;
#pragma unroll
          for (int r = 0; r < 16; ++r) tmp[r] = 0.f;
          tmp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lb[0], la[io][0], tmp, 0, 0, 0);
          if (nk > 1) tmp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lb[1], la[io][1], tmp, 0, 0, 0);
          if (nk > 2) tmp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lb[2], la[io][2], tmp, 0, 0, 0);
          finish(io, j, tmp);
        }
      }
    }
    }
  };

  // ---- register epilogue (every launch without column statistics): no staging through LDS.  The transposed accumulators give
  // a lane four consecutive columns per register quad; v_permlane32_swap between the quads q and q+1 of the two half-waves
  // leaves lane l < 32 with columns 8q .. 8q+7 and lane l + 32 with columns 8(q+1) .. 8(q+1)+7 of tile row (l & 31): the same
  // 8-column fp32 chunk the staged path hands to a thread, stored as 16 bytes (a wave's store covers 32 rows x 32 bytes).
  // The staged path cost two LDS passes at the 64-79 B/clk of LDS stores plus 2-3 workgroup barriers — 9000-17000 cycles of a
  // 20 us launch (profiles/r03_w8_timeline.txt); here the waves leave the K loop for their stores independently.
  //   K groups (KG = 2): the groups swap HALF of their partial sums through LDS (kg 0 finishes fragment band 0 of the wave
  //   tile, kg 1 band 1), lane-private slots, one barrier.
  //   split-K: the slabs hold the accumulator registers as they are (lane-private float4 slots, coalesced).
  constexpr int OWN = KG == 2 ? FM / 2 : FM;          // fragment bands a wave finishes
  constexpr bool DIRECT_OK = KG == 1 || (FM % 2 == 0 && 8 * OWN * FN * 4096 <= SMEM_BYTES);
  static_assert(CS || DIRECT_OK, "this configuration has no register epilogue");
  if constexpr (!CS) {
    {
      auto quad = [](const f32x16& a, int q) { return make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]); };
      const int ib0 = KG == 2 ? kg * OWN : 0;          // first band of the wave tile among this wave's own
      const bf16_t* Rp = (const bf16_t*)p.R;
      const bf16_t* rbp = (const bf16_t*)p.rowbias;
      const float* biasp = (const float*)p.bias;
      const int half = lane >> 5;
      // per-chunk operands, requested PD chunks ahead of their use.  Every request is an unconditional load: an absent operand
      // reads the zero page (offset mask 0), a lane outside the tile reads row / column 0 — with the loads inside `if (ok)` /
      // `if (bias)` branches and a `cur = nxt` hand-over the compiler drained vmcnt to 0 once per chunk, i.e. every chunk
      // waited for the PREVIOUS chunk's store to be acknowledged and for its own operands' round trip (round 6, found in the ISA)
      struct Pre {
        float4 b0, b1;
        bf16x8 rb, rv;
      };
      auto chunk_pos = [&](int io, int j, int qp, unsigned& row, int& ccol, bool& ok, bool& rk) {
        row = (unsigned)m0 + wr * TM + (ib0 + io) * 32 + (lane & 31);
        ccol = n0 + wc * TN + j * 32 + 8 * (qp + half);
        ok = row < (unsigned)M && (ccol - n0) < nbase;
        rk = p.n_split > 0 && ccol >= p.n_split;
      };
      const float* bias_q = biasp ? biasp : (const float*)g_w8_zero;
      const bf16_t* rb_q = rbp ? rbp : (const bf16_t*)g_w8_zero;
      const bf16_t* r_q = Rp ? Rp : (const bf16_t*)g_w8_zero;
      const unsigned bias_m = biasp ? ~0u : 0u, rb_m = rbp ? ~0u : 0u, r_m = Rp ? ~0u : 0u;
      unsigned rbrow[OWN];                               // row-bias row of a band's tile row (one division per band, not per chunk)
#pragma unroll
      for (int io = 0; io < OWN; ++io) {
        rbrow[io] = 0;
        if (rbp) rbrow[io] = min((unsigned)m0 + wr * TM + (ib0 + io) * 32 + (lane & 31), (unsigned)M - 1) / (unsigned)p.rows_per_rb * (unsigned)p.ldrb;
      }
      constexpr int NCH = 2 * OWN * FN, PD = 2;           // chunks; prefetch distance
      // chunks whose residual rows are requested right after the K loop (the K-group tiles have the registers for it; the
      // kernels without K groups spill with any)
      constexpr int NEARLY = (KG == 2 && LR != 0) ? (NCH < 8 ? NCH : 8) : 0;
      auto prefetch = [&](int ch, Pre& pf) {
        unsigned row;
        int ccol;
        bool ok, rk;
        const int io = (ch >> 1) / FN;
        chunk_pos(io, (ch >> 1) % FN, 2 * (ch & 1), row, ccol, ok, rk);
        const bool use = ok & !rk;
        const unsigned cu = use ? (unsigned)ccol : 0u, ru = use ? row : 0u;
        pf.b0 = *(const float4*)(bias_q + (cu & bias_m));
        pf.b1 = *(const float4*)(bias_q + ((cu & bias_m) + 4));
        pf.rb = *(const bf16x8*)(rb_q + ((rbrow[io] + cu) & rb_m));
        if (ch >= NEARLY) pf.rv = *(const bf16x8*)(r_q + ((ru * (unsigned)p.ldr + cu) & r_m));
      };
      Pre pre[PD + 1];
      bf16x8 rve[NEARLY > 0 ? NEARLY : 1];
      if (KG == 2 && role == 0 && LR == 0) prefetch(0, pre[0]);      // (in flight under the K-group exchange)
      if constexpr (KG == 2) {
        constexpr int PER = OWN * FN * 4 * 64;        // float4 slots per (receiving group, wave pair)
        float4* X = (float4*)smem;
        const int pair = wr + WM * wc;
        float4* xs = X + ((1 - kg) * 4 + pair) * PER + lane;
        const float4* xr = X + (kg * 4 + pair) * PER + lane;
        if (kg == 0) {
#pragma unroll
          for (int io = 0; io < OWN; ++io)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) xs[((io * FN + j) * 4 + q) * 64] = quad(acc[OWN + io][j], q);
        } else {
#pragma unroll
          for (int io = 0; io < OWN; ++io)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) xs[((io * FN + j) * 4 + q) * 64] = quad(acc[io][j], q);
        }
        lds_barrier();
#pragma unroll
        for (int io = 0; io < OWN; ++io)
#pragma unroll
          for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 x = xr[((io * FN + j) * 4 + q) * 64];
              f32x16& d = acc[io][j];
              if (kg == 0) {
                d[4 * q] += x.x; d[4 * q + 1] += x.y; d[4 * q + 2] += x.z; d[4 * q + 3] += x.w;
              } else {                                  // (kg 1 finishes band OWN + io: moved down, so that a wave's own bands
                const f32x16& o = acc[OWN + io][j];     //  are acc[0 .. OWN) from here on)
                d[4 * q] = o[4 * q] + x.x; d[4 * q + 1] = o[4 * q + 1] + x.y; d[4 * q + 2] = o[4 * q + 2] + x.z; d[4 * q + 3] = o[4 * q + 3] + x.w;
              }
            }
      }
      if ((dbg & 8)) tl[5] = __builtin_readcyclecounter();
      // the residual rows of the first NEARLY chunks are requested from inside the rank phase, as soon as its own operand loads
      // are out (`early`, see rank_phase): the launches of a level move in lock-step, so every workgroup of the chip asks for its
      // residual tile at the same moment and the chunk loop ran at HBM speed (profiles/r06_w8_epilogue_timeline_before.txt: 7 900
      // cycles with a residual, 5 000 without); issued there the rows arrive under the phase's LDS / MFMA / VALU work.  (Requested
      // before the K-group exchange, or ahead of younger loads that the phase waits for, they only moved the wait.)
      auto early_rows = [&]() {
#pragma unroll
        for (int ch = 0; ch < NEARLY; ++ch) {
          unsigned row;
          int ccol;
          bool ok, rk;
          chunk_pos((ch >> 1) / FN, (ch >> 1) % FN, 2 * (ch & 1), row, ccol, ok, rk);
          const bool use = ok & !rk;
          rve[ch] = *(const bf16x8*)(r_q + ((((use ? row : 0u) * (unsigned)p.ldr) + (use ? (unsigned)ccol : 0u)) & r_m));
        }
      };
      if (role != 0) {
        // slab slot of (wave, own band io, j, quad q, lane): coalesced 1 KiB per wave instruction
        const long long slot0 = ((long long)wave * OWN * FN * 4) * 64 + lane;
        float4* sl = (float4*)slab0 + slot0;
        constexpr long long ZS = (long long)BM * BN / 4; // float4 per split slab
        if (writer) {
#pragma unroll
          for (int io = 0; io < OWN; ++io)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) sl[(long long)bz * ZS + ((io * FN + j) * 4 + q) * 64] = quad(acc[io][j], q);
        } else {
          // ordered sum (0 + slab_0 + ... + own at position bz + ... + slab_{S-1}: independent of which split came last), a few
          // fragments at a time so that one batch of loads is in flight per slab
          constexpr int NFR = OWN * FN;
          constexpr int GRP = FM * FN > 6 ? 1 : (NFR % 3 == 0 ? 3 : (NFR % 2 == 0 ? 2 : 1));
#pragma unroll
          for (int g0 = 0; g0 < NFR; g0 += GRP) {
            float4 t[GRP * 4];
#pragma unroll
            for (int x = 0; x < GRP * 4; ++x) t[x] = make_float4(0.f, 0.f, 0.f, 0.f);
            auto add_slab = [&](int z) {
              float4 ld[GRP * 4];
#pragma unroll
              for (int x = 0; x < GRP * 4; ++x) ld[x] = sl[(long long)z * ZS + ((g0 * 4) + x) * 64];
#pragma unroll
              for (int x = 0; x < GRP * 4; ++x) {
                t[x].x += ld[x].x; t[x].y += ld[x].y; t[x].z += ld[x].z; t[x].w += ld[x].w;
              }
            };
            // two slabs per round trip (round 6: the reducer's slab reads are dependent round trips to another XCD's L2 / the
            // Infinity Cache, ~1.5 us each; six to eight splits spent 10 us of a 27 us launch here) — same summation order
            auto add_slabs2 = [&](int z) {
              float4 la[GRP * 4], lb[GRP * 4];
#pragma unroll
              for (int x = 0; x < GRP * 4; ++x) la[x] = sl[(long long)z * ZS + ((g0 * 4) + x) * 64];
#pragma unroll
              for (int x = 0; x < GRP * 4; ++x) lb[x] = sl[(long long)(z + 1) * ZS + ((g0 * 4) + x) * 64];
#pragma unroll
              for (int x = 0; x < GRP * 4; ++x) {
                t[x].x += la[x].x; t[x].y += la[x].y; t[x].z += la[x].z; t[x].w += la[x].w;
              }
#pragma unroll
              for (int x = 0; x < GRP * 4; ++x) {
                t[x].x += lb[x].x; t[x].y += lb[x].y; t[x].z += lb[x].z; t[x].w += lb[x].w;
              }
            };
            constexpr bool PAIRS = KG == 2;          // (the K-group tiles hold 3 - 4 fragments per wave: registers to spare; the
                                                     //  128x384 / 256x256 kernels spilled with a second slab in flight)
            {
              int z = 0;
              if constexpr (PAIRS)
                for (; z + 1 < bz; z += 2) add_slabs2(z);
              for (; z < bz; ++z) add_slab(z);
            }
#pragma unroll
            for (int x = 0; x < GRP * 4; ++x) {
              const float4 own = quad(acc[(g0 + x / 4) / FN][(g0 + x / 4) % FN], x % 4);
              t[x].x += own.x; t[x].y += own.y; t[x].z += own.z; t[x].w += own.w;
            }
            {
              int z = bz + 1;
              if constexpr (PAIRS)
                for (; z + 1 < splits; z += 2) add_slabs2(z);
              for (; z < splits; ++z) add_slab(z);
            }
#pragma unroll
            for (int x = 0; x < GRP * 4; ++x) {
              f32x16& d = acc[(g0 + x / 4) / FN][(g0 + x / 4) % FN];
              const int q = x % 4;
              d[4 * q] = t[x].x; d[4 * q + 1] = t[x].y; d[4 * q + 2] = t[x].z; d[4 * q + 3] = t[x].w;
            }
          }
        }
      }
      if (!writer) {
        if constexpr (LR != 0) {
          // (after the K-group exchange and the split-K reduction: the accumulators are complete — mode 2 multiplies them)
          if (p.lr_mode != 0) rank_phase(ib0, std::integral_constant<int, OWN>{}, early_rows);
          else early_rows();
        } else {
          early_rows();
        }
        if (KG == 2 && (dbg & 8)) tl[6] = __builtin_readcyclecounter();       // (exchange / slab reduction / rank phase done)
        if (KG == 1 || role != 0 || LR != 0) prefetch(0, pre[0]);    // (the slab reduction / the rank phase need the registers)
#pragma unroll
        for (int c = 1; c < PD; ++c)
          if (c < NCH) prefetch(c, pre[c]);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {               // chunk = (fragment, quad pair)
          const int io = (ch >> 1) / FN, j = (ch >> 1) % FN, qp = 2 * (ch & 1);
          if (ch + PD < NCH) prefetch(ch + PD, pre[(ch + PD) % (PD + 1)]);
          Pre& cur = pre[ch % (PD + 1)];
          if (ch < NEARLY) cur.rv = rve[ch];
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {                // (every lane takes part in the swap: before any row / column guard)
            const float fa = acc[io][j][4 * qp + e], fb = acc[io][j][4 * (qp + 1) + e];     // (copies: a bit_cast straight from a
            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, fa),      //  vector element read element 0)
                                                             __builtin_bit_cast(unsigned, fb), false, false);
            const unsigned s0 = sw[0], s1 = sw[1];
            v[e] = __builtin_bit_cast(float, s0);
            v[4 + e] = __builtin_bit_cast(float, s1);
          }
          unsigned row;
          int ccol;
          bool ok, rk;
          chunk_pos(io, j, qp, row, ccol, ok, rk);
          // straight-line arithmetic (absent operands are zeros, lanes outside the tile compute and do not store); only the
          // stores sit behind the row / column guard (x + 0 differs from x only in the sign of an exact zero)
          float w[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = v[e] * alpha;
          const bf16x8 rkv = pack8bf(w);               // rank columns: second output block, alpha only
          w[0] += cur.b0.x; w[1] += cur.b0.y; w[2] += cur.b0.z; w[3] += cur.b0.w;
          w[4] += cur.b1.x; w[5] += cur.b1.y; w[6] += cur.b1.z; w[7] += cur.b1.w;
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] += bf2f((unsigned short)cur.rb[e]);
          if (act_silu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = silu_f(w[e]);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] += beta * bf2f((unsigned short)cur.rv[e]);
          const bf16x8 ov = pack8bf(w);
          if (ok) {
            if (rk) *(bf16x8*)((bf16_t*)p.D2 + row * (unsigned)p.ldd2 + (ccol - p.n_split)) = rkv;
            else *(bf16x8*)((bf16_t*)p.D + row * (unsigned)p.ldd + ccol) = ov;
          }
          if (KG == 1 && (dbg & 8) && ch == 1) tl[6] = __builtin_readcyclecounter();
        }
      }
    }
  } else {
  float cb[8];                                     // per-column additive terms: bias (+ the tile's row-bias row when it has one)
#pragma unroll
  for (int e = 0; e < 8; ++e) cb[e] = 0.f;
  const bf16_t* rowbias = nullptr;                 // non-null: row-bias differs inside the tile, looked up per row
  if (cact && !rankcol) {
    if (p.bias) {
      const float4 b0 = *(const float4*)((const float*)p.bias + col), b1 = *(const float4*)((const float*)p.bias + col + 4);
      cb[0] = b0.x; cb[1] = b0.y; cb[2] = b0.z; cb[3] = b0.w; cb[4] = b1.x; cb[5] = b1.y; cb[6] = b1.z; cb[7] = b1.w;
    }
    if (p.rowbias) {
      const unsigned rlo = (unsigned)m0 / (unsigned)p.rows_per_rb;
      const unsigned rhi = (unsigned)(min((long long)M, m0 + BM) - 1) / (unsigned)p.rows_per_rb;
      if (rlo == rhi) {
        const bf16x8 tb = *(const bf16x8*)((const bf16_t*)p.rowbias + rlo * (unsigned)p.ldrb + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) cb[e] += bf2f((unsigned short)tb[e]);
      } else {
        rowbias = (const bf16_t*)p.rowbias;
      }
    }
  }
  // ---- column statistics of the stored tile (T2VGemm.colsum, GroupNorm fusion): every thread accumulates its 8 columns over
  // the rows it writes; the RPI threads of a column chunk are combined in fixed order through LDS after the last pass
  if constexpr (LR != 0) {
    // staged path: the term is added to ONE of the partial sums that meet later (K group 0 of split 0); mode 2 needs complete
    // accumulators and therefore KG == 1 and a single split (launch_w8)
    if (p.lr_mode != 0 && (KG == 1 || kg == 0) && bz == 0) rank_phase(0, std::integral_constant<int, FM>{}, [] {});
  }
  CsState cst;
  cs_init(cst, p, cs_mode, cact && !rankcol, m0, col, Nb);
  const bf16_t* CX = (cs_mode == 2 && cact && !rankcol) ? (const bf16_t*)p.cs_x : nullptr;
  const bf16_t* R = (cact && !rankcol && !writer) ? (const bf16_t*)p.R : nullptr;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int ib = ps / WRP, h = ps % WRP;
    // residual chunks of this pass: requested now, consumed after the staging barriers
    bf16x8 rv[ITERS];                                // (cs_mode 2 excludes a residual: the same registers hold the x rows)
    // (one batch of unconditional loads: a thread's rows outside the pass / the matrix read the last valid row and are not used)
    if (CX && role != 1) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int rl = min(r0 + RPI * it, PROWS - 1);
        const unsigned row = min((unsigned)m0 + tile_row(ib, h, rl), (unsigned)M - 1);
        rv[it] = *(const bf16x8*)(CX + row * (unsigned)p.cs_ldx + col);
      }
    } else if (R) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int rl = min(r0 + RPI * it, PROWS - 1);
        const unsigned row = min((unsigned)m0 + tile_row(ib, h, rl), (unsigned)M - 1);
        rv[it] = *(const bf16x8*)(R + row * (unsigned)p.ldr + col);
      }
    }
    if (ps > 0) lds_barrier();                       // (LDS-only: the residual rows requested above stay in flight)
    const bool mine = (wr / WPP) == h;
    const int wrl = wr % WPP;
    if (mine && (KG == 1 || kg == 1)) {
#pragma unroll
      for (int fi = 0; fi < FG; ++fi)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int rl = (fi * WPP + wrl) * 32 + (lane & 31);
            const int cl = wc * TN + j * 32 + 8 * q + 4 * (lane >> 5);
            const f32x16& a = acc[ib * FG + fi][j];
            *(float4*)(sC + rl * SB + cl) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
          }
    }
    lds_barrier();
    if constexpr (KG == 2) {
      if (mine && kg == 0) {
#pragma unroll
        for (int fi = 0; fi < FG; ++fi)
#pragma unroll
          for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int rl = (fi * WPP + wrl) * 32 + (lane & 31);
              const int cl = wc * TN + j * 32 + 8 * q + 4 * (lane >> 5);
              const f32x16& a = acc[ib * FG + fi][j];
              float4 t = *(const float4*)(sC + rl * SB + cl);
              t.x += a[4 * q]; t.y += a[4 * q + 1]; t.z += a[4 * q + 2]; t.w += a[4 * q + 3];
              *(float4*)(sC + rl * SB + cl) = t;
            }
      }
      lds_barrier();
    }
    if ((dbg & 8) && ps == 0) tl[5] = __builtin_readcyclecounter();
    if (cact) {
      // all staged rows of this thread are read in one batch (the row guards below are divergent branches: with the reads
      // inside them every iteration paid its own LDS round trip)
      constexpr int CH = NPASS > 1 ? 4 : ITERS;      // batch size (a multi-pass tile still holds live accumulators)
#pragma unroll
      for (int it0 = 0; it0 < ITERS; it0 += CH) {
      float4 sv[CH][2];
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int rl = min(r0 + RPI * (it0 + q), PROWS - 1);
        sv[q][0] = *(const float4*)(sC + rl * SB + cc * 8);
        sv[q][1] = *(const float4*)(sC + rl * SB + cc * 8 + 4);
      }
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int it = it0 + q;
        if (it >= ITERS) continue;
        const int rl = r0 + RPI * it;
        const int trow = tile_row(ib, h, rl);
        const unsigned row = (unsigned)m0 + trow;
        if (rl >= PROWS || row >= (unsigned)M) continue;
        float v[8];
        {
          const float4 a = sv[q][0], b = sv[q][1];
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        if (role != 0) {
          float* sp = slab0 + (long long)trow * BN + cc * 8;
          if (writer) {
            float* wp = sp + (long long)bz * (BM * BN);
            *(float4*)wp = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(wp + 4) = make_float4(v[4], v[5], v[6], v[7]);
            continue;
          }
          float tot[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) tot[e] = 0.f;
          for (int z = 0; z < splits; ++z) {           // fixed order; this workgroup's partial takes its own place
            if (z == bz) {
#pragma unroll
              for (int e = 0; e < 8; ++e) tot[e] += v[e];
            } else {
              const float4 a = *(const float4*)(sp + (long long)z * (BM * BN));
              const float4 b = *(const float4*)(sp + (long long)z * (BM * BN) + 4);
              tot[0] += a.x; tot[1] += a.y; tot[2] += a.z; tot[3] += a.w; tot[4] += b.x; tot[5] += b.y; tot[6] += b.z; tot[7] += b.w;
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = tot[e];
        }
        if (alpha != 1.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= alpha;
        }
        if (rankcol) {                               // rank columns: second output block, alpha only
          *(bf16x8*)((bf16_t*)p.D2 + row * (unsigned)p.ldd2 + (col - p.n_split)) = pack8bf(v);
          continue;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += cb[e];
        if (rowbias) {
          const bf16x8 tb = *(const bf16x8*)(rowbias + (row / (unsigned)p.rows_per_rb) * (unsigned)p.ldrb + col);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf2f((unsigned short)tb[e]);
        }
        if (act_silu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        }
        if (R && !CX) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += beta * bf2f((unsigned short)rv[it][e]);
        }
        const bf16x8 ov = pack8bf(v);
        *(bf16x8*)((bf16_t*)p.D + row * (unsigned)p.ldd + col) = ov;
        if (cs_mode != 0) cs_add(cst, cs_mode, ov, rv[it], p.cs_silu, row, col, cs_mode == 2 && p.cs_drop_p > 0.f);
      }
      }
    }
    if ((dbg & 8) && ps == 0) tl[6] = __builtin_readcyclecounter();
  }
  if (cs_mode != 0 && role != 1) {
    __syncthreads();                                 // staging buffer free: [RPI][BN][2] partials of the column chunks
    cs_flush(cst, p, (float*)smem, cact && !rankcol, r0, RPI, cc, BN, tid, NT, n0, nbase, Nb, tm, BM);
  }
  }
  if (dbg & 8) tl[7] = __builtin_readcyclecounter();
  if (writer) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's slab rows have left
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the compiler may drop the fence's own wait, guide G16)
      __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if ((dbg & 8) && p.workspace && tid == 0 && splits == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the stores of this wave have left
    tl[4] = __builtin_readcyclecounter();
    unsigned long long* o = (unsigned long long*)((unsigned char*)p.workspace + 65536) + (long long)blockIdx.x * 8;   // (behind the split-K counters)
    for (int q = 0; q < 8; ++q) o[q] = tl[q];
  }
}

template <int BM, int BN, int WM, int WN, int KG, int NSTAGE, int SCHED, int BKT, bool CS, int LR = 0>
int launch_w8(const T2VGemm& p, int nstep, int splits, hipStream_t s) {
  constexpr int RING = NSTAGE * (BM + BN) * BKT * 2;
  constexpr int EPI = EpiPlan<BN, WM, BM / WM / 32>::BYTES;
  // (LR = 2: + the rank waves' keep-bit plane slots, see MASK_OFF in the kernel)
  constexpr int SMEM = (RING > EPI ? RING : EPI) + 16 + (LR == 2 ? WM * KG * NSTAGE * (BM / WM) * 8 : 0);
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  T2V_CHECK_ARG(!(p.colsum && p.cs_mode == 2) || (!p.R && p.cs_x && p.cs_sums && p.cs_gamma && p.cs_beta && p.cs_G > 0 &&
                                                   p.cs_domain_rows % BM == 0 && p.cs_ldx % 8 == 0),
                "t2v_gemm_w8: colsum mode 2 needs x / sums / gamma / beta, no residual, and tiles inside a domain");
  T2V_CHECK_ARG(p.n_split <= 0 || p.n_split % (64 / (BKT / 8)) == 0,
                "t2v_gemm_w8: n_split=%d must be a multiple of %d for this configuration", p.n_split, 64 / (BKT / 8));
  const bool lr2 = LR == 1 && p.lr_mode == 2;
  const bool lr3 = LR == 2;
  if constexpr (LR == 2) {
    T2V_CHECK_ARG(p.lr_mode == 3, "t2v_gemm_w8: lr_mode %d on a mode-3 kernel", p.lr_mode);
    T2V_CHECK_ARG(p.lr_rp >= 8 && p.lr_rp <= 64 && p.lr_rp % 8 == 0 && p.lr_b && p.lr_ldb % 8 == 0 && p.alpha == 1.f && p.lr_scale == 1.f,
                  "t2v_gemm_w8: lr_mode 3 needs a padded rank of 8..64, lr_b (scale folded in) and alpha == lr_scale == 1");
    T2V_CHECK_ARG(p.a_mode == T2V_A_DENSE && p.n_split <= 0 && p.B2 && p.D2 && p.ldb2 % 8 == 0 && p.ldd2 % 8 == 0 && p.ldd2 >= p.lr_rp &&
                      p.lr_taps <= 1 && p.b2_klen <= 0,
                  "t2v_gemm_w8: lr_mode 3 takes a dense A, the up factor (rank-major) in B2 and dt in D2, n_split = 0");
    T2V_CHECK_ARG(p.lr_drop_p >= 0.f && p.lr_drop_p < 1.f, "t2v_gemm_w8: lr_drop_p must be in [0, 1)");
    T2V_CHECK_ARG(p.lr_group_cols == 0 || (p.lr_group_cols % 64 == 0 && p.K % p.lr_group_cols == 0 && p.K / p.lr_group_cols <= 3),
                  "t2v_gemm_w8: lr_group_cols (mode 3: the members' K width) must divide K into at most 3 members of whole stages");
    T2V_CHECK_ARG((long long)p.M * (p.lr_group_cols > 0 ? p.lr_group_cols : p.K) < (1ll << 34), "t2v_gemm_w8: mask matrix beyond 2^34 elements");
    T2V_CHECK_ARG((long long)p.N * p.lr_ldb < 0x7ff00000ll && (long long)p.lr_rp * p.ldb2 * 2 < 0x7ff00000ll && (long long)p.M * p.ldd2 < 0x7ff00000ll,
                  "t2v_gemm_w8: lr_mode 3 offsets");
    T2V_CHECK_ARG(KG == 1 || !CS, "t2v_gemm_w8: lr_mode 3 with column statistics needs a configuration without K groups");
    T2V_CHECK_ARG(BN - 32 * ((p.lr_rp + 31) / 32) >= 32, "t2v_gemm_w8: tile too narrow for the rank fragments");
    T2V_CHECK_ARG(!p.lr_plane || ((p.lr_group_cols > 0 ? p.lr_group_cols : p.K) % 64 == 0 && ((uintptr_t)p.lr_plane & 3) == 0 &&
                                  (long long)p.M * p.K / 8 < 0x7ff00000ll),
                  "t2v_gemm_w8: lr_plane (mode 3) needs members of whole 64-column stages and a plane below 2 GiB");
    if (CS) splits = 1;
  } else if constexpr (LR == 1) {
    T2V_CHECK_ARG(p.lr_mode == 1 || p.lr_mode == 2, "t2v_gemm_w8: lr_mode %d", p.lr_mode);
    T2V_CHECK_ARG(p.lr_rp >= 8 && p.lr_rp <= (p.lr_mode == 1 ? 48 : 32) && p.lr_rp % 8 == 0 && p.lr_b && p.lr_ldb % 8 == 0 && p.alpha == 1.f,
                  "t2v_gemm_w8: rank-wide epilogue term needs a padded rank of 8..32 (mode 1: ..48), lr_b and alpha == 1");
    T2V_CHECK_ARG(p.lr_group_cols == 0 || (p.lr_mode == 2 && p.lr_group_cols % 32 == 0 && p.N % p.lr_group_cols == 0 && p.N / p.lr_group_cols <= 3),
                  "t2v_gemm_w8: lr_group_cols (mode 2 only) must divide N into at most 3 members of whole fragments");
    T2V_CHECK_ARG(p.lr_drop_p >= 0.f && p.lr_drop_p < 1.f, "t2v_gemm_w8: lr_drop_p must be in [0, 1)");
    T2V_CHECK_ARG((long long)p.N * p.lr_ldb < 0x7ff00000ll, "t2v_gemm_w8: lr_b too large for 32-bit offsets");
    T2V_CHECK_ARG(p.lr_drop_p == 0.f || ((p.lr_group_cols > 0 ? p.lr_group_cols : p.N) % 8 == 0 &&
                                         (long long)p.M * (p.lr_group_cols > 0 ? p.lr_group_cols : p.N) < (1ll << 34)),
                  "t2v_gemm_w8: the mask matrix of a dropped rank-wide term must have a width of whole 8-column chunks and < 2^34 elements");
    if (p.lr_mode == 1) {
      T2V_CHECK_ARG(p.lr_a && p.lr_lda % 8 == 0 && p.lr_lda >= p.lr_rp && p.lr_taps >= 1, "t2v_gemm_w8: lr_mode 1 needs lr_a [rows, lr_lda >= lr_rp]");
      T2V_CHECK_ARG((long long)p.M * p.lr_lda < 0x7ff00000ll, "t2v_gemm_w8: lr_a too large for 32-bit offsets");
      if (p.lr_taps > 1) {
        const T2VConvGeom& g = p.geom;
        T2V_CHECK_ARG(p.a_mode == T2V_A_CONV && p.lr_taps == g.KH * g.KW && g.sy == 1 && g.sx == 1 && g.tdiv == 1 && g.up == 0 &&
                          g.Hv == g.Ho && g.Wv == g.Wo,
                      "t2v_gemm_w8: a windowed rank-wide term follows the A gather's own stride-1 same-size window");
        T2V_CHECK_ARG(p.lr_drop_p == 0.f && p.lr_scale == 1.f, "t2v_gemm_w8: a windowed rank-wide term takes no mask / scale (fold the scale into lr_b)");
      }
    } else {
      T2V_CHECK_ARG(KG == 1 || !CS, "t2v_gemm_w8: lr_mode 2 with column statistics needs a configuration without K groups");
      T2V_CHECK_ARG(p.n_split <= 0 && p.B2 && p.D2 && p.ldb2 % 8 == 0 && p.ldd2 % 8 == 0 && p.ldd2 >= p.lr_rp && p.lr_taps <= 1 && p.b2_klen <= 0,
                    "t2v_gemm_w8: lr_mode 2 takes the down factor in B2 / D2 with n_split = 0");
      T2V_CHECK_ARG((long long)p.lr_rp * 3 * p.ldb2 * 2 < 0x7ff00000ll && (long long)p.M * p.ldd2 < 0x7ff00000ll, "t2v_gemm_w8: lr_mode 2 offsets");
      T2V_CHECK_ARG(!p.lr_plane || ((p.lr_group_cols > 0 ? p.lr_group_cols : p.N) % 32 == 0 && ((uintptr_t)p.lr_plane & 3) == 0),
                    "t2v_gemm_w8: lr_plane (mode 2) needs an output width of whole 32-column fragments");
      if (CS) splits = 1;                          // the staged epilogue reduces the splits after the rank phase (t2v_gemm never
                                                   // pairs the two: colsum_bm() answers 0 for a split configuration)
    }
  } else {
    T2V_CHECK_ARG(p.lr_mode == 0, "t2v_gemm_w8: this configuration has no rank-wide epilogue term (lr_mode %d)", p.lr_mode);
  }
  auto kern = gemm_w8_kernel<BM, BN, WM, WN, KG, NSTAGE, SCHED, BKT, CS, LR>;
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
  if (lr3) {                                       // every tile: up to nstep base columns; the rank fragments sit at the tile's end
    const int lim = BN - 32 * ((p.lr_rp + 31) / 32);
    if (nstep > lim) nstep = lim;
    ntn = (p.N + nstep - 1) / nstep;
  } else if (lr2) {                                // every tile: up to nstep base columns + one fragment of rank columns
    if (nstep > BN - 32) nstep = BN - 32;
    if (p.lr_group_cols > 0)                       // a tile never straddles two members of a projection group
      while (p.lr_group_cols % nstep != 0) nstep -= 32;
    ntn = (p.N + nstep - 1) / nstep;
  } else {
    while ((long long)(ntn - 1) * nstep + BN < p.N) ++ntn;      // the last tile takes up to BN columns
  }
  // K splits: every split gets at least NSTAGE stages; the slabs + counters must fit the caller's scratch (first 64 KB =
  // counters, zero when the scratch is first handed over; the kernels leave them zero)
  const int nt_all = p.K / BKT;
  if (splits < 1) splits = 1;
  while (splits > 1 && (nt_all / splits < NSTAGE || !p.workspace || (long long)ntm * ntn > 8000 ||
                        65536ll + (long long)ntm * ntn * splits * BM * BN * 4 > (long long)p.workspace_bytes))
    --splits;
  static const int dbg = [] { const char* e = getenv("T2V_W8_DBG"); return e ? atoi(e) : 0; }();
  static const int force_raster = [] { const char* e = getenv("T2V_W8_RASTER"); return e ? atoi(e) : -1; }();   // A/B switch
  T2VGemm q = p;
  {
    const double bytesA = (double)p.M * (p.a_mode == T2V_A_CONV ? p.geom.C : p.K) * 2.0;      // unique activation bytes
    const double bytesB = (double)p.N * p.K * 2.0;
    q.raster_n = force_raster >= 0 ? force_raster : (bytesB + 8.0 * bytesA < bytesA + 8.0 * bytesB ? 1 : 0);
  }
  T2V_LAUNCH(kern, dim3(ntm * ntn * splits), dim3(512), SMEM, s, q, nstep, ntn, splits, dbg);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

}  // namespace

// Number of W8 configurations and a pinned-configuration launch (tile table / tuning / diagnostics).  The caller (gemm.hip)
// has already validated the descriptor and checked that the lean loader and the bf16 epilogue apply.
#if W8_PART == 0
extern "C" int t2v_gemm_w8_configs(void) { return 23; }
// tile rows of a configuration (t2v_gemm_colsum_rows)
int t2v_gemm_w8_bm(int cfg) {
  static const int bm[] = {128, 128, 256, 128, 128, 256, 128, 256, 256, 128, 256, 128, 128, 256, 128, 128, 128, 128, 256, 128, 128, 128, 128};
  return cfg >= 0 && cfg < (int)(sizeof(bm) / sizeof(bm[0])) ? bm[cfg] : 0;
}
#endif
// The instantiations are spread over three translation units (build_ext.py compiles this file with -DW8_PART=0 / 1 / 2: the
// plain kernels + the entry points / the LR = 1 kernels (modes 1, 2) / the LR = 2 kernels (mode 3)), so that a rebuild takes the
// time of the largest third.
int t2v_gemm_w8_launch_plain(const T2VGemm& p, int cfg, int nstep, int splits, bool staged, hipStream_t s);
int t2v_gemm_w8_launch_lr(const T2VGemm& p, int cfg, int nstep, int splits, bool staged, hipStream_t s);
int t2v_gemm_w8_launch_lr3(const T2VGemm& p, int cfg, int nstep, int splits, bool staged, hipStream_t s);
#if W8_PART == 0
int t2v_gemm_w8_launch(const T2VGemm& p, int cfg, int nstep, int splits, hipStream_t s) {
  // the staged-epilogue kernels serve the launches that ask for column statistics (and T2V_W8_STAGED=1: A/B runs, tests)
  static const bool force_staged = [] { const char* e = getenv("T2V_W8_STAGED"); return e && atoi(e) != 0; }();
  const bool staged = p.colsum != nullptr || force_staged;
  if (p.lr_mode == 3) return t2v_gemm_w8_launch_lr3(p, cfg, nstep, splits, staged, s);
  if (p.lr_mode != 0) return t2v_gemm_w8_launch_lr(p, cfg, nstep, splits, staged, s);
  return t2v_gemm_w8_launch_plain(p, cfg, nstep, splits, staged, s);
}
int t2v_gemm_w8_launch_plain(const T2VGemm& p, int cfg, int nstep, int splits, bool staged, hipStream_t s) {
  switch (cfg) {
    //                       BM   BN  WM WN KG NS SCHED BK
    case 0: return launch_w8<128, 384, 2, 2, 2, 2, 0, 64, true>(p, nstep, splits, s);      // wave 64x192, K groups, classic ring
    case 1: return launch_w8<128, 384, 4, 2, 1, 2, 0, 64, true>(p, nstep, splits, s);      // wave 32x192
    case 2: return launch_w8<256, 256, 4, 2, 1, 2, 0, 64, true>(p, nstep, splits, s);      // wave 64x128
    case 3: return launch_w8<128, 192, 2, 2, 2, 3, 0, 64, true>(p, nstep, splits, s);      // wave 64x96, K groups, 3 stages
    case 4: return launch_w8<128, 256, 2, 2, 2, 3, 0, 64, true>(p, nstep, splits, s);      // wave 64x128, K groups, 3 stages
    // ping-pong, fragments read in the M phase
    case 5: return launch_w8<256, 256, 4, 2, 1, 2, 2, 64, true>(p, nstep, splits, s);
    case 6: return launch_w8<128, 384, 2, 2, 2, 2, 2, 64, true>(p, nstep, splits, s);
    // ping-pong with prefetched fragments, 32-deep stages
    case 7: return launch_w8<256, 256, 4, 2, 1, 4, 3, 32, true>(p, nstep, splits, s);      // wave 64x128, 4 x 32 KB
    case 8: return launch_w8<256, 384, 4, 2, 1, 3, 3, 32, true>(p, nstep, splits, s);      // wave 64x192, 3 x 40 KB (register-bound: spills)
    case 9: return launch_w8<128, 384, 4, 2, 1, 4, 3, 32, true>(p, nstep, splits, s);      // wave 32x192, 4 x 32 KB
    case 10: return launch_w8<256, 128, 4, 2, 1, 4, 3, 32, true>(p, nstep, splits, s);     // wave 64x64, 4 x 24 KB
    case 11: return launch_w8<128, 256, 2, 4, 1, 4, 3, 32, true>(p, nstep, splits, s);     // wave 64x64, 4 x 24 KB
    // classic ring with the refill pieces spread over the k16 steps
    case 12: return staged ? launch_w8<128, 384, 4, 2, 1, 2, 4, 64, true>(p, nstep, splits, s) : launch_w8<128, 384, 4, 2, 1, 2, 4, 64, false>(p, nstep, splits, s);
    case 13: return staged ? launch_w8<256, 256, 4, 2, 1, 2, 4, 64, true>(p, nstep, splits, s) : launch_w8<256, 256, 4, 2, 1, 2, 4, 64, false>(p, nstep, splits, s);
    case 14: return staged ? launch_w8<128, 192, 2, 2, 2, 3, 4, 64, true>(p, nstep, splits, s) : launch_w8<128, 192, 2, 2, 2, 3, 4, 64, false>(p, nstep, splits, s);
    case 15: return launch_w8<128, 384, 2, 2, 2, 2, 4, 64, true>(p, nstep, splits, s);
    case 16: return staged ? launch_w8<128, 256, 2, 2, 2, 3, 4, 64, true>(p, nstep, splits, s) : launch_w8<128, 256, 2, 2, 2, 3, 4, 64, false>(p, nstep, splits, s);
    // software-pipelined classic ring (fragments one k16 step ahead, barrier before the last step, refill in portions)
    case 17: return staged ? launch_w8<128, 384, 4, 2, 1, 2, 5, 64, true>(p, nstep, splits, s) : launch_w8<128, 384, 4, 2, 1, 2, 5, 64, false>(p, nstep, splits, s);
    case 18: return staged ? launch_w8<256, 256, 4, 2, 1, 2, 5, 64, true>(p, nstep, splits, s) : launch_w8<256, 256, 4, 2, 1, 2, 5, 64, false>(p, nstep, splits, s);
    case 19: return staged ? launch_w8<128, 192, 2, 2, 2, 3, 5, 64, true>(p, nstep, splits, s) : launch_w8<128, 192, 2, 2, 2, 3, 5, 64, false>(p, nstep, splits, s);
    case 20: return staged ? launch_w8<128, 256, 2, 2, 2, 3, 5, 64, true>(p, nstep, splits, s) : launch_w8<128, 256, 2, 2, 2, 3, 5, 64, false>(p, nstep, splits, s);
    // 128x384 with the waves 2 x 4 (wave tile 64x96 instead of 32x192: 5 instead of 7 fragment reads per 6 MFMAs — the
    // 32x192 arrangement needs more LDS bandwidth per k16 step (56 KB of fragments + 16 KB of DMA) than its MFMAs take)
    case 21: return staged ? launch_w8<128, 384, 2, 4, 1, 2, 4, 64, true>(p, nstep, splits, s) : launch_w8<128, 384, 2, 4, 1, 2, 4, 64, false>(p, nstep, splits, s);
    case 22: return staged ? launch_w8<128, 384, 2, 4, 1, 2, 5, 64, true>(p, nstep, splits, s) : launch_w8<128, 384, 2, 4, 1, 2, 5, 64, false>(p, nstep, splits, s);
    // (four 32-deep stages under the classic spread-refill ring — three stages in flight instead of one — were measured as
    //  configurations 23-26 and removed: within +-5 % of the two-stage 64-deep rings on every signature,
    //  profiles/r03_w8_ablation.txt)
    default: t2v_set_error("t2v_gemm_w8: unknown configuration %d", cfg); return T2V_EINVAL;
  }
}
#elif W8_PART == 1
int t2v_gemm_w8_launch_lr(const T2VGemm& p, int cfg, int nstep, int splits, bool staged, hipStream_t s) {   // rank-wide epilogue term: modes 1, 2
    switch (cfg) {
      case 12: return staged ? launch_w8<128, 384, 4, 2, 1, 2, 4, 64, true, 1>(p, nstep, splits, s) : launch_w8<128, 384, 4, 2, 1, 2, 4, 64, false, 1>(p, nstep, splits, s);
      case 17: return staged ? launch_w8<128, 384, 4, 2, 1, 2, 5, 64, true, 1>(p, nstep, splits, s) : launch_w8<128, 384, 4, 2, 1, 2, 5, 64, false, 1>(p, nstep, splits, s);
      case 21: return staged ? launch_w8<128, 384, 2, 4, 1, 2, 4, 64, true, 1>(p, nstep, splits, s) : launch_w8<128, 384, 2, 4, 1, 2, 4, 64, false, 1>(p, nstep, splits, s);
      case 22: return staged ? launch_w8<128, 384, 2, 4, 1, 2, 5, 64, true, 1>(p, nstep, splits, s) : launch_w8<128, 384, 2, 4, 1, 2, 5, 64, false, 1>(p, nstep, splits, s);
      // K groups: the register epilogue runs the rank phase after the groups have met (mode 2 too); the staged one adds a
      // mode-1 term to group 0's partial sums
      case 14: return staged ? launch_w8<128, 192, 2, 2, 2, 3, 4, 64, true, 1>(p, nstep, splits, s) : launch_w8<128, 192, 2, 2, 2, 3, 4, 64, false, 1>(p, nstep, splits, s);
      case 16: return staged ? launch_w8<128, 256, 2, 2, 2, 3, 4, 64, true, 1>(p, nstep, splits, s) : launch_w8<128, 256, 2, 2, 2, 3, 4, 64, false, 1>(p, nstep, splits, s);
      case 19: return staged ? launch_w8<128, 192, 2, 2, 2, 3, 5, 64, true, 1>(p, nstep, splits, s) : launch_w8<128, 192, 2, 2, 2, 3, 5, 64, false, 1>(p, nstep, splits, s);
      case 20: return staged ? launch_w8<128, 256, 2, 2, 2, 3, 5, 64, true, 1>(p, nstep, splits, s) : launch_w8<128, 256, 2, 2, 2, 3, 5, 64, false, 1>(p, nstep, splits, s);
      default: t2v_set_error("t2v_gemm_w8: configuration %d has no rank-wide epilogue term (12, 14, 16, 17, 19-22 do; the 256x256 LR kernels spilled and were removed in round 6)", cfg); return T2V_EINVAL;
    }
}
#else
int t2v_gemm_w8_launch_lr3(const T2VGemm& p, int cfg, int nstep, int splits, bool staged, hipStream_t s) {  // dt computed by the launch itself (mode 3)
    switch (cfg) {
      case 12: return staged ? launch_w8<128, 384, 4, 2, 1, 2, 4, 64, true, 2>(p, nstep, splits, s) : launch_w8<128, 384, 4, 2, 1, 2, 4, 64, false, 2>(p, nstep, splits, s);
      case 17: return staged ? launch_w8<128, 384, 4, 2, 1, 2, 5, 64, true, 2>(p, nstep, splits, s) : launch_w8<128, 384, 4, 2, 1, 2, 5, 64, false, 2>(p, nstep, splits, s);
      case 21: return staged ? launch_w8<128, 384, 2, 4, 1, 2, 4, 64, true, 2>(p, nstep, splits, s) : launch_w8<128, 384, 2, 4, 1, 2, 4, 64, false, 2>(p, nstep, splits, s);
      case 22: return staged ? launch_w8<128, 384, 2, 4, 1, 2, 5, 64, true, 2>(p, nstep, splits, s) : launch_w8<128, 384, 2, 4, 1, 2, 5, 64, false, 2>(p, nstep, splits, s);
      case 14: T2V_CHECK_ARG(!staged, "t2v_gemm_w8: lr_mode 3 with column statistics needs a KG = 1 configuration"); return launch_w8<128, 192, 2, 2, 2, 3, 4, 64, false, 2>(p, nstep, splits, s);
      case 16: T2V_CHECK_ARG(!staged, "t2v_gemm_w8: lr_mode 3 with column statistics needs a KG = 1 configuration"); return launch_w8<128, 256, 2, 2, 2, 3, 4, 64, false, 2>(p, nstep, splits, s);
      case 19: T2V_CHECK_ARG(!staged, "t2v_gemm_w8: lr_mode 3 with column statistics needs a KG = 1 configuration"); return launch_w8<128, 192, 2, 2, 2, 3, 5, 64, false, 2>(p, nstep, splits, s);
      case 20: T2V_CHECK_ARG(!staged, "t2v_gemm_w8: lr_mode 3 with column statistics needs a KG = 1 configuration"); return launch_w8<128, 256, 2, 2, 2, 3, 5, 64, false, 2>(p, nstep, splits, s);
      default: t2v_set_error("t2v_gemm_w8: configuration %d has no lr_mode 3 instantiation (12, 14, 16, 17, 19-22 do)", cfg); return T2V_EINVAL;
    }
}
#endif
