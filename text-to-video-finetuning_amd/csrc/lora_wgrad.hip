// LoRA factor gradients of one layer in one streaming launch (t2v_lora_wgrad, include/t2v_abi.h).
//
// Both gradients are K-major contractions over the activation rows with a rank-wide (<=16 per pass) left operand:
//   out[16, taps, C] += alpha * sum_rows S[src(row, tap), 0:16]^T * Big[row, 0:C]
// (dU: S = t, Big = dy, taps = 1;  dD: S = dt, Big = x, taps = KH*KW).  The window of a conv is applied to the narrow
// operand S, so the wide operand is read exactly once: HBM-bound on `Big`, MFMA work = the algorithmic 2*16*taps*C*rows.
//
// Workgroup = 4 wave64; tile = 64 rows x 64 columns of Big per step, wave w owns columns [16w, 16w+16) for every tap.
// v_mfma_f32_16x16x32_bf16: A = S^T (16 x 32 rows), B = Big (32 rows x 16 cols).  Both operands are K-major in memory,
// so both go through LDS and ds_read_tr16_b64; the contraction index is permuted (lane group g, element j <-> row
// 16*(j>>2) + 4g + (j&3)) so that one transposed read touches 16 consecutive staged rows (conflict-free with the
// 160-byte row pitch).  Next step's global loads are issued into registers before the current step's MFMAs.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"

namespace {

constexpr int KR = 64;      // rows staged per step (two k=32 MFMA blocks)
constexpr int PITCH = 160;  // bytes per staged row of the wide tile: 128 data + 32 pad
constexpr int SP = 32;      // bytes per staged row of a narrow tile (16 bf16)

struct Prob {
  const bf16_t* big;
  long long ldbig;
  int C;
  const bf16_t* s;
  long long lds;
  float* out;
  long long ldo;
  int tiles;
  // dropout mask on the wide operand (the dU problem of a layer whose LoRA branch is dropped: Big = mask (.) dy / (1-p),
  // utils/lora.py:49,119; the mask is regenerated from (seed, row * C + column), never stored): mp > 0 switches it on
  float mp;
  unsigned long long mseed;
  const unsigned long long* mepoch;
};

struct Args {
  Prob u, d;
  long long rows;
  int chunk_rows;
  int rk;  // valid rank columns in this pass (8 or 16)
  int conv;
  T2VConvGeom g;
  float alpha;
};

template <int TAPS>
__device__ __forceinline__ void wgrad_body(const Prob& P, long long r_begin, long long r_end, int c0, const T2VConvGeom& g,
                                           int rk, float alpha, unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned char* sBig = smem;
  unsigned char* sS = smem + KR * PITCH;
  constexpr int NS = (TAPS * 128 + 255) / 256;
  f32x4 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 rbA[2], rsA[NS], rbB[2], rsB[NS];          // two steps of operands in flight (round 6, see the loop below)
  const bool masked = P.mp > 0.f;
  const DropKey dkey = drop_key(masked ? eff_seed(P.mseed, P.mepoch) : 0ull, P.mp);
  if (masked) alpha *= 1.f / (1.f - P.mp);         // the kept elements' 1/(1-p) rides in the output scale: the operand is a bit select
  const int srow = (tid & 127) >> 1, shalf = tid & 1, stap0 = tid >> 7;
  const unsigned hw = (unsigned)(g.Hv * g.Wv);

  auto fetch = [&](long long row0, bf16x8 (&rb)[2], bf16x8 (&rs)[NS]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + 256 * i, row = id >> 3, cc = id & 7;
      const long long gr = row0 + row;
      const int col = c0 + cc * 8;
      rb[i] = (gr < r_end && col < P.C) ? *(const bf16x8*)(P.big + gr * P.ldbig + col) : zero8;
      if (masked && gr < r_end && col < P.C) {
        const unsigned kb = drop_bits8(dkey, (unsigned long long)gr * P.C + col);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (!((kb >> e) & 1u)) rb[i][e] = 0;
      }
    }
    const long long q = row0 + srow;
    const bool okq = q < r_end && shalf * 8 < rk;
    int n = 0, iy = 0, ix = 0;
    if (TAPS > 1) {
      const unsigned uq = (unsigned)q;
      n = (int)(uq / hw);
      const unsigned rem = uq - (unsigned)n * hw;
      iy = (int)(rem / (unsigned)g.Wv);
      ix = (int)(rem - (unsigned)iy * (unsigned)g.Wv);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int tap = stap0 + 2 * i;
      if (tap < TAPS) {
        long long src = q;
        bool v = okq;
        if (TAPS > 1) {
          const int ky = tap / g.KW, kx = tap - ky * g.KW;
          const int oy = iy - ky + g.py, ox = ix - kx + g.px;
          v = v && (unsigned)oy < (unsigned)g.Ho && (unsigned)ox < (unsigned)g.Wo;
          src = ((long long)n * g.Ho + oy) * g.Wo + ox;
        }
        rs[i] = v ? *(const bf16x8*)(P.s + src * P.lds + shalf * 8) : zero8;
      }
    }
  };
  auto stash = [&](const bf16x8 (&rb)[2], const bf16x8 (&rs)[NS]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + 256 * i, row = id >> 3, cc = id & 7;
      *(bf16x8*)(sBig + row * PITCH + cc * 16) = rb[i];
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int tap = stap0 + 2 * i;
      if (tap < TAPS) *(bf16x8*)(sS + (tap * KR + srow) * SP + shalf * 16) = rs[i];
    }
  };
  const int gq = lane >> 4, li = lane & 15, jq = li >> 2, q4 = li & 3;
  auto compute = [&]() {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int rbase = 32 * h2 + 4 * gq + jq;
      const unsigned char* pb = sBig + rbase * PITCH + w * 32 + q4 * 8;
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)pb);
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(pb + 16 * PITCH));
      const bf16x8 b = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const unsigned char* pa = sS + (t * KR + rbase) * SP + q4 * 8;
        bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)pa);
        bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(pa + 16 * SP));
        const bf16x8 a = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
      }
    }
  };

  // Two steps of global loads stay in flight per workgroup (round 6).  With one (rounds 1 - 5) a workgroup had 8 KB of the wide
  // operand outstanding, ~40 KB per CU at the five workgroups the LDS footprint admits: 10 MB chip-wide against the ~16 MB that
  // 8 TB/s x ~2 us of latency asks for — the batched launches of a C2 step streamed at 3.4 TB/s (profiles/r06_mid_window.txt:
  // 4.68 ms for 15.8 GB).
  fetch(r_begin, rbA, rsA);
  if (r_begin + KR < r_end) fetch(r_begin + KR, rbB, rsB);
  for (long long row0 = r_begin; row0 < r_end; row0 += 2 * KR) {
    __syncthreads();
    stash(rbA, rsA);
    __syncthreads();
    if (row0 + 2 * KR < r_end) fetch(row0 + 2 * KR, rbA, rsA);
    compute();
    if (row0 + KR >= r_end) break;
    __syncthreads();
    stash(rbB, rsB);
    __syncthreads();
    if (row0 + 3 * KR < r_end) fetch(row0 + 3 * KR, rbB, rsB);
    compute();
  }
  const int col = c0 + 16 * w + li;
  if (col < P.C) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * gq + r;
        if (j < rk) atomicAdd(P.out + (long long)j * P.ldo + (long long)t * P.C + col, alpha * acc[t][r]);
      }
  }
}


// ---- dt = (mask (.) dy) U^T / (1-p): the gradient of the down-projection output of a layer whose LoRA branch is dropped
// (utils/lora.py:49,119: y = base(x) + scale * dropout(up(down(x))), so d(up input) = (mask (.) dy / (1-p)) U).  One streaming
// pass over dy: the mask is regenerated from (seed, row * N + column) and applied as a bit select on the loaded chunk, no masked
// copy of dy is ever written (round 3: a mask pass (read + write) plus a skinny GEMM (read) = three passes and two launches).
// v_mfma_f32_16x16x32_bf16 with A = 16 rows x 32 columns of dy, B = the same 32 columns of 16 rank rows of U; the contraction
// index is permuted so that a lane reads 2 x 16 contiguous bytes of its row (k group g <-> columns 16g .. 16g+15 of a 64-column
// block, MFMA step s takes the s-th 8).
// Round 5 shape of the launch (profiles/r05_branch_probe.txt: the round-4 kernel ran 364 launches of a C2 step at 0.1 - 3.2 TB/s,
// ~12 us whatever the size below M = 8192 — 16 .. 64 workgroups of 4 waves with two dependent load rounds each): a workgroup is
// NW = 4 .. 16 waves laid out WR x WC, a wave owns ROWG row groups of 16 rows and every WC-th 64-column block with UNR blocks in
// flight; the launcher picks (ROWG, WC, NW) so that >= ~2048 waves are in flight and a wave needs ONE load round where the
// column count allows (N = 320: UNR = 5).  With WC > 1 the partial sums meet in LDS, added in wave order (bit-reproducible).
template <int RG, int ROWG, int UNR, int MAXT>      // rank groups of 16, row groups per wave, 64-column blocks in flight per wave, block size bound
__global__ __launch_bounds__(MAXT) void lora_drop_dt_kernel(const bf16_t* __restrict__ dy, long long lddy, const bf16_t* __restrict__ U,
                                                             long long ldu, bf16_t* __restrict__ dt, long long lddt, long long M, int N,
                                                             int rp, int WC, float p, unsigned long long seed_in,
                                                             const unsigned long long* __restrict__ epoch, long long u_member_stride,
                                                             unsigned long long seed1, unsigned long long seed2) {
  extern __shared__ __attribute__((aligned(16))) float red_dyn[];     // [NW][ROWG*16][RG*16 + 1] (WC > 1 only)
  constexpr int RPW = ROWG * 16, PITCH_R = RG * 16 + 1;
  // projection group: blockIdx.y = member (own column block of dy, own up factor, own seed, own rank columns of dt)
  if (blockIdx.y > 0) {
    dy += (long long)blockIdx.y * N;
    U += (long long)blockIdx.y * u_member_stride;
    dt += (long long)blockIdx.y * rp;
    seed_in = blockIdx.y == 1 ? seed1 : seed2;
  }
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int NW = (int)blockDim.x >> 6;
  const int WR = NW / WC, wr = w / WC, wc = w - wr * WC;
  const int li = lane & 15, g4 = lane >> 4;
  const long long row0 = ((long long)blockIdx.x * WR + wr) * RPW;
  const DropKey dkey = drop_key(eff_seed(seed_in, epoch), p);
  const float ks = 1.f / (1.f - p);
  f32x4 acc[ROWG][RG];
#pragma unroll
  for (int a = 0; a < ROWG; ++a)
#pragma unroll
    for (int j = 0; j < RG; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const int nblk = (N + 63) >> 6;
  for (int b0 = wc; b0 < nblk; b0 += WC * UNR) {
    bf16x8 a[UNR][ROWG][2], u[UNR][RG][2];
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const int col = (b0 + q * WC) * 64 + g4 * 16;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bool cok = (b0 + q * WC) < nblk && col + s * 8 < N;
#pragma unroll
        for (int rg = 0; rg < ROWG; ++rg) {
          const long long row = row0 + rg * 16 + li;
          a[q][rg][s] = (cok && row < M) ? *(const bf16x8*)(dy + row * lddy + col + s * 8) : zero8;
        }
#pragma unroll
        for (int j = 0; j < RG; ++j) {
          const int rk = j * 16 + li;
          u[q][j][s] = (cok && rk < rp) ? *(const bf16x8*)(U + (long long)rk * ldu + col + s * 8) : zero8;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const int col = (b0 + q * WC) * 64 + g4 * 16;
      if ((b0 + q * WC) >= nblk) break;             // wave-uniform
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int rg = 0; rg < ROWG; ++rg) {
          const long long row = row0 + rg * 16 + li;
          bf16x8 g = a[q][rg][s];
          if (row < M && col + s * 8 < N) {
            const unsigned kb = drop_bits8(dkey, (unsigned long long)row * N + col + s * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (!((kb >> e) & 1u)) g[e] = 0;
          }
#pragma unroll
          for (int j = 0; j < RG; ++j) acc[rg][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g, u[q][j][s], acc[rg][j], 0, 0, 0);
        }
    }
  }
  // accumulator: lane holds rows 4*g4 + r (r = 0..3) of the row group, rank 16j + li
  if (WC == 1) {
#pragma unroll
    for (int rg = 0; rg < ROWG; ++rg)
#pragma unroll
      for (int j = 0; j < RG; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long row = row0 + rg * 16 + 4 * g4 + r;
          const int rk = j * 16 + li;
          if (row < M && rk < rp) dt[row * lddt + rk] = f2bf(acc[rg][j][r] * ks);
        }
    return;
  }
  float* red = red_dyn + (long long)w * RPW * PITCH_R;
#pragma unroll
  for (int rg = 0; rg < ROWG; ++rg)
#pragma unroll
    for (int j = 0; j < RG; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(rg * 16 + 4 * g4 + r) * PITCH_R + j * 16 + li] = acc[rg][j][r];
  __syncthreads();
  // WR row bands of RPW rows x (RG*16) ranks, each summed over its WC waves in wave order
  for (int i = tid; i < WR * RPW * RG * 16; i += (int)blockDim.x) {
    const int rk = i % (RG * 16), rr = (i / (RG * 16)) % RPW, band = i / (RG * 16 * RPW);
    float v = 0.f;
    for (int c = 0; c < WC; ++c) v += red_dyn[((long long)(band * WC + c) * RPW + rr) * PITCH_R + rk];
    const long long row = ((long long)blockIdx.x * WR + band) * RPW + rr;
    if (row < M && rk < rp) dt[row * lddt + rk] = f2bf(v * ks);
  }
}

template <int TAPS_D>
__global__ __launch_bounds__(256) void lora_wgrad_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[KR * PITCH + TAPS_D * KR * SP];
  const int ntiles = a.u.tiles + a.d.tiles;
  const int tile = blockIdx.x % ntiles, chunk = blockIdx.x / ntiles;
  const long long r_begin = (long long)chunk * a.chunk_rows;
  const long long r_end = std::min(a.rows, r_begin + a.chunk_rows);
  if (r_begin >= r_end) return;
  if (tile < a.u.tiles)
    wgrad_body<1>(a.u, r_begin, r_end, tile * 64, a.g, a.rk, a.alpha, smem);
  else
    wgrad_body<TAPS_D>(a.d, r_begin, r_end, (tile - a.u.tiles) * 64, a.g, a.rk, a.alpha, smem);
}

// Many layers in ONE launch (t2v_lora_wgrad_batch): a device table of per-pass arguments + the first workgroup of every
// pass; a workgroup finds its pass by binary search and runs the same body.  The per-layer launches of a C2 step are
// 568 kernels of ~15 us whose ramp and tail dominate (most layers move < 10 MB); batched they stream back to back.
// Round 6: the table is split by window size.  MAXTAPS = 1 (Linear / 1x1 layers: most of a step's passes) compiles the one-tap
// body only — 86 registers and 12 KB of LDS instead of the nine-tap body's 180 / 28 KB, i.e. five workgroups per CU with two
// steps of loads in flight each instead of three with one (24 KB -> 80 KB outstanding per CU); MAXTAPS = 9 takes the passes of
// the windowed layers (their dU tiles run the one-tap body there).
template <int MAXTAPS>
__global__ __launch_bounds__(256, 3) void lora_wgrad_batch_kernel(const Args* __restrict__ jobs, const int* __restrict__ first, int njobs) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[KR * PITCH + MAXTAPS * KR * SP];
  int lo = 0, hi = njobs - 1;
  const int b = (int)blockIdx.x;
  while (lo < hi) {                          // largest j with first[j] <= b (wave-uniform: scalar loads)
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= b) lo = mid;
    else hi = mid - 1;
  }
  const Args a = jobs[lo];
  const int rel = b - first[lo];
  const int ntiles = a.u.tiles + a.d.tiles;
  const int tile = rel % ntiles, chunk = rel / ntiles;
  const long long r_begin = (long long)chunk * a.chunk_rows;
  const long long r_end = std::min(a.rows, r_begin + a.chunk_rows);
  if (r_begin >= r_end) return;
  if (tile < a.u.tiles) {
    wgrad_body<1>(a.u, r_begin, r_end, tile * 64, a.g, a.rk, a.alpha, smem);
    return;
  }
  const int c0 = (tile - a.u.tiles) * 64;
  if constexpr (MAXTAPS == 1) {
    wgrad_body<1>(a.d, r_begin, r_end, c0, a.g, a.rk, a.alpha, smem);
  } else {
    const int taps = a.g.KH * a.g.KW;
    if (taps == 1) wgrad_body<1>(a.d, r_begin, r_end, c0, a.g, a.rk, a.alpha, smem);
    else if (taps == 3) wgrad_body<3>(a.d, r_begin, r_end, c0, a.g, a.rk, a.alpha, smem);
    else wgrad_body<9>(a.d, r_begin, r_end, c0, a.g, a.rk, a.alpha, smem);
  }
}

// argument checks + the passes (one per 16 rank rows) of one layer; returns the number of passes appended or a negative code
int wgrad_passes(const T2VLoraWgrad& p, Args* out, int* blocks) {
  T2V_CHECK_ARG(p.rows > 0 && p.rows < (1LL << 31) && p.t && p.dy && p.dU && p.dt && p.x && p.dD, "t2v_lora_wgrad: bad args");
  T2V_CHECK_ARG(p.rp >= 8 && p.rp <= 32 && p.rp % 8 == 0, "t2v_lora_wgrad: padded rank must be 8, 16, 24 or 32 (got %d)", p.rp);
  T2V_CHECK_ARG(p.drop_p >= 0.f && p.drop_p < 1.f, "t2v_lora_wgrad: dropout probability must be in [0, 1)");
  T2V_CHECK_ARG(p.N > 0 && p.N % 8 == 0 && p.C > 0 && p.C % 8 == 0 && p.ldt % 8 == 0 && p.lddt % 8 == 0 && p.lddy % 8 == 0 &&
                    p.ldx % 8 == 0,
                "t2v_lora_wgrad: N, C and the bf16 leading dimensions must be multiples of 8 (N=%d C=%d)", p.N, p.C);
  int taps = 1;
  T2VConvGeom g = p.geom;
  if (p.conv) {
    taps = g.KH * g.KW;
    T2V_CHECK_ARG(taps == 1 || taps == 3 || taps == 9, "t2v_lora_wgrad: window of %d taps not supported", taps);
    T2V_CHECK_ARG(g.sy == 1 && g.sx == 1 && g.tdiv == 1 && g.up == 0 && g.Hv == g.Ho && g.Wv == g.Wo && g.Hv > 0 && g.Wv > 0,
                  "t2v_lora_wgrad: only stride-1 same-size windows (use the K-major GEMM for the others)");
    T2V_CHECK_ARG(p.rows % ((long long)g.Hv * g.Wv) == 0, "t2v_lora_wgrad: rows is not a whole number of %dx%d images", g.Hv, g.Wv);
  } else {
    g = T2VConvGeom{};
    g.Hv = g.Wv = g.Ho = g.Wo = g.KH = g.KW = 1;
  }
  T2V_CHECK_ARG(p.lddu >= p.N && p.lddd >= (long long)taps * p.C, "t2v_lora_wgrad: output leading dimensions too small");
  int n = 0;
  for (int r0 = 0; r0 < p.rp; r0 += 16, ++n) {     // one pass per 16 rank rows
    Args& a = out[n];
    a.rk = std::min(16, p.rp - r0);
    a.u = Prob{(const bf16_t*)p.dy, p.lddy, p.N, (const bf16_t*)p.t + r0, p.ldt, p.dU + (long long)r0 * p.lddu, p.lddu, (p.N + 63) / 64,
               p.drop_p, p.drop_seed, p.drop_p > 0.f ? t2v_drop_epoch : nullptr};
    a.d = Prob{(const bf16_t*)p.x, p.ldx, p.C, (const bf16_t*)p.dt + r0, p.lddt, p.dD + (long long)r0 * p.lddd, p.lddd, (p.C + 63) / 64,
               0.f, 0ull, nullptr};
    a.rows = p.rows;
    a.conv = p.conv;
    a.g = g;
    a.alpha = p.alpha;
    const int ntiles = a.u.tiles + a.d.tiles;
    long long nsteps = (p.rows + KR - 1) / KR;
    long long nchunks = std::max<long long>(1, std::min<long long>(nsteps, (384 + ntiles - 1) / ntiles));
    a.chunk_rows = (int)(((nsteps + nchunks - 1) / nchunks) * KR);
    nchunks = (p.rows + a.chunk_rows - 1) / a.chunk_rows;
    blocks[n] = (int)(nchunks * ntiles);
  }
  return n;
}

}  // namespace

extern "C" long long t2v_lora_wgrad_batch_bytes(int nlayers) { return (long long)nlayers * 2 * (long long)(sizeof(Args) + sizeof(int)) + 64; }

extern "C" int t2v_lora_wgrad_batch(const T2VLoraWgrad* descs, int nlayers, void* host_staging, void* device_table,
                                    long long table_bytes, t2v_stream_t stream) {
  T2V_CHECK_ARG(descs && nlayers > 0 && host_staging && device_table, "t2v_lora_wgrad_batch: bad args");
  T2V_CHECK_ARG(table_bytes >= t2v_lora_wgrad_batch_bytes(nlayers), "t2v_lora_wgrad_batch: table buffers too small (%lld bytes for %d layers)",
                table_bytes, nlayers);
  // staging layout: [2*nlayers] Args, then [2*nlayers] first-workgroup indices (host builds it, ONE async copy brings it over;
  // inside a stream capture the copy becomes a graph node that re-reads the same pinned staging buffer at every replay)
  // (round 6: two tables — the passes of one-tap layers in front, those of windowed layers behind them, each with its own
  //  first-workgroup indices; one launch per non-empty table)
  Args* jobs = (Args*)host_staging;
  int* first = (int*)((unsigned char*)host_staging + (size_t)nlayers * 2 * sizeof(Args));
  auto light = [&](const T2VLoraWgrad& d) { return !d.conv || d.geom.KH * d.geom.KW == 1; };
  int nl = 0;
  for (int l = 0; l < nlayers; ++l)
    if (light(descs[l])) nl += (descs[l].rp + 15) / 16;
  int il = 0, ih = nl;
  long long tl = 0, th = 0;
  for (int l = 0; l < nlayers; ++l) {
    int blocks[2];
    Args tmp[2];
    const int n = wgrad_passes(descs[l], tmp, blocks);
    if (n < 0) return n;
    const bool lt = light(descs[l]);
    for (int q = 0; q < n; ++q) {
      int& idx = lt ? il : ih;
      long long& tot = lt ? tl : th;
      jobs[idx] = tmp[q];
      first[idx] = (int)tot;
      tot += blocks[q];
      ++idx;
    }
  }
  T2V_CHECK_ARG(il == nl && ih <= 2 * nlayers, "t2v_lora_wgrad_batch: pass count mismatch");
  const int nh = ih - nl;
  T2V_CHECK_ARG(tl < (1LL << 31) && th < (1LL << 31), "t2v_lora_wgrad_batch: too many workgroups");
  const size_t args_bytes = (size_t)nlayers * 2 * sizeof(Args);
  if (hipMemcpyAsync(device_table, host_staging, args_bytes + (size_t)ih * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream) !=
      hipSuccess) {
    t2v_set_error("t2v_lora_wgrad_batch: staging copy failed: %s", hipGetErrorString(hipGetLastError()));
    return T2V_ELAUNCH;
  }
  const Args* djobs = (const Args*)device_table;
  const int* dfirst = (const int*)((const unsigned char*)device_table + args_bytes);
  // (measurement hook: the start event goes to the first launch, the stop event to the last)
  if (nl > 0 && nh > 0) {
    T2V_LAUNCH_FIRST(lora_wgrad_batch_kernel<1>, dim3((unsigned)tl), dim3(256), 0, (hipStream_t)stream, djobs, dfirst, nl);
    T2V_CHECK_LAUNCH();
    T2V_LAUNCH_LAST(lora_wgrad_batch_kernel<9>, dim3((unsigned)th), dim3(256), 0, (hipStream_t)stream, djobs + nl, dfirst + nl, nh);
  } else if (nl > 0) {
    T2V_LAUNCH(lora_wgrad_batch_kernel<1>, dim3((unsigned)tl), dim3(256), 0, (hipStream_t)stream, djobs, dfirst, nl);
  } else {
    T2V_LAUNCH(lora_wgrad_batch_kernel<9>, dim3((unsigned)th), dim3(256), 0, (hipStream_t)stream, djobs + nl, dfirst + nl, nh);
  }
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

extern "C" int t2v_lora_wgrad(const T2VLoraWgrad* pp, t2v_stream_t stream) {
  T2V_CHECK_ARG(pp, "t2v_lora_wgrad: null descriptor");
  Args pass[2];
  int blocks[2];
  const int n = wgrad_passes(*pp, pass, blocks);
  if (n < 0) return n;
  for (int q = 0; q < n; ++q) {
    const Args& a = pass[q];
    const int taps = a.g.KH * a.g.KW;
    dim3 grid((unsigned)blocks[q]);
    if (taps == 1)
      T2V_LAUNCH(lora_wgrad_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (taps == 3)
      T2V_LAUNCH(lora_wgrad_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
      T2V_LAUNCH(lora_wgrad_kernel<9>, grid, dim3(256), 0, (hipStream_t)stream, a);
    T2V_CHECK_LAUNCH();
  }
  return T2V_OK;
}

static int lora_drop_dt_launch(const void* dy, long long lddy, const void* U, long long ldu, long long u_member_stride, void* dt,
                               long long lddt, long long M, int N, int rp, int nmem, float drop_p, const unsigned long long* seeds,
                               t2v_stream_t stream) {
  T2V_CHECK_ARG(dy && U && dt && M > 0 && N > 0 && N % 8 == 0 && lddy % 8 == 0 && ldu % 8 == 0 && lddt >= (long long)rp * nmem, "t2v_lora_drop_dt: bad args");
  T2V_CHECK_ARG(rp >= 8 && rp <= 32 && rp % 8 == 0, "t2v_lora_drop_dt: padded rank must be 8, 16, 24 or 32 (got %d)", rp);
  T2V_CHECK_ARG(nmem >= 1 && nmem <= 3 && seeds, "t2v_lora_drop_dt: 1..3 members");
  T2V_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "t2v_lora_drop_dt: dropout probability must be in [0, 1)");
  // launch shape: ROWG = 1 (16 rows per wave) until the rows alone give ~4096 waves; columns split over WC waves (1 .. 16, never
  // more than the 64-column blocks) until ~2048 waves are in flight; a workgroup = max(4, WC) waves (WR x WC)
  const int nblk = (N + 63) / 64;
  const int rowg = (M / 32) * nmem >= 4096 ? 2 : 1;
  const long long wrows = (M + 16 * rowg - 1) / (16 * rowg) * nmem;
  int wc = 1;
  while (wc < 16 && wc * 2 <= nblk && wrows * wc < 2048) wc *= 2;
  static const int force_wc = [] { const char* e = getenv("T2V_DT_WC"); return e ? atoi(e) : 0; }();          // A/B switches
  static const int force_rowg = [] { const char* e = getenv("T2V_DT_ROWG"); return e ? atoi(e) : 0; }();
  if (force_wc == 1 || force_wc == 2 || force_wc == 4 || force_wc == 8 || force_wc == 16) wc = force_wc;
  const int ROWG = wc > 4 ? 1 : ((force_rowg == 1 || force_rowg == 2) ? force_rowg : rowg);
  const int NW = wc > 4 ? wc : 4, WR = NW / wc;
  const long long blocks = (M + 16LL * ROWG * WR - 1) / (16LL * ROWG * WR);
  T2V_CHECK_ARG(blocks < (1LL << 31), "t2v_lora_drop_dt: too many rows");
  const unsigned long long s0 = seeds[0], s1 = nmem > 1 ? seeds[1] : 0ull, s2 = nmem > 2 ? seeds[2] : 0ull;
  const int RGn = rp <= 16 ? 1 : 2;
  const size_t lds = wc > 1 ? (size_t)NW * ROWG * 16 * (RGn * 16 + 1) * sizeof(float) : 0;
  const dim3 grid((unsigned)blocks, (unsigned)nmem), block((unsigned)NW * 64);
#define T2V_DT_ARGS grid, block, lds, (hipStream_t)stream, (const bf16_t*)dy, lddy, (const bf16_t*)U, ldu, (bf16_t*)dt, lddt, M, N, rp, wc, drop_p, s0, \
                    t2v_drop_epoch, u_member_stride, s1, s2
  // (workgroups of more than 4 waves are compiled for 1024 threads = 128 registers per lane: ROWG = 1 only, which is what the
  //  shape rule above gives them — the rows of such launches are few)
#define T2V_DT_LAUNCH(RG_, ROWG_, UNR_)                                                                   \
  do {                                                                                                    \
    if (NW == 4) t2v_launch_timed(3, lora_drop_dt_kernel<RG_, ROWG_, UNR_, 256>, T2V_DT_ARGS);            \
    else t2v_launch_timed(3, lora_drop_dt_kernel<RG_, 1, UNR_, 1024>, T2V_DT_ARGS);                      \
  } while (0)
  // blocks in flight per wave: all of a wave's share when it is <= 5 (N = 320 on one wave column: one load round), else 4 / 3
  const int per = (nblk + wc - 1) / wc;
  if (RGn == 1) {
    if (ROWG == 1) {
      if (per == 5) T2V_DT_LAUNCH(1, 1, 5);
      else if (per <= 2) T2V_DT_LAUNCH(1, 1, 2);
      else T2V_DT_LAUNCH(1, 1, 4);
    } else {
      if (per == 5) T2V_DT_LAUNCH(1, 2, 5);
      else if (per <= 2) T2V_DT_LAUNCH(1, 2, 2);
      else T2V_DT_LAUNCH(1, 2, 4);
    }
  } else {
    if (ROWG == 1) {
      if (per <= 2) T2V_DT_LAUNCH(2, 1, 2);
      else T2V_DT_LAUNCH(2, 1, 3);
    } else {
      if (per <= 2) T2V_DT_LAUNCH(2, 2, 2);
      else T2V_DT_LAUNCH(2, 2, 3);
    }
  }
#undef T2V_DT_LAUNCH
#undef T2V_DT_ARGS
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

extern "C" int t2v_lora_drop_dt(const void* dy, long long lddy, const void* U, long long ldu, void* dt, long long lddt, long long M,
                                int N, int rp, float drop_p, unsigned long long drop_seed, t2v_stream_t stream) {
  return lora_drop_dt_launch(dy, lddy, U, ldu, 0, dt, lddt, M, N, rp, 1, drop_p, &drop_seed, stream);
}

extern "C" int t2v_lora_drop_dt_group(const void* dy, long long lddy, const void* U, long long ldu, long long u_member_stride, void* dt,
                                      long long lddt, long long M, int N, int rp, int nmem, float drop_p,
                                      const unsigned long long* seeds, t2v_stream_t stream) {
  return lora_drop_dt_launch(dy, lddy, U, ldu, u_member_stride, dt, lddt, M, N, rp, nmem, drop_p, seeds, stream);
}
