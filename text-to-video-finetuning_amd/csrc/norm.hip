// norm.hip — GroupNorm(+SiLU) and LayerNorm, forward and backward, over channels-last bf16 token matrices.
// HBM-bound streaming kernels: 16-byte (8 x bf16) loads per lane, fp32 statistics, fixed-order LDS reductions
// (no floating-point atomics on the statistics).  A GroupNorm "domain" is the set of rows one statistic spans
// (H*W rows for per-frame norms, F*H*W rows for the 5-D temporal norms) — see t2v_abi.h.
#include "common.h"

namespace {

// ------------------------------------------------------------------ GroupNorm statistics
// grid (nsplit, ndomains); each block reduces a slab of rows of one domain over all channels and writes its
// per-group partial; the last block of a domain to finish sums the partials in index order (see the hand-off at the end of
// the kernel) => bit-reproducible statistics without a separate finalize launch.
constexpr int GN_MAX_SPLIT = 256;

// FLAGS (backward only; round 6, as in gn_apply below): bit 0 SiLU, bit 1 dropout, bit 2 parameter gradients — compile-time
template <bool BWD, int FLAGS = 0>
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                        const bf16_t* __restrict__ dy, long long lddy,
                                                        int rows_per_domain, int C, int G, const float* __restrict__ sums,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int silu, float drop_p, unsigned long long drop_seed_in,
                                                        const unsigned long long* __restrict__ drop_epoch,
                                                        float* __restrict__ partial, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, float* __restrict__ out,
                                                        unsigned* __restrict__ counters) {
  __shared__ float sval[256 * 17];
  __shared__ int s_last;   // per-thread (8 sums, 8 second sums), row stride 17 to dodge bank conflicts
  constexpr bool SILU = (FLAGS & 1) != 0, DROP = (FLAGS & 2) != 0, PG = (FLAGS & 4) != 0;
  const DropKey dkey = drop_key(DROP ? eff_seed(drop_seed_in, drop_epoch) : 0ull, drop_p);
  const int d = blockIdx.y, tid = threadIdx.x;
  const int nchunks = C >> 3;
  const int tpr = min(nchunks, 256), rpp = 256 / tpr;
  const int ncb = (nchunks + tpr - 1) / tpr;          // column blocks (C > 2048 -> more than one)
  const int rl = tid / tpr;
  const int cpg = C / G;
  const int rows_per_split = (rows_per_domain + gridDim.x - 1) / gridDim.x;
  const int rbeg = blockIdx.x * rows_per_split, rend = min(rows_per_domain, rbeg + rows_per_split);
  float gacc = 0.f;                                   // thread t < 2G owns (group t>>1, moment t&1)
  for (int cb = 0; cb < ncb; ++cb) {
    const int cc = cb * tpr + tid % tpr;
    const bool active = (rl < rpp) && (cc < nchunks);
    float s0[8], s1[8];
    float a0[8], a1[8];                               // per-channel parameter-gradient sums (full finetune only)
#pragma unroll
    for (int e = 0; e < 8; ++e) s0[e] = s1[e] = a0[e] = a1[e] = 0.f;
    if (active) {
      float mean[8], rstd[8], gm[8], bt[8];
      if (BWD) {
        const float cnt = (float)rows_per_domain * cpg;
        int gi = (cc * 8) / cpg, rem = (cc * 8) - gi * cpg;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = cc * 8 + e;
          float mu = sums[(d * G + gi) * 2] / cnt;
          float var = fmaxf(sums[(d * G + gi) * 2 + 1] / cnt - mu * mu, 0.f);
          mean[e] = mu;
          rstd[e] = rsqrtf(var + eps);
          gm[e] = gamma[c];
          bt[e] = beta[c];
          if (++rem == cpg) {
            rem = 0;
            ++gi;
          }
        }
      }
      const float ks = DROP ? 1.f / (1.f - drop_p) : 1.f;
      constexpr int UN = 4;                 // independent 16-byte loads in flight per operand
      for (int r0 = rbeg + rl; r0 < rend; r0 += UN * rpp) {
        bf16x8 xq[UN], gq[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int rr = r0 + u * rpp;
          if (rr < rend) {
            const long long row = (long long)d * rows_per_domain + rr;
            xq[u] = *(const bf16x8*)(x + row * ldx + cc * 8);
            if (BWD) gq[u] = *(const bf16x8*)(dy + row * lddy + cc * 8);
          }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int rr = r0 + u * rpp;
          if (rr >= rend) continue;
          const long long row = (long long)d * rows_per_domain + rr;
          const bf16x8 xv = xq[u];
          if (!BWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float v = bf2f((unsigned short)xv[e]);
              s0[e] += v;
              s1[e] += v * v;
            }
          } else {
            const bf16x8 gv = gq[u];
            const unsigned kb = DROP ? drop_bits8(dkey, (unsigned long long)row * C + cc * 8) : 0xffu;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float xh = (bf2f((unsigned short)xv[e]) - mean[e]) * rstd[e];
              float dz = bf2f((unsigned short)gv[e]);
              if (DROP) dz = ((kb >> e) & 1u) ? dz * ks : 0.f;
              if (SILU) {
                float zz = xh * gm[e] + bt[e];
                float sg = sigmoid_f(zz);
                dz *= sg * (1.f + zz * (1.f - sg));
              }
              s0[e] += dz * gm[e];           // sum dxh
              s1[e] += dz * gm[e] * xh;      // sum dxh * xh
              if (PG) {
                a0[e] += dz * xh;
                a1[e] += dz;
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sval[tid * 17 + e] = s0[e];
      sval[tid * 17 + 8 + e] = s1[e];
    }
    __syncthreads();
    if (tid < 2 * G) {
      const int g = tid >> 1, which = tid & 1;
      const int clo = max(g * cpg, cb * tpr * 8), chi = min((g + 1) * cpg, min(C, (cb + 1) * tpr * 8));
      for (int c = clo; c < chi; ++c) {
        const int chl = (c >> 3) - cb * tpr, e = c & 7;
        for (int q = 0; q < rpp; ++q) gacc += sval[(q * tpr + chl) * 17 + which * 8 + e];
      }
    }
    __syncthreads();
    if (BWD && PG) {
      // Parameter gradients d(gamma), d(beta) (config C3): the row lanes of a chunk column are summed through LDS in lane
      // order and the workgroup writes ONE partial row [2][C] with plain stores (`dgamma` is the partial buffer here);
      // param_grad_reduce_kernel adds the workgroups' rows in index order.  The first version issued 16 float atomics per
      // thread onto the same 2C addresses: 300 us per launch, 50 ms of the C3 step (profiles/r03_c3_kernel_stats.txt).
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sval[tid * 17 + e] = a0[e];
        sval[tid * 17 + 8 + e] = a1[e];
      }
      __syncthreads();
      const int ccc = cb * tpr + tid;
      if (tid < tpr && ccc < nchunks) {
        float* pg = dgamma + ((long long)d * gridDim.x + blockIdx.x) * 2 * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t0 = 0.f, t1 = 0.f;
          for (int q = 0; q < rpp; ++q) {
            t0 += sval[(q * tpr + tid) * 17 + e];
            t1 += sval[(q * tpr + tid) * 17 + 8 + e];
          }
          pg[ccc * 8 + e] = t0;
          pg[C + ccc * 8 + e] = t1;
        }
      }
      __syncthreads();
    }
  }
  // ---- last-arriving block of this domain sums the per-split partials IN FIXED ORDER (bit-reproducible whichever
  // block is last).  Hand-off in the write-through form of the agent-scope recipe (guide G16): the 4-byte partials are
  // sc1 stores (relaxed agent-scope atomic stores -> no L2 write-back fence), every wave drains its stores, one lane
  // takes a ticket; the last arriver reads the partials with sc1 loads.
  if (tid < 2 * G)
    __hip_atomic_store(partial + (((long long)d * gridDim.x + blockIdx.x) * G) * 2 + tid, gacc, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    unsigned ticket = __hip_atomic_fetch_add(counters + d, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (ticket == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (s_last) {
    // 256 threads = (2G statistics) x nparts slices of the split range; every thread sums its slice in index order with
    // 8 loads in flight, then the slices are combined in slice order: a fixed association whichever block is last.
    const int nst = 2 * G, nparts = max(1, 256 / nst);
    const int v = tid % nst, part = tid / nst;
    float a = 0.f;
    if (part < nparts) {
      const unsigned per = (gridDim.x + nparts - 1) / nparts;
      const unsigned sb = part * per, se = min(gridDim.x, sb + per);
      const float* pp = partial + (long long)d * gridDim.x * G * 2 + v;
      for (unsigned s0 = sb; s0 < se; s0 += 8) {
        float vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          vv[u] = (s0 + u < se) ? __hip_atomic_load(pp + (long long)(s0 + u) * G * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) a += vv[u];
      }
    }
    sval[tid] = a;
    __syncthreads();
    if (tid < nst) {
      float tot = 0.f;
      for (int q = 0; q < nparts; ++q) tot += sval[q * nst + tid];
      out[(long long)d * G * 2 + tid] = tot;
    }
    if (tid == 0) counters[d] = 0u;          // re-armed for the next call (stream order makes it visible)
  }
}

// ------------------------------------------------------------------ GroupNorm apply (fwd) / dx (bwd)
// grid (nsplit, ndomains) like the statistics pass: a thread owns ONE 8-channel chunk column and walks rows, so the
// per-channel affine terms (mean, rstd, gamma, beta and the backward sums) are formed once in registers and the row loop
// is pure streaming — four independent 16-byte loads in flight per operand.
// FLAGS (round 6): bit 0 SiLU, bit 1 dropout, bit 2 (backward) a pass-through addend — compile-time, not run-time: as uniform
// run-time flags every ELEMENT of the row loop carried its own `s_cbranch` around the transcendental / the select (8 per 16-byte
// chunk, found in the ISA), which also kept the eight exp -> rcp chains of a chunk from interleaving.
template <bool BWD, int FLAGS>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                        const bf16_t* __restrict__ dy, long long lddy,
                                                        bf16_t* __restrict__ y, long long ldy, int rows_per_domain, int C, int G,
                                                        const float* __restrict__ sums, const float* __restrict__ bsums,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int silu, float drop_p, unsigned long long drop_seed_in,
                                                        const unsigned long long* __restrict__ drop_epoch,
                                                        const bf16_t* __restrict__ addend, long long ldadd) {
  constexpr bool SILU = (FLAGS & 1) != 0, DROP = (FLAGS & 2) != 0, ADD = BWD && (FLAGS & 4) != 0;
  const DropKey dkey = drop_key(DROP ? eff_seed(drop_seed_in, drop_epoch) : 0ull, drop_p);
  const int d = blockIdx.y, tid = threadIdx.x;
  const int nchunks = C >> 3;
  const int tpr = min(nchunks, 256), rpp = 256 / tpr;
  const int ncb = (nchunks + tpr - 1) / tpr;
  const int rl = tid / tpr, cpg = C / G;
  const int rows_per_split = (rows_per_domain + gridDim.x - 1) / gridDim.x;
  const int rbeg = blockIdx.x * rows_per_split, rend = min(rows_per_domain, rbeg + rows_per_split);
  const float icnt = 1.f / ((float)rows_per_domain * cpg);
  const float ks = DROP ? 1.f / (1.f - drop_p) : 1.f;
  constexpr int UN = 4;
  for (int cb = 0; cb < ncb; ++cb) {
    const int cc = cb * tpr + tid % tpr;
    if (rl >= rpp || cc >= nchunks) continue;
    float mu[8], rs[8], gm[8], bt[8], b1[8], b2[8];
    {
      int gi = (cc * 8) / cpg, rem = (cc * 8) - gi * cpg;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* sp = sums + ((long long)d * G + gi) * 2;
        const float m = sp[0] * icnt;
        mu[e] = m;
        rs[e] = rsqrtf(fmaxf(sp[1] * icnt - m * m, 0.f) + eps);
        gm[e] = gamma[cc * 8 + e];
        bt[e] = beta[cc * 8 + e];
        if (BWD) {
          const float* bp = bsums + ((long long)d * G + gi) * 2;
          b1[e] = bp[0] * icnt;
          b2[e] = bp[1] * icnt;
        }
        if (++rem == cpg) {
          rem = 0;
          ++gi;
        }
      }
      if (!BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {     // y = x * sc + sh
          gm[e] *= rs[e];
          bt[e] -= mu[e] * gm[e];
        }
      }
    }
    const long long base = (long long)d * rows_per_domain;
    for (int r = rbeg + rl; r < rend; r += UN * rpp) {
      bf16x8 xv[UN], gv[UN], av[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int rr = r + u * rpp;
        if (rr < rend) {
          xv[u] = *(const bf16x8*)(x + (base + rr) * ldx + cc * 8);
          if (BWD) gv[u] = *(const bf16x8*)(dy + (base + rr) * lddy + cc * 8);
          if (ADD) av[u] = *(const bf16x8*)(addend + (base + rr) * ldadd + cc * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int rr = r + u * rpp;
        if (rr >= rend) continue;
        const long long row = base + rr;
        bf16x8 ov;
        const unsigned kb = DROP ? drop_bits8(dkey, (unsigned long long)row * C + cc * 8) : 0xffu;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float out;
          if (!BWD) {
            float zz = bf2f((unsigned short)xv[u][e]) * gm[e] + bt[e];
            if (SILU) zz = silu_f(zz);
            if (DROP) zz = ((kb >> e) & 1u) ? zz * ks : 0.f;
            out = zz;
          } else {
            const float xh = (bf2f((unsigned short)xv[u][e]) - mu[e]) * rs[e];
            float dz = bf2f((unsigned short)gv[u][e]);
            if (DROP) dz = ((kb >> e) & 1u) ? dz * ks : 0.f;
            if (SILU) {
              const float zz = xh * gm[e] + bt[e];
              const float sg = sigmoid_f(zz);
              dz *= sg * (1.f + zz * (1.f - sg));
            }
            out = rs[e] * (dz * gm[e] - b1[e] - xh * b2[e]);
            if (ADD) out += bf2f((unsigned short)av[u][e]);      // gradient of a pass-through (residual) use of x
          }
          ov[e] = (short)f2bf(out);
        }
        *(bf16x8*)(y + row * ldy + cc * 8) = ov;
      }
    }
  }
}

// ------------------------------------------------------------------ LayerNorm: one wave per row
// LN_MAXCH chunks of 8 per lane (C <= 512 * LN_MAXCH) and LN_RW rows per wave and trip are template parameters: the common
// C = 320 rows need one chunk per lane and leave registers for eight rows in flight

// what an out-of-range 16-byte chunk reads.  (Round 6, found in the ISA: the row loads of both LayerNorm kernels were predicated
// `if (ch < nch && row < rows) x = load` — every load sat behind an exec-mask branch and the compiler put `s_waitcnt vmcnt(0)` at each
// merge, so the "eight rows in flight per wave" went out ONE AT A TIME, each behind the previous row's round trip.  The loads are
// unconditional now: an out-of-range chunk reads this page, and the statistics of the backward come from a clamped row.)
__device__ __attribute__((aligned(16))) unsigned g_ln_zero_page[8];

template <int LN_MAXCH, int LN_RW_F>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, long long ldx, bf16_t* __restrict__ y,
                                                      long long ldy, int rows, int C, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nch = C >> 3;
  // the affine terms of this lane's channels, once per wave (round 6: inside the row loop they were re-loaded for every row —
  // four loads and a full wait per row between the reduction and the store)
  float gm[LN_MAXCH][8], bt[LN_MAXCH][8];
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int ch = min(lane + 64 * i, nch - 1);           // (scalar loads: a parameter inside the flat buffer is only 4-byte aligned)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gm[i][e] = gamma[ch * 8 + e];
      bt[i][e] = beta[ch * 8 + e];
    }
  }
  // LN_RW_F rows per wave and trip: the 16-byte loads of all of them are issued before the first row is reduced (one row per
  // wave left a single load round trip in flight per wave — 2.2 TB/s at level 0)
  for (long long row0 = ((long long)blockIdx.x * 4 + wv) * LN_RW_F; row0 < rows; row0 += (long long)gridDim.x * 4 * LN_RW_F) {
    bf16x8 xq[LN_RW_F][LN_MAXCH];
    const bf16_t* zpage = (const bf16_t*)g_ln_zero_page;
#pragma unroll
    for (int r = 0; r < LN_RW_F; ++r)
#pragma unroll
      for (int i = 0; i < LN_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        const bf16_t* src = (ch < nch && row0 + r < rows) ? x + (row0 + r) * ldx + ch * 8 : zpage;
        xq[r][i] = *(const bf16x8*)src;
      }
#pragma unroll
    for (int r = 0; r < LN_RW_F; ++r) {
      const long long row = row0 + r;
      if (row >= rows) break;
      float v[LN_MAXCH][8];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < LN_MAXCH; ++i) {
        int ch = lane + 64 * i;
        if (ch < nch) {
          const bf16x8 xv = xq[r][i];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[i][e] = bf2f((unsigned short)xv[e]);
            s += v[i][e];
          }
        }
      }
      const float mu = wave_sum(s) / C;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < LN_MAXCH; ++i) {
        int ch = lane + 64 * i;
        if (ch < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float dlt = v[i][e] - mu;
            q += dlt * dlt;
          }
        }
      }
      const float rs = rsqrtf(wave_sum(q) / C + eps);
      if (lane == 0 && stats) {
        stats[row * 2] = mu;
        stats[row * 2 + 1] = rs;
      }
#pragma unroll
      for (int i = 0; i < LN_MAXCH; ++i) {
        int ch = lane + 64 * i;
        if (ch < nch) {
          bf16x8 ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = (short)f2bf((v[i][e] - mu) * rs * gm[i][e] + bt[i][e]);
          *(bf16x8*)(y + row * ldy + ch * 8) = ov;
        }
      }
    }
  }
}

template <int LN_MAXCH, int LN_RW_B, bool PG>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                      const bf16_t* __restrict__ dy, long long lddy,
                                                      bf16_t* __restrict__ dx, long long lddx, int rows, int C,
                                                      const float* __restrict__ gamma, const float* __restrict__ stats,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      const bf16_t* __restrict__ addend, long long ldadd) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nch = C >> 3;
  float ag[PG ? LN_MAXCH : 1][8], ab[PG ? LN_MAXCH : 1][8];
  if constexpr (PG) {
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) ag[i][e] = ab[i][e] = 0.f;
  }
  for (long long row0 = ((long long)blockIdx.x * 4 + wv) * LN_RW_B; row0 < rows; row0 += (long long)gridDim.x * 4 * LN_RW_B) {
    bf16x8 xq[LN_RW_B][LN_MAXCH], gq[LN_RW_B][LN_MAXCH], aq[LN_RW_B][LN_MAXCH];
    float mus[LN_RW_B], rss[LN_RW_B];
    const bf16_t* zpage = (const bf16_t*)g_ln_zero_page;
#pragma unroll
    for (int r = 0; r < LN_RW_B; ++r) {
      const long long row = row0 + r;
      const bool rok = row < rows;
      const long long rc = rok ? row : rows - 1;           // (a clamped row: its values are never used)
      mus[r] = stats[rc * 2];
      rss[r] = stats[rc * 2 + 1];
#pragma unroll
      for (int i = 0; i < LN_MAXCH; ++i) {
        const int ch = lane + 64 * i;
        const bool ok = rok && ch < nch;
        xq[r][i] = *(const bf16x8*)(ok ? x + row * ldx + ch * 8 : zpage);
        gq[r][i] = *(const bf16x8*)(ok ? dy + row * lddy + ch * 8 : zpage);
        if (addend) aq[r][i] = *(const bf16x8*)(ok ? addend + row * ldadd + ch * 8 : zpage);
      }
    }
#pragma unroll
    for (int r = 0; r < LN_RW_B; ++r) {
      const long long row = row0 + r;
      if (row >= rows) break;
      const float mu = mus[r], rs = rss[r];
      float xh[LN_MAXCH][8], dh[LN_MAXCH][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < LN_MAXCH; ++i) {
        int ch = lane + 64 * i;
        if (ch < nch) {
          const bf16x8 xv = xq[r][i], gv = gq[r][i];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float h = (bf2f((unsigned short)xv[e]) - mu) * rs;
            float g = bf2f((unsigned short)gv[e]);
            if constexpr (PG) {
              ag[i][e] += g * h;
              ab[i][e] += g;
            }
            float dd = g * gamma[ch * 8 + e];
            xh[i][e] = h;
            dh[i][e] = dd;
            s1 += dd;
            s2 += dd * h;
          }
        }
      }
      s1 = wave_sum(s1) / C;
      s2 = wave_sum(s2) / C;
#pragma unroll
      for (int i = 0; i < LN_MAXCH; ++i) {
        int ch = lane + 64 * i;
        if (ch < nch) {
          bf16x8 ov;
          if (addend) {                                     // + gradient of a pass-through (residual) use of x
            const bf16x8 av = aq[r][i];
#pragma unroll
            for (int e = 0; e < 8; ++e)
              ov[e] = (short)f2bf(rs * (dh[i][e] - s1 - xh[i][e] * s2) + bf2f((unsigned short)av[e]));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (short)f2bf(rs * (dh[i][e] - s1 - xh[i][e] * s2));
          }
          *(bf16x8*)(dx + row * lddx + ch * 8) = ov;
        }
      }
    }
  }
  if constexpr (PG) {
    // the four waves' per-channel sums are added through LDS in wave order, then ONE partial row [2][C] per workgroup with
    // plain stores (`dgamma` is the partial buffer; param_grad_reduce_kernel adds the rows in index order).  The first
    // version issued 2C float atomics per WAVE onto the same addresses: 820 us per launch, 81 ms of the C3 step.
    extern __shared__ float lnred[];                     // [2][C]
    for (int w = 0; w < 4; ++w) {
      if (wv == w) {
#pragma unroll
        for (int i = 0; i < LN_MAXCH; ++i) {
          const int ch = lane + 64 * i;
          if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              if (w == 0) {
                lnred[ch * 8 + e] = ag[i][e];
                lnred[C + ch * 8 + e] = ab[i][e];
              } else {
                lnred[ch * 8 + e] += ag[i][e];
                lnred[C + ch * 8 + e] += ab[i][e];
              }
            }
          }
        }
      }
      __syncthreads();
    }
    float* pg = dgamma + (long long)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < 2 * C; c += 256) pg[c] = lnred[c];
  }
}

// d(gamma)[c] += sum_w partial[w][0][c], d(beta)[c] += sum_w partial[w][1][c]: 32 columns x 8 row slices per workgroup, every
// slice summed in index order, the slices combined in slice order (fixed association: bit-reproducible)
__global__ __launch_bounds__(256) void param_grad_reduce_kernel(const float* __restrict__ partial, int nrows, int C,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;                   // column of the [2C]-wide partial rows
  float a = 0.f;
  if (c < 2 * C) {
    const int per = (nrows + 7) / 8;
    const int r0 = sl * per, r1 = min(nrows, r0 + per);
    for (int r = r0; r < r1; r += 4) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (r + u < r1) ? partial[(long long)(r + u) * 2 * C + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) a += v[u];
    }
  }
  red[sl][cl] = a;
  __syncthreads();
  if (sl == 0 && c < 2 * C) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) tot += red[q][cl];
    if (c < C) dgamma[c] += tot;
    else dbeta[c - C] += tot;
  }
}

// ------------------------------------------------------------------ GroupNorm statistics from GEMM-epilogue column sums
// grid (G, ndomains), 256 threads: thread i owns items i, i+256, .. of the (tile, column-of-group) list — a fixed assignment
// with the loads of up to 8 items in flight — and the partials are added in a fixed two-level order: bit-reproducible.
__global__ __launch_bounds__(256) void gn_finish_kernel(const float* __restrict__ colsum, int rows_per_domain, int C, int G,
                                                         float* __restrict__ out) {
  __shared__ float sred[256 * 2];
  const int bmt = ((const int*)colsum)[0], nb = ((const int*)colsum)[1];
  const int g = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G;
  float a0 = 0.f, a1 = 0.f;
  if (bmt > 0 && rows_per_domain % bmt == 0 && nb == C) {
    const int tpd = rows_per_domain / bmt;
    const float* base = colsum + 4 + ((long long)d * tpd * nb + g * cpg) * 2;
    const int items = tpd * cpg;
    for (int i0 = tid; i0 < items; i0 += 256 * 8) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        const int t = i / cpg, c = i - t * cpg;
        v[u] = i < items ? *(const float2*)(base + ((long long)t * nb + c) * 2) : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 += v[u].x;
        a1 += v[u].y;
      }
    }
  } else {
    a0 = a1 = __builtin_nanf("");                   // a caller bug must not pass silently
  }
  sred[tid * 2] = a0;
  sred[tid * 2 + 1] = a1;
  __syncthreads();
  float seg_tot = 0.f;
  if (tid < 32) {                                   // 16 segments x 2 moments: segment sums in index order
    const int seg = tid >> 1, w = tid & 1;
    for (int q = 0; q < 16; ++q) seg_tot += sred[(seg * 16 + q) * 2 + w];
  }
  __syncthreads();
  if (tid < 32) sred[tid] = seg_tot;
  __syncthreads();
  if (tid < 2) {
    float tot = 0.f;
    for (int q = 0; q < 16; ++q) tot += sred[q * 2 + tid];
    out[((long long)d * G + g) * 2 + tid] = tot;
  }
}

int gn_check(const char* fn, int C, int G, long long ldx) {
  if (C <= 0 || G <= 0 || C % G != 0 || C % 8 != 0 || ldx % 8 != 0) {
    t2v_set_error("%s: need C%%G==0, C%%8==0, ld%%8==0 (C=%d G=%d ld=%lld)", fn, C, G, ldx);
    return T2V_EINVAL;
  }
  return T2V_OK;
}
// row slabs of the apply passes: >= 4 rows per thread, up to ~2048 blocks (no cross-block reduction, so no split cap)
int gn_apply_splits(int ndomains, int rows_per_domain, int C) {
  static const int rows_per_thread = [] { const char* e = getenv("T2V_GN_APPLY_ROWS"); return e ? max(1, atoi(e)) : 4; }();   // (4: measured optimum of {4, 8, 16} at C2)
  int rpp = 256 / min(C >> 3, 256);
  int want = max(1, 2048 / max(1, ndomains));
  return max(1, min(want, rows_per_domain / (rpp * rows_per_thread)));
}
int gn_splits(int ndomains, int rows_per_domain, int C) {
  int rpp = 256 / min(C >> 3, 256);
  int want = max(1, 2048 / max(1, ndomains));
  int maxs = max(1, rows_per_domain / (rpp * 8));
  return max(1, min(GN_MAX_SPLIT, min(want, maxs)));
}
}  // namespace

// workspace = [GN_MAX_DOMAINS] arrival counters (zero on first use; the kernels leave them zero) followed by
// [ndomains][GN_MAX_SPLIT][G][2] partial sums.  The counters sit at a FIXED place so that partials of another call (other
// ndomains/G) can never land on them.
constexpr int GN_MAX_DOMAINS = 4096;
extern "C" long long t2v_gn_workspace_floats(int ndomains, int G) {
  return GN_MAX_DOMAINS + (long long)ndomains * GN_MAX_SPLIT * G * 2;
}

extern "C" int t2v_gn_stats(const void* x, long long ldx, int ndomains, int rows_per_domain, int C, int G, float* sums,
                            float* workspace, t2v_stream_t stream) {
  if (int e = gn_check("t2v_gn_stats", C, G, ldx)) return e;
  T2V_CHECK_ARG(x && sums && workspace && ndomains > 0 && rows_per_domain > 0 && G <= 128, "t2v_gn_stats: bad args");
  const int ns = gn_splits(ndomains, rows_per_domain, C);
  dim3 grid(ns, ndomains);
  T2V_CHECK_ARG(ndomains <= GN_MAX_DOMAINS, "t2v_gn_stats: more than %d domains", GN_MAX_DOMAINS);
  unsigned* counters = (unsigned*)workspace;
  T2V_LAUNCH(gn_stats_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, nullptr, 0,
                     rows_per_domain, C, G, nullptr, nullptr, nullptr, 0.f, 0, 0.f, 0ull, (const unsigned long long*)nullptr, workspace + GN_MAX_DOMAINS, nullptr,
                     nullptr, sums, counters);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

extern "C" int t2v_gn_finish(const float* colsum, int ndomains, int rows_per_domain, int C, int G, float* sums,
                             t2v_stream_t stream) {
  if (int e = gn_check("t2v_gn_finish", C, G, 8)) return e;
  T2V_CHECK_ARG(colsum && sums && ndomains > 0 && ndomains <= 65535 && rows_per_domain > 0, "t2v_gn_finish: bad args");
  T2V_LAUNCH(gn_finish_kernel, dim3(G, ndomains), dim3(256), 0, (hipStream_t)stream, colsum, rows_per_domain, C, G, sums);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

extern "C" int t2v_gn_apply(const void* x, long long ldx, void* y, long long ldy, int ndomains, int rows_per_domain, int C,
                            int G, const float* sums, const float* gamma, const float* beta, float eps, int silu,
                            float drop_p, unsigned long long drop_seed, t2v_stream_t stream) {
  if (int e = gn_check("t2v_gn_apply", C, G, ldx)) return e;
  T2V_CHECK_ARG(x && y && sums && gamma && beta && ldy % 8 == 0, "t2v_gn_apply: bad args");
  T2V_CHECK_ARG(ndomains > 0 && ndomains <= 65535 && rows_per_domain > 0, "t2v_gn_apply: bad domain grid");
  dim3 grid(gn_apply_splits(ndomains, rows_per_domain, C), ndomains);
#define T2V_GN_FWD(F_)                                                                                                         \
  T2V_LAUNCH((gn_apply_kernel<false, F_>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, nullptr, 0, (bf16_t*)y, \
             ldy, rows_per_domain, C, G, sums, nullptr, gamma, beta, eps, silu, drop_p, drop_seed, t2v_drop_epoch,                 \
             (const bf16_t*)nullptr, 0LL)
  switch ((silu ? 1 : 0) | (drop_p > 0.f ? 2 : 0)) {
    case 0: T2V_GN_FWD(0); break;
    case 1: T2V_GN_FWD(1); break;
    case 2: T2V_GN_FWD(2); break;
    default: T2V_GN_FWD(3); break;
  }
#undef T2V_GN_FWD
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

static int ln_bwd_grid(int rows, int C, bool with_pg) {
  const int rw = C <= 512 ? 8 : (C <= 1024 ? 4 : 2);
  return min((rows + 4 * rw - 1) / (4 * rw), with_pg ? 2048 : 16384);
}
// sizes of the caller-owned scratch for the per-workgroup parameter-gradient partial rows (Face 2: the library owns no memory)
extern "C" long long t2v_gn_bwd_pg_floats(int ndomains, int rows_per_domain, int C) {
  return (long long)gn_splits(ndomains, rows_per_domain, C) * ndomains * 2 * C;
}
extern "C" long long t2v_layernorm_bwd_pg_floats(int rows, int C) { return (long long)ln_bwd_grid(rows, C, true) * 2 * C; }

extern "C" int t2v_gn_bwd_stats(const void* x, long long ldx, const void* dy, long long lddy, int ndomains,
                                int rows_per_domain, int C, int G, const float* sums, const float* gamma, const float* beta,
                                float eps, int silu, float drop_p, unsigned long long drop_seed, float* bsums, float* workspace,
                                float* dgamma, float* dbeta, float* pg_workspace, t2v_stream_t stream) {
  if (int e = gn_check("t2v_gn_bwd_stats", C, G, ldx)) return e;
  T2V_CHECK_ARG(x && dy && sums && gamma && beta && bsums && workspace && lddy % 8 == 0 && G <= 128, "t2v_gn_bwd_stats: bad args");
  T2V_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "t2v_gn_bwd_stats: dgamma/dbeta must both be set or NULL");
  const int ns = gn_splits(ndomains, rows_per_domain, C);
  dim3 grid(ns, ndomains);
  T2V_CHECK_ARG(ndomains <= GN_MAX_DOMAINS, "t2v_gn_bwd_stats: more than %d domains", GN_MAX_DOMAINS);
  unsigned* counters = (unsigned*)workspace;
  float* pg = nullptr;
  const int nwg = ns * ndomains;
  if (dgamma) {
    pg = pg_workspace;
    T2V_CHECK_ARG(pg, "t2v_gn_bwd_stats: dgamma / dbeta need pg_workspace (t2v_gn_bwd_pg_floats() = %lld floats)", (long long)nwg * 2 * C);
  }
#define T2V_GN_BSTATS(F_)                                                                                                        \
  do {                                                                                                                           \
    if (dgamma)                                                                                                                  \
      T2V_LAUNCH_FIRST((gn_stats_kernel<true, (F_) | 4>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,           \
                       (const bf16_t*)dy, lddy, rows_per_domain, C, G, sums, gamma, beta, eps, silu, drop_p, drop_seed,           \
                       t2v_drop_epoch, workspace + GN_MAX_DOMAINS, pg, dbeta, bsums, counters);                                  \
    else                                                                                                                         \
      T2V_LAUNCH((gn_stats_kernel<true, (F_)>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (const bf16_t*)dy, \
                 lddy, rows_per_domain, C, G, sums, gamma, beta, eps, silu, drop_p, drop_seed, t2v_drop_epoch,                    \
                 workspace + GN_MAX_DOMAINS, pg, dbeta, bsums, counters);                                                        \
  } while (0)
  switch ((silu ? 1 : 0) | (drop_p > 0.f ? 2 : 0)) {
    case 0: T2V_GN_BSTATS(0); break;
    case 1: T2V_GN_BSTATS(1); break;
    case 2: T2V_GN_BSTATS(2); break;
    default: T2V_GN_BSTATS(3); break;
  }
#undef T2V_GN_BSTATS
  T2V_CHECK_LAUNCH();
  if (dgamma) {
    T2V_LAUNCH_LAST(param_grad_reduce_kernel, dim3((2 * C + 31) / 32), dim3(256), 0, (hipStream_t)stream, (const float*)pg, nwg, C, dgamma, dbeta);
    T2V_CHECK_LAUNCH();
  }
  return T2V_OK;
}

extern "C" int t2v_gn_bwd_apply(const void* x, long long ldx, const void* dy, long long lddy, void* dx, long long lddx,
                                int ndomains, int rows_per_domain, int C, int G, const float* sums, const float* bsums,
                                const float* gamma, const float* beta, float eps, int silu, float drop_p,
                                unsigned long long drop_seed, const void* addend, long long ldadd, t2v_stream_t stream) {
  if (int e = gn_check("t2v_gn_bwd_apply", C, G, ldx)) return e;
  T2V_CHECK_ARG(!addend || ldadd % 8 == 0, "t2v_gn_bwd_apply: addend leading dimension must be a multiple of 8");
  T2V_CHECK_ARG(x && dy && dx && sums && bsums && gamma && beta && lddy % 8 == 0 && lddx % 8 == 0,
                "t2v_gn_bwd_apply: bad args");
  T2V_CHECK_ARG(ndomains > 0 && ndomains <= 65535 && rows_per_domain > 0, "t2v_gn_bwd_apply: bad domain grid");
  dim3 grid(gn_apply_splits(ndomains, rows_per_domain, C), ndomains);
#define T2V_GN_BWD(F_)                                                                                                         \
  T2V_LAUNCH((gn_apply_kernel<true, F_>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, \
             (bf16_t*)dx, lddx, rows_per_domain, C, G, sums, bsums, gamma, beta, eps, silu, drop_p, drop_seed, t2v_drop_epoch,     \
             (const bf16_t*)addend, ldadd)
  switch ((silu ? 1 : 0) | (drop_p > 0.f ? 2 : 0) | (addend ? 4 : 0)) {
    case 0: T2V_GN_BWD(0); break;
    case 1: T2V_GN_BWD(1); break;
    case 2: T2V_GN_BWD(2); break;
    case 3: T2V_GN_BWD(3); break;
    case 4: T2V_GN_BWD(4); break;
    case 5: T2V_GN_BWD(5); break;
    case 6: T2V_GN_BWD(6); break;
    default: T2V_GN_BWD(7); break;
  }
#undef T2V_GN_BWD
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

extern "C" int t2v_layernorm_fwd(const void* x, long long ldx, void* y, long long ldy, int rows, int C, const float* gamma,
                                 const float* beta, float eps, float* stats, t2v_stream_t stream) {
  T2V_CHECK_ARG(x && y && gamma && beta && rows > 0, "t2v_layernorm_fwd: bad args");
  T2V_CHECK_ARG(C % 8 == 0 && C <= 2048 && ldx % 8 == 0 && ldy % 8 == 0, "t2v_layernorm_fwd: need C%%8==0, C<=2048 (C=%d)", C);
  auto go = [&](auto kern, int rw) {
    const int grid = min((rows + 4 * rw - 1) / (4 * rw), 16384);
    T2V_LAUNCH(kern, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, C, gamma, beta, eps,
               stats);
  };
  if (C <= 512) go(ln_fwd_kernel<1, 8>, 8);
  else if (C <= 1024) go(ln_fwd_kernel<2, 4>, 4);
  else go(ln_fwd_kernel<4, 2>, 2);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

extern "C" int t2v_layernorm_bwd(const void* x, long long ldx, const void* dy, long long lddy, void* dx, long long lddx,
                                 int rows, int C, const float* gamma, const float* stats, float* dgamma, float* dbeta,
                                 float* pg_workspace, const void* addend, long long ldadd, t2v_stream_t stream) {
  T2V_CHECK_ARG(!addend || ldadd % 8 == 0, "t2v_layernorm_bwd: addend leading dimension must be a multiple of 8");
  T2V_CHECK_ARG(x && dy && dx && gamma && stats && rows > 0, "t2v_layernorm_bwd: bad args");
  T2V_CHECK_ARG(C % 8 == 0 && C <= 2048 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0,
                "t2v_layernorm_bwd: need C%%8==0, C<=2048 (C=%d)", C);
  T2V_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "t2v_layernorm_bwd: dgamma/dbeta must both be set or NULL");
  const int grid = ln_bwd_grid(rows, C, dgamma != nullptr);
  float* pg = nullptr;
  if (dgamma) {
    pg = pg_workspace;
    T2V_CHECK_ARG(pg, "t2v_layernorm_bwd: dgamma / dbeta need pg_workspace (t2v_layernorm_bwd_pg_floats() = %lld floats)", (long long)grid * 2 * C);
  }
  auto go = [&](auto kern, auto kern_pg) {
    if (dgamma)
      T2V_LAUNCH_FIRST(kern_pg, dim3(grid), dim3(256), 2 * C * 4, (hipStream_t)stream, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy,
                       (bf16_t*)dx, lddx, rows, C, gamma, stats, pg, dbeta, (const bf16_t*)addend, ldadd);
    else
      T2V_LAUNCH(kern, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, (bf16_t*)dx,
                 lddx, rows, C, gamma, stats, pg, dbeta, (const bf16_t*)addend, ldadd);
  };
  if (C <= 512) go(ln_bwd_kernel<1, 8, false>, ln_bwd_kernel<1, 8, true>);
  else if (C <= 1024) go(ln_bwd_kernel<2, 4, false>, ln_bwd_kernel<2, 4, true>);
  else go(ln_bwd_kernel<4, 2, false>, ln_bwd_kernel<4, 2, true>);
  T2V_CHECK_LAUNCH();
  if (dgamma) {
    T2V_LAUNCH_LAST(param_grad_reduce_kernel, dim3((2 * C + 31) / 32), dim3(256), 0, (hipStream_t)stream, (const float*)pg, grid, C, dgamma, dbeta);
    T2V_CHECK_LAUNCH();
  }
  return T2V_OK;
}
