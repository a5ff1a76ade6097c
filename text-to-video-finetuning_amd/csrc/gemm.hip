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
#include "common.h"

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

template <int BM, int BN, int WM, int WN, bool AT, bool BT>
__global__ __launch_bounds__(256) void gemm_kernel(const T2VGemm p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
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
  const int ntiles = gridDim.x;
  int t;
  {
    int q = ntiles >> 3, r = ntiles & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = t / ntn, tn = t - tm * ntn;
  const long long m0 = (long long)tm * BM;
  const int n0 = tn * BN;

  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* B = (const bf16_t*)p.B;
  int kbeg = 0, kend = p.K;
  const int z = blockIdx.z;
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
  if constexpr (!BT) {
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      int n = n0 + (tid >> 3) + 32 * i;
      bok[i] = n < N;
      brow[i] = B + (long long)(bok[i] ? n : 0) * p.ldb;
    }
  } else if (p.b_conv) {
    int nn = n0 + (tid % (BN / 8)) * 8;
    btap = nn / g.C;
    bc = nn - btap * g.C;
  }

  bf16x8 ra[NCA], rb[NCB];
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto load_tiles = [&](int k0) {
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
          ra[i] = v ? *(const bf16x8*)(A + sr * p.lda + c) : zero8;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NCA; ++i) ra[i] = (aok[i] && kok) ? *(const bf16x8*)(arow[i] + kidx) : zero8;
      }
    } else {
      constexpr int CPR = BM / 8, RPP = 256 / CPR;
      const long long mm = m0 + (tid % CPR) * 8;
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        int kk = k0 + tid / CPR + RPP * i;
        bool v = (kk < kend) && (mm < M);
        ra[i] = v ? *(const bf16x8*)(A + (long long)kk * p.lda + mm) : zero8;
      }
    }
    if constexpr (!BT) {
      const int kidx = k0 + (tid & 7) * 8;
      const bool kok = kidx < kend;
#pragma unroll
      for (int i = 0; i < NCB; ++i) rb[i] = (bok[i] && kok) ? *(const bf16x8*)(brow[i] + kidx) : zero8;
    } else {
      constexpr int CPR = BN / 8, RPP = 256 / CPR;
      const int nn = n0 + (tid % CPR) * 8;
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        int kk = k0 + tid / CPR + RPP * i;
        bool v = (kk < kend) && (nn < N);
        if (p.b_conv) {
          Pos ps = decompose(kk, kend, g);
          bool v2;
          long long sr = src_row(ps, btap, g, v2);
          v = v && v2;
          rb[i] = v ? *(const bf16x8*)(B + sr * p.ldb + bc) : zero8;
        } else {
          rb[i] = v ? *(const bf16x8*)(B + (long long)kk * p.ldb + nn) : zero8;
        }
      }
    }
  };

  auto store_tiles = [&](int stage) {
    unsigned char* sA = smem + stage * STAGE;
    unsigned char* sB = sA + A_BYTES;
    if constexpr (!AT) {
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        int row = (tid >> 3) + 32 * i;
        *(bf16x8*)(sA + row * 128 + ((((tid & 7) ^ ((row >> 1) & 7))) << 4)) = ra[i];
      }
    } else {
      constexpr int CPR = BM / 8, RPP = 256 / CPR;
#pragma unroll
      for (int i = 0; i < NCA; ++i) {
        int kr = tid / CPR + RPP * i;
        *(bf16x8*)(sA + (kr * LDA_T + (tid % CPR) * 8) * 2) = ra[i];
      }
    }
    if constexpr (!BT) {
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        int row = (tid >> 3) + 32 * i;
        *(bf16x8*)(sB + row * 128 + ((((tid & 7) ^ ((row >> 1) & 7))) << 4)) = rb[i];
      }
    } else {
      constexpr int CPR = BN / 8, RPP = 256 / CPR;
#pragma unroll
      for (int i = 0; i < NCB; ++i) {
        int kr = tid / CPR + RPP * i;
        *(bf16x8*)(sB + (kr * LDB_T + (tid % CPR) * 8) * 2) = rb[i];
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
  load_tiles(kbeg);
  store_tiles(0);
  __syncthreads();
  for (int it = 0; it < nt; ++it) {
    const int cur = it & 1;
    if (it + 1 < nt) load_tiles(kbeg + (it + 1) * BK);
    compute(cur);
    if (it + 1 < nt) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue
  const float* bias = (const float*)p.bias;
  const bf16_t* rowbias = (const bf16_t*)p.rowbias;
  const bf16_t* R = p.R ? (const bf16_t*)p.R + zoffR : nullptr;
  const float keep_scale = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int col = n0 + wc * (FN * 32) + j * 32 + (lane & 31);
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long row = m0 + wr * (FM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        float v = p.alpha * acc[i][j][r];
        if (p.drop_p > 0.f) {
          unsigned long long idx = ((unsigned long long)z * M + row) * N + col;
          v = drop_keep(p.drop_seed, idx, p.drop_p) ? v * keep_scale : 0.f;
        }
        v += bv;
        if (rowbias) v += bf2f(rowbias[(row / p.rows_per_rb) * p.ldrb + col]);
        if (p.act == T2V_ACT_SILU) v = silu_f(v);
        if (R) v += p.beta * bf2f(R[row * p.ldr + col]);
        const long long di = zoffD + row * p.ldd + col;
        if (p.out_mode == T2V_OUT_BF16)
          ((bf16_t*)p.D)[di] = f2bf(v);
        else if (p.out_mode == T2V_OUT_F32)
          ((float*)p.D)[di] = v;
        else
          atomicAdd((float*)p.D + di, v);
      }
    }
  }
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

// pick the tile that minimises (waves of workgroups) x (tile cost); ~2 workgroups resident per CU
template <bool AT, bool BT>
int dispatch(const T2VGemm& p, hipStream_t s) {
  const long long zdim = p.split_k > 1 ? p.split_k : (p.batch > 1 ? p.batch : 1);
  auto cost = [&](int bm, int bn, double eff) {
    long long tiles = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * zdim;
    long long waves = (tiles + 511) / 512;
    return (double)waves * bm * bn / eff;
  };
  double c0 = cost(128, 128, 1.0), c1 = cost(128, 64, 0.8), c2 = cost(64, 64, 0.55);
  if (c0 <= c1 && c0 <= c2) return launch<128, 128, 2, 2, AT, BT>(p, s);
  if (c1 <= c2) return launch<128, 64, 2, 2, AT, BT>(p, s);
  return launch<64, 64, 2, 2, AT, BT>(p, s);
}

}  // namespace

extern "C" int t2v_gemm(const T2VGemm* pp, t2v_stream_t stream) {
  T2V_CHECK_ARG(pp != nullptr, "t2v_gemm: null descriptor");
  const T2VGemm& p = *pp;
  T2V_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "t2v_gemm: bad dims M=%d N=%d K=%d", p.M, p.N, p.K);
  T2V_CHECK_ARG(p.A && p.B && p.D, "t2v_gemm: null operand");
  T2V_CHECK_ARG((p.a_trans && p.b_trans) || p.K % 8 == 0, "t2v_gemm: K=%d must be a multiple of 8", p.K);
  T2V_CHECK_ARG(p.lda % 8 == 0 && p.ldb % 8 == 0, "t2v_gemm: lda=%lld ldb=%lld must be multiples of 8", p.lda, p.ldb);
  T2V_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0, "t2v_gemm: operands must be 16-byte aligned");
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
  hipStream_t s = (hipStream_t)stream;
  if (!p.a_trans && !p.b_trans) return dispatch<false, false>(p, s);
  if (p.a_trans && p.b_trans) return dispatch<true, true>(p, s);
  if (!p.a_trans && p.b_trans) return dispatch<false, true>(p, s);
  t2v_set_error("t2v_gemm: a_trans=1 with b_trans=0 is not instantiated");
  return T2V_EINVAL;
}
