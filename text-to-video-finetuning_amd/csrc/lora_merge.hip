// lora_merge.hip — effective weights of every LoRA-wrapped layer in ONE launch (t2v_lora_merge, include/t2v_abi.h).
//
//   W_eff[n, tap, c] = W[n, tap, c] + scale * sum_j U[j, n] * D[j, tap, c]          (utils/lora.py:57-62,134-139,211-216:
//   base(x) + scale * up(down(x)) == x (*) W_eff^T with dropout off and the identity selector — the same merge the
//   reference's collapse_lora performs at save time, utils/lora.py:781-815)
//
// written as bf16 in the two layouts the implicit-GEMM kernels stream:
//   forward        Wf[n, tap*Cp + c]
//   backward-data  Wb[c, (taps-1-tap)*Np + n]      (flipped taps)
// from the fp32 master of the frozen base weight (GEMM layout) and the fp32 LoRA factors in the trainer's flat buffer:
// one rounding of the exact sum, so factors far below one bf16 ulp of W still move W_eff in expectation.
//
// HBM-bound streaming pass: per 64x64 tile 16 KiB fp32 in, 2 x 8 KiB bf16 out, rank x 4096 fp32 FMAs on the VALU.
// A step's table (574 layers of the ModelScope UNet = ~3.5e5 tiles) is one grid; tile -> job through a device map.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

constexpr int TN = 64, TC = 64, RMAX = 32;

__global__ __launch_bounds__(256) void lora_merge_kernel(const T2VLoraMergeJob* __restrict__ jobs,
                                                         const int* __restrict__ tile_job) {
  __shared__ __attribute__((aligned(16))) float sU[RMAX][TN];
  __shared__ __attribute__((aligned(16))) float sD[RMAX][TC];
  __shared__ __attribute__((aligned(16))) unsigned short sT[TC][TN + 8];   // transposed bf16 tile (row pitch 144 B)
  const int tid = threadIdx.x;
  const int jb = tile_job[blockIdx.x];
  const T2VLoraMergeJob J = jobs[jb];
  int t = (int)blockIdx.x - J.tile0;
  const int ntc = (J.Cp + TC - 1) / TC, ntn = (J.Np + TN - 1) / TN;
  const int tap = t / (ntn * ntc);
  t -= tap * ntn * ntc;
  const int n0 = (t / ntc) * TN, c0 = (t % ntc) * TC;
  const long long K = (long long)J.taps * J.Cp;
  // stage the factor tiles (zero beyond the layer's extent)
  for (int i = tid; i < J.rp * TN; i += 256) {
    const int j = i / TN, n = i - j * TN;
    sU[j][n] = (n0 + n < J.Np) ? J.up[(long long)j * J.ldu + n0 + n] : 0.f;
  }
  for (int i = tid; i < J.rp * TC; i += 256) {
    const int j = i / TC, c = i - j * TC;
    sD[j][c] = (c0 + c < J.Cp) ? J.down[(long long)j * K + (long long)tap * J.Cp + c0 + c] : 0.f;
  }
  __syncthreads();
  const int cc = (tid & 7) * 8;            // this thread's 8-column chunk
  const bool cok = c0 + cc < J.Cp;         // Cp % 8 == 0: a chunk is entirely inside or outside
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int nl = (tid >> 3) + 32 * half, n = n0 + nl;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    const bool ok = cok && n < J.Np;
    if (ok) {
      for (int j = 0; j < J.rp; ++j) {
        const float u = sU[j][nl];
        const float4 d0 = *(const float4*)&sD[j][cc], d1 = *(const float4*)&sD[j][cc + 4];
        v[0] += u * d0.x; v[1] += u * d0.y; v[2] += u * d0.z; v[3] += u * d0.w;
        v[4] += u * d1.x; v[5] += u * d1.y; v[6] += u * d1.z; v[7] += u * d1.w;
      }
      const float* wp = J.w32 + (long long)n * K + (long long)tap * J.Cp + c0 + cc;
      const float4 w0 = *(const float4*)wp, w1 = *(const float4*)(wp + 4);
      v[0] = w0.x + J.scale * v[0]; v[1] = w0.y + J.scale * v[1]; v[2] = w0.z + J.scale * v[2]; v[3] = w0.w + J.scale * v[3];
      v[4] = w1.x + J.scale * v[4]; v[5] = w1.y + J.scale * v[5]; v[6] = w1.z + J.scale * v[6]; v[7] = w1.w + J.scale * v[7];
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(v[e]);
      *(bf16x8*)((bf16_t*)J.wf + (long long)n * J.ldwf + (long long)tap * J.Cp + c0 + cc) = o;
#pragma unroll
      for (int e = 0; e < 8; ++e) sT[cc + e][nl] = (unsigned short)o[e];
    }
  }
  if (!J.wb) return;
  __syncthreads();
  // transposed store: thread -> row c of the backward layout, 8-column chunk of n (Np % 8 == 0)
  const int nc = (tid & 7) * 8;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int cl = (tid >> 3) + 32 * half;
    if (c0 + cl < J.Cp && n0 + nc < J.Np) {
      const bf16x8 o = *(const bf16x8*)&sT[cl][nc];
      *(bf16x8*)((bf16_t*)J.wb + (long long)(c0 + cl) * J.ldwb + (long long)(J.taps - 1 - tap) * J.Np + n0 + nc) = o;
    }
  }
}


// ---- transposed bf16 copies of the factors for the rank-wide epilogue term of the GEMM kernels (T2VGemm.lr_b): with the
// wrappers' dropout active the branch cannot be merged into W; the base layer's launch then adds s mask (t U^T) (forward) and
// s dt (*) D^T (backward-data) itself and wants the factor ROW-per-output-column, ranks contiguous:
//   upT[n, j]            = U[j, n]                                   [Np, rk]          (rk = 16 or 32: ranks padded with zeros)
//   dnT[c, tap*rk + j]   = scale * D[j, (taps-1-tap)*Cp + c]         [Cp, taps*rk]     (flipped taps: the backward-data window)
// One launch for all layers: chunk (8 ranks of one row) -> job by binary search over the jobs' first chunks.
__global__ __launch_bounds__(256) void lora_prep_kernel(const T2VLoraPrepJob* __restrict__ jobs, int njobs, long long total) {
  const long long ch = (long long)blockIdx.x * 256 + threadIdx.x;
  if (ch >= total) return;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].chunk0 <= ch) lo = mid;
    else hi = mid - 1;
  }
  const T2VLoraPrepJob J = jobs[lo];
  long long r = ch - J.chunk0;
  const int cpr_u = J.rk / 8;
  const long long nu = (long long)J.Np * cpr_u;
  bf16x8 o;
  if (r < nu) {                                    // up: row n, ranks j0 .. j0+7
    const int j0 = (int)(r / J.Np) * 8, n = (int)(r % J.Np);             // (consecutive threads: consecutive n — coalesced reads)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (j0 + e < J.rp) ? (short)f2bf(J.up[(long long)(j0 + e) * J.ldu + n]) : (short)0;
    *(bf16x8*)((bf16_t*)J.upT + (long long)n * J.rk + j0) = o;
    return;
  }
  r -= nu;
  const long long per_tap = (long long)J.Cp * (J.rkd / 8);
  const int tap = (int)(r / per_tap);
  r -= (long long)tap * per_tap;
  const int j0 = (int)(r / J.Cp) * 8, c = (int)(r % J.Cp);
  const long long K = (long long)J.taps * J.Cp;
#pragma unroll
  for (int e = 0; e < 8; ++e)
    o[e] = (j0 + e < J.rp) ? (short)f2bf(J.scale * J.down[(long long)(j0 + e) * K + (long long)(J.taps - 1 - tap) * J.Cp + c]) : (short)0;
  *(bf16x8*)((bf16_t*)J.dnT + (long long)c * J.ldt + (long long)tap * J.rkd + j0) = o;
}

}  // namespace

// Host-side planning: validates the jobs, assigns each its first tile (`tile0`) and fills the tile -> job map.
// Returns the total number of tiles (the grid of t2v_lora_merge), or a negative T2V_E* code.  With tile_job == NULL only
// counts.  The caller uploads both arrays to the device once; they stay valid while the buffers they point to do.
extern "C" long long t2v_lora_merge_plan(T2VLoraMergeJob* jobs, int njobs, int* tile_job, long long capacity) {
  T2V_CHECK_ARG(jobs && njobs > 0, "t2v_lora_merge_plan: no jobs");
  long long total = 0;
  for (int i = 0; i < njobs; ++i) {
    T2VLoraMergeJob& j = jobs[i];
    T2V_CHECK_ARG(j.w32 && j.up && j.down && j.wf, "t2v_lora_merge_plan: job %d has a null operand", i);
    T2V_CHECK_ARG(j.Np > 0 && j.Cp > 0 && j.Np % 8 == 0 && j.Cp % 8 == 0 && j.taps > 0 && j.rp > 0 && j.rp <= RMAX,
                  "t2v_lora_merge_plan: job %d: Np=%d Cp=%d must be multiples of 8, rank %d <= %d", i, j.Np, j.Cp, j.rp, RMAX);
    T2V_CHECK_ARG(j.ldwf % 8 == 0 && j.ldwf >= (long long)j.taps * j.Cp && ((uintptr_t)j.wf & 15) == 0 && ((uintptr_t)j.w32 & 15) == 0,
                  "t2v_lora_merge_plan: job %d: forward output must be 16-byte aligned, ldwf %% 8 == 0", i);
    T2V_CHECK_ARG(!j.wb || (j.ldwb % 8 == 0 && j.ldwb >= (long long)j.taps * j.Np && ((uintptr_t)j.wb & 15) == 0),
                  "t2v_lora_merge_plan: job %d: backward output must be 16-byte aligned, ldwb %% 8 == 0", i);
    const long long nt = (long long)((j.Np + TN - 1) / TN) * ((j.Cp + TC - 1) / TC) * j.taps;
    T2V_CHECK_ARG(total + nt < (1ll << 31), "t2v_lora_merge_plan: too many tiles");
    j.tile0 = (int)total;
    if (tile_job) {
      T2V_CHECK_ARG(total + nt <= capacity, "t2v_lora_merge_plan: tile map capacity %lld too small", capacity);
      for (long long t = 0; t < nt; ++t) tile_job[total + t] = i;
    }
    total += nt;
  }
  return total;
}

extern "C" int t2v_lora_merge(const T2VLoraMergeJob* jobs_dev, int njobs, const int* tile_job_dev, long long ntiles,
                              t2v_stream_t stream) {
  T2V_CHECK_ARG(jobs_dev && tile_job_dev && njobs > 0 && ntiles > 0 && ntiles < (1ll << 31), "t2v_lora_merge: bad arguments");
  hipLaunchKernelGGL(lora_merge_kernel, dim3((unsigned)ntiles), dim3(256), 0, (hipStream_t)stream, jobs_dev, tile_job_dev);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}

// chunks of one prep job (host side: the caller accumulates them into T2VLoraPrepJob.chunk0 and the launch's total)
extern "C" long long t2v_lora_prep_chunks(int Np, int Cp, int taps, int rk, int rkd) {
  return (long long)Np * (rk / 8) + (long long)Cp * taps * (rkd / 8);
}

extern "C" int t2v_lora_prep(const T2VLoraPrepJob* jobs_dev, int njobs, long long total_chunks, t2v_stream_t stream) {
  T2V_CHECK_ARG(jobs_dev && njobs > 0 && total_chunks > 0 && (total_chunks + 255) / 256 < (1ll << 31), "t2v_lora_prep: bad arguments");
  hipLaunchKernelGGL(lora_prep_kernel, dim3((unsigned)((total_chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs, total_chunks);
  T2V_CHECK_LAUNCH();
  return T2V_OK;
}
