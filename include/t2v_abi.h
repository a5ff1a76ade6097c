/* t2v_abi.h — C ABI of the MI355X-native denoising-step kernels (libt2v_hip.so).
 *
 * The reference (ExponentialML/Text-To-Video-Finetuning) has no FFI of its own: its hot path
 * reaches native code through PyTorch leaf modules.  This header IS the drop-in
 * boundary underneath the module-tree contract: every entry point replaces the native kernel
 * that one reference call site dispatches to (cited per function).  Conventions:
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer;
 *   - asynchronous on the given hipStream_t, no internal synchronisation, no allocation
 *     (graph-capture safe); caller owns all memory incl. workspaces.  t2v_gemm picks its tile from a shipped table
 *     (t2v_gemm_tune_import) or a heuristic and never synchronises; only a TUNING run (T2V_GEMM_AUTOTUNE=live) times
 *     candidates of unknown signatures on first use, to produce that table (t2v_gemm_tune_export);
 *   - returns 0 on success, negative T2V_E* on error; t2v_last_error() gives a thread-local text;
 *   - activations are "token matrices": row-major [rows, ld] bf16, channels contiguous
 *     (channels-last).  rows = images*H*W.  `ld` = row stride in ELEMENTS;
 *   - accumulation, statistics, softmax, loss and optimizer state are fp32.
 */
#ifndef T2V_ABI_H
#define T2V_ABI_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* t2v_stream_t; /* hipStream_t */

#define T2V_OK 0
#define T2V_EINVAL (-1)
#define T2V_ELAUNCH (-2)
#define T2V_ABI_VERSION 9   /* bumped whenever a struct layout or a signature changes (native.py checks it) */

int t2v_abi_version(void);
const char* t2v_last_error(void);

/* ---- sliding-window gather geometry (Conv2d 3x3/1x1/stride-2, Conv3d (3,1,1), their bwd-data) ----
 * Output position p=(n,oy,ox) on the Ho x Wo grid, tap (ky,kx):
 *   vy = oy*sy + ky - py ; vx = ox*sx + kx - px
 *   if tdiv==2: position is valid only if vy,vx even; vy/=2, vx/=2      (bwd-data of a stride-2 conv)
 *   valid iff 0<=vy<Hv && 0<=vx<Wv ; source row = (n*(Hv>>up) + (vy>>up))*(Wv>>up) + (vx>>up)
 * `up`=1 reads a nearest-2x-upsampled view of a half-resolution source (Upsample2D,
 * models/unet_3d_blocks.py:742,852).  The (3,1,1) temporal conv (TemporalConvLayer,
 * models/unet_3d_blocks.py:308-314) is the case n=b, y=frame, x=pixel, KH=3, KW=1, py=1, px=0. */
typedef struct {
  int C;          /* channels per tap (K = KH*KW*C) */
  int Hv, Wv;     /* (virtual) source grid */
  int Ho, Wo;     /* output grid */
  int KH, KW;
  int sy, sx, py, px;
  int tdiv;       /* 1 or 2 */
  int up;         /* 0 or 1 */
} T2VConvGeom;

enum { T2V_A_DENSE = 0, T2V_A_CONV = 1 };
enum { T2V_OUT_BF16 = 0, T2V_OUT_F32 = 1, T2V_OUT_F32_ATOMIC = 2 };
enum { T2V_ACT_NONE = 0, T2V_ACT_SILU = 1 };

/* D[M,N] = act( alpha * sum_k A[m,k]*B[n,k] + bias[n] + rowbias[m / rows_per_rb, n] ) + beta * R[m,n]
 *
 * Replaces (under torch autocast bf16, fp32 accumulate): F.linear / F.conv2d / F.conv3d of every
 * nn.Linear / nn.Conv2d / nn.Conv3d on the path (ResnetBlock2D.conv1/conv2/conv_shortcut/time_emb_proj,
 * TemporalConvLayer.conv1-4, Attention.to_q/k/v/out, GEGLU.proj, FeedForward, proj_in/out, conv_in/out,
 * Down/Upsample2D.conv — wired at models/unet_3d_blocks.py:295-361,458-513,597-628,693-744,827-854 and
 * models/unet_3d_condition.py:132-152,249-251), their backward-data passes, and — with a_trans/b_trans —
 * the weight-gradient reductions (dW[n,k] = sum_m dY[m,n] X[m,k]) incl. the LoRA factor gradients
 * (utils/lora.py:57-62,134-139,211-216).
 *   a_trans=0: A is [M, K] (lda) or, a_mode=CONV, gathered rows of a [rows, lda] source, K ordered (tap, c)
 *   a_trans=1: A is stored [K, M] (lda = stride between k rows)
 *   b_trans=0: B is [N, K] (ldb)      b_trans=1: B stored [K, N] (ldb); with b_conv=1 the k rows of B
 *              are gathered positions and N is ordered (tap, c)  (conv weight gradient)
 * K%8==0, lda/ldb/ldd/ldr %8==0 (16-byte chunks); M,N arbitrary (edge tiles are predicated).
 * batch>1: blockIdx.z batches with element strides; split_k>1 requires out_mode=F32_ATOMIC. */
typedef struct {
  int M, N, K;
  const void* A; long long lda; int a_mode; int a_trans;
  const void* B; long long ldb; int b_trans; int b_conv;
  T2VConvGeom geom;
  void* D; long long ldd; int out_mode;
  const void* bias;            /* fp32 [N] or NULL */
  const void* rowbias; long long ldrb; int rows_per_rb;   /* bf16 [M/rows_per_rb, N] or NULL */
  const void* R; long long ldr;  /* bf16 residual or NULL */
  float alpha, beta;
  int act;
  int batch; long long strideA, strideB, strideD, strideR;
  int split_k;
  /* optional dropout on (alpha*acc) BEFORE bias/residual — LoRA branch (utils/lora.py:49,119) */
  float drop_p; unsigned long long drop_seed;
  /* optional second weight/output block (b_trans=0 only): output columns n >= n_split use rows (n - n_split) of B2 and
   * are written, scaled by alpha only, to D2[m, n - n_split] — the LoRA down-projection rides in the base layer's launch
   * (one read of the activations for `linear(x)` and `lora_down(x)`, utils/lora.py:59-60).  n_split <= 0: unused. */
  const void* B2; long long ldb2; int n_split; void* D2; long long ldd2;
  /* b_trans=1 only: B is a conv weight stored [r, taps, C] and is read as the flipped-tap transpose
   * B[n=c, k=(tap', j)] = W[j, taps-1-tap', c]  (backward-data of a LoRA down conv without a transposed copy);
   * uses geom.C (= r) and geom.KH*geom.KW of the A gather. */
  int b_tapflip;
  /* optional K window of the second weight block: B2 row j holds K indices [b2_k0, b2_k0 + b2_klen) only (B2[j][k - b2_k0]),
   * zero elsewhere — a 1x1 projection riding in a windowed (conv gather) launch reads only the tap whose gathered row is
   * the row itself: `dt = dy U` inside the backward-data launch of a stride-1 conv.  b2_klen <= 0: B2 rows span all of K. */
  int b2_k0, b2_klen;
  /* optional caller-owned fp32 scratch (>= M*N*4 bytes): lets the library split K across workgroups for deep-K launches
   * with few output tiles (partials are accumulated in the scratch, a finalize pass applies the epilogue; the 8-wave kernels
   * reduce inside the launch and keep arrival counters in the first 64 KB, which must be ZERO when the scratch is first handed
   * over and are left zero by every launch).  ws_split is internal (set 0). */
  void* workspace; size_t workspace_bytes; int ws_split;
  /* internal (set 0): tile rasterisation chosen by the library — 0: an XCD owns a run of M-tiles x all N-tiles (activation
   * panels stay in its L2, every XCD streams the whole weight matrix); 1: an XCD owns a run of N-tiles x all M-tiles (each
   * weight column block is fetched by ONE XCD — the layers whose weights outweigh their activations: 8x8 / 4x4 levels) */
  int raster_n;
  /* dropout epoch: device counter folded into drop_seed (see t2v_set_dropout_epoch); NULL = the process-wide one (or none) */
  const unsigned long long* drop_epoch;
  /* optional column statistics of the stored output (GroupNorm fusion: the statistics pass of the F.group_norm that consumes
   * this layer's output — or, in backward, of the GroupNorm whose output this layer consumed — rides in the epilogue, see
   * t2v_gn_finish).  colsum: fp32 [4 + ceil(M / BMt) * Nb * 2], Nb = n_split > 0 ? n_split : N; word 0 = BMt (int: the tile
   * rows of the kernel that ran), word 1 = Nb, then per (tile row, column) two fixed-order sums over the tile's rows of the
   * bf16-ROUNDED stored values y:
   *   cs_mode 1: (sum y, sum y^2)
   *   cs_mode 2: y = dL/d(GroupNorm output); with xh = (cs_x - mean) * rstd of the forward statistics cs_sums [ndomains, G, 2]
   *              (a domain = cs_domain_rows consecutive rows), z = xh * gamma + beta, dz = y * silu'(z) if cs_silu else y:
   *              (sum dz * gamma, sum dz * gamma * xh) — the two sums of t2v_gn_bwd_stats.
   * Honoured only by the kernels t2v_gemm_colsum_rows() reports (ask first; 0 = not available for this descriptor, leave
   * colsum NULL); tiles never straddle a domain when BMt divides cs_domain_rows. */
  float* colsum; int cs_mode; int cs_domain_rows;
  const void* cs_x; long long cs_ldx; const float* cs_sums; const float* cs_gamma; const float* cs_beta; float cs_eps;
  int cs_G; int cs_silu;
  /* ABI v5 — optional rank-wide epilogue term: the LoRA branch of a wrapped layer folded into the base layer's launch
   * (utils/lora.py:57-62,134-139,211-216 and their autograd) when it cannot be merged into the weight, i.e. with the wrapper's
   * dropout active (the reference's default train mode, utils/lora.py:35,49,89,119):
   *   D[m, n] += lr_scale * mask(m, n) * sum_{tap, j} LA[src(m, tap), j] * LB[n, tap * lr_rk + j]
   * LB = lr_b, bf16 [N, lr_ldb]: row n = output column, K ordered (tap, rank), lr_rk = 16 (lr_rp <= 16) or 32 ranks per tap,
   * zero padded.  mask(m, n) = keep(lr_drop_seed, m * N + n) / (1 - lr_drop_p) of t2v_dropout_mask's protocol (1 when
   * lr_drop_p == 0), regenerated in the epilogue.  Needs alpha == 1, bf16 output, and an 8-wave kernel (t2v_gemm selects one).
   *   lr_mode 1: LA = lr_a, bf16 [rows, lr_lda], read from memory; lr_taps > 1: src(m, tap) = the source row `geom` (the A
   *              gather's own stride-1 same-size window) gives for output row m under `tap`, zero outside — the backward-data
   *              launch of a wrapped layer: dx = dy (*) W^T + s dt (*) D^T with dt = lr_a, LB = the flipped-tap transpose of D.
   *   lr_mode 2: LA = x (*) B2^T computed BY THIS LAUNCH: every column tile carries the lr_rp rows of B2 (bf16 [lr_rp, ldb2],
   *              the down factor in the layout of B) as extra weight rows behind its base columns; their accumulators t are
   *              rounded to bf16, written to D2 [M, ldd2] (the saved down-projection of the backward) and multiplied with LB in
   *              the epilogue: y = x W^T + s mask (t U^T) in one launch, with no pass over y.  N counts the base columns only;
   *              n_split must be 0, lr_taps 1.
   *   lr_mode 3 (ABI v8): the backward-data launch of a dropped LINEAR wrapper (or projection group) that forms its own dt:
   *              A = dy (dense), B = W^T layout, N = C_in; B2 = the up factor(s) rank-major, bf16 [lr_rp, ldb2] over K (members of a
   *              group side by side / block diagonal with the zeros stored), lr_rp <= 64 total ranks.  The launch computes
   *                dt[m, j] = 1/(1-p) sum_k keep(seed_i, m * W + (k - i W)) dy[m, k] B2[j, k]        (i = k / W, the member of column k)
   *              with W = lr_group_cols (the members' K width; 0: one member, W = K) — the mask each member drew in its forward
   *              launch (lr_drop_seed, lr_group_seed[0..1]) is applied to the A fragments of the rank columns' MFMAs, hashed in
   *              the K loop — writes dt (bf16) to D2 [M, ldd2] and adds dt LB^T (LB = lr_b [N, lr_ldb] = (s D)^T, the scale folded
   *              in: lr_scale must be 1) to D in its epilogue: dx = dy W + s dt D with NO separate t2v_lora_drop_dt launch.
   *              n_split must be 0, lr_taps 1, M * W < 2^34. */
  int lr_mode; int lr_rp; int lr_taps;
  const void* lr_a; long long lr_lda;
  const void* lr_b; long long lr_ldb;
  float lr_scale; float lr_drop_p; unsigned long long lr_drop_seed;
  /* ABI v6 — projection groups under lr_mode 2 (to_q / to_k / to_v of an attention read the same input: ONE launch with
   * N = n * lr_group_cols): the column tiles of member i = n0 / lr_group_cols carry rows [i*lr_rp, (i+1)*lr_rp) of B2 (the members'
   * down factors back to back), write their t to columns [i*lr_rp, ..) of D2, and draw the mask of member i from its OWN seed
   * (member 0: lr_drop_seed, members 1, 2: lr_group_seed[0..1]) with index m * lr_group_cols + (n - i*lr_group_cols) — every
   * member keeps the mask it has as a stand-alone layer.  0: no groups.  lr_mode 1 accepts lr_rp up to 48 (three members' dt
   * side by side: dx = dy_cat W_cat + s dt_cat D_cat). */
  int lr_group_cols; unsigned long long lr_group_seed[2];
  /* colsum mode 2 for a norm that DROPS behind its SiLU (TemporalConvLayer's GroupNorm -> SiLU -> Dropout(0.1),
   * models/unet_3d_blocks.py:312: t2v_gn_apply's drop_p): the incoming gradient is masked first, dz = keep(cs_drop_seed,
   * m * Nb + n) ? y / (1 - cs_drop_p) : 0 — what t2v_gn_bwd_stats does with its drop_p / drop_seed.  0: no dropout. */
  float cs_drop_p; unsigned long long cs_drop_seed;
  /* ABI v8 — keep-bit plane of the dropped LoRA branch (optional, may be NULL): the forward launch (lr_mode 2), which hashes the
   * mask of its output anyway, WRITES the keep bits; the backward-data launch of the same layer (lr_mode 3) READS them instead of
   * hashing the mask again inside its K loop.  For a mask matrix [M, W] (W % 32 == 0) the plane holds M * W / 8 bytes, fragment-
   * major: 16-bit word (j, row, h) at byte ((j * M + row) * 2 + h) * 2, j = column >> 5, h = 0 / 1; bit b of it = keep of column
   * 32 j + 8 (b >> 2) + 4 h + (b & 3) — the register order of the 8-wave kernels' transposed accumulators, so that a wave's 64
   * words of a fragment are one 128-byte line.  Projection groups: member i's plane (W = lr_group_cols) starts at byte
   * i * M * W / 8.  The plane is per FORWARD (the dropout epoch is already folded in): no seed is needed to read it. */
  void* lr_plane;
} T2VGemm;
/* Kernel selection is the library's: the shipped tile table / heuristic picks a 4-wave or an 8-wave tiled kernel; a plain dense
 * NN descriptor with M <= 96 and K % 64 == 0 (no rank columns, statistics, rank-wide term or batch; bf16 output) runs on the
 * skinny weight-streaming kernel (the CLIP text tower's 77-token layers, train.py:784-790; T2V_GEMM_SKINNY=0 switches it off) —
 * same arithmetic (fp32 accumulation, one bf16 rounding in the epilogue), fixed summation order. */
int t2v_gemm(const T2VGemm* p, t2v_stream_t stream);
/* Tile rows BMt of the kernel t2v_gemm will run for this descriptor if that kernel can emit `colsum`, else 0 (deterministic:
 * shipped tile table / heuristic; 0 during live tuning runs). */
int t2v_gemm_colsum_rows(const T2VGemm* p);
/* 1 if t2v_gemm runs this descriptor WITH its rank-wide epilogue term (lr_mode != 0: the 8-wave kernels' domain — K%64, window
 * C%64, bf16 output, 32-bit offsets; mode 3 additionally: dense A, <= 64 ranks, members of whole 64-column stages), else 0: the
 * caller then evaluates the LoRA branch with separate launches. */
int t2v_gemm_lr_ok(const T2VGemm* p);
/* Pinned launch on one of the 8-wave, one-workgroup-per-CU configurations of csrc/gemm_w8.hip (what t2v_gemm selects through
 * the tile table for lean descriptors: K%64==0, window C%64==0, bf16 output, no dropout / batch); for tuning runs, the
 * configuration probes (scripts/w8_probe.py) and the kernel tests.  cfg: configuration index (t2v_gemm_w8_configs() of them);
 * nstep: column step of the tile grid (multiple of 32, <= the configuration's BN; 0 = BN); splits: K splits reduced inside the
 * launch through `workspace` (first 64 KB = arrival counters, zero when the scratch is first handed over; the kernels leave
 * them zero) — clamped to what K and the scratch admit. */
int t2v_gemm_w8(const T2VGemm* p, int cfg, int nstep, int splits, t2v_stream_t stream);
int t2v_gemm_w8_configs(void);
/* Dropout epoch.  Every dropout-capable entry point (t2v_gemm epilogue, t2v_gn_apply / t2v_gn_bwd_*, t2v_lowrank_update_drop,
 * t2v_dropout_mask) takes its seed BY VALUE, so a captured HIP graph would replay the same masks every step.  With an epoch
 * registered, the launches issued afterwards also carry the ADDRESS of this 8-byte device counter and use
 * seed ^ splitmix64(*counter + 0x9E3779B97F4A7C15); the caller bumps the counter once per optimisation step (inside the graph), forward
 * and backward of one step read the same value.  NULL unregisters (seeds are used as given: the protocol of oracle/dropout.py). */
int t2v_set_dropout_epoch(const unsigned long long* device_counter);
/* Tuned tile table: text, one line per problem signature ("M N K a_mode n_split out_mode has_res batch KH KW sy tdiv up C
 * tile stages split").  export: writes at most `cap` bytes (NUL-terminated) and returns the size needed; import: merges the
 * lines into the table and returns the number of entries accepted. */
long long t2v_gemm_tune_export(char* buf, long long cap);
int t2v_gemm_tune_import(const char* text);
/* Measurement hook (bench.py's rooflines): hand over a pair of hipEvent_t that the NEXT instrumented entry point called by this
 * thread fills with its kernel's own begin / end timestamps (hipExtLaunchKernelGGL): t2v_gemm (single-kernel NN launches; a
 * two-kernel split-K launch leaves the pair untouched), t2v_gn_*, t2v_layernorm_*, t2v_attn_fwd, t2v_attn_bwd (start of its
 * first kernel, end of its last), t2v_lora_wgrad(_batch).  t2v_launch_timing_consumed() tells whether the pair was filled (1/0)
 * and clears it. */
int t2v_launch_timing_events(void* start_event, void* stop_event);
int t2v_launch_timing_consumed(void);
/* Two independent K-major problems (a_trans = b_trans = 1, fp32 atomic output) in ONE launch: the two LoRA factor
 * gradients dU = s t^T dy and dD = s dt^T x_col of a wrapped layer (backward of utils/lora.py:57-62). */
int t2v_gemm_pair(const T2VGemm* a, const T2VGemm* b, t2v_stream_t stream);

/* direct small-channel conv (Cin<=8 or Cout<=8): conv_in 4->320, conv_out 320->4
 * (models/unet_3d_condition.py:132-134,249-251), VAE conv_in 3->128, conv_out 512->8, quant_conv 8->8.
 * x: [rows, ldx] bf16 channels-last (or fp32 NCHW when x_nchw_f32=1), w: fp32 [Cout, KH, KW, Cin], y: bf16 [rows_out, ldy] */
typedef struct {
  const void* x; long long ldx; int x_nchw_f32;
  const float* w; const float* bias;
  void* y; long long ldy; int y_nchw_f32;
  int nimg, Cin, Cout; T2VConvGeom geom;
} T2VSmallConv;
int t2v_smallconv(const T2VSmallConv* p, t2v_stream_t stream);

/* ---- GroupNorm (+SiLU) over channels-last data.  Replaces F.group_norm(+F.silu) of ResnetBlock2D.norm1/2,
 * TemporalConvLayer.conv*[0:2], Transformer2DModel.norm, TransformerTemporalModel.norm, conv_norm_out
 * (models/unet_3d_condition.py:239-243,488-490).  A "domain" = the rows one statistic spans: H*W rows
 * (per-frame norms) or F*H*W rows (5-D temporal norms).  sums: fp32 [ndomains, G, 2] = (sum, sumsq), written (not
 * accumulated).  Statistics are reduced in a fixed order (no atomics): results are bit-reproducible.
 * workspace: fp32 scratch of t2v_gn_workspace_floats(ndomains, G) elements, ZERO-filled before its first use (it holds
 * per-split partials and the arrival counters of the last-block reduction; the kernels leave the counters zero). */
long long t2v_gn_workspace_floats(int ndomains, int G);
int t2v_gn_stats(const void* x, long long ldx, int ndomains, int rows_per_domain, int C, int G,
                 float* sums, float* workspace, t2v_stream_t stream);
/* statistics from the per-tile column sums a GEMM epilogue left in `colsum` (T2VGemm.colsum, either mode): sums[d, g, :] =
 * fixed-order sum over the tiles of domain d and the columns of group g.  Needs rows_per_domain % BMt == 0 (checked by the
 * caller against t2v_gemm_colsum_rows; the kernel writes NaN otherwise). */
int t2v_gn_finish(const float* colsum, int ndomains, int rows_per_domain, int C, int G, float* sums, t2v_stream_t stream);
int t2v_gn_apply(const void* x, long long ldx, void* y, long long ldy, int ndomains, int rows_per_domain, int C, int G,
                 const float* sums, const float* gamma, const float* beta, float eps, int silu,
                 float drop_p, unsigned long long drop_seed, t2v_stream_t stream);
/* backward: bsums fp32 [ndomains,G,2] = (sum dxh, sum dxh*xh), written; dgamma/dbeta fp32 [C] accumulate (may be NULL).
 * With dgamma/dbeta set (full finetune) the launch is followed by a fixed-order reduction of per-workgroup partial rows held in
 * `pg_workspace`: caller-owned fp32 scratch of t2v_gn_bwd_pg_floats(ndomains, rows_per_domain, C) elements (no initialisation
 * needed; NULL when dgamma/dbeta are NULL) — like every other scratch of this ABI the library never allocates it (ABI v5; v3
 * kept a library-owned hipMalloc buffer, whose first use inside a stream capture failed).  Same for t2v_layernorm_bwd with
 * t2v_layernorm_bwd_pg_floats(rows, C).  No float atomics: the parameter gradients are bit-reproducible. */
long long t2v_gn_bwd_pg_floats(int ndomains, int rows_per_domain, int C);
long long t2v_layernorm_bwd_pg_floats(int rows, int C);
int t2v_gn_bwd_stats(const void* x, long long ldx, const void* dy, long long lddy, int ndomains, int rows_per_domain,
                     int C, int G, const float* sums, const float* gamma, const float* beta, float eps, int silu,
                     float drop_p, unsigned long long drop_seed,
                     float* bsums, float* workspace, float* dgamma, float* dbeta, float* pg_workspace, t2v_stream_t stream);
int t2v_gn_bwd_apply(const void* x, long long ldx, const void* dy, long long lddy, void* dx, long long lddx,
                     int ndomains, int rows_per_domain, int C, int G, const float* sums, const float* bsums,
                     const float* gamma, const float* beta, float eps, int silu,
                     float drop_p, unsigned long long drop_seed, const void* addend, long long ldadd, t2v_stream_t stream);
/* `addend` (bf16 [rows, C], may be NULL) is added to dx: the gradient arriving through a pass-through use of x — the
 * residual branch `x + f(norm(x))` of ResnetBlock2D / TemporalConvLayer / the transformers — so that the sum autograd
 * would do in a separate pass rides in this one.  Same for t2v_layernorm_bwd. */

/* ---- LayerNorm over the last dim (BasicTransformerBlock.norm1/2/3).  stats: fp32 [rows,2] = (mean, rstd). */
int t2v_layernorm_fwd(const void* x, long long ldx, void* y, long long ldy, int rows, int C, const float* gamma,
                      const float* beta, float eps, float* stats, t2v_stream_t stream);
int t2v_layernorm_bwd(const void* x, long long ldx, const void* dy, long long lddy, void* dx, long long lddx, int rows,
                      int C, const float* gamma, const float* stats, float* dgamma, float* dbeta, float* pg_workspace,
                      const void* addend, long long ldadd, t2v_stream_t stream);

/* ---- scaled-dot-product attention core, head_dim 64, no mask (AttnProcessor2_0, train.py:138-139).
 * One kernel serves temporal self (S=F, strided over frames), spatial self (S=H*W) and cross (Sk=77)
 * attention through strides (ELEMENTS): element (batch b, position s, head h, dim d) of X lives at
 *   X + (b / x_bdiv) * x_bstride_hi + (b % x_bdiv) * x_bstride_lo + s * x_sstride + h*64 + d
 * lse: fp32 [nbatch, heads, Sq] log-sum-exp (saved for backward). */
typedef struct {
  const void* ptr; long long bstride_hi, bstride_lo, sstride; int bdiv;
} T2VAttnOperand;
typedef struct {
  int nbatch, heads, Sq, Sk; float scale;
  T2VAttnOperand q, k, v, o;
  float* lse;
  /* backward only */
  T2VAttnOperand d_o, dq, dk, dv;
  float* delta;   /* fp32 [nbatch, heads, Sq] workspace: rowsum(dO*O) */
  int causal;     /* ABI v7: 1 = query i sees keys 0..i only (self-attention, Sq == Sk): the CLIP text tower of train.py:784-790 */
} T2VAttn;
int t2v_attn_fwd(const T2VAttn* p, t2v_stream_t stream);
int t2v_attn_bwd(const T2VAttn* p, t2v_stream_t stream);

/* ---- the temporal self-attention unit of TransformerTemporalModel as ONE forward-only launch (ABI v9, csrc/temporal_fused.hip):
 *   out = x + softmax_F((LN(x) Wq^T)(LN(x) Wk^T)^T * scale)(LN(x) Wv^T) Wo^T + bo
 * = `norm1 -> attn1 -> + residual` / `norm2 -> attn2 -> + residual` of the BasicTransformerBlock inside every temp_attention and
 * transformer_in (reference: models/unet_3d_blocks.py:331-340,491-500,726-735, models/unet_3d_condition.py:147-152,407-411; the
 * leaf arithmetic is diffusers' BasicTransformerBlock / Attention with double_self_attention).  x / out: bf16 token matrices whose
 * rows are ordered (batch b, frame f, pixel): row = (b F + f) HW + pixel; a sequence is the F rows of one pixel.  wqkv: bf16
 * [3C, C] = the rows of to_q, to_k, to_v (nn.Linear layout, K contiguous); wo: bf16 [C, C] = to_out.0's weight with its INPUT
 * index permuted inside every group of 16: stored position 16 g + 8 a + 4 b + c holds input 16 g + 8 b + 4 a + c (a, b in {0, 1},
 * c in 0..3) — the order in which a lane of the O^T accumulators holds head dims, so that a fragment is one 16-byte LDS read;
 * bo, gamma, beta: fp32 [C] (bo may be NULL).  C = heads * 64.  Nothing is kept for a backward: the no-grad forward (sampling, train.py:908-958) only.
 * t2v_temporal_fused_ok: 1 if the library has a kernel for this width / clip length. */
typedef struct {
  const void* x; long long ldx;
  void* out; long long ldo;
  const void* wqkv; const void* wo;
  const float* bo; const float* gamma; const float* beta;
  float eps, scale;
  int B, F, HW, C;
  int ablate;      /* measurement only (scripts/temporal_fused_probe.py), 0 in every product call: bit 0 = no output pass (residual read,
                    * stores), bit 1 = no LayerNorm input pass, bit 2 = no weight traffic (LDS-DMA), bit 3 = no MFMA k-loops */
} T2VTemporalFused;
int t2v_temporal_fused_fwd(const T2VTemporalFused* p, t2v_stream_t stream);
int t2v_temporal_fused_ok(int C, int F);

/* ---- row softmax (VAE mid-block single-head attention, d=512: scores are materialised per frame through the
 * batched GEMM).  In place allowed.  y[r,:] = softmax(x[r,:cols]) ---- */
int t2v_softmax_rows(const void* x, long long ldx, void* y, long long ldy, long long rows, int cols, t2v_stream_t stream);
/* y = keep(seed, row*cols+col) ? x/(1-p) : 0 — the mask the GEMM epilogue applies to a LoRA branch (utils/lora.py:49,119) */
int t2v_dropout_mask(const void* x, long long ldx, void* y, long long ldy, long long rows, int cols, float p,
                     unsigned long long seed, t2v_stream_t stream);

/* rank-r update y[m,n] += scale * sum_j t[m,j] * U[j,n]  (bf16, r in {8,16,24,32,48,64,96}) — the LoRA up-projection
 * `lora_up(lora_down(x)) * scale` added onto the base layer's output (utils/lora.py:57-62) and the linear case of its
 * backward dx += dt D, as an HBM-bound streaming pass. */
int t2v_lowrank_update(void* y, long long ldy, const void* t, long long ldt, const void* U, long long ldu, long long M, int N,
                       int r, float scale, t2v_stream_t stream);

/* the same with dropout on the update (train mode of the wrappers, dropout_p = 0.1 by default: utils/lora.py:35,49,89,119):
 * y[m,n] += keep(seed, m*N + n) ? scale/(1-p) * (t U)[m,n] : 0 — the mask protocol of t2v_dropout_mask / the GEMM epilogue */
int t2v_lowrank_update_drop(void* y, long long ldy, const void* t, long long ldt, const void* U, long long ldu, long long M, int N,
                            int r, float scale, float drop_p, unsigned long long drop_seed, t2v_stream_t stream);

/* windowed rank-r update y[q, c] += scale * sum_tap sum_j t[p(q,tap), j] * D[j, tap*N + c]  (r in {8,16,24,32}; stride-1
 * same-size window of 3 or 9 taps, p(q,tap) as in T2VLoraWgrad) — the backward-data of a LoRA down conv, `dx += dt (*) D^T`
 * (autograd of utils/lora.py:134-139,211-216), streaming over dx with the window applied to the rank-wide operand. */
int t2v_lowrank_window_update(void* y, long long ldy, const void* t, long long ldt, const void* D, long long ldd,
                              const T2VConvGeom* geom, long long M, int N, int r, float scale, t2v_stream_t stream);

/* ---- LoRA factor gradients of one layer in one streaming launch (the weight-gradient half of
 * LoraInjectedLinear/Conv2d/Conv3d, utils/lora.py:57-62,134-139,211-216, as autograd derives it):
 *   dU[j, n]      += alpha * sum_m t[m, j]  * dy[m, n]                       (up factor, 1x1)
 *   dD[j, tap, c] += alpha * sum_q dt[p(q, tap), j] * x[q, c]               (down factor, same window as the base layer)
 * where p(q, tap) is the output position that reads input position q under `tap` (zero when it falls outside the image).
 * Both are K-major contractions over the rows; the activation operands (dy, x) are streamed ONCE — the window is applied
 * to the rank-wide operand dt instead of the 9x-replicated input gather a weight-gradient GEMM would do.
 * fp32 outputs are accumulated with atomics (like T2V_OUT_F32_ATOMIC).  Restrictions: rank padded to a multiple of 8
 * (<= 32); a conv geometry must be stride 1, same-size (Hv==Ho, Wv==Wo), tdiv==1, up==0, KH*KW in {1,3,9}. */
typedef struct T2VLoraWgrad {
  long long rows;                 /* activation rows (positions) */
  int rp;                         /* padded rank: 8, 16, 24 or 32 */
  int conv;                       /* 0: linear; 1: window geometry in `geom` */
  const void* t;   long long ldt;    /* bf16 [rows, rp]  saved down-projection */
  const void* dy;  long long lddy;   /* bf16 [rows, N] */
  int N;
  float* dU;       long long lddu;   /* fp32 [rp, lddu] */
  const void* dt;  long long lddt;   /* bf16 [rows, rp]  gradient of the down-projection output */
  const void* x;   long long ldx;    /* bf16 [rows, C]   layer input */
  int C;
  float* dD;       long long lddd;   /* fp32 [rp, lddd], column = tap*C + c */
  T2VConvGeom geom;
  float alpha;
  /* ABI v4: dropout on the LoRA branch (utils/lora.py:49,119).  drop_p > 0: the dU contraction reads mask (.) dy / (1-p) with
   * the mask of t2v_dropout_mask (index = row * N + column) regenerated on the fly — no masked copy of dy exists. */
  float drop_p;
  unsigned long long drop_seed;
} T2VLoraWgrad;
int t2v_lora_wgrad(const T2VLoraWgrad* p, t2v_stream_t stream);
/* dt[M, rp] = (mask (.) dy / (1-p)) U^T — gradient of the down-projection output of a layer whose LoRA branch is dropped
 * (autograd of utils/lora.py:57-62 with the dropout of :49 active).  U = bf16 [rp, ldu] rank-major (the up factor as the bank
 * stores it), mask = t2v_dropout_mask's (index = row * N + column), regenerated in registers: dy is read once, nothing of its
 * size is written.  rp in {8, 16, 24, 32}. */
int t2v_lora_drop_dt(const void* dy, long long lddy, const void* U, long long ldu, void* dt, long long lddt, long long M, int N,
                     int rp, float drop_p, unsigned long long drop_seed, t2v_stream_t stream);
/* the same for the nmem <= 3 members of a projection group in ONE launch: member i reads columns [i*N, (i+1)*N) of dy, the up
 * factor at U + i * u_member_stride (elements), its own seed, and writes columns [i*rp, (i+1)*rp) of dt */
int t2v_lora_drop_dt_group(const void* dy, long long lddy, const void* U, long long ldu, long long u_member_stride, void* dt,
                           long long lddt, long long M, int N, int rp, int nmem, float drop_p, const unsigned long long* seeds,
                           t2v_stream_t stream);
/* The same for MANY layers in one launch (the train step queues the descriptors of a backward pass and flushes them in a few
 * batches: 568 launches of ~15 us become streaming work without per-layer ramps and tails).  `host_staging` = pinned host
 * memory and `device_table` = device memory, each of t2v_lora_wgrad_batch_bytes(nlayers) bytes, owned by the caller and left
 * untouched until the launch has run (inside a stream capture the staging copy is a graph node that re-reads the staging
 * buffer at every replay: keep one staging/table pair per captured batch).  Every operand of every descriptor must stay
 * alive until the launch has run. */
long long t2v_lora_wgrad_batch_bytes(int nlayers);
int t2v_lora_wgrad_batch(const T2VLoraWgrad* descs, int nlayers, void* host_staging, void* device_table, long long table_bytes,
                         t2v_stream_t stream);

/* ---- LoRA merge: effective weights of all wrapped layers of a model in ONE streaming launch.
 *   W_eff[n, tap, c] = W[n, tap, c] + scale * sum_j U[j, n] * D[j, tap, c]
 * With dropout off and the identity selector, `base(x) + scale*up(down(x))` (utils/lora.py:57-62,134-139,211-216) equals
 * `x (*) W_eff^T` and its backward-data `dy (*) W_eff`; the reference performs the same merge in collapse_lora
 * (utils/lora.py:781-815).  The train step refreshes W_eff once per optimisation step from the fp32 master of the frozen
 * weight and the fp32 factors, so every wrapped layer runs as a plain N = C_out GEMM (no rank columns in the tile grid, no
 * rank-update passes).  Outputs: bf16 forward layout wf[n, tap*Cp + c] and (optional) backward-data layout
 * wb[c, (taps-1-tap)*Np + n].  All pointers are DEVICE pointers; the job table itself lives in device memory too. */
typedef struct T2VLoraMergeJob {
  const float* w32;            /* fp32 [Np, taps*Cp]  frozen base weight, GEMM (forward) layout, zero padded */
  const float* up;   long long ldu;   /* fp32 U[j, n] = up[j*ldu + n], j < rp (the bank stores the up factor transposed) */
  const float* down;           /* fp32 D[j, tap*Cp + c], j < rp */
  void* wf;  long long ldwf;   /* bf16 out, row stride in elements */
  void* wb;  long long ldwb;   /* bf16 out or NULL */
  int Np, Cp, taps, rp;        /* Np, Cp multiples of 8; rp <= 32 */
  float scale;
  int tile0;                   /* filled by t2v_lora_merge_plan */
} T2VLoraMergeJob;
/* host-side: validate `jobs` (host copy, device pointers inside), assign tile0, fill the tile->job map (or count only when
 * tile_job == NULL).  Returns the total tile count (grid size) or a negative error. */
long long t2v_lora_merge_plan(T2VLoraMergeJob* jobs, int njobs, int* tile_job, long long capacity);
int t2v_lora_merge(const T2VLoraMergeJob* jobs_dev, int njobs, const int* tile_job_dev, long long ntiles, t2v_stream_t stream);

/* ---- transposed bf16 copies of the LoRA factors of all wrapped layers in ONE launch: the operands of the rank-wide epilogue
 * term of t2v_gemm (T2VGemm.lr_b) when the wrappers' dropout is active and the branch cannot be merged into the weight.
 *   upT[n, j]          = U[j, n]                              bf16 [Np, rk]        (forward: y += s mask (t U^T))
 *   dnT[c, tap*rkd + j] = scale * D[j, (taps-1-tap)*Cp + c]   bf16 [Cp, taps*rkd]  (backward-data: dx += s dt (*) D^T)
 * rk = 16 (rp <= 16) or 32, ranks >= rp zero.  Device pointers; the job table lives in device memory.  chunk0 = sum of
 * t2v_lora_prep_chunks() of the jobs before this one. */
typedef struct T2VLoraPrepJob {
  const float* up;   long long ldu;   /* fp32 U[j, n] = up[j*ldu + n], j < rp */
  const float* down;                  /* fp32 D[j, tap*Cp + c], j < rp */
  void* upT; void* dnT;
  int Np, Cp, taps, rp, rk;
  float scale;
  long long chunk0;
  long long ldt;                      /* row stride of dnT in elements (>= taps*rkd; a projection group's members share rows) */
  int rkd;                            /* ranks per tap in dnT (multiple of 8, >= rp): rk for a layer of its own, the padded rank
                                         itself for the members of a projection group, whose blocks sit side by side */
} T2VLoraPrepJob;
long long t2v_lora_prep_chunks(int Np, int Cp, int taps, int rk, int rkd);
int t2v_lora_prep(const T2VLoraPrepJob* jobs_dev, int njobs, long long total_chunks, t2v_stream_t stream);

/* ---- elementwise ---- */
/* GEGLU gate: y[m, j] = x[m, j] * gelu_erf(x[m, inner + j])  (FeedForward/GEGLU) */
int t2v_geglu_fwd(const void* x, long long ldx, void* y, long long ldy, int rows, int inner, t2v_stream_t stream);
int t2v_geglu_bwd(const void* x, long long ldx, const void* dy, long long lddy, void* dx, long long lddx, int rows,
                  int inner, t2v_stream_t stream);
/* y = gelu(x) on a flat bf16 array and its backward — the MLP activation of the CLIP text tower (train.py:784-790; kind 0: exact
 * erf GELU, the ModelScope / OpenCLIP ViT-H text config; kind 1: quick_gelu x * sigmoid(1.702 x), OpenAI CLIP configs) */
int t2v_gelu_fwd(const void* x, void* y, long long n, int kind, t2v_stream_t stream);
int t2v_gelu_bwd(const void* x, const void* dy, void* dx, long long n, int kind, t2v_stream_t stream);
/* y[g, c] = sum over the `rows_per_group` consecutive rows of group g of x[., c]  (bf16 in, fp32 accumulate, bf16 out): the
 * gradient of a per-video row-bias (time embedding) and of keys / values shared by the frames of a video */
int t2v_rowgroup_sum(const void* x, long long ldx, void* y, long long ldy, int groups, int rows_per_group, int cols, float* workspace,
                     t2v_stream_t stream);
/* row splits the launch uses when `workspace` (caller-owned fp32 scratch of groups * t2v_rowgroup_splits() * cols elements, no
 * initialisation needed) is given; NULL workspace: one workgroup column per group */
int t2v_rowgroup_splits(int groups, int rows_per_group, int cols);
/* y = silu(x) on a flat bf16 array (temb activation, ResnetBlock2D) and its backward */
int t2v_silu_fwd(const void* x, void* y, long long n, t2v_stream_t stream);
int t2v_silu_bwd(const void* x, const void* dy, void* dx, long long n, t2v_stream_t stream);
/* strided 2-D copy/add of bf16 token matrices (skip concat torch.cat, models/unet_3d_blocks.py:764,861; grad accumulation) */
int t2v_copy2d(const void* x, long long ldx, void* y, long long ldy, int rows, int cols, int accumulate,
               t2v_stream_t stream);
/* 2x2 sum-pool of a [nimg, 2H, 2W, C] gradient into [nimg, H, W, C] (backward of nearest-2x upsample) */
int t2v_pool2x2_sum(const void* x, long long ldx, void* y, long long ldy, int nimg, int H, int W, int C,
                    t2v_stream_t stream);
/* layout/dtype conversion at the 4-channel boundary: NCHW-like fp32 [n, C, rows] <-> channels-last bf16 [n*rows, ld] */
int t2v_f32_planar_to_bf16_cl(const float* x, void* y, long long ldy, int n, int C, long long rows, t2v_stream_t stream);
int t2v_bf16_cl_to_f32_planar(const void* x, long long ldx, float* y, int n, int C, long long rows, t2v_stream_t stream);
/* fp32 <-> bf16 flat casts (weight preparation, gradient hand-off) */
int t2v_cast_f32_to_bf16(const float* x, void* y, long long n, t2v_stream_t stream);
int t2v_cast_bf16_to_f32(const void* x, float* y, long long n, int accumulate, t2v_stream_t stream);

/* ---- loss and optimizer (train.py:827, 868-879) ---- */
/* loss[0] += mean((pred-target)^2); dpred = 2*(pred-target)/n * gscale   (fp32, F.mse_loss) */
int t2v_mse_fwd_bwd(const float* pred, const float* target, long long n, float* loss, float* dpred, float gscale,
                    t2v_stream_t stream);
/* out[0] += sum(x^2) over a flat fp32 buffer (global grad-norm, accelerator.clip_grad_norm_).  Fixed-order two-stage
 * reduction (bit-reproducible: data-parallel replicas must clip identically); workspace: 2048 floats of caller scratch. */
int t2v_sumsq(const float* x, long long n, float* out, float* workspace, t2v_stream_t stream);
/* fused AdamW on flat fp32 buffers (torch.optim.AdamW semantics, train.py:238-249,598-604);
 * sumsq: device pointer to sum of squared grads (clip coefficient = min(1, max_norm/(sqrt(sumsq)+1e-6))) or NULL;
 * grads are scaled by grad_scale first (1/world_size after a SUM all-reduce). step = 1-based device counter incremented here. */
int t2v_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
              float wd, const float* sumsq, float max_norm, float grad_scale, int* step, t2v_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
