/*
 * storygen_hip.h — C ABI of libstorygen_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for StoryGen's
 * denoising hot path.
 *
 * The reference (haoningwu3639/StoryGen) is pure Python on diffusers 0.13.1 and has NO FFI of its own
 * (SURVEY §0 F1); its operator boundary is "whatever torch kernel an nn.Module call resolves to".  Each entry
 * point below therefore names the reference *call site(s)* it replaces (file:line under /root/reference) —
 * that is the interface a maintainer binds (ctypes stub in INTEGRATION.md).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types, no exceptions.
 *   - every function returns 0 on success or a negative SG_E* code; sg_last_error() gives the thread-local
 *     message.  Shapes/alignment are validated on the host before anything is enqueued.
 *   - all device work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no hidden
 *     allocation, no hidden synchronisation => every entry point is hipGraph-capturable.
 *   - the library owns no device memory.  Callers pass scratch explicitly (see *_workspace_bytes).
 *   - activations are fp16, channels-last: an image tensor is [B, H, W, C] with an explicit pixel stride
 *     `ld*` (elements) so channel-slices of wider buffers can be read/written in place; token matrices are
 *     [rows, cols] row-major with a row stride.  Statistics, softmax and all accumulators are fp32.
 *   - weights are fp16: linear [N, K] (PyTorch layout), conv3x3 [Cout, 3, 3, Cin] (KRSC; repacked by the
 *     host from PyTorch's [Cout, Cin, 3, 3]).
 *   - every pointer that is loaded 16 bytes at a time (activations, weights, residuals) must be 16-byte
 *     aligned and the matching ld/K/C a multiple of 8 elements; violations return SG_EINVAL.
 */
#ifndef STORYGEN_HIP_H
#define STORYGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_OK        0
#define SG_EINVAL   (-1)   /* bad shape / alignment / null pointer */
#define SG_EUNSUP   (-2)   /* valid request this build has no kernel for (e.g. head_dim not in {40,80,160}) */
#define SG_ELAUNCH  (-3)   /* hipLaunchKernel / hipGetLastError failed */
#define SG_EARCH    (-4)   /* device is not gfx950 */

typedef void* sg_stream_t;   /* hipStream_t */
typedef uint16_t sg_half;    /* IEEE binary16 storage */

int          sg_version(void);                 /* ABI version, currently 1 */
const char*  sg_last_error(void);              /* thread-local, never NULL */
/* gcnArch number of the current device (950 for MI355X); negative SG_* on failure. */
int          sg_device_arch(void);
/* number of compute units of the current device (256 on MI355X) */
int          sg_device_cus(void);

/* ------------------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C[M, N'] = epi( A[M,K] @ W[N,K]^T )
 * Replaces every nn.Linear / 1x1 nn.Conv2d on the path:
 *   CrossAttention.to_q/to_k/to_v/to_out[0]   model/attention.py:175-183,192-199,215-223 (+ residual adds :262,277,291-293)
 *   GEGLU.proj + FeedForward.net[2]           model/attention.py:381-393,342 (+ residual :300)
 *   Transformer2DModel.proj_in / proj_out     model/attention.py:59,83,101,121-123
 *   ResnetBlock2D.conv_shortcut (1x1)         diffusers resnet, instantiated model/unet_2d_blocks.py:331-344 etc.
 * epilogue (fp32, then one rounding to fp16):
 *   SG_EPI_LINEAR : v = acc + bias[n] + rowbias[(m / rows_per_batch) * rowbias_ld + n] + res1[m,n] + res2[m,n]
 *                   (every term optional); N' = N
 *   SG_EPI_GEGLU  : W rows (and bias) are stored interleaved in groups of 64: rows [64u, 64u+32) are "value"
 *                   outputs 32u..32u+31 and rows [64u+32, 64u+64) the matching "gate" outputs;
 *                   C[m, 32u+j] = (val + bias_v) * gelu_erf(gate + bias_g); N' = N/2 (exact erf GELU).
 * split_k > 1 needs `workspace` of sg_gemm_workspace_bytes(M, N, split_k) bytes (fp32 partial tiles, reduced
 * by a second kernel that applies the epilogue).  split_k == 0 lets the library choose: it only considers
 * splits whose partial tiles fit the workspace it was given (M*N*4 bytes per split), so any fixed scratch
 * buffer — or NULL to forbid splitting — is valid; sg_gemm_workspace_bytes(M, N, 0) is the most it can use.
 */
#define SG_EPI_LINEAR 0
#define SG_EPI_GEGLU  1

/* dtype flags of sg_gemm_desc / sg_conv3x3_desc: the UNet's residual stream is kept in fp32 while every MFMA
 * operand is fp16, so outputs and residual inputs can each be fp32 (float*) instead of fp16. */
#define SG_F_OUT_F32   1   /* C / y is float* (row stride still in elements) */
#define SG_F_RES1_F32  2   /* res1 is const float* */
#define SG_F_RES2_F32  4   /* res2 is const float* */

typedef struct sg_gemm_desc {
    const sg_half* A;  int64_t lda;
    const sg_half* W;  int64_t ldw;
    void*          C;  int64_t ldc;   /* fp16, or fp32 with SG_F_OUT_F32 */
    sg_half*       C2; int64_t ldc2;  /* optional second, fp16 copy of the output (NULL = none) */
    int32_t M, N, K;
    int32_t epilogue;                 /* SG_EPI_* */
    int32_t flags;                    /* SG_F_* */
    const sg_half* bias;              /* [N] or NULL */
    const float*   rowbias;           /* fp32 [batches, rowbias_ld] or NULL */
    int64_t        rowbias_ld;
    int32_t        rows_per_batch;    /* rows of A per rowbias row (>=1) */
    int32_t        split_k;           /* 0 = auto, 1 = none, >1 = forced */
    int32_t        tile_m, tile_n;    /* 0, 0 = library heuristic; else one of 256x128, 128x128, 256x64, 128x64, 64x128,
                                         64x64 (a tuning hint: results are identical up to fp32 summation order) */
    int32_t        tile_waves;        /* 0, or the tile's wave count: 64x64 per wave = (tile_m/64)*(tile_n/64); round 6: 4 with a 64x64 tile /
                                         8 with a 64x128 tile = the 32x32-per-wave latency kernel (mma_lat_kernel: deep LDS ring, for launches
                                         that are a short dependent chain; linear epilogue only — a GEGLU launch falls back to the heuristic).
                                         8 with a 512x128 or 256x256 tile = eight waves of 128x64 on a 2-stage ring (mma_fat_kernel: the
                                         throughput form for the largest launches; linear and GEGLU epilogues, no split-K).
                                         0, 0, -1 = heuristic tile but never the latency kernel.  Anything else is rejected.  With 0, 0, 0 the library
                                         picks it itself for GEMMs of at most 640
                                         64x64 tiles and 8 - 64 K slabs, paired launches excepted (sg_debug_set_option "lat_*"). */
    const void*    res1; int64_t ldr1;   /* fp16, or fp32 with SG_F_RES1_F32 */
    const void*    res2; int64_t ldr2;   /* fp16, or fp32 with SG_F_RES2_F32 */
    void*          workspace; size_t workspace_bytes;
    float*         stats;             /* optional: GroupNorm partial statistics of the output, see below (NULL = none) */
    int32_t        stats_batch_rows;  /* rows of C per image (B*HW rows = B images); tiles must not straddle images */
    /* LayerNorm folded into the GEMM that consumes it (BasicTransformerBlock norm1..norm4 -> to_q / to_k / to_v / ff.net.0,
     * model/attention.py:250,268,283,298), see "LayerNorm fold" below.  ln_mode 0 = none. */
    int32_t        ln_mode;           /* 1: rows of A are the normalised tokens; 2: rows of W are (the V^T = W_v X^T product) */
    int32_t        ln_parts;          /* K / 64 */
    float          ln_eps;
    const float*   ln_stats;          /* [tokens][P][2] fp32, P = ln_parts rounded up to even: per token and 64-channel block (sum, M2
                                         about the block mean); the padding block of an odd ln_parts is never used */
    const float*   ln_c;              /* fp32 [N] (mode 1) / [M] (mode 2):  c_n = sum_k (gamma (.) W)_nk  of the fp16-rounded folded weight */
    const float*   ln_d;              /* fp32 [N] / [M]:  d_n = sum_k beta_k W_nk (+ bias_n) */
    float*         ln_stats_out;      /* producer side: write the partials of THIS output, [M][P][2] with P = N/64 rounded up to even
                                         (NULL = none; needs N % 64 == 0) */
    void*          ln_guard;          /* optional device uint32 of sticky SG_LN_GUARD_* flags (atomic OR, never cleared by the library),
                                         or NULL: see "LayerNorm fold" below */
} sg_gemm_desc;

/* The two assumptions of the LayerNorm fold, checked where the numbers are (the reference's LayerNorm reads an fp32 tensor and has
 * neither limit, model/attention.py:250,268,283,298):
 *   SG_LN_GUARD_RANGE   a producer (ln_stats_out) wrote a 64-column block that may hold |x| >= 65504: its fp16 copy C2 was SATURATED
 *                       (C2 beside an fp32 C is always clamped to +-65504, never inf), so consumers of the copy see clipped values;
 *   SG_LN_GUARD_OFFSET  a consumer (ln_mode) met a token with |mean| / sigma > SG_LN_GUARD_RATIO: the fp16 rounding of the raw copy,
 *                       2^-11 |x|, is then more than SG_LN_GUARD_RATIO * 2^-11 = 7.8e-3 of the normalised value.
 * A caller that sees a flag should rerun the block with a LayerNorm launch (sg_layernorm_f16 on the fp32 tensor). */
#define SG_LN_GUARD_RANGE   1u
#define SG_LN_GUARD_OFFSET  2u
#define SG_LN_GUARD_RATIO   16.0f

int    sg_gemm_f16(const sg_gemm_desc* d, sg_stream_t stream);
/* GroupNorm statistics as an epilogue (north-star: "GroupNorm/SiLU ... as fused epilogue kernels"; the consumers are ResnetBlock2D
 * norm1 / norm2, Transformer2DModel.norm model/attention.py:55,99 and conv_norm_out model/unet_2d_condition.py:259-262,477-479):
 * with `stats` set, the fused linear epilogue also writes, for every output row tile t (T rows) and column n,
 *     stats[(t*2 + 0)*N + n] = sum over the tile's rows of C[m, n],   stats[(t*2 + 1)*N + n] = sum of C[m, n]^2
 * (final fp32 values: bias, row bias and residuals included, before any fp16 rounding; deterministic, no atomics), which
 * sg_groupnorm_nhwc_f16 accepts instead of its own statistics pass (sg_groupnorm_desc.pstats).  T is the launch's tile height; a
 * split-K launch writes them from its second (reduction) pass instead, with T = 256, 128 or 64 rows (needs N % 64 == 0).
 * sg_gemm_stats_tile_rows / sg_conv3x3_stats_tile_rows return T for a descriptor (the plan is a pure function of the descriptor),
 * or 0 when that launch cannot emit statistics (GEGLU, a tile that does not divide the image, split-K with N % 64 != 0) — in
 * which case a launch with `stats` set fails with SG_EINVAL.  Size: (M / T) * 2 * N floats (T >= 64). */
int    sg_gemm_stats_tile_rows(const sg_gemm_desc* d);
/* LayerNorm fold.  Instead of a LayerNorm launch between two GEMMs, the producer of the stream tensor x (proj_in, attn1.to_out,
 * attn2/3.to_out) also writes x's fp16 copy (C2) and, through ln_stats_out, per token and 64-channel block the sum and the M2
 * (sum of squared deviations from the block mean) of its FINAL fp32 values; the consumer runs on the raw copy with the weight
 * W' = gamma (.) W (rounded to fp16 once) and its epilogue evaluates
 *     LN(x) W^T + b  =  rstd_t (x W'^T - mean_t c) + d
 * with mean_t / rstd_t merged from the partials (Chan's parallel formula: every block is centred on its own mean first, so no
 * cancellation against the token mean; eps inside the square root as torch.nn.LayerNorm).  GEGLU applies it to values and gates
 * before the gate activation.  A GEMM with ln_mode never splits K.  Results equal the unfused LayerNorm -> GEMM up to the fp16
 * rounding of x and W' instead of LN(x) and W (tests/test_kernels_gpu.py::test_gemm_layernorm_fold). */
/* Two independent GEMMs in ONE launch (same results as two sg_gemm_f16 calls): the pairs of projections that share an
 * activation operand and are each too small to fill the chip — attn1.to_q|to_k with attn1.to_v (model/attention.py:250-262 via
 * CrossAttention), attn2.to_q with attn3.to_q (:266-276,281-290), attn3.to_k with attn3.to_v of a harvested context.  The two
 * problems run concurrently: their outputs must not overlap and, if both carry a workspace, the workspaces must be disjoint.
 * tile_m / tile_n of d0 apply to both (0, 0 = heuristic); falls back to two launches when a problem needs the generic kernel. */
int    sg_gemm_pair_f16(const sg_gemm_desc* d0, const sg_gemm_desc* d1, sg_stream_t stream);
size_t sg_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k);

/* ------------------------------------------------------------------------------------------------------------
 * 3x3 convolution, padding 1, NHWC fp16, as implicit GEMM on MFMA:
 *   y[b, oy, ox, co] = epi( sum_{ky,kx,ci} x[b, iy, ix, ci] * w[co, ky, kx, ci] )
 *   iy = oy*stride + ky - 1 (stride in {1,2}); with upsample2x the input is first nearest-neighbour upsampled
 *   by 2 (F.interpolate(scale_factor=2, mode="nearest")) — folded into the gather as iy>>1.
 * Replaces ResnetBlock2D.conv1/conv2 (diffusers resnet; model/unet_2d_blocks.py:331-344,461-474,552-564,
 * 688-700,222-234,252-264), Downsample2D.conv (stride 2; :361-368,478-485) and Upsample2D (interpolate + conv;
 * :582,705,656-658,730-732).  Epilogue = SG_EPI_LINEAR terms of sg_gemm_desc (bias; rowbias = the
 * time_emb_proj(silu(temb)) broadcast add of conv1; res1 = the block's residual/shortcut add of conv2),
 * indexed by output pixel m = (b*Ho + oy)*Wo + ox with rows_per_batch = Ho*Wo.
 * Cin must be a multiple of 64 (conv_in/conv_out have their own entry points).
 * x_padded = 1: `x` is the start of a buffer [B, H+2, W+2, Cin] whose one-pixel border is zero (interior pixel
 * (y, x) at row y+1, column x+1).  Padding then needs no predicate and the kernel streams its tiles with LDS-DMA
 * through a multi-stage pipeline; x_padded = 0 reads an unpadded [B, H, W, Cin] tensor with bounds checks.
 */
typedef struct sg_conv3x3_desc {
    const sg_half* x;  int64_t ldx;   /* [B, H, W, Cin], pixel stride ldx */
    const sg_half* w;                 /* [Cout, 3, 3, Cin] */
    void*          y;  int64_t ldy;   /* [B, Ho, Wo, Cout], pixel stride ldy; fp16, or fp32 with SG_F_OUT_F32 */
    int32_t B, H, W, Cin, Cout;
    int32_t flags;                    /* SG_F_OUT_F32 | SG_F_RES1_F32 */
    int32_t stride;                   /* 1 or 2 */
    int32_t upsample2x;               /* 0 or 1 (then stride must be 1) */
    int32_t x_padded;                 /* 0 or 1, see above */
    const sg_half* bias;              /* [Cout] or NULL */
    const float*   rowbias; int64_t rowbias_ld;   /* fp32 [B, rowbias_ld] or NULL */
    const void*    res1; int64_t ldr1;            /* [B, Ho, Wo, Cout] or NULL; fp16, or fp32 with SG_F_RES1_F32 */
    int32_t        split_k;           /* as in sg_gemm_desc */
    int32_t        tile_m, tile_n, tile_waves;   /* as in sg_gemm_desc */
    void*          workspace; size_t workspace_bytes;   /* sg_gemm_workspace_bytes(B*Ho*Wo, Cout, split_k) */
    float*         stats;             /* optional GroupNorm partial statistics of y, as sg_gemm_desc.stats (image = Ho*Wo rows) */
    int32_t        defer_reduce;      /* 1: a split-K launch stops after writing its fp32 partial tiles [splits][B*Ho*Wo][Cout] into the
                                         workspace: bias / rowbias / res1 / y are NOT applied or written — the consumer does it
                                         (sg_groupnorm_desc.split_*: the one-launch GroupNorm sums the slices in slice order while it
                                         loads its slab, so conv1 -> norm2 of a ResnetBlock2D at the 16x16 / 8x8 levels costs no
                                         second pass).  Only valid when sg_conv3x3_planned_splits(d) > 1 (else SG_EINVAL). */
} sg_conv3x3_desc;

int sg_conv3x3_nhwc_f16(const sg_conv3x3_desc* d, sg_stream_t stream);
int sg_conv3x3_stats_tile_rows(const sg_conv3x3_desc* d);
/* Number of K slices sg_conv3x3_nhwc_f16 will use for this descriptor (1 = no split-K; the plan is a pure function of the descriptor
 * and the development options).  No launch happens.  < 0 on an invalid descriptor. */
int sg_conv3x3_planned_splits(const sg_conv3x3_desc* d);
/* The decomposition a launch with this descriptor will use (host-only, nothing is launched): out[6] = {tile rows, tile columns,
 * K slices, workgroups, threads per workgroup, 1 if the LDS-DMA kernel applies else 0}.  For measurement tooling: a profiler
 * reports (kernel instantiation, grid size) classes, this tells which problems fall into which (tools/traffic_from_pmc.py puts
 * the ALGORITHMIC bytes per launch next to the measured HBM bytes of every class).  No reference counterpart (the reference
 * calls cuDNN / cuBLAS through torch and never sees a launch geometry). */
int sg_gemm_launch_plan(const sg_gemm_desc* d, int32_t* out);
int sg_conv3x3_launch_plan(const sg_conv3x3_desc* d, int32_t* out);

/* conv_in: x fp32 NCHW [B, Cin<=8, H, W] -> y NHWC [B,H,W,Cout] (fp16, or fp32 when y_f32), 3x3 pad 1
 * (unet_2d_condition.py:124,411).
 * w_kn: fp16 [9*Cin, Cout] with k = (ky*3+kx)*Cin + ci (host repack); bias fp16 [Cout]; Cout % 8 == 0. */
int sg_conv_in_f16(const float* x_nchw, const sg_half* w_kn, const sg_half* bias, void* y, int64_t ldy, int32_t y_f32,
                   int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, sg_stream_t stream);
/* conv_out: x fp16 NHWC [B,H,W,Cin] -> y fp32 NCHW [B, Cout<=4, H, W], 3x3 pad 1 (unet_2d_condition.py:268,480).
 * w: fp16 [Cout, 3, 3, Cin]; bias fp16 [Cout]; Cin % 8 == 0. */
int sg_conv_out_f16(const sg_half* x, int64_t ldx, const sg_half* w, const sg_half* bias, float* y_nchw,
                    int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused GEGLU feed-forward of a BasicTransformerBlock (round 4):
 *   y[m, :] = Linear2( a * gelu(g) ) + b2 + x[m, :],   [a | g] = Linear1( LayerNorm(x[m, :]) ) + b1
 * Replaces norm3 -> ff.net.0 (GEGLU: Linear(C, 8C), chunk, gelu) -> ff.net.2 (Linear(4C, C)) -> + hidden_states of
 * model/attention.py:298-300 (FeedForward / GEGLU: :365-393) in ONE launch: the [M, 4C] intermediate is never materialised and the
 * LayerNorm is evaluated in registers (exact two-pass).  x is the fp32 residual stream; y is fp16 (it only feeds proj_out).
 * wpack = the weight stream of storygen_amd/repack.ff_fused_pack(gamma (.) W1 interleaved, W1 beta + b1, W2): per 32 hidden units the
 * byte-for-byte LDS images of the W1 rows (values | gates, K-slab-wise, rows bit-permuted) and of W2's columns, followed by the chunk's
 * 64 fp32 bias terms (layout: repack.ff_fused_pack's docstring); sg_ff_fused_pack_bytes(C) bytes (0: C not supported).
 * Built for C = 320 (the 64x64 level, where M = B * 4096 rows give every wave 32 rows of its own); other widths return SG_EUNSUP and
 * the caller runs the two GEMMs.  GELU uses erf to 1.5e-7 (Abramowitz-Stegun 7.1.26). */
typedef struct sg_ff_desc {
    const float*   x;  int64_t ldx;      /* [M, C] fp32 */
    const void*    wpack; size_t wpack_bytes;
    const sg_half* b2;                   /* [C] bias of the second linear */
    sg_half*       y;  int64_t ldy;      /* [M, C] fp16 ([M, 2C] with hidden_split) */
    int32_t        M, C;
    float          eps;                  /* LayerNorm eps */
    int32_t        hidden_split;         /* 0: y = Linear2(geglu) + b2 + x in one workgroup per 128 tokens.  2 (round 5): two workgroups per 128
                                            tokens, each over half of the 4C hidden units; y is [M, 2C]: columns [0, C) = first half's sum
                                            + b2 + x, columns [C, 2C) = the second half's partial sum alone.  Their sum is the result;
                                            the consumer adds them for free by contracting [y_a | y_b] with [W | W] (proj_out,
                                            model/attention.py:121-123) — for launches too small to fill the chip (M <= 16 k tokens) */
} sg_ff_desc;

size_t sg_ff_fused_pack_bytes(int32_t C);
int sg_ff_geglu_fused_f16(const sg_ff_desc* d, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused attention (flash-style, online softmax in fp32, no mask, no dropout):
 *   O[b, i, h*D + :] = softmax_j( scale * Q[b,i,h,:] . K[kb,j,h,:] ) @ V[kb,j,h,:]
 * Replaces the attention core of CrossAttention for attn1 / attn2 / attn3 — the default CrossAttnProcessor
 * (baddbmm -> softmax -> bmm) or xformers.memory_efficient_attention selected at inference.py:58-64; call
 * sites model/attention.py:255-260,271-276,285-290.  Heads are interleaved in the channel dimension exactly
 * as head_to_batch_dim expects (head h = channels [h*D, (h+1)*D)), so Q/K are read straight from the
 * projection GEMM outputs and O is written heads-merged.  D in {40, 80, 160} (SD-1.5: C/8); any Nq, Nk >= 1.
 * V is passed TRANSPOSED: vt[kb][h*D + d][j], keys contiguous (row stride ldvt, batch stride bsvt) — the host gets it
 * for free by running the V projection with swapped operands (VT = Wv . X^T).  Every vt row must hold finite values
 * up to Nk rounded up to a multiple of 8 (ldvt >= that).
 * kv_batches: 0 or B = one K/V per query batch; 0 < kv_batches < B: query batch b reads K/V batch
 * (b < kv_batches ? b : b - (B - kv_batches)) — the image-conditioned CFG branches of the main pass share one
 * context.  Strides in elements: token stride ld*, batch stride bs*.
 */
typedef struct sg_attn_desc {
    const sg_half* q;  int64_t ldq, bsq;
    const sg_half* k;  int64_t ldk, bsk;
    const sg_half* vt; int64_t ldvt, bsvt;
    sg_half*       o;  int64_t ldo, bso;
    int32_t B, H, Nq, Nk, D;
    int32_t kv_batches;
    float   scale;
    /* Optional SHORT K / V rows (0 / NULL = none): the first kv2_batches of the kv_batches K/V rows live in (k2, vt2) — row j at
     * k2 + j bsk2, vt2 + j bsvt2, same ldk / ldvt — and hold Nk2 keys each; K/V row j >= kv2_batches is row j - kv2_batches of
     * (k, vt) with Nk keys.  StoryGen's main pass uses it for the zero-image CFG branch in `multi-image-condition` mode: its R
     * context frames are R copies of one feature map (model/pipeline.py:390-397,425-430), and softmax over R copies of the same keys
     * equals softmax over one copy, so that row carries HW keys instead of R * HW. */
    const sg_half* k2;  int64_t bsk2;
    const sg_half* vt2; int64_t bsvt2;
    int32_t Nk2, kv2_batches;
} sg_attn_desc;

int sg_attn_fwd_f16(const sg_attn_desc* d, sg_stream_t stream);

/* Two attentions with the same query geometry (B, H, Nq, D) in ONE launch: the text cross-attention (attention.py:271-276) and the
 * image cross-attention (:285-290) of one BasicTransformerBlock read different K / V^T and write different outputs, but neither
 * depends on the other, so their workgroups share a grid (the short text problem fills the tail of the long image one).  Results
 * are bit-identical to two sg_attn_fwd_f16 calls.  Descriptors of different geometry are accepted and run as two launches. */
int sg_attn_fwd_pair_f16(const sg_attn_desc* d0, const sg_attn_desc* d1, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU) over NHWC fp16, fp32 statistics.
 * Replaces nn.GroupNorm(32, C) [+ SiLU]: ResnetBlock2D.norm1/norm2 + nonlinearity, conv_norm_out + conv_act
 * (unet_2d_condition.py:259-262,477-479; eps 1e-5) and Transformer2DModel.norm (attention.py:55,99; eps 1e-6).
 * Two launches: per-(batch, pixel-chunk, group) shifted partial sums -> normalise+affine(+SiLU).
 * workspace: sg_groupnorm_workspace_bytes(B, groups) bytes.
 */
typedef struct sg_groupnorm_desc {
    const void*    x;  int64_t ldx;   /* [B, HW, C] channels-last, row stride ldx; fp16, or fp32 when x_f32 != 0 */
    int32_t        x_f32;
    int32_t        y_pad_w;           /* 0: y is [B, HW, C].  W > 0: y is the zero-bordered [B, H+2, W+2, C] input of
                                         sg_conv3x3 (x_padded = 1); only its interior is written */
    sg_half*       y;  int64_t ldy;
    sg_half*       xcopy; int64_t ldxc;   /* optional raw fp16 copy of x (operand of the 1x1 conv_shortcut), or NULL */
    const sg_half* gamma; const sg_half* beta;
    int32_t B, HW, C, groups;
    float   eps;
    int32_t silu;
    void*   workspace; size_t workspace_bytes;
    /* Statistics from the producers' epilogues (sg_gemm_desc.stats / sg_conv3x3_desc.stats) instead of a statistics pass over x.
     * Up to two sources, because x may be a channel concatenation [h | skip] (model/unet_2d_blocks.py:609,626,716) whose halves
     * were written by different launches: source i covers channels [pstats_c0[i], pstats_c0[i] + pstats_nc[i]) of x with
     * partials over pstats_rows[i] pixels each (pstats_nc = the producer's N).  The sources must cover all C channels.
     * pstats[0] == NULL: none (the kernel makes its own pass).  Only the wide variant uses them (sg_groupnorm_uses_pstats). */
    const float* pstats[2];
    int32_t pstats_rows[2], pstats_c0[2], pstats_nc[2];
    /* x as the UNREDUCED output of a split-K convolution (sg_conv3x3_desc.defer_reduce), one-launch variant only
     * (sg_groupnorm_is_fused): x[b, p, c] = sum over the split_count fp32 slices split_ws[z][b*HW + p][c] (slice order) + split_bias[c]
     * + split_rowbias[b][c] + split_res[b*HW + p][c] — the additions of the convolution's own second pass, in its order, so the
     * values are bit-identical to reduce-then-normalise.  split_out (optional) receives the reduced tensor (fp32, or fp16 when
     * split_out_f32 = 0; with an fp16 split_out — or none and split_round_f16 = 1 — the normalisation sees the fp16-rounded values,
     * exactly as if the tensor had been stored and read back).  x / ldx / x_f32 are ignored.  split_ws == NULL: x is a plain tensor. */
    const float*   split_ws; int32_t split_count;
    const sg_half* split_bias;
    const float*   split_rowbias; int64_t split_rowbias_ld;
    const void*    split_res; int64_t split_ldr; int32_t split_res_f32;
    void*          split_out; int64_t split_ldo; int32_t split_out_f32;
    int32_t        split_round_f16;
} sg_groupnorm_desc;

int sg_groupnorm_nhwc_f16(const sg_groupnorm_desc* d, sg_stream_t stream);
/* 1 when a GroupNorm of this shape runs as the one-launch kernel (slab of one (batch, group) in registers), else 0 */
int sg_groupnorm_is_fused(int32_t HW, int32_t C, int32_t groups);
size_t sg_groupnorm_workspace_bytes(int32_t B, int32_t groups);
/* 1 if a GroupNorm of this shape would consume producer statistics (the two-launch "wide" variant), 0 if it runs the
 * single-launch register-resident variant, which reads x once anyway: producers then need not emit any. */
int sg_groupnorm_uses_pstats(int32_t HW, int32_t C, int32_t groups);

/* LayerNorm over the last dim of x[M, C] (row stride ldx), eps, affine; optionally a second affine output from
 * the same statistics (norm2 and norm4 both normalise the post-self-attention state, attention.py:268,283).
 * x is fp16, or fp32 when x_f32 != 0 (the residual stream).
 * Replaces nn.LayerNorm at attention.py:188,206-212,225-229,234 (used :250,268,283,298). */
int sg_layernorm_f16(const void* x, int64_t ldx, int32_t x_f32, int32_t M, int32_t C, float eps,
                     const sg_half* gamma1, const sg_half* beta1, sg_half* y1, int64_t ldy1,
                     const sg_half* gamma2, const sg_half* beta2, sg_half* y2, int64_t ldy2, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Time embedding pieces (unet_2d_condition.py:392-398 and ResnetBlock2D.time_emb_proj).
 * sg_timestep_embed_f32: out[b, :] = [cos(t_b f_j) | sin(t_b f_j)] (flip_sin_to_cos) or [sin|cos]; `freqs` is the
 *   fp32 table exp(-ln(1e4) j/(half-shift)), j < dim/2, computed by the host exactly as diffusers' Timesteps.
 * sg_linear_rows_f32: y[b, n] = act_out( sum_k act_in(x[b,k]) * W[n,k] + bias[n] ) for a handful of rows
 *   (B <= 16), x/y fp32, W/bias fp16 — TimestepEmbedding.linear_1/linear_2 and all 22 time_emb_proj at once
 *   (W = row-concatenation).  act: 0 none, 1 SiLU.
 */
int sg_timestep_embed_f32(const float* t, const float* freqs, float* out, int32_t B, int32_t dim,
                          int32_t flip_sin_to_cos, sg_stream_t stream);
int sg_linear_rows_f32(const float* x, int64_t ldx, const sg_half* W, int64_t ldw, const sg_half* bias, float* y,
                       int64_t ldy, int32_t B, int32_t N, int32_t K, int32_t act_in, int32_t act_out,
                       sg_stream_t stream);
/* out[b, 0:N] = table[j, 0:N] for the row j whose table_keys[j] == keys[b] (exact fp32 comparison); rows of NaN when no key matches.
 * The denoising loop knows every timestep at prepare() time (pipeline.py:410-415: `timesteps`, and t // 10 for the reference frames), so
 * the chain above — Timesteps -> TimestepEmbedding -> 22 x time_emb_proj(silu(emb)) (unet_2d_condition.py:392-398, ResnetBlock2D) — is
 * evaluated once per distinct timestep with the two functions above and each UNet call reads its rows (bit-identical to recomputing:
 * those kernels treat rows independently).  N % 4 == 0; keys / table_keys / table / out fp32. */
int sg_lookup_rows_f32(const float* keys, int32_t B, const float* table_keys, int32_t T, const float* table, int64_t ldt,
                       float* out, int64_t ldo, int32_t N, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * fp8 (OCP e4m3) attention for head dim 40 — BASELINE config 5 ("768x768 latent, 5 prior-frame context, fp8 MFMA attention
 * path"): the D = 40 self- and image cross-attention core, model/attention.py:255-260,285-290 at the 96x96 level.
 * v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales, fp32 softmax; results differ from sg_attn_fwd_f16 by e4m3 rounding of
 * Q, K, V and P (3 mantissa bits): ~2-4e-2 relative on the attention output (tests/test_kernels_gpu.py states the bound).
 *   sg_attn_f8_bytes(B, H, N, transposed)  size of a packed operand image
 *   sg_attn_f8_pack   fp16 operand exactly as sg_attn_fwd_f16 takes it -> e4m3 image:
 *                     transposed = 0: Q or K  [B, N, H*40] (ld = token stride, bs = batch stride) -> [B][H][N][64] (bytes 40..63 zero)
 *                     transposed = 1: V^T     [B, H*40, >= N rounded up to 8] (ld = row stride)   -> [B][H][64][N rounded up to 64],
 *                                     keys permuted inside every 64-key tile into the MFMA's contraction order, keys >= N zero
 *   sg_attn_fwd_f8_d40  O[b, q, h*40 + d] fp16 from the three images; kv_batches as in sg_attn_desc. */
size_t sg_attn_f8_bytes(int32_t B, int32_t H, int32_t N, int32_t transposed);
int sg_attn_f8_pack(const sg_half* src, int64_t ld, int64_t bs, void* dst, int32_t B, int32_t H, int32_t N, int32_t transposed,
                    sg_stream_t stream);
int sg_attn_fwd_f8_d40(const void* q8, const void* k8, const void* vt8, sg_half* o, int64_t ldo, int64_t bso, int32_t B, int32_t H,
                       int32_t Nq, int32_t Nk, int32_t kv_batches, float scale, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Sampling-loop elementwise steps (model/pipeline.py:412-461), all fp32 NCHW [*,C,H,W] with `n` = elements per
 * sample.  Per-step scalars live in DEVICE memory (`coef`) so a captured hipGraph can be replayed for every step.
 * sg_add_noise_f32 (scheduler.add_noise, :419-427): out[u] = coef[2u]*src[u] + coef[2u+1]*noise[u % N] for the U
 *     stacked reference-pass inputs (zero-image and prior-frame latents; `noise` holds the N shared noise samples,
 *     :409), coef[2u..2u+1] = {sqrt(abar_t_u), sqrt(1-abar_t_u)} — each sample may sit at its own timestep.
 * sg_cfg_ddim_step_f32 (:457-461): eps = e_u + coef[0](e_i - e_u) + coef[1](e_a - e_i)   (eps3 = [e_u|e_i|e_a])
 *     x0 = (x - coef[3]*eps)/coef[2];  x <- coef[4]*x0 + coef[5]*eps
 *     (coef = {s_img, s_txt, sqrt(abar_t), sqrt(1-abar_t), sqrt(abar_prev), sqrt(1-abar_prev)}); updates
 *     `latents` in place and also writes the three-fold replicated UNet input `latents3` if non-NULL.
 */
int sg_add_noise_f32(const float* src, const float* noise, const float* coef, float* out, int32_t U, int32_t N,
                     int64_t n, sg_stream_t stream);
int sg_cfg_ddim_step_f32(const float* eps3, float* latents, float* latents3, const float* coef, int32_t N,
                         int64_t n, sg_stream_t stream);
/* sg_cfg_plms_step_f32: the same guidance combine followed by the PNDM / PLMS update of diffusers' PNDMScheduler with
 *     skip_prk_steps (model/pipeline.py:7-16 accepts it, ckpt/stable-diffusion-v1-5/scheduler/scheduler_config.json names
 *     it; call site :461): coef = {s_img, s_txt, A, Bc, w0, w1, w2, w3, slot_cur, slot1, slot2, slot3, push, use_kept,
 *     keep} (15 floats, integers stored as floats).  e' = w0 eps + w1 history[slot1] + w2 history[slot2] + w3 history[slot3];
 *     if push: history[slot_cur] <- eps (history = 4 x [N*n] ring of past guided epsilons); x_src = use_kept ? kept :
 *     latents; if keep: kept <- latents; latents <- A x_src - Bc e' (and the three-fold `latents3`, if non-NULL). */
int sg_cfg_plms_step_f32(const float* eps3, float* latents, float* latents3, float* history, float* kept,
                         const float* coef, int32_t N, int64_t n, sg_stream_t stream);

/* Strided, batched 2-D copy of rows: dst[b][r][0:cols] = src[b][r][0:cols] (cols % 8 == 0; strides in elements).
 * mode 0: fp16 -> fp16, 1: fp32 -> fp32, 2: fp32 -> fp16 (cast).
 * Replaces torch.cat([hidden, skip], dim=1) (unet_2d_blocks.py:609,626,716), the feature `.clone()`s
 * (attention.py:263; unet_2d_condition.py:428-429,445,468-470) and the token-axis concat of per-frame
 * features (pipeline.py:440-443) — each becomes a write into its slot of a preallocated buffer. */
int sg_copy_rows(void* dst, int64_t ldd, int64_t bsd, const void* src, int64_t lds, int64_t bss, int32_t batches,
                 int32_t rows, int32_t cols, int32_t mode, sg_stream_t stream);

/* x [B,H,W,C] (fp16, or fp32 when x_f32) -> interior of the zero-bordered fp16 buffer y [B,H+2,W+2,C] that
 * sg_conv3x3 (x_padded = 1) consumes; the border is never written (allocate it zeroed once).  Feeds the
 * Downsample2D / Upsample2D convolutions, whose input is a residual-stream tensor rather than a GroupNorm output. */
int sg_pad_cast_f16(const void* x, int64_t ldx, int32_t x_f32, sg_half* y, int64_t ldy, int32_t B, int32_t H, int32_t W,
                    int32_t C, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * The networks either side of the loop: CLIP text encoder and AutoencoderKL.  Their GEMMs, convolutions,
 * GroupNorms and LayerNorms are the entry points above; these are the remaining pieces.
 *
 * sg_softmax_rows_f16: p[m, j] = softmax_j(scale * s[m, j]), j < N, fp32 scores -> fp16 probabilities; columns [N, N rounded
 *   up to 8) of p are zero-filled (ldp >= that), so p is directly the A operand of the PV sg_gemm_f16.  With two sg_gemm_f16
 *   calls this is the single-head AttentionBlock of the VAE mid block (diffusers 0.13.1 models/attention.py AttentionBlock,
 *   instantiated by AutoencoderKL; reference call sites /root/reference/model/pipeline.py:198-205,392,401).
 * sg_attn_small_f16: O[b,i,h*D+:] = softmax_j(scale * Q[b,i,h,:].K[b,j,h,:] + key_bias[b,j] (+ causal mask j <= i)) V[b,j,h,:]
 *   for short sequences (T <= 128, D <= 64), heads interleaved in channels, V NOT transposed — CLIPAttention of the text
 *   encoder (transformers 4.27.4 modeling_clip.py; /root/reference/model/pipeline.py:137,183).  key_bias: fp32 [B, T] or NULL.
 * sg_act_rows_f16: in place x = x*sigmoid(1.702x) (SG_ACT_QUICK_GELU, CLIPMLP "quick_gelu") or erf GELU (SG_ACT_GELU).
 * sg_embed_tokens_f32: out[r,:] = tok[ids[r],:] + pos[r % T,:] (CLIPTextEmbeddings), fp32 tables and output; ids are not
 *   range-checked on the device (the host validates them).
 * sg_gaussian_sample_f32: out = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale — DiagonalGaussianDistribution.sample()
 *   followed by the 0.18215 factor (pipeline.py:392-393,401-402); noise == NULL gives mean * scale (.mode()).
 */
#define SG_ACT_QUICK_GELU 0
#define SG_ACT_GELU       1
int sg_softmax_rows_f16(const float* s, int64_t lds, sg_half* p, int64_t ldp, int32_t M, int32_t N, float scale,
                        sg_stream_t stream);
int sg_attn_small_f16(const sg_half* q, int64_t ldq, int64_t bsq, const sg_half* k, int64_t ldk, int64_t bsk, const sg_half* v,
                      int64_t ldv, int64_t bsv, sg_half* o, int64_t ldo, int64_t bso, const float* key_bias, int32_t B,
                      int32_t H, int32_t T, int32_t D, float scale, int32_t causal, sg_stream_t stream);
int sg_act_rows_f16(sg_half* x, int64_t ldx, int32_t M, int32_t N, int32_t act, sg_stream_t stream);
int sg_embed_tokens_f32(const int64_t* ids, const float* tok, const float* pos, float* out, int64_t ldo, int32_t rows, int32_t T,
                        int32_t C, sg_stream_t stream);
int sg_gaussian_sample_f32(const float* mean, const float* logvar, const float* noise, float* out, float scale, int64_t n,
                           sg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Optimizer step of the training loop: /root/reference/train_StorySalon_stage2.py:186-205 builds torch.optim.AdamW
 * or bitsandbytes' AdamW8bit over the trainable (attn3) parameters, :328-333 clips the global gradient norm and steps.
 *
 * sg_sumsq_f32: *out = sum_i x[i]^2, deterministic (two launches, `scratch` of sg_sumsq_scratch_floats() floats).
 * sg_adamw_f32: one tensor's torch.optim.AdamW update (decoupled weight decay, bias correction, no amsgrad) on fp32 states.
 * sg_adamw8bit: the same update with both moments stored as 8-bit codes in blocks of 2048 elements (per-block absmax, signed /
 *   unsigned dynamic-tree code books q_code1 / q_code2 of 256 ascending fp32 entries) — block-wise 8-bit Adam as published for
 *   bitsandbytes (Dettmers et al. 2021); code arrays of n bytes, absmax arrays of sg_adamw8bit_blocks(n) floats.
 * Gradients are first multiplied by grad_scale (1 / loss scale); with `sumsq` set (n_sumsq per-tensor sums of squares of the
 * UNSCALED-by-grad_scale gradients of ALL trainable tensors) they are also clipped to the global norm max_norm as
 * torch.nn.utils.clip_grad_norm_ does: g *= min(1, max_norm / (||g|| + 1e-6)) — on the device, no host read.  step counts from 1.
 */
typedef struct sg_adamw_desc {
    float*       param;  const float* grad;  int64_t n;
    float*       exp_avg; float* exp_avg_sq;                 /* sg_adamw_f32 */
    uint8_t*     code1;  uint8_t* code2;                      /* sg_adamw8bit: 8-bit states ...             */
    float*       absmax1; float* absmax2;                     /* ... their per-block scales ...             */
    const float* q_code1; const float* q_code2;               /* ... and the two 256-entry code books       */
    float        lr, beta1, beta2, eps, weight_decay;
    int32_t      step;
    float        grad_scale;
    const float* sumsq;  int32_t n_sumsq;  float max_norm;     /* NULL / 0 / 0 = no clipping */
} sg_adamw_desc;
size_t  sg_sumsq_scratch_floats(void);
int     sg_sumsq_f32(const float* x, int64_t n, float* out, float* scratch, sg_stream_t stream);
int     sg_adamw_f32(const sg_adamw_desc* d, sg_stream_t stream);
size_t  sg_adamw8bit_blocks(int64_t n);
int     sg_adamw8bit(const sg_adamw_desc* d, sg_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Backward pass of the stage-2 training step (BASELINE config 4; /root/reference/train_StorySalon_stage2.py:322-327:
 * accelerator.backward(loss) through the main UNet pass, weight gradients for the attn3 modules only, :170-177).
 * STATUS: compiled for gfx950, not yet run on hardware (round 1's GPU budget was spent before they were written);
 * formulas and layer order are pinned on the CPU by oracle/storygen_backward.py + tests/test_oracle_backward.py.
 * The contractions reuse the forward entry points: linear dgrad = sg_gemm_f16 with the transposed weight, weight
 * gradient dW[n,k] = sum_m dy[m,n] x[m,k] = sg_gemm_f16(A = dy^T, W = x^T) on sg_transpose_f16 outputs, convolution
 * dgrad = sg_conv3x3_nhwc_f16 with the 180-degree-rotated, channel-swapped weight (stride 2: on sg_zero_stuff_f16's
 * output; nearest-2x upsampling: followed by sg_sum2x2_f32).
 */

/* LayerNorm backward: out[M,C] (fp32) = res_scale * res + rstd * (g - mean(g) - xhat * mean(g * xhat)),
 * g = dy1 * gamma1 (+ dy2 * gamma2: norm2 and norm4 share their input, attention.py:268,283).  x fp16 / fp32 (x_f32),
 * dy fp16 / fp32 (dy_f32, both pairs alike); res optional.  The affine parameters are frozen (no dgamma / dbeta). */
int sg_layernorm_bwd_f16(const void* x, int64_t ldx, int32_t x_f32, const void* dy1, int64_t lddy1, const sg_half* gamma1,
                         const void* dy2, int64_t lddy2, const sg_half* gamma2, int32_t dy_f32, const float* res,
                         int64_t ldr, float res_scale, float* out, int64_t ldo, int32_t M, int32_t C, float eps,
                         sg_stream_t stream);

/* GEGLU backward (attention.py:381-393).  proj / dproj are [M, N8] in the 32/32-interleaved column layout of the
 * forward GEMM's GEGLU weight (blocks of 32 value columns followed by their 32 gate columns), du = d(val * gelu(gate))
 * is [M, N8/2]: dval = du * gelu(gate), dgate = du * val * gelu'(gate), exact erf GELU. */
int sg_geglu_bwd_f16(const sg_half* proj, int64_t ldp, const sg_half* du, int64_t ldu, sg_half* dproj, int64_t lddp,
                     int32_t M, int32_t N8, sg_stream_t stream);

/* GroupNorm(+SiLU) backward over NHWC: d->out = (res +) dx, dx = rstd * (g - mean_g(g) - xhat * mean_g(g * xhat)),
 * g = dy * act'(xhat * gamma + beta) * gamma.  Output either fp32 [B, HW, C] (out_f32, optional fp32 residual
 * gradient `res`) or fp16, optionally into the interior of the zero-bordered [B, H+2, out_pad_w+2, C] image the next
 * dgrad convolution reads.  Needs >= 8 channels per group and C <= 2560 (SG_EUNSUP otherwise). */
typedef struct {
    const void* x; int64_t ldx; int32_t x_f32;          /* the forward input [B, HW, C] */
    const void* dy; int64_t lddy; int32_t dy_f32;       /* gradient w.r.t. the GroupNorm(+SiLU) output */
    const sg_half* gamma; const sg_half* beta;
    const float* res; int64_t ldr;                       /* optional, fp32 output only */
    void* out; int64_t ldo; int32_t out_f32; int32_t out_pad_w;
    int32_t B, HW, C, groups;
    float   eps;
    int32_t silu;
    void*   workspace; size_t workspace_bytes;           /* sg_groupnorm_bwd_workspace_bytes(B, groups) */
} sg_groupnorm_bwd_desc;

int sg_groupnorm_bwd_nhwc_f16(const sg_groupnorm_bwd_desc* d, sg_stream_t stream);
size_t sg_groupnorm_bwd_workspace_bytes(int32_t B, int32_t groups);

/* Attention forward that also stores lse2[b, h, q] (fp32, [B, H, Nq]): the log2-domain log-sum-exp row,
 * max + log2(sum), from which the backward recomputes P = exp2(scale * log2(e) * S - lse2).  Same descriptor, same O. */
int sg_attn_fwd_lse_f16(const sg_attn_desc* d, float* lse2, sg_stream_t stream);

/* Attention backward (oracle/storygen_backward.py::attention_core_bwd):
 *   dV = P^T dO,  dP = dO V^T,  dS = P * (dP - delta),  dQ = scale * dS K,  dK = scale * dS^T Q,  delta = rowsum(dO * O).
 * sg_attn_bwd_prep_f32 packs ld2[b,h,q] = (lse2, delta); sg_attn_bwd_dq_f16 writes dQ token-major [B, Nq, H*D];
 * sg_attn_bwd_dkv_f16 writes dK and dV TRANSPOSED ([B][H*D][Nk], keys contiguous) — the layout the weight-gradient
 * GEMMs want as their A operand.  Inputs: q, k, v, dout token-major ([B, N, H*D], token stride ld*, batch stride bs*);
 * kt (dq) and qt, dot (dkv) are the transposed copies ([B][H*D][N], row stride ld*t) the host gets from projection GEMMs
 * with swapped operands.  No K/V batch sharing (training has no CFG).  D in {40, 80, 160}; dkv needs Nq % 8 == 0; dq
 * accepts any Nk (text: 77) provided every kt row is finite up to Nk rounded up to 8 (ldkt >= that).
 * Text cross-attention (frozen K/V inputs and weights) needs only the dq call. */
typedef struct {
    const sg_half* q;    int64_t ldq, bsq;
    const sg_half* qt;   int64_t ldqt, bsqt;     /* dkv only */
    const sg_half* k;    int64_t ldk, bsk;
    const sg_half* kt;   int64_t ldkt, bskt;     /* dq only */
    const sg_half* v;    int64_t ldv, bsv;       /* token-major (not the forward's V^T) */
    const sg_half* dout; int64_t lddo, bsdo;
    const sg_half* dot;  int64_t lddot, bsdot;   /* dkv only */
    const float*   ld2;                          /* [B, H, Nq][2] from sg_attn_bwd_prep_f32 */
    sg_half* dq;  int64_t lddq, bsdq;            /* dq output */
    sg_half* dkt; int64_t lddkt, bsdkt;          /* dkv outputs */
    sg_half* dvt; int64_t lddvt, bsdvt;
    int32_t B, H, Nq, Nk, D;
    float   scale;
} sg_attn_bwd_desc;

int sg_attn_bwd_prep_f32(const sg_half* o, int64_t ldo, int64_t bso, const sg_half* dout, int64_t lddo, int64_t bsdo,
                         const float* lse2, float* ld2, int32_t B, int32_t H, int32_t Nq, int32_t D, sg_stream_t stream);
int sg_attn_bwd_dq_f16(const sg_attn_bwd_desc* d, sg_stream_t stream);
int sg_attn_bwd_dkv_f16(const sg_attn_bwd_desc* d, sg_stream_t stream);

/* dst[c][m] = src[m][c], fp16 out, fp16 / fp32 in (M, C multiples of 8). */
int sg_transpose_f16(const void* src, int64_t lds, int32_t src_f32, sg_half* dst, int64_t ldd, int32_t M, int32_t C,
                     sg_stream_t stream);
/* The same for B images in one launch: dst[b][c][m] = src[b][m][c]; bs_src / bs_dst = batch strides in elements (multiples of 8).
 * The attention operands of a training step (K^T, Q^T, dO^T, V^T per batch row: /root/reference/model/attention.py:171-185 computes
 * them as strided views) are B such images. */
int sg_transpose_batched_f16(const void* src, int64_t lds, int64_t bs_src, int32_t src_f32, sg_half* dst, int64_t ldd, int64_t bs_dst,
                             int32_t B, int32_t M, int32_t C, sg_stream_t stream);
/* Backward of nearest-2x upsampling: dx[b,y,x,:] (+)= sum of the 2x2 block of du [B, 2H, 2W, C] (fp32). */
int sg_sum2x2_f32(const float* du, int64_t ldu, float* dx, int64_t ldx, int32_t B, int32_t H, int32_t W, int32_t C,
                  int32_t accumulate, sg_stream_t stream);
/* dy [B, Ho, Wo, C] -> interior of the zero-bordered fp16 image y [B, 2Ho+2, 2Wo+2, C] with dy on the even positions
 * and zeros elsewhere: the input of the stride-1 convolution that computes a stride-2 convolution's dgrad. */
int sg_zero_stuff_f16(const void* dy, int64_t lddy, int32_t dy_f32, sg_half* y, int64_t ldy, int32_t B, int32_t Ho, int32_t Wo,
                      int32_t C, sg_stream_t stream);
/* loss = mean(((pred - noise) * (1 - mask))^2) and its gradient d_pred (train_StorySalon_stage2.py:325), n elements. */
int sg_mse_grad_f32(const float* pred, const float* noise, const float* mask, float* d_pred, float* loss, int64_t n,
                    sg_stream_t stream);

/* Diagnostic: raw per-lane MFMA register dump used by tests/test_kernels_gpu.py::test_mfma_fragment_layout to pin the fragment layout
 * assumptions of the kernels above.  out: fp32 [64 lanes][16 regs] of D = A(32x16) @ B(16x32) with
 * A[i][k] = a[i*16+k], B[k][j] = b[k*32+j] loaded with the kernels' own lane mapping. */
int sg_debug_mfma_32x32x16(const sg_half* a, const sg_half* b, float* out, sg_stream_t stream);
/* Diagnostic for BASELINE config 5 (fp8 attention, not built): one v_mfma_scale_f32_32x32x64_f8f6f4 on raw operands — a, b:
 * [64 lanes][32 bytes] of fp8 e4m3, scale_a / scale_b: E8M0 exponents (127 = 1.0); out: fp32 [64 lanes][16 registers].
 * tools/probe_mfma_f8.py derives the (lane, byte) -> (row / column, k) operand maps from it on the device. */
int sg_debug_mfma_f8_32x32x64(const void* a, const void* b, float* out, int32_t scale_a, int32_t scale_b, sg_stream_t stream);
/* Test hook: force the GEMM/conv tile shape (bm, bn in {256x128, 128x128, 256x64, 128x64, 64x128, 64x64}; 0,0 = automatic) and
 * optionally disable the LDS-DMA pipelined kernel (no_pipe = 1), so the parity tests can cover every code path.
 * Process-global; not for production use. */
int sg_debug_set_tile(int32_t bm, int32_t bn, int32_t no_pipe);
/* Host-side self-test of the multiply-shift divisors the GEMM / conv prologues use instead of integer division (made per launch
 * from tile counts, image sizes, channel chunks): returns the number of mismatches against n / d over a sweep (0 = correct). */
int sg_debug_fastdiv_selftest(void);
/* Mainloop anatomy (development): the same launch as sg_gemm_f16 / sg_conv3x3_nhwc_f16, through an instrumented instantiation of
 * the pipelined kernel that stamps s_memtime around the phases of every 64-deep slab.  prof: 10 uint64 per wave,
 * [block][wave][10] = {slabs, vmcnt wait, barrier, first fragment reads + k-step 0, k-step 1 up to the DMA issue, DMA issue, rest
 * of the slab, prologue, epilogue, total} in shader cycles; prof_bytes >= blocks * waves * 80.  tools/anatomy.py prints it. */
/* Only in a library built with SG_BUILD_EXPERIMENTS (python -m storygen_amd.build --experiments); the product library carries no
 * instrumented kernels and answers SG_EINVAL. */
int sg_debug_gemm_anatomy(const sg_gemm_desc* d, void* prof, size_t prof_bytes, sg_stream_t stream);
int sg_debug_conv_anatomy(const sg_conv3x3_desc* d, void* prof, size_t prof_bytes, sg_stream_t stream);
/* the same for sg_ff_geglu_fused_f16: 8 uint64 per wave ([workgroup][wave][8]: iterations, vmcnt wait, barrier, DMA issue + d1, GEMM1 (+ GEGLU),
 * GEMM2, prologue, total cycles) */
int sg_debug_ff_anatomy(const sg_ff_desc* d, void* prof, size_t prof_bytes, sg_stream_t stream);
/* Development options — kernel-variant selectors for the tuning / anatomy tools and the parity tests.  Process-global; the
 * library never reads the environment (storygen_amd/ops.py maps the SG_* variables of its tools onto this call).  name / value:
 *   "tile_m", "tile_n"   force a GEMM / conv tile (same effect as sg_debug_set_tile)      "no_pipe", "no_split"  1 = disable
 *   "no_nmajor"          1 = M-major tile order everywhere
 *   "attn_sub2", "attn_prio", "attn_d80" (0..2), "attn_d160" (0..4; 4 = key-split workgroups at Nq <= 256, default)  attention instantiation selectors
 *   "gn_no_fused", "gn_wide", "gn_fused_max"                                               GroupNorm kernel selection
 *   "lat_tiles" (640; 0 = only on a tile hint), "lat_min_kt" (8), "lat_max_kt" (64), "lat_stages" (4 | 8), "lat_wide" (0 | 1: M <= "lat_wide_m"
 *   takes the 64x128 / 6-stage form), "lat_mask" (62: bit 1 = paired launches, off by default)   when a GEMM runs the 32x32-per-wave kernel
 *   "fat_m" (0 = off)    convolutions of at least this many output rows take the 128x64-per-wave tiles (512x128 / 256x256, eight waves,
 *                        2-stage ring) when no tile is hinted; tile hint (512, 128, 8) / (256, 256, 8) selects them per launch
 *   "reset"              every option back to its default */
int sg_debug_set_option(const char* name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* STORYGEN_HIP_H */
