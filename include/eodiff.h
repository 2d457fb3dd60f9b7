/*
 * eodiff.h -- C ABI of libeodiff.so: the MI355X (gfx950) replacement for the stock-torch op call
 * sites on the EODiffusion hot path.
 *
 * Contract: stateless, stream-ordered entry points.  Every function returns 0 on success and a
 * negative EOD_E* code on failure (message via eod_last_error()).  The caller owns every buffer
 * (raw device pointers); nothing is allocated, freed or synchronised inside, so every call is
 * hipGraph-capturable.  `stream` is a hipStream_t passed as void*.  No torch types cross here.
 *
 * Activation layout inside the library is channels-last ("NHWC": [N][H][W][C], C contiguous) in
 * the storage dtype of the precision mode (EOD_F32: exact-fp32 MFMA; EOD_F16: fp16 storage,
 * fp16 MFMA, fp32 accumulate).  The reference API layout (NCHW fp32) is converted at the edges.
 *
 * Each entry point cites the reference call site (file:line in furio1999/EO_Diffusion) it replaces.
 */
#ifndef EODIFF_H
#define EODIFF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EOD_OK 0
#define EOD_EINVAL (-1)  /* bad shape / alignment / argument */
#define EOD_ELAUNCH (-2) /* HIP launch error */
#define EOD_ENOSYS (-3)  /* variant not built */

enum { EOD_F32 = 0, EOD_F16 = 1 };

const char* eod_last_error(void);
/* ABI revision of this header: bumped whenever an entry point's arguments, a descriptor layout or the size / meaning of a caller-provided
 * state buffer changes (round 3 -> 4: eod_gn_finalize, eod_gn_apply, eod_gn_bwd_apply, eod_attention_fwd_nat, the 4 + Cout slot weight-scale
 * buffer of eod_pack_conv_weight_split, the 4-int state of eod_adamw_step_guarded).  eod_version() returns the value the library was built
 * with; a binding compares it with the header it mirrors at load time (eo_diffusion_amd/_lib.py does) instead of finding out by an
 * out-of-bounds device write. */
#define EOD_ABI_VERSION 104
int eod_version(void);
/* Kernel-selection options ("skip_fuse", "head", "halo_bn256", "halo_splitk", "first": 1 / 0; "gn_fuse_max_cout": n, -1 = default; "halo_tpw":
 * pixel tiles per workgroup of the streaming halo instances, 1 = off = default, 0 = chosen per launch): every option has one
 * measured-best default, the other arm computes the same function on another kernel (same-box A/B runs, per-switch parity tests).
 * Read from the environment (EOD_SKIP_FUSE, EOD_HEAD, EOD_HALO_BN256, EOD_GN_FUSE_MAX_COUT, EOD_HALO_TPW, EOD_HALO_SPLITK, EOD_FIRST) at first use; returns the previous value,
 * or EOD_EINVAL for an unknown name.  Plans built before a change keep the kernels they were built with. */
int eod_set_option(const char* name, int value);
int eod_get_option(const char* name);
/* sizeof() of the descriptor structs as compiled (1 conv, 2 gemm, 3 temb, 4 small, 5 op): lets a
 * foreign-language binding verify its mirror of the layouts at load time. */
int eod_struct_size(int kind);

/* ------------------------------------------------------------------------------------------
 * k1/k2/k3/k4/k9/k10: implicit-GEMM convolution on MFMA.
 * Replaces nn.Conv2d call sites: ResBlock.in_layers[2]/out_layers[3] unet_openai.py:315,341;
 * skip_connection 1x1 :352; Downsample.op (stride 2) :262-264; Upsample.conv after nearest-2x
 * F.interpolate :227,236 (the 2x is virtual: upsample=1 indexes h>>1,w>>1); th.cat :773 is
 * virtual too (two A sources x | x2); first/last conv :609,742.
 * y[n,ho,wo,co] = alpha * sum_{tap,c} A(n, ho*stride-pad+dy, wo*stride-pad+dx, c) * w[tap][co][c]
 *                 + bias[co] + cbias[n*cbias_stride + co] + res[n,ho,wo,co]
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* x;       /* A source 0, NHWC [N][H][W][C0], storage dtype                      */
    const void* x2;      /* A source 1 (virtual concat) [N][H][W][C1], or NULL when C1 == 0     */
    const void* w;       /* packed weights [ksize*ksize][Cout][C0+C1], storage dtype            */
    const float* bias;   /* [Cout] fp32 or NULL                                                 */
    const float* cbias;  /* per-sample bias (timestep embedding, unet_openai.py:374-383) or NULL */
    const void* res;     /* residual, NHWC [N][Ho][Wo][Cout], storage dtype, or NULL (:385)      */
    void* y;             /* output: NHWC storage dtype (out_nchw_f32=0) or NCHW fp32 (=1)        */
    int64_t cbias_stride;
    int32_t dtype;       /* EOD_F32 | EOD_F16 */
    int32_t N, H, W;     /* input spatial dims as stored (before the virtual 2x upsample)        */
    int32_t C0, C1, Cout;
    int32_t ksize;       /* 1 or 3 */
    int32_t stride;      /* 1 or 2 */
    int32_t pad;         /* 0 or 1 */
    int32_t upsample;    /* 1: A is the nearest-2x upsampling of x (H,W are the stored dims);
                            2: A is x with zeros inserted between the pixels (z[2i][2j] = x[i][j]): the input of the
                            backward-data conv of a stride-2 conv (training path);
                            3: as 1, computed as four 2x2-tap parity classes with pre-summed weights (see eod_conv_up4_ok);
                            4: the backward-data of 3 (see eod_conv_up4_bwd_ok): x = dY on the (H x W) = (2 Ho x 2 Wo) grid */
    int32_t pad_tl;      /* 1: extra zero row/col on top/left after upsampling (3x3 -> 7x7 hack,
                            unet_openai.py:237-239)                                              */
    int32_t Ho, Wo;      /* output spatial dims */
    int32_t out_nchw_f32;
    float alpha;
    float* stats;        /* optional OUT: per-channel GroupNorm partial sums of y, [N][stats_slots][Cout][2] (sum, sumsq),
                            produced by the epilogue (saves the statistics read pass of the next GroupNorm); NULL = off */
    int32_t stats_slots; /* must equal eod_conv_stats_slots(d) when stats != NULL */
    int32_t gn_silu;     /* with gn_scale_shift: 1 = SiLU after the affine normalisation */
    const float* gn_scale_shift; /* optional: fuse GroupNorm32 (+FiLM) (+SiLU) of the conv INPUT (in_layers[0:2] / out_layers[0:2],
                            unet_openai.py:312-316,336-343): {scale, shift} per (image, input channel of the virtual concat),
                            [N][C0+C1][2] fp32 as written by eod_gn_finalize.  x / x2 are then the UN-normalised tensors; the
                            normalised activation is never written to HBM.  Only where eod_conv_gn_fusable(d) == 1. */
    void* workspace;     /* caller-owned scratch of eod_conv_workspace_size(d) bytes (split-K partial tiles of small maps) */
    int64_t workspace_bytes;
    int32_t w_tapmajor;  /* 1: thin-input 3x3 conv (the UNet's first conv, unet_openai.py:609: 3 / 7 / 13 image channels):
                            w is [Cout][ldk] with k = tap*C0 + c (eod_pack_conv_weight_tapmajor, ldk =
                            eod_conv_tapmajor_ldk) and the K loop runs over the flattened [tap][C0] axis -- ceil(9*C0/BK)
                            K-steps instead of 9 mostly-zero ones.  Needs C1 == 0, no upsample, C0 in {1,2,4} 16-byte chunks */
    int32_t w_split;     /* 1: "fp32x3" product -- fp32 storage, every product as three fp16 MFMAs on split operands (hi + lo):
                            w is the output of eod_pack_conv_weight_split, w_scale its scale pair.  Only where
                            eod_conv_split_ok(d) == 1 (fp32, C0 and C1 multiples of 8 -- or w_tapmajor with eod_pack_conv_weight_tapmajor_split);
                            rel. error ~2^-22 per product */
    const float* w_scale; /* device pointer to the scale buffer written by eod_pack_conv_weight_split (w_split only): float {s, 1/(16 s), -, -}
                           * followed by one int32 row exponent d_j per output row (4 + Cout slots; the parity-class form: 4 + 4 Cout):
                           * row j is packed under s * 2^d_j, the epilogue multiplies column j by 2^-d_j / (16 s) */
    const float* a_bound; /* w_split: bound table [N][32] fp32 (device) of the tensor the conv SPLITS -- x | x2 as the conv sees them, i.e. behind
                            the fused GroupNorm + SiLU when gn_scale_shift is set: entry maximum per image >= max|element|.  The kernel
                            derives a power-of-two operand scale per image from it (no host synchronisation), so the product is
                            fp32-grade for inputs of any magnitude.  Written by eod_gn_finalize (ab_norm / ab_raw) or eod_act_bound.
                            NULL = the caller guarantees |element| < 4094 (fixed scale 16) */
    /* ResBlock skip connection fused into out_layers' conv (unet_openai.py:352, 385: `return self.skip_connection(x) + h`): y gets
     * sum_c skip_w[co][c] * X(n, ho, wo, c) on top of the 3x3 conv, X = the block input (skip_x | skip_x2 as a virtual concat) at the
     * output resolution.  The skip tensor is never written: same accumulators, the K loop simply continues over X's channels.
     * skip_w = the 1x1 weight packed like `w` (eod_pack_conv_weight, ksize 1: [Cout][skip_C0 + skip_C1]; w_split: BOTH weights packed
     * by eod_pack_conv_weight_split_pair -- one common scale); the skip conv's bias can go through cbias with cbias_stride = 0.  Only where
     * eod_conv_skip_ok(d) == 1 (3x3 / stride 1 / halo-tile geometry, Cout > 64, C1 == 0, no `res`); NULL = off. */
    const void* skip_x;
    const void* skip_x2;
    const void* skip_w;
    int32_t skip_C0, skip_C1;
    const float* skip_bound; /* w_split: bound table [N][32] of skip_x | skip_x2 (see a_bound; the launch runs on the smaller of the two
                            scales because both phases feed one accumulator); NULL = |element| < 4094 guaranteed */
    int32_t x_presplit;  /* w_split, generic-kernel geometries (1x1, stride 2): x | x2 are PRE-SPLIT -- 4 bytes per element, every 8
                            channels as [8 x fp16 hi | 8 x fp16 lo] of s_n * x with s_n = the power-of-two scale that a_bound gives for
                            image n -- written by a producer that knew the bound beforehand (eod_gn_apply split_out, the fp32-storage
                            attention).  The conv DMAs the rows straight into its LDS image: no split pass per K-step. */
    const float* y_presplit_bound; /* w_split, generic-kernel geometries whose tiles lie inside one image (Ho*Wo % 128 == 0), Cout % 8 == 0:
                            write y PRE-SPLIT, scaled per image from this table [N][32] (an a-priori bound of y: eod_bound_affine; exponent
                            range of the attention kernels, s >= 2^-48) -- the qkv projection in front of eod_attention_fwd_nat(in_presplit).
                            NULL = plain fp32 output */
} eod_conv_desc;
int eod_conv2d_igemm(const eod_conv_desc* d, void* stream);
/* number of partial-sum slots per image the epilogue of this conv would write, or 0 if it cannot (tiles that straddle
 * images, NCHW output): the caller sizes `stats` with it. */
int eod_conv_stats_slots(const eod_conv_desc* d);
int eod_conv_gn_fusable(const eod_conv_desc* d);
int eod_conv_split_ok(const eod_conv_desc* d);
int eod_conv_skip_ok(const eod_conv_desc* d);
/* upsample = 3 (Upsample.conv, unet_openai.py:236-241, in 4/9 of the MACs): output pixel (2i+p, 2j+q) of the 3x3 conv over the nearest-2x
 * image reads only stored rows {i-1+p, i+p} and columns {j-1+q, j+q}, so each parity class (p, q) is a 2x2-tap conv of the STORED map with
 * summed taps (rows: p = 0 -> [w0 | w1+w2], p = 1 -> [w0+w1 | w2]; columns alike).  w is then ONE packed tensor (eod_pack_conv_weight /
 * _split) of a [4*Cout][Cin][3][3] weight whose row block 2p+q holds class (p, q)'s kernel in the tap slots (dy' in {p, p+1}, dx' in
 * {q, q+1}) and zeros elsewhere; Cout stays the real channel count.  1 where this form is available (fp16, or fp32 with w_split). */
int eod_conv_up4_ok(const eod_conv_desc* d);
/* the [4*Cout][Cin][3][3] fp32 class-kernel tensor of that form from the OIHW weight (device side: the training step re-forms it from
 * the live parameter every step, then packs it like any weight) */
int eod_conv_up4_weights(const float* w_oihw, float* wc, int Cout, int Cin, void* stream);
/* upsample = 4 (fp16): dX of that conv straight from dY -- per class (p, q) a 2x2-tap conv of the stride-2 view dY[2i+p][2j+q] with the
 * transposed class kernels, the four accumulated in one dense (Ho x Wo) tile (no (2Ho x 2Wo) intermediate, no 2x2 sum pool).  w =
 * eod_pack_conv_weight_dgrad (cout_pad = 4*C0) of the class-kernel tensor of eod_conv_up4_weights; Cout = channels of dX. */
int eod_conv_up4_bwd_ok(const eod_conv_desc* d);
int64_t eod_conv_workspace_size(const eod_conv_desc* d);

/* ------------------------------------------------------------------------------------------
 * k3/k7/k8: batched GEMM on MFMA,  C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][n][k] (+bias)(+res)
 * A and B are both K-contiguous ("NT").  Used for the attention block (unet_openai.py:427-433):
 * qkv / proj_out Conv1d(1x1) :414,422 and the two einsums of QKVAttention(Legacy) :476-480.
 * batch index z = b0 * nb1 + b1, each operand has one stride per level (elements).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* a;
    const void* b;
    const float* bias; /* fp32, or NULL */
    const void* res;   /* same layout/dtype as c, or NULL */
    void* c;
    int64_t lda, ldb, ldc;
    int64_t sa0, sa1, sb0, sb1, sc0, sc1; /* batch strides (elements) */
    int32_t dtype;     /* dtype of a, b (and res / c unless c_f32) */
    int32_t M, N, K;   /* K must be a multiple of 16 bytes worth of elements */
    int32_t nb0, nb1;
    int32_t bias_mode; /* 0 none, 1 per column n, 2 per row m, 3 = softmax-backward epilogue: bias is [batch][M], and
                          c = res * (alpha*acc - bias[z][m])  (dS = P * (dP - rowsum(dP*P)), unet_openai.py:479);
                          4 = softmax-rebuild epilogue: c = exp(alpha*acc - bias[z][m])  (P from the scores and their log-sum-exp) */
    int32_t c_f32;     /* 1: store c as fp32 regardless of dtype */
    float alpha;
    int32_t x3;        /* 1 (dtype EOD_F32, K % 8 == 0): every product as three fp16 MFMAs on operands split into hi + lo halves in LDS
                          (the fp32x3 precision mode, see eod_conv_desc.w_split) */
    const float* a_bound; /* x3: bound tables [nb0][32] of a / b (see eod_conv_desc.a_bound; indexed by the OUTER batch index b0 = image);
                          NULL = that operand is known to stay below 4094 in magnitude (softmax weights) */
    const float* b_bound;
} eod_gemm_desc;
int eod_gemm_nt(const eod_gemm_desc* d, void* stream);

/* weights: OIHW fp32 (Conv2d.weight, unet_openai.py:21-25) -> [tap][Cout][cin_pad] storage dtype */
int eod_pack_conv_weight(const float* w_oihw, void* dst, int dtype, int Cout, int Cin, int ksize,
                         int cin_pad, void* stream);
/* split-fp16 weights of the fp32x3 product (eod_conv_desc.w_split): [tap][Cout][cin_pad] at 4 bytes per element, every 8 input
 * channels as [8 x fp16 hi | 8 x fp16 lo] of s_j*w, s_j = s * 2^d_j: s = 2^k per tensor and d_j >= 0 per output row, both chosen on the
 * device so that every row's largest value lies in (2^12, 2^13] (each output channel keeps its own 22 bits whatever the other rows hold);
 * scale (device, 4 + Cout slots of 4 bytes) receives float {s, 1/(16 s), -, -} and int32 d_j.  cin_pad % 8 == 0.  No host synchronisation. */
int eod_pack_conv_weight_split(const float* w_oihw, void* dst, float* scale, int Cout, int Cin, int ksize, int cin_pad, void* stream);
/* two weights that feed ONE accumulator (eod_conv_desc.skip_w: out_layers' 3x3 conv + the 1x1 skip_connection, unet_openai.py:341,352):
 * both packed as above with one common scale s taken over the two tensors.  w2 is [Cout][Cin2] (1x1), dst2 [Cout][Cin2] at 4 bytes. */
int eod_pack_conv_weight_split_pair(const float* w_oihw, void* dst, const float* w2_oi, void* dst2, float* scale, int Cout, int Cin,
                                    int ksize, int cin_pad, int Cin2, void* stream);
/* thin-input variant: OIHW fp32 -> [Cout][ldk], k = tap*cin_pad + c, zero padded (see eod_conv_desc.w_tapmajor) */
int eod_pack_conv_weight_tapmajor(const float* w_oihw, void* dst, int dtype, int Cout, int Cin, int cin_pad, void* stream);
/* the same in the split-fp16 pair format (fp32 storage, eod_conv_desc.w_tapmajor together with w_split); scale as above */
int eod_pack_conv_weight_tapmajor_split(const float* w_oihw, void* dst, float* scale, int Cout, int Cin, int cin_pad, void* stream);
int eod_conv_tapmajor_ldk(int C0, int dtype);
/* generic strided 2-D cast-copy: dst[r][c] = (dtype) src[row_map[r]*ld_src + c]; row_map may be NULL
 * (identity); a negative row_map entry yields a zero row (K padding of attention heads) */
int eod_pack_rows(const float* src, int64_t ld_src, const int32_t* row_map, void* dst, int64_t ld_dst,
                  int dtype, int rows, int cols, void* stream);

/* API-edge layout conversion (x.type(self.dtype) / th.cat([x,cond],1), unet_openai.py:756,767):
 * dst NHWC [N][H][W][c_pad] = concat(src0 NCHW fp32 [N][C0], src1 NCHW fp32 [N][C1]), zero padded. */
int eod_nchw_to_nhwc(const float* src0, int C0, const float* src1, int C1, void* dst, int dtype,
                     int N, int H, int W, int c_pad, void* stream);
int eod_nhwc_to_nchw(const void* src, int dtype, float* dst, int N, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * k5: GroupNorm32(32, C) [+ SiLU]  (unet_openai.py:11-13, 71-78, 312-316, 336-343, 739-743)
 * three stream-ordered phases, deterministic (no atomics):
 *   partial : per (n, pixel-chunk p, channel) sum / sum-of-squares       -> part[N][P][Ctot][2]
 *             (skipped when the producing conv already emitted them: eod_conv_desc.stats)
 *   finalize: per (n, group) mean/rstd over BOTH concat sources (part0 [N][P0][C0][2] | part1 [N][P1][C1][2],
 *             part1 may be NULL), folded with gamma/beta (and the
 *             FiLM scale/shift of use_scale_shift_norm :377-381) into scale/shift[N][Ctot]
 *   apply   : y = act(x * scale + shift), written at channel offset `coff` of a Ctot-wide tensor
 * ------------------------------------------------------------------------------------------ */
int eod_gn_partial(const void* x, int dtype, int N, int HW, int C, float* part, int P, int Ctot,
                   int coff, void* stream);
int eod_gn_finalize(const float* part0, int P0, int C0, const float* part1, int P1, int C1, int N, int64_t HW,
                    int groups, float eps, const float* gamma, const float* beta, const float* film,
                    int64_t film_stride, float* scale_shift, float* ab_raw, float* ab_norm, void* stream);
/* ab_raw / ab_norm (optional OUT, [N][32] fp32, groups <= 32): bound tables for the split-fp16 consumers of the tensor(s) the statistics
 * were taken of (eod_conv_desc.a_bound / skip_bound): entry g of image n = an upper bound of max|x| (ab_raw) and of max|x*scale + shift|
 * (ab_norm; also bounds its SiLU) over group g, from the largest partial sum of squares -- no extra pass over the tensor.
 * eod_act_bound writes the same table for a tensor without a GroupNorm in front of its consumer: from the partial sums a conv epilogue
 * emitted (part0 [N][P0][C0][2] | part1), or -- part0 == NULL -- as the exact max|x| of x ([N][per_image] elements, any finite fp32;
 * accumulate = 1: entry-wise maximum with what the table already holds -- the further sources of a virtual concat). */
int eod_act_bound(const void* x, int dtype, int N, int64_t per_image, const float* part0, int P0, int C0, const float* part1, int P1,
                  int C1, float* ab, int accumulate, void* stream);
/* A-priori table of a linear layer's OUTPUT (a producer that writes pre-split needs its scale before it has seen its values):
 * |y_r| <= (max_r sum_c |w[r][c]|) * max|x| + max|b|.  eod_weight_l1max: coef = {max row L1 norm of w [rows][cols], max|bias|} (device,
 * once per plan); eod_bound_affine: ab_out[n][j] = ab_in[n][j] * coef[0] + coef[1]. */
int eod_weight_l1max(const float* w, int rows, int cols, const float* bias, float* coef, void* stream);
int eod_bound_affine(const float* ab_in, const float* coef, float* ab_out, int N, void* stream);
int eod_gn_apply(const void* x, int dtype, int N, int HW, int C, const float* scale_shift, int Ctot,
                 int coff, int silu, void* y, const float* split_bound, void* stream);
/* split_bound (EOD_F32, C / Ctot / coff multiples of 8): write y PRE-SPLIT for a split-fp16 consumer (eod_conv_desc.x_presplit):
 * [8 x fp16 hi | 8 x fp16 lo] per 8 channels of s_n * y, s_n from this bound table [N][32] (eod_gn_finalize's ab_norm). NULL = plain. */

/* ------------------------------------------------------------------------------------------
 * k8: fused ("flash"-style) QKVAttention / QKVAttentionLegacy forward (unet_openai.py:465-481, 497-515):
 *   out[n, t, h*d + j] = sum_s softmax_s((q_t . k_s) / sqrt(d)) v[s, j]     (scale = d^-1/4 on q and on k, :475-478)
 * online softmax in fp32, the T x T matrix is never materialised.  fp16 storage, head dim <= 64.
 *   qk  [N*T][ld_qk] : q heads at column h*dpad, k heads at k_off + h*dpad (each head zero-padded to dpad channels)
 *   vT  [N][C][ldt]  : V transposed (keys contiguous, zero beyond T)
 *   out [N*T][C]
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const void* qk;
    const void* vT;
    void* out;
    float* lse;          /* optional OUT [N][heads][T]: log-sum-exp of every score row (natural log) -- what the training path
                            needs to rebuild P = exp(S - lse) in the backward without keeping the T x T matrix */
    int64_t ld_qk, ldt;
    int32_t dtype, N, T, C, heads, d, dpad, k_off;
} eod_attn_desc;
int eod_attention_fwd(const eod_attn_desc* d, void* stream);

/* k8 (softmax of QKVAttention, unet_openai.py:479/513): p[r][0..ldp) = softmax(s[r][0..n)), zero pad */
int eod_softmax_rows(const float* s, int64_t lds, void* p, int64_t ldp, int dtype, int64_t rows, int n,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * k6: timestep embedding MLP (unet_openai.py:81-99, 597-602, 604-605, 329-335, 763-766).
 *   emb   = W2 * silu(W1 * [cos(t f) | sin(t f)] + b1) + b2 (+ label_emb[y])
 *   out   = Wcat * silu(emb) + bcat             (all ResBlock emb_layers batched into one GEMV)
 * all fp32.  freqs[half] is the host-computed fp32 table of :92-94.  t is int64 [N]; with t_f32 != 0 it points at fp32 [N] instead
 * (`timesteps[:, None].float()`, :95: the reference's embedding also takes fractional timesteps).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const int64_t* t;
    const float* freqs;
    const float* w1; const float* b1; /* [E][D], [E] */
    const float* w2; const float* b2; /* [E][E], [E] */
    const float* label_emb;           /* [num_classes][E] or NULL */
    const int64_t* y;                 /* [N] or NULL */
    const float* wcat; const float* bcat; /* [J][E], [J] */
    float* h1;   /* scratch [N][E] */
    float* emb;  /* out [N][E] (pre-SiLU, as the reference's emb) */
    float* out;  /* out [N][J] */
    int32_t N, D, E, J;
    int32_t t_f32, _pad;
} eod_temb_desc;
int eod_time_embed(const eod_temb_desc* d, void* stream);
/* the sinusoid alone (timestep_embedding, unet_openai.py:81-99; the reference's standalone helper): out [N][dim] fp32 =
 * [cos(t f) | sin(t f)] (+ a zero column when dim is odd); t fp32 [N] (may be fractional), freqs[dim/2] as above */
int eod_timestep_embedding(const float* t, const float* freqs, float* out, int N, int dim, void* stream);

/* ------------------------------------------------------------------------------------------
 * k11-k15: fused sampler updates.  fp32, built with -ffp-contract=off so that every rounding of
 * the reference's torch op sequence is reproduced bit-for-bit.  x/pred/noise/out are [N][CHW] fp32.
 * tables are the EODiffusion buffers (model.py:23-32) of length T, t is int64 [N].  A timestep outside [0, T) never reads
 * behind the tables: that sample's output is filled with NaN (the reference's gather raises there).
 * ------------------------------------------------------------------------------------------ */
/* model.py:94-98 (== ddpm.py:279-282) */
int eod_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp,
                 const float* sqrt_1m_acp, float* out, int N, int64_t chw, int T, void* stream);
/* model.py:58-60  x_t <- mask*q_sample(gt,t,noise) + (1-mask)*x_t ; mask is [N][1][H][W] */
int eod_repaint_mix(const float* x_t, const float* gt, const float* mask, const float* noise,
                    const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp, float* out,
                    int N, int C, int64_t hw, int T, void* stream);
/* model.py:126-150 (clip=1) / :101-122 (clip=0), including the batch-wide `t.min()>0` branch */
int eod_ddpm_step(const float* x_t, const float* pred, const float* noise, const int64_t* t,
                  const float* betas, const float* alphas, const float* acp, const float* sqrt_1m_acp,
                  float* out, int N, int64_t chw, int T, int clip, void* stream);
/* ddim.py:192-206; the four scalars are the fp32-rounded table entries of :192-195 */
int eod_ddim_step(const float* x, const float* e_t, const float* noise, float a_t, float a_prev,
                  float sigma_t, float sqrt_1m_at, float temperature, float* x_prev, float* pred_x0,
                  int64_t numel, void* stream);
/* classifier-free guidance of p_sample_ddim (ddim.py:177-181): out = e_uncond + scale * (e_cond - e_uncond) */
int eod_cfg_combine(const float* e_uncond, const float* e_cond, float scale, float* out, int64_t numel, void* stream);
/* table-driven DDPM step of the LDM-derived sampler: DDPM.p_sample ddpm.py:248-255 with predict_start_from_noise :221-225,
 * q_posterior :227-234 (tables of register_schedule :122-162, fp32 [T]); noise masked per sample at t == 0.  t outside [0, T): NaN. */
int eod_ldm_p_sample(const float* x, const float* eps, const float* noise, const int64_t* t, const float* sqrt_recip_acp,
                     const float* sqrt_recipm1_acp, const float* post_coef1, const float* post_coef2,
                     const float* post_logvar, float* out, int N, int64_t chw, int T, int clip, void* stream);
/* k15: counter-based N(0,1): Philox4x32-10 keyed by seed, counter = (element/4, sample0+n, step, stream_id)
 * -> results are invariant to how samples are sharded over ranks. */
int eod_randn_philox(float* out, int N, int64_t chw, uint64_t seed, int64_t sample0, int32_t step,
                     int32_t stream_id, void* stream);

/* harness-side elementwise ops of inference.py, fp32, bit-exact vs the torch expressions:
 *   eod_repaint_cond   :100-109  cond [N][C+1][hw] = cat(image [N][C][hw], invert ? 1 - mask : mask), mask [N][1][hw]
 *   eod_postprocess    :128      mode 0: y = clip(x, 0, 1) (data in [0,1]);  mode 1: y = (x + 1) / 2 (data in [-1,1])
 *   eod_masked_preview :134      out = image * clip(mask + lift, 0, 1)   (lift = 0.7 in the reference) */
int eod_repaint_cond(const float* image, const float* mask, float* cond, int N, int C, int64_t hw, int invert, void* stream);
int eod_postprocess(const float* x, float* y, int64_t numel, int mode, void* stream);
int eod_masked_preview(const float* image, const float* mask, float* out, int N, int C, int64_t hw, float lift, void* stream);

/* ------------------------------------------------------------------------------------------
 * Native executor: run a pre-built program (array of tagged descriptors) on one stream without
 * returning to the host language between launches (replaces the ~100-180 Python-dispatched
 * ATen launches per UNet forward).
 * ------------------------------------------------------------------------------------------ */
enum {
    EOD_OP_CONV = 1, EOD_OP_GEMM = 2, EOD_OP_GN_PARTIAL = 3, EOD_OP_GN_FINALIZE = 4,
    EOD_OP_GN_APPLY = 5, EOD_OP_SOFTMAX = 6, EOD_OP_TEMB = 7, EOD_OP_TO_NHWC = 8, EOD_OP_TO_NCHW = 9,
    EOD_OP_POOL = 10, EOD_OP_ATTN = 11,
    EOD_OP_TRANSPOSE = 12, /* eod_transpose_gather (training forward: transposed q|k|v for the attention GEMMs) */
    EOD_OP_ATTN_NAT = 13,  /* eod_attention_fwd_nat */
    EOD_OP_DROPOUT = 14,   /* eod_dropout (training forward) */
    EOD_OP_ACT_BOUND = 15, /* eod_act_bound */
    EOD_OP_BOUND_AFFINE = 16 /* eod_bound_affine */
};
typedef struct {
    const void* p[8];
    int64_t l[4];
    int32_t i[10];
    float f[2];
} eod_small_desc; /* argument pack of the small ops, field use documented in csrc/program.hip */
typedef struct {
    int32_t kind;
    int32_t _pad;
    union {
        eod_conv_desc conv;
        eod_gemm_desc gemm;
        eod_temb_desc temb;
        eod_attn_desc attn;
        eod_small_desc small;
    } u;
} eod_op;
int eod_program_run(const eod_op* ops, int n_ops, void* stream);
/* Measurement variant: brackets every op with HIP events recorded on `stream` (the stream the kernels run
 * on).  timer = eod_timer_create(n_ops, max_runs); after a stream sync, eod_timer_read() returns the number
 * of recorded runs and fills ms[n_ops] with each op's duration summed over those runs. */
void* eod_timer_create(int n_ops, int max_runs);
void eod_timer_destroy(void* timer);
int eod_timer_read(void* timer, float* ms);
/* bracket only the ops whose mask byte is non-zero (each event pair idles the stream for a few microseconds) */
int eod_timer_set_mask(void* timer, const unsigned char* mask, int n_ops);
int eod_program_run_timed(const eod_op* ops, int n_ops, void* stream, void* timer);
/* resblock_updown helpers (unet_openai.py:320-325): mode 0 = 2x2 average pool, 1 = nearest 2x,
 * 2 = 2x2 sum pool (backward of nearest 2x), 3 = nearest 2x times 0.25 (backward of the average pool), 4 = crop the last row
 * (pad_tl & 1) / column (pad_tl & 2) -- training path */
int eod_resample2x(const void* x, int dtype, int N, int H, int W, int C, int mode, int pad_tl, void* y,
                   void* stream);

/* ------------------------------------------------------------------------------------------
 * Training path (train.py:109-124: forward, nn.MSELoss, loss.backward(), AdamW.step(), EMA update).
 * The backward of every UNet block runs on the forward's MFMA kernels (see csrc/train.hip):
 *   backward-data    = eod_conv2d_igemm(dY, eod_pack_conv_weight_dgrad(W))        (stride 2: upsample = 2)
 *   backward-weights = eod_gemm_nt over the pixel axis on eod_transpose_gather'ed operands + eod_wgrad_reduce
 * ------------------------------------------------------------------------------------------ */
/* OIHW fp32 -> [taps-1-tap][ci - ci0][cout_pad]: packed weights of the conv dY -> dX[:, ci0:ci0+nci] */
int eod_pack_conv_weight_dgrad(const float* w_oihw, void* dst, int dtype, int Cout, int Cin, int ksize, int ci0, int nci,
                               int cout_pad, void* stream);
/* every weight re-pack of a training step in one launch: jobs (device array) of kind 0 = eod_pack_conv_weight (cpad = cin_pad) or
 * kind 1 = eod_pack_conv_weight_dgrad (cpad = cout_pad); the caller cuts each job's destination into blocks of EOD_PACK_CHUNK
 * elements: blk_job[b] = job index, blk_first[b] = first destination element of block b (device arrays, nblocks entries) */
#define EOD_PACK_CHUNK 4096
typedef struct {
    const float* w;  /* OIHW fp32 parameter */
    void* dst;       /* packed destination, storage dtype */
    int32_t kind, Cout, Cin, taps, ci0, nci, cpad, _pad;
} eod_pack_job;
int eod_pack_jobs(const eod_pack_job* jobs, const int32_t* blk_job, const int64_t* blk_first, int nblocks, int dtype, void* stream);
/* NHWC [N][H][W][C] -> [C][ld_dst], column k = (n*(Ho+2*row_pad) + ho + row_pad)*Wo + wo holds
 * src[n][ho*stride - pad + dy][wo*stride - pad + dx][c] (ups: of the nearest-2x image), zero outside / in pad rows */
int eod_transpose_gather(const void* src, int dtype, int N, int H, int W, int C, void* dst, int64_t ld_dst, int Ho, int Wo,
                         int stride, int pad, int dy, int dx, int ups, int row_pad, void* stream);
/* seg[s][c] = scale * sum of x[c][s*seg_len .. (s+1)*seg_len)   (bias / timestep-embedding gradients) */
int eod_rowsum_segments(const void* x, int dtype, int C, int64_t ld, int nseg, int64_t seg_len, float scale, float* seg,
                        int64_t seg_ld, void* stream);
int eod_colsum(const float* seg, int S, int C, float* out, void* stream); /* out[c] = sum_s seg[s][c] */
/* the same gradients from eod_gn_partial's per-(image, slab, channel) sums of the NHWC gradient (no transposed copy):
 * dbias[c] = scale * sum_{n,p} part[n][p][c][0] (c < cvalid), demb[n][c] = sum_p part[n][p][c][0]; either may be NULL */
int eod_channel_sums_finish(const float* part, int N, int P, int C, int cvalid, float scale, float* dbias, float* demb,
                            int64_t demb_ld, float* scratch /* N*cvalid floats when dbias != NULL */, void* stream);
/* dW_oihw[co][ci0+ci][tap] = scale * sum_s partial[s][tap][co][ci]  (partial: fp32 [S][taps][Cout][ldp]) */
int eod_wgrad_reduce(const float* partial, int S, int ksize, int Cout, int nci, int ldp, int ci0, int Cin, float scale,
                     float* dw_oihw, void* stream);
/* dedicated backward-weights kernel for 3x3 / stride-1 / pad-1 convs (fp16, Wo % 64 == 0, channels % 8 == 0): reads dY
 * [N][Ho][Wo][Cy] and X [N][H][W][Cx] (ups: conv input = nearest-2x of X) in place -- no transposed copies -- and writes the
 * fp32 partial tiles partial[S][9][Cout][ldp] that eod_wgrad_reduce sums (csrc/train.hip: conv3x3_wgrad_kernel) */
int eod_conv3x3_wgrad(const void* dy, const void* x, int dtype, int N, int H, int W, int Cx, int Ho, int Wo, int Cy, int Cout,
                      int ups, float* partial, int ldp, int S, void* stream);
/* ups = 2: the same for a conv whose input is the nearest-2x upsampling of X, in the parity-class form (see eod_conv_up4_ok): 16 tap
 * planes partial[S][((2p+q)*2 + a)*2 + b][Cout][ldp] from the stride-2 views of dY (4/9 of the MACs of ups = 1);
 * eod_wgrad_reduce(ksize = 4) sums the splits into [Cout][Cin][16], eod_wgrad_up4_map folds those into dW_oihw [Cout][Cin][3][3] */
int eod_wgrad_up4_map(const float* t16, int Cout, int Cin, float* dw_oihw, void* stream);
/* ups = 3: the same kernel for a STRIDE-2 3x3 conv (Downsample.op, unet_openai.py:262) of an even map: X is [N][H = 2 Ho][W = 2 Wo][Cx],
 * gathered at pixel stride 2 (two column phases per row tap); ordinary 9-plane partial tiles -> eod_wgrad_reduce(ksize 3) */
/* the same for a 1x1 / stride-1 conv (skip connections, attention projections; F.conv2d / conv1d backward-weights behind
 * unet_openai.py:345,409,413): dW[co][ci] = sum_pix dY[pix][co] X[pix][ci] over npix = N*H*W pixel-major rows, split over S
 * pixel ranges into partial[S][1][Cout][ldp] (fp16, channels % 8 == 0; csrc/train.hip: gemm_tn_kernel) */
int eod_conv1x1_wgrad(const void* dy, const void* x, int dtype, int64_t npix, int Cx, int Cy, int Cout, float* partial, int ldp,
                      int S, void* stream);
/* GroupNorm32 (+SiLU) backward (unet_openai.py:11-13,312-316): see csrc/train.hip for the algebra */
int eod_gn_mean_rstd(const float* part0, int P0, int C0, const float* part1, int P1, int C1, int N, int64_t HW, int groups,
                     float eps, float* mean_rstd, void* stream);
int eod_gn_bwd_partial(const void* x, const void* dy, const float* scale_shift, int dtype, int N, int HW, int C, float* part,
                       int P, int Ctot, int coff, int silu, void* stream);
/* film / dfilm (optional): FiLM of use_scale_shift_norm (:377-381), film[n] = [scale | shift] with row stride film_stride;
 * dfilm[n] = [d scale | d shift] */
int eod_gn_bwd_finalize(const float* part, int P, int Ctot, int N, int64_t HW, int groups, const float* mean_rstd,
                        const float* gamma, const float* beta, const float* film, int64_t film_stride, float* dfilm,
                        int64_t dfilm_stride, float* coef, float* gb, void* stream);
int eod_gn_bwd_params(const float* gb, int N, int Ctot, float scale, float* dgamma, float* dbeta, void* stream);
int eod_gn_bwd_apply(const void* x, const void* dy, const float* scale_shift, const float* coef, const void* add, int dtype,
                     int N, int HW, int C, int Ctot, int coff, int silu, void* dx, float* csum, void* stream);
/* csum (optional OUT): per-(image, slab, channel) sums of the stored dx, [N][eod_gn_bwd_apply_slabs(...)][C][2] (sum at [0]): dx is the
 * output gradient of the conv that produced x, and that conv's bias / timestep-projection gradients are these sums
 * (eod_channel_sums_finish) -- no separate eod_gn_partial pass over dY. */
int eod_gn_bwd_apply_slabs(int dtype, int N, int HW, int C);
int eod_add(const void* a, const void* b, void* y, int dtype, int64_t n, void* stream);
/* nn.Dropout of ResBlock.out_layers (unet_openai.py:339): y = x * keep / (1 - p); keep = Philox4x32-10(seed; element/4, layer, step)
 * -- forward and backward call it with the same key (x = activation / x = gradient), the mask is never stored */
int eod_dropout(const void* x, void* y, int dtype, int64_t n, float p, uint64_t seed, uint32_t layer, uint32_t step, void* stream);
/* out[(i0*n1 + i1)*n2 + i2] = sum_{j<d} a[off + j] * b[off + j], off = i0*s0 + i1*s1 + i2*s2 (elements): the attention
 * backward's D[n][head][t] = sum_j dO * O over one head's channels (rowsum(dP * P) without forming dP) */
int eod_rowdot(const void* a, const void* b, int dtype, int64_t n0, int64_t n1, int64_t n2, int64_t s0, int64_t s1, int64_t s2, int d,
               float* out, void* stream);
/* x[i] *= s (fp32, in place).  Training keeps dS = P (dP - D) of the materialised attention backward on a 2^12 scale so that it stays
 * a NORMAL fp16 number at T in the thousands (P ~ 1/T): D is scaled with this before the dS GEMM, whose alpha carries the same factor */
int eod_scale_f32(float* x, int64_t n, float s, void* stream);
/* softmax backward on rows (QKVAttention, unet_openai.py:479): dS[r][j] = P[r][j] * (dP[r][j] - sum_k dP[r][k] P[r][k]),
 * P / dS storage dtype with row stride ldp, dP fp32 with row stride lds; columns n..ldp-1 of dS are written as zeros */
/* fused attention forward on the NATURAL qkv layout [N][T][3C] (channel = q_off / k_off / v_off + head*head_stride + j; legacy
 * order: 0, d, 2d, 3d; new order: 0, C, 2C, d), head dim a multiple of 8 and <= 512, any T; out [N][T][C]; lse optional
 * [N][heads][T].  The T x T weights of unet_openai.py:476-480 / 508-514 never exist in HBM.  Head dims above 64 (the one 512-channel
 * head of the train.py:50 architecture's middle block, unet_openai.py:675-681) run on csrc/attn_wide.hip: the head dim split over the
 * waves of a workgroup, EOD_F16 and EOD_F32 (split-fp16 products) only, no pre-split tensors.
 *   EOD_F16: K / V tiles staged row-major by LDS-DMA, V^T through transposed LDS reads (csrc/attn_bwd.hip)
 *   EOD_F32: fp32 in / out, fp32 online softmax, both contractions as three fp16 MFMAs per product on operands split into
 *            hi + lo halves (fp32-grade, ~2^-22 per product; csrc/attn_x3.hip) */
enum { EOD_ATTN_OUT_PRESPLIT = 1, EOD_ATTN_IN_PRESPLIT = 2, EOD_ATTN_EXACT_F32 = 4 };
int eod_attention_fwd_nat(const void* qkv, void* out, float* lse, int dtype, int N, int T, int C, int heads, int d, int q_off, int k_off,
                          int v_off, int head_stride, const float* qkv_bound, int flags, void* stream);
/* EOD_F32 only.  qkv_bound: bound table [N][32] of the qkv tensor (eod_conv_desc.a_bound); NULL = |q|, |k|, |v| < 4094 guaranteed.
 * flags: EOD_ATTN_OUT_PRESPLIT: `out` is written pre-split for a split-fp16 conv (eod_conv_desc.x_presplit with a_bound = qkv_bound:
 *        rows of out are convex combinations of v rows, so the table of qkv bounds them);
 *        EOD_ATTN_IN_PRESPLIT: qkv arrives pre-split (eod_conv_desc.y_presplit_bound = qkv_bound): q fragments are loaded as they are,
 *        K / V tiles staged by LDS-DMA, no split arithmetic in the kernel;
 *        EOD_ATTN_EXACT_F32: IEEE fp32 products (v_mfma_f32_32x32x2_f32, csrc/attn_f32.hip) instead of the split-fp16 ones -- the
 *        fused kernel of the exact `fp32` precision mode (no bound table, no pre-split tensors). */
/* flash-style attention backward (fp16, head dim a multiple of 8 and <= 64, any T): dqkv [N][T][3C] from qkv [N][T][3C] (channel = q_off /
 * k_off / v_off + head*head_stride + j), dO [N][T][C], the forward's log-sum-exp lse [N][heads][T] (eod_attn_desc.lse) and
 * D[n][h][t] = sum_j dO*O (eod_rowdot).  P is rebuilt tile by tile in registers: nothing T x T touches HBM (csrc/attn_bwd.hip) */
int eod_attention_bwd(const void* qkv, const void* dO, const float* lse, const float* D, void* dqkv, int dtype, int N, int T, int C,
                      int heads, int d, int q_off, int k_off, int v_off, int head_stride, void* stream);
/* C[m][n] = alpha * sum_k A[k][m] * B[k][n], both operands K-major ([K][lda], [K][ldb]), fp16, batched with two stride levels
 * (batch z = b0*nb1 + b1): dV = P^T dO and dK = dS^T Q of the attention backward without transposing the T x T matrices */
int eod_gemm_tn(const void* a, int64_t lda, const void* b, int64_t ldb, void* c, int64_t ldc, int dtype, int M, int N, int K, float alpha,
                int nb0, int nb1, int64_t sa0, int64_t sa1, int64_t sb0, int64_t sb1, int64_t sc0, int64_t sc1, void* stream);
int eod_softmax_bwd_rows(const void* P, int64_t ldp, const float* dP, int64_t lds, void* dS, int dtype, int64_t rows, int n,
                         void* stream);
/* dense layers of the timestep-embedding MLP (unet_openai.py:597-602,329-335), fp32: dW, db (scaled) and / or din;
 * act_in: the layer's input is 0 = in, 1 = SiLU(in), 2 = sinusoid(t); pre != NULL multiplies din by SiLU'(pre) */
int eod_linear_bwd_small(const float* dout, int64_t ld_dout, const float* in, const int64_t* t, const float* freqs, const float* w,
                         const float* pre, int N, int K, int J, int act_in, float scale, float* dW, float* db, float* din,
                         float* scratch /* optional, 32*N*K floats: splits the J loop of din */, void* stream);
int eod_temb_pre1(const int64_t* t, const float* freqs, const float* w1, const float* b1, int N, int D, int E, float* pre1,
                  void* stream);
/* nn.Embedding backward (label_emb :604-605): dW[c][e] = scale * sum over the rows n with y[n] == c of dout[n][e] */
int eod_embedding_bwd(const float* dout, const int64_t* y, int N, int E, int classes, float scale, float* dW, void* stream);
/* nn.MSELoss(reduction='mean') (train.py:86,117): loss[0] = mean((pred-target)^2), dpred = 2*(pred-target)/n (optional);
 * scratch: scratch_len (>= 1, up to 1024 used) floats of per-block partial sums (fixed-order two-level sum) */
int eod_mse_loss(const float* pred, const float* target, int64_t n, float* loss, float* dpred, float* scratch, int scratch_len,
                 void* stream);
/* torch.optim.AdamW single-tensor step (train.py:75,119; algorithm of the pinned PyTorch 1.13, eo_diffusion.yml:114) on flat
 * fp32 buffers; `step` is the 1-based update count */
int eod_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int step, void* stream);
/* the same step behind an overflow guard (fp16 training with a static loss scale): if any gradient is inf / NaN the step is SKIPPED
 * (p, m, v untouched).  state (device, 4 ints, zero-initialised by the caller once) = {this step was skipped, steps skipped so far,
 * two floats of the kernel's own}; scratch: scratch_len (>= 1, up to 8192 used) ints.  `step` = the 1-based number of CALLS; the bias
 * corrections use step - (steps skipped so far), i.e. the number of moment updates, computed on the device: a skipped step never
 * reaches the optimizer, as with torch.cuda.amp.GradScaler.  No host synchronisation. */
int eod_adamw_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                           double weight_decay, int step, int* state, int* scratch, int scratch_len, void* stream);
/* EMA of script_utils/utils.py:56-67: avg = decay*avg + (1-decay)*p */
int eod_ema_update(float* avg, const float* p, int64_t n, double decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif
