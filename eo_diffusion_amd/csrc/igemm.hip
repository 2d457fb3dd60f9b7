// Implicit-GEMM convolution and batched NT GEMM on the gfx950 matrix cores.
//
// One kernel family serves every contraction on the path (k1 3x3 conv, k2 stride-2
// conv, k3 1x1 / Conv1d, k4 edge convs, k7/k8 attention GEMMs, k9 virtual nearest-2x, k10 virtual concat):
//
//     C[m][n] = alpha * sum_k A[m][k] * B[n][k]   (+ bias, + per-sample bias, + residual)
//
//  * A rows are output pixels (conv: gathered from the NHWC input per filter tap, zero outside the
//    image; gemm: plain rows), B rows are output channels ([tap][Cout][Cin] packed weights).
//  * 256 threads = 4 waves per workgroup, one BM x BN output tile, 32x32 MFMA sub-tiles per wave:
//      fp16 storage : v_mfma_f32_32x32x16_f16  (8 halves / lane / operand, fp32 accumulate)
//      fp32 storage : v_mfma_f32_32x32x2_f32   (exact fp32, 4 instructions per 16-byte chunk)
//    Both consume the SAME 16-byte-chunk LDS image (K-step = 128 bytes per row = 8 chunks): lane
//    (r = lane&31, h = lane>>5) reads chunk 2s+h of row r for k-substep s, so the kernel body is
//    byte-oriented and the dtype only appears in Mma<T>.
//  * Staging is LDS-DMA: `global_load_lds_dwordx4` writes 64 lanes x 16 B = 8 rows x 128 B straight into
//    LDS with no VGPR round trip and no ds_write.  The LDS destination of that instruction is lane-linear,
//    so the bank-conflict swizzle lives on the SOURCE address (guide rule 21): lane (row r, slot j) fetches
//    global chunk j ^ ((r>>1)&7); readers apply the same XOR.  With 128-byte rows every ds_read_b128 lane
//    group then touches 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).
//    The DMA goes through buffer descriptors (`buffer_load_dwordx4 ... lds`): the per-lane part of the address
//    is one 32-bit byte offset relative to the tile's first image / row, the per-K-step part (filter tap,
//    channel chunk) is a wave-uniform SGPR offset, and masked elements (image border = conv zero padding,
//    K / M / N tails) simply use an out-of-range offset: the hardware range check then writes ZEROS into LDS
//    (verified on gfx950), so no lane branches around its load and there is no pointer select.
//  * 2-stage LDS ring, ONE barrier per K-step: the DMA of K-step t+1 is in flight while the MFMAs of step
//    t run; 64-70 KiB of LDS per workgroup leaves two workgroups per CU to cover each other's waits.
//  * K loop order is (concat source, channel chunk) OUTER, filter tap INNER: the 9 taps of one channel
//    chunk re-read the same (BM + halo) x 128 B of input, which stays in the CU's L1.  Per-slot state is a
//    32-bit byte offset and a 9-bit tap-validity mask; (tap, chunk, source) are scalar counters: no 64-bit
//    arithmetic and no division in the loop (the first version spent 8 VALU + 9 SALU per MFMA there).
//  * M tiles are TH x TW pixel patches when the feature map allows it (halo reuse in L1), linear runs
//    of BM pixels otherwise (ragged shapes: 28x28, 7x7, 3x3...).
//  * Epilogue: per-column bias / timestep bias are added in the MFMA C layout (lane = output column), the
//    tile is transposed through LDS (fp32) and written as full 16-byte chunks along the channel axis with
//    the residual added on the way: whole 128/256-byte lines per row instead of 2-byte scattered stores.
//  * workgroup -> tile mapping is XCD-aware (bijective remap, guide T1): the N-tiles of one M-tile and
//    neighbouring M-tiles land on the same XCD's L2.
#include <type_traits>

#include "common.h"

struct IgemmP {
    const char* a0;
    const char* a1;
    const char* b;
    const float* bias;
    const float* cbias;
    const char* res;
    char* y;
    long long cbias_stride;
    // conv geometry
    int N, H, W, C0, C1, Cin, Cout, KS, stride, pad, ups, pad_tl, Ho, Wo, HoWo, Heff, Weff;
    int tw_log2, th;  // patch mode when tw_log2 >= 0
    int tiles_pw, tiles_pi;
    // gemm geometry (elements)
    long long lda, ldb, ldc, sa0, sa1, sb0, sb1, sc0, sc1;
    int nb1, bias_mode, c_f32;
    // common
    long long M;
    int Ncols, K, taps, KT, tiles_m, tiles_n, out_nchw;
    int kc0, kc1;  // conv: channel chunks of source 0 / source 1 (K-steps = taps * (kc0 + kc1))
    int tapmajor_log2;  // >= 0: thin-input mode, K index = tap * C0 + c (C0 = EPC << tapmajor_log2), weights [Cout][ldk]
    int ldk;
    const float* gn_ss;  // halo kernel: fused GroupNorm of the INPUT: {scale, shift} per (image, input channel), or NULL
    int gn_silu;
    float* stats;  // optional GroupNorm partial sums of the output: [N][stats_P][Cout][2]
    int stats_P, tiles_per_image;
    int splitk;    // conv: K-steps are split over gridDim.y workgroups; raw fp32 partial tiles go to `y` (= workspace)
    int splitk_per;  // halo kernel: channel chunks per K slice (the last slice takes the rest + the fused skip phase)
    float alpha;
    const float* w_scale;  // split-fp16 mode: device {s, 1/(s*A_SCALE)} of the packed weights (eod_pack_conv_weight_split)
    const int* w_rexp;     // ... and its per-row exponents d_j (csrc/misc.hip: row_exp_kernel): column j is multiplied by 2^-d_j on top
    int w_row0;            // weight row of output column 0 (the parity-class upsample conv: class * Cout)
    // row-decode grid: the M axis enumerates (image, Hd x Wd) positions.  Normally that is the output map (Hd = Ho, Wd = Wo).  Parity
    // mode (par = 1; zero-insertion upsampling = the backward-data of a stride-2 conv): one launch per output parity class (par_y,
    // par_x), rows enumerate the (Ho/2) x (Wo/2) positions of that class, output pixel = (2 hd + par_y, 2 wd + par_x), and the K loop
    // walks only the taps that meet stored (even) input positions for that class: taplist holds them, 4 bits each, `taps` of them.
    int Hd, Wd, HWd, par, par_y, par_x;
    unsigned taplist;
    // fused 1x1 skip connection of a ResBlock (halo kernel, SKIP instances): after the 3x3 K loop the same accumulators take
    // sum_c sx[pixel][c] * b2[co][c] over the block input (one or two sources = the virtual concat), see eod_conv_desc.skip_x
    const char* sx0;
    const char* sx1;
    const char* b2;  // packed [Cout][SC0 + SC1] (the 1x1 weight; split mode: same scale as `b`)
    int SC0, SC1, skc0, skc1;
    int n_base;  // halo kernel: first output column of this launch (a conv of 384 columns runs as a 256-column and a 128-column launch)
    int tpw;     // halo kernel, STREAM instances: consecutive pixel tiles (of one image and one N-tile) per workgroup; the others: 1
    // split-fp16 products: bound tables (common.h) of the activation operands -> a power-of-two operand scale per image, derived in
    // the kernel.  conv: a_bound [N][32] of the conv input, skip_bound of the fused skip conv's input; gemm (x3): a_bound / b_bound
    // [nb0][32] of the two operands.  NULL = fixed scale 16 (the caller guarantees |x| < 4094).
    const float* a_bound;
    const float* skip_bound;
    const float* b_bound;
    int ab_tab_off;  // generic kernel: byte offset in LDS of the per-image {s, 16/s} table of a tile that straddles images
    int a_ps;        // generic split kernel: the A operand arrives PRE-SPLIT from HBM (eod_conv_desc.x_presplit): no rewrite in LDS
    const float* y_ps_bound;  // generic split kernel: write the output pre-split, scale per image from this table (y_presplit_bound)
};

template <typename T> struct Mma;
template <> struct Mma<half_t> {
    static __device__ __forceinline__ void run(const i32x4& a, const i32x4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const i32x4& a, const i32x4& b, f32x16& c) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], c, 0, 0, 0);
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// fp32 THROUGH THE fp16 MATRIX PIPE ("fp32x3" precision mode).  v_mfma_f32_32x32x2_f32 runs at 1/16 of the fp16 MFMA rate, so
// the exact-fp32 convolution is pinned at 157 TFLOP/s.  Here every fp32 operand is split into two fp16 values,
//     x = hi + lo,   hi = fp16(x),  lo = fp16(x - hi)          (x - hi is exact in fp32; hi + lo carries ~22 significant bits)
// and a product is three fp16 MFMAs into ONE fp32 accumulator:  a*b ~ a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  (the dropped lo*lo term is
// 2^-22 relative).  Measured on gfx950: fp16 SUBNORMAL inputs are honoured by the MFMA and produced by v_cvt_f16_f32
// (tools/probe/mfma_denorm.hip), so `lo` keeps an absolute resolution of 2^-25 / scale even where it underflows the normal range.
// Scaling (exact powers of two, undone by alpha in the epilogue):
//   * weights: s = 2^k per tensor with max|w|*s in (2^12, 2^13], times 2^d_j per output row so that EVERY row's maximum lies there (pack
//     time, device side: csrc/misc.hip row_exp_kernel; the epilogues multiply column j by 2^-d_j) -> every output channel keeps 22 bits
//     (inside a row: every weight within 2^-17 of the row's largest one);
//   * activations: per IMAGE, s_a = 2^k derived in the kernel from the bound table of the tensor (common.h: eod_gn_finalize /
//     eod_act_bound write an upper bound B of max|x| per image; B s_a in [2^14, 2^15)), so every element of a tensor of any magnitude
//     stays inside the fp16 range and lo_a resolves 2^-34 of the image's largest element or better.  This holds for EVERY consumer,
//     including the ones whose input is not behind a GroupNorm: the first conv on x_t, the (fused) 1x1 skip convs over the raw block
//     input, Downsample / Upsample convs, proj_out, q / k / v.  Without a table (NULL) the scale is the fixed 16 of round 2 and the
//     caller guarantees |x| < 65504/16 = 4094.
// LDS image: operands stay 4 bytes per element.  Each pair of 16-byte chunks (8 consecutive k) is rewritten IN PLACE as
// [8 x hi | 8 x lo]: weights are packed that way in HBM, activation patches are converted once per staged element by the wave
// that DMA'd them (lanes l and l^1 hold the two chunks of a pair and swap halves through DPP).  A K-step of 128 bytes = 32 k =
// 2 MFMA sub-steps; lane half h of sub-step s reads pair 2s+h: hi at chunk 2(2s+h), lo at chunk 2(2s+h)+1 -- the same
// ds_read_b128 / XOR-swizzle machinery as the other modes, 16 reads and 24 MFMAs per K-step and wave.
// Measured and rejected here (A/B on MI355X, tools/conv_bench.py): reading both sub-steps' fragments ahead of the MFMAs (two register
// sets, pinned order; 246-256 VGPRs): -1..-3 %; rewriting the patch piece before instead of after a step's MFMAs: +-0; an 8-wave
// 16x16-pixel tile with a 3-stage weight ring (half the weight DMA per MFMA) on 16x16x32: +-1 %; a pixel permutation inside the
// 16-row MFMA blocks that makes the patch reads bank-conflict free (PMC: SQ_LDS_BANK_CONFLICT 26 % -> 1 % of LDS_IDX_ACTIVE):
// fp16 +-0 (LDS is not the limiter), fp32x3 -13 % (57-62 spilled VGPRs); with the registers the column-keyed swizzle freed (213
// VGPRs), reading the NEXT tap's patch fragments at the end of a tap, across the wait + barrier: +-0.5 %.  PMC of the split kernel (tools/pmc_stall.txt): waves are
// 32 % issuing, 44 % stalled on issue (the matrix pipe shared by two waves), 24 % parked at s_waitcnt / s_barrier.
// ---------------------------------------------------------------------------------------------------------------------------
#define EOD_SPLIT_ASCALE 16.0f  // operand scale without a bound table
// four fp32 values (one 16-byte chunk, already multiplied by the activation scale) -> packed {hi[4]} , {lo[4]} (2 dwords each)
__device__ __forceinline__ void split4(const f32x4& f, int (&hi)[2], int (&lo)[2]) {
    half4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = (half_t)f[e];
        l[e] = (half_t)(f[e] - (float)h[e]);
    }
    const auto hb = __builtin_bit_cast(__attribute__((ext_vector_type(2))) int, h);
    const auto lb = __builtin_bit_cast(__attribute__((ext_vector_type(2))) int, l);
    hi[0] = hb[0]; hi[1] = hb[1]; lo[0] = lb[0]; lo[1] = lb[1];
}
// The same split of s * v (s a power of two, so s * v is exact): hi = fp16(s v), lo = fp16(s v - hi), in EIGHT instructions for four
// values -- v_fma_mixlo / mixhi_f16 multiply in fp32, add the (negated fp16) third operand and write one half of the destination, so the
// scale multiply, the conversions back to fp32 and the packing of the C form (14 instructions as hipcc emits it) disappear.  Same bits:
// s v and s v - hi are exact in fp32, each is rounded to fp16 once (round to nearest even, fp16 subnormals kept: the mode the casts use).
// (the DPP moves of the exchange read SELECTS of these results, instructions hipcc sees and pads itself)
__device__ __forceinline__ void split4_scaled(const f32x4& v, float s, int (&hi)[2], int (&lo)[2]) {
    int h01, h23, l01, l23;
    asm("v_fma_mixlo_f16 %0, %4, %5, 0\n\t"
        "v_fma_mixlo_f16 %1, %4, %7, 0\n\t"
        "v_fma_mixhi_f16 %0, %4, %6, 0\n\t"
        "v_fma_mixhi_f16 %1, %4, %8, 0\n\t"
        "v_fma_mixlo_f16 %2, %4, %5, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %4, %7, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %4, %6, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %4, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23)
        : "v"(s), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    hi[0] = h01; hi[1] = h23; lo[0] = l01; lo[1] = l23;
}
// the lane holding the EVEN chunk of a pair keeps [hi_even | hi_odd], the lane holding the ODD chunk keeps [lo_even | lo_odd]
__device__ __forceinline__ i32x4 split_pair_exchange_scaled(const f32x4& v, float s, bool odd_chunk) {
    int hi[2], lo[2];
    split4_scaled(v, s, hi, lo);
    const int s0 = odd_chunk ? hi[0] : lo[0], s1 = odd_chunk ? hi[1] : lo[1];
    const int r0 = __builtin_amdgcn_mov_dpp(s0, 0xB1, 0xF, 0xF, true), r1 = __builtin_amdgcn_mov_dpp(s1, 0xB1, 0xF, 0xF, true);
    return odd_chunk ? i32x4{r0, r1, lo[0], lo[1]} : i32x4{hi[0], hi[1], r0, r1};
}
__device__ __forceinline__ i32x4 split_pair_exchange(const f32x4& f, bool odd_chunk) {
    int hi[2], lo[2];
    split4(f, hi, lo);
    const int s0 = odd_chunk ? hi[0] : lo[0], s1 = odd_chunk ? hi[1] : lo[1];
    // lane ^ 1 through DPP quad_perm [1, 0, 3, 2] (a VALU move; __shfl_xor would go through the LDS crossbar: ds_bpermute + wait)
    const int r0 = __builtin_amdgcn_mov_dpp(s0, 0xB1, 0xF, 0xF, true), r1 = __builtin_amdgcn_mov_dpp(s1, 0xB1, 0xF, 0xF, true);
    return odd_chunk ? i32x4{r0, r1, lo[0], lo[1]} : i32x4{hi[0], hi[1], r0, r1};
}

typedef __attribute__((address_space(3))) void lds_void;

#define EOD_OOB 0x80000000u   // per-lane offset == num_records (out of range even if soffset were added without wrap): zeros
#define EOD_WINDOW 0x80000000u  // num_records of every descriptor: 2 GiB window behind its base

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)EOD_WINDOW, 0x00020000);
}
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, char* lds_dst_wave_uniform) {
    // 64 lanes x 16 B -> lds_dst + lane*16 (the destination is wave-uniform base + lane*16 by hardware)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_dst_wave_uniform, 16, voff, soff, 0, 0);
}

// tile origin (wave-uniform): conv tiles are addressed relative to the tile's first image
struct TileGeom {
    int tile_m, n_first, rem_first, ty0, tx0;
    long long offA, offB, offC;  // gemm batch offsets (elements)
};

// row r of the tile -> (image index relative to n_first, ho, wo); false if past the end
template <int BM>
__device__ __forceinline__ bool decode_row(const IgemmP& p, const TileGeom& g, int r, int& nrel, int& ho, int& wo) {
    if (p.tw_log2 >= 0) {
        nrel = 0;
        ho = g.ty0 + (r >> p.tw_log2);
        wo = g.tx0 + (r & ((1 << p.tw_log2) - 1));
        const bool inside = wo < p.Wd;  // (halo kernel on 8-wide maps: the right half of the 8 x 16 tile lies behind the map)
        if (p.par) {
            ho = 2 * ho + p.par_y;
            wo = 2 * wo + p.par_x;
        }
        return inside;
    }
    int rem = g.rem_first + r;
    nrel = 0;
    if (rem >= p.HWd) {  // a tile may straddle images (or span several when the map is tiny)
        nrel = rem / p.HWd;
        rem -= nrel * p.HWd;
    }
    ho = rem / p.Wd;
    wo = rem - ho * p.Wd;
    if (p.par) {
        ho = 2 * ho + p.par_y;
        wo = 2 * wo + p.par_x;
    }
    return (long long)g.tile_m * BM + r < p.M;
}

template <bool CONV, int BM>
__device__ __forceinline__ TileGeom make_geom(const IgemmP& p, int tile_m) {
    TileGeom g = {};
    g.tile_m = tile_m;
    if constexpr (CONV) {
        if (p.tw_log2 >= 0) {
            g.n_first = tile_m / p.tiles_pi;
            const int t = tile_m - g.n_first * p.tiles_pi;
            const int ty = t / p.tiles_pw;
            g.ty0 = ty * p.th;
            g.tx0 = (t - ty * p.tiles_pw) << p.tw_log2;
        } else {
            const long long m0 = (long long)tile_m * BM;
            g.n_first = (int)(m0 / p.HWd);
            g.rem_first = (int)(m0 - (long long)g.n_first * p.HWd);
        }
    } else {
        const int z = blockIdx.y;
        const int b0 = z / p.nb1, b1 = z - b0 * p.nb1;
        g.offA = b0 * p.sa0 + b1 * p.sa1;
        g.offB = b0 * p.sb0 + b1 * p.sb1;
        g.offC = b0 * p.sc0 + b1 * p.sc1;
    }
    return g;
}

#ifdef EOD_STAMP
// Diagnostic build only (-DEOD_STAMP, tools/debug/halo_stamps.py): wave 0 of every workgroup of conv3x3_halo_kernel stamps the 100 MHz
// wall counter and the shader clock at entry, after the prologue, after the K loop and at exit.  The values go to a buffer of their own
// that no kernel reads; the product library is built without this.
__device__ unsigned long long g_eod_stamp[65536][16];
#define EOD_STAMP_AT(k)                                                                    \
    do {                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 65536) {                                      \
            g_eod_stamp[blockIdx.x][2 * (k)] = __builtin_amdgcn_s_memrealtime();           \
            g_eod_stamp[blockIdx.x][2 * (k) + 1] = __builtin_amdgcn_s_memtime();           \
        }                                                                                  \
    } while (0)
extern "C" int eod_debug_read_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_eod_stamp), (size_t)n * 16 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#else
#define EOD_STAMP_AT(k) do { } while (0)
#endif
#ifdef EOD_TSTAMP
// Diagnostic build only (-DEOD_TSTAMP, tools/debug/halo_timeline.py): wave 1 of a workgroup stamps the shader clock at four points of every
// K-step of its second chunk (into the 512 filler bytes behind the last patch row of that chunk's buffer), copied to a buffer of its own
// when the chunk ends.
__device__ unsigned long long g_eod_tstamp[2048][40];
extern "C" int eod_debug_read_tstamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_eod_tstamp), (size_t)n * 40 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#define EOD_TSTAMP_AT(idx)                                                                                      \
    do {                                                                                                        \
        if (GN && wave == 1 && cc == 1 && k == 0) {                                                             \
            const unsigned long long ts_ = __builtin_amdgcn_s_memtime();                                        \
            if (lane == 0) reinterpret_cast<unsigned long long*>(sA + ABUF + PR * 128)[idx] = ts_;              \
        }                                                                                                       \
    } while (0)
#else
#define EOD_TSTAMP_AT(idx) do { } while (0)
#endif
// XCD-aware tile mapping (bijective remap, guide T1)
__device__ __forceinline__ void map_tile(const IgemmP& p, int& tile_m, int& tile_n) {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tile_n = swz % p.tiles_n;
    tile_m = swz / p.tiles_n;
}

// =============================================================================================
// Shared epilogue: bias / per-sample (timestep) bias in the MFMA C layout (lane = column), transpose through LDS
// (fp32, wave-private slab), 16-byte stores along the channel axis with the residual added on the way.
// Precondition: every wave has passed a barrier after its last LDS read of the operand ring.
// =============================================================================================
// accumulator layout of the two MFMA shapes (C/D maps of cdna_hip_programming.md section 3): MS = 32: 32x32 tiles, 16 registers,
// col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5);  MS = 16: 16x16 tiles, 4 registers, col = lane & 15, row = 4 (lane >> 4) + r
template <int MS> struct AccLayout {
    static constexpr int R = MS == 32 ? 16 : 4;
    typedef typename std::conditional<MS == 32, f32x16, f32x4>::type vec;
    static __device__ __forceinline__ int row(int r, int lane) { return MS == 32 ? (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) : 4 * (lane >> 4) + r; }
    static __device__ __forceinline__ int col(int lane) { return MS == 32 ? (lane & 31) : (lane & 15); }
};

// rowtab (generic split kernel, tiles that straddle images): LDS table of {s, 16/s} per image of the tile; a row's values are
// multiplied by ITS image's 16/s on top of p.alpha
template <typename T, bool CONV, int BM, int BN, int WAVES_M, int WAVES_N, bool OUTF32, int MS = 32>
__device__ __forceinline__ void igemm_epilogue(const IgemmP& p, const TileGeom& g,
                                               typename AccLayout<MS>::vec (&acc)[BM / WAVES_M / MS][BN / WAVES_N / MS], char* smem, int wave,
                                               int lane, int n0, const float* rowtab = nullptr, int rowtab_n = 0,
                                               const float* pre_bcol = nullptr) {
    constexpr int ES = sizeof(T);
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / MS, TN = WN / MS, R = AccLayout<MS>::R;
    constexpr int EP_LD = WN + 4;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = AccLayout<MS>::col(lane);
    // =========================== epilogue ===========================
    // 1) C layout (lane = column): alpha, per-column bias; conv: per-sample bias needs the row's image.
    int col[TN];
    bool cok[TN];
    float bcol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        col[j] = n0 + wn * WN + j * MS + lr;
        cok[j] = col[j] < p.Ncols;
        // pre_bcol (patch-mode kernels): bias + per-sample bias of the tile's image, loaded at kernel ENTRY -- fetched here, the two
        // dependent global loads cost every workgroup 3-4 us of memory latency between its last MFMA and its first store
        // (tools/debug/halo_stamps.py: 5.2 us from the end of the K loop to the end of the LDS transpose, 1-2 us without them)
        bcol[j] = pre_bcol ? pre_bcol[j] : ((cok[j] && p.bias && p.bias_mode == 1) ? p.bias[col[j]] : 0.0f);
    }
    float cmul[TN];  // split-fp16 weights: 2^-d_j of the column's weight row (1 elsewhere)
#pragma unroll
    for (int j = 0; j < TN; ++j) cmul[j] = (CONV && p.w_rexp && cok[j]) ? ldexpf(1.0f, -p.w_rexp[p.w_row0 + col[j]]) : 1.0f;
    // per-sample (timestep) bias: one value per (image, column).  A tile almost always lies inside one image
    // (always in patch mode); then it is folded into bcol once instead of being fetched per row.
    bool cb_per_row = false;
    if constexpr (CONV) {
        if (p.cbias && !pre_bcol) {
            const bool one_image = (p.tw_log2 >= 0) || (g.rem_first + BM <= p.HWd);
            if (one_image) {
                const float* cbp = p.cbias + (long long)g.n_first * p.cbias_stride;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (cok[j]) bcol[j] += cbp[col[j]];
            } else {
                cb_per_row = true;
            }
        }
    }

    if (CONV && p.out_nchw) {
        // tiny-Cout head writing the API layout (NCHW fp32) directly: scalar stores, negligible bytes
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = wm * WM + i * MS + AccLayout<MS>::row(r, lane);
                int nrel, ho, wo;
                if (!decode_row<BM>(p, g, row, nrel, ho, wo)) continue;
                const int n = g.n_first + nrel;
                const float ralpha = rowtab ? p.alpha * rowtab[2 * min(nrel, rowtab_n - 1) + 1] : p.alpha;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!cok[j]) continue;
                    float v = acc[i][j][r] * (ralpha * cmul[j]) + bcol[j];
                    if (cb_per_row) v += p.cbias[(long long)n * p.cbias_stride + col[j]];
                    reinterpret_cast<float*>(p.y)[((long long)n * p.Cout + col[j]) * p.HoWo + (long long)ho * p.Wo + wo] = v;
                }
            }
        return;
    }

    // 3) row-major read-back: each lane handles ITER 16-byte output chunks (one row each)
    constexpr int EPO = OUTF32 ? 4 : 8;          // elements per 16-byte output chunk
    constexpr int CPR = WN / EPO;                // chunks per row of the wave's slab
    constexpr int ITER = WM * CPR / 64;          // chunks per lane
    static_assert(64 % CPR == 0 && (WM * CPR) % 64 == 0, "epilogue mapping");
    typedef typename std::conditional<OUTF32, float, half_t>::type OT;
    const int ncol0 = n0 + wn * WN;
    const int cj = (lane % CPR) * EPO;           // a lane always owns the same column chunk
    const int c = ncol0 + cj;
    const bool c_ok = c < p.Ncols;
    const bool c_full = c + EPO <= p.Ncols;
    long long off[ITER];
    bool ok[ITER];
    i32x4 rv[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int rw = it * (64 / CPR) + lane / CPR;
        ok[it] = c_ok;
        if constexpr (CONV) {
            int nrel = 0, ho = 0, wo = 0;
            ok[it] = ok[it] && decode_row<BM>(p, g, wm * WM + rw, nrel, ho, wo);
            off[it] = ((((long long)(g.n_first + nrel) * p.Ho + ho) * p.Wo + wo)) * p.Cout + c;
        } else {
            const long long m = (long long)g.tile_m * BM + wm * WM + rw;
            ok[it] = ok[it] && m < p.M;
            off[it] = g.offC + m * p.ldc + c;
        }
        rv[it] = i32x4{0, 0, 0, 0};
    }
    // residual: issue ALL loads now, so that their latency overlaps the LDS transpose below (a rolled loop would
    // expose one HBM round trip per chunk)
    const bool vec = c_full && ((p.Cout * (int)sizeof(OT)) % 16 == 0 || !CONV);
    if (p.res) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const OT* ro = reinterpret_cast<const OT*>(p.res) + off[it];
            if (ok[it]) {
                if (vec && ((reinterpret_cast<uintptr_t>(ro) & 15) == 0)) {
                    rv[it] = *reinterpret_cast<const i32x4*>(ro);
                } else {
                    OT tmp[EPO];
                    for (int e = 0; e < EPO; ++e) tmp[e] = (c + e < p.Ncols) ? ro[e] : (OT)0.0f;
                    rv[it] = *reinterpret_cast<const i32x4*>(tmp);
                }
            }
        }
    }

    // 2) transpose through LDS: each wave owns a [WM][EP_LD] fp32 slab (the ring is free after the last barrier)
    float* slab = reinterpret_cast<float*>(smem) + wave * (WM * EP_LD);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int rw = i * MS + AccLayout<MS>::row(r, lane);  // row inside the wave's slab
            float cb[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) cb[j] = 0.0f;
            float ralpha = p.alpha;
            if constexpr (CONV) {
                if (rowtab) {
                    int nrel, ho, wo;
                    decode_row<BM>(p, g, wm * WM + rw, nrel, ho, wo);
                    ralpha = p.alpha * rowtab[2 * min(nrel, rowtab_n - 1) + 1];
                }
                if (cb_per_row) {
                    int nrel, ho, wo;
                    // rows behind the last image (M tail of the last tile) are never stored: they must not index the table either -- image
                    // N of a [N][stride] table lies behind its allocation (found by tests/test_gpu_fuzz_archs.py as an intermittent fault)
                    const bool rok = decode_row<BM>(p, g, wm * WM + rw, nrel, ho, wo);
                    const float* cbp = p.cbias + (long long)(g.n_first + nrel) * p.cbias_stride;
#pragma unroll
                    for (int j = 0; j < TN; ++j) cb[j] = (rok && cok[j]) ? cbp[col[j]] : 0.0f;
                }
            } else {
                if (p.bias && p.bias_mode >= 2) {
                    const long long m = (long long)g.tile_m * BM + wm * WM + rw;
                    // mode 3 (softmax-backward epilogue): one value per (batch, row), SUBTRACTED; the result is then multiplied by res
                    // (mode 4, softmax rebuild: the same subtraction, then exp)
                    const float br = m < p.M ? (p.bias_mode >= 3 ? -p.bias[(long long)blockIdx.y * p.M + m] : p.bias[m]) : 0.0f;
#pragma unroll
                    for (int j = 0; j < TN; ++j) cb[j] = br;
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) slab[rw * EP_LD + j * MS + lr] = acc[i][j][r] * (ralpha * cmul[j]) + bcol[j] + cb[j];
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's slab writes are done (the slab is wave-private)
    __builtin_amdgcn_wave_barrier();
    EOD_STAMP_AT(4);

    // pre-split output: the image's power-of-two scale from the a-priori table of y (one image per tile: checked by the launcher)
    float ps_scale = 1.0f;
    if constexpr (CONV && OUTF32) {
        if (p.y_ps_bound) ps_scale = ab_scale_of(ab_wave_bound(p.y_ps_bound, g.n_first), EOD_AB_KMIN_ATTN).s;
    }
    // optional GroupNorm partial statistics of the STORED values (sum / sum of squares per output channel over the
    // wave's WM rows): accumulated per lane over its ITER rows, combined across the lanes that share a column chunk.
    const bool want_stats = CONV && p.stats != nullptr;
    float st_s[EPO], st_q[EPO];
#pragma unroll
    for (int e = 0; e < EPO; ++e) st_s[e] = st_q[e] = 0.0f;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int rw = it * (64 / CPR) + lane / CPR;
        const float* sp = slab + rw * EP_LD + cj;
        float v[EPO];
        {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v0[e];
            if constexpr (EPO == 8) {
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 + e] = v1[e];
            }
        }
        OT o_[EPO];
        {
            OT rr[EPO];
            *reinterpret_cast<i32x4*>(rr) = rv[it];
#pragma unroll
            for (int e = 0; e < EPO; ++e) {
                float ov = v[e] + (float)rr[e];
                if constexpr (!CONV) {
                    if (p.bias_mode == 3) ov = v[e] * (float)rr[e];
                    if (p.bias_mode == 4) ov = sizeof(T) == 2 ? __expf(v[e]) : expf(v[e]);
                }
                o_[e] = (OT)ov;
            }
        }
        if constexpr (CONV && OUTF32) {
            if (p.y_ps_bound) {
                // PRE-SPLIT output (eod_conv_desc.y_presplit_bound): this lane's 4 channels and the neighbouring lane's (the other chunk
                // of the 8-channel group, same row) become [8 x hi | 8 x lo] of s * y -- every lane takes part in the exchange
                f32x4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = (float)o_[e];
                const i32x4 w4 = split_pair_exchange_scaled(v4, ps_scale, ((cj >> 2) & 1) != 0);
                if (ok[it]) *reinterpret_cast<i32x4*>(reinterpret_cast<OT*>(p.y) + off[it]) = w4;
                continue;
            }
        }
        if (!ok[it]) continue;
        OT* yo = reinterpret_cast<OT*>(p.y) + off[it];
        if (vec && ((reinterpret_cast<uintptr_t>(yo) & 15) == 0)) {
            *reinterpret_cast<i32x4*>(yo) = *reinterpret_cast<const i32x4*>(o_);
        } else {
            for (int e = 0; e < EPO && c + e < p.Ncols; ++e) yo[e] = o_[e];
        }
        if (want_stats) {
#pragma unroll
            for (int e = 0; e < EPO; ++e) {
                const float x = (float)o_[e];
                st_s[e] += x;
                st_q[e] += x * x;
            }
        }
    }
    if constexpr (CONV) {
        if (want_stats) {
            // lanes l, l+CPR, l+2CPR, ... own the same columns: butterfly over the lane bits above log2(CPR)
#pragma unroll
            for (int e = 0; e < EPO; ++e) {
#pragma unroll
                for (int o = 32; o >= CPR; o >>= 1) {
                    st_s[e] += __shfl_xor(st_s[e], o);
                    st_q[e] += __shfl_xor(st_q[e], o);
                }
            }
            if (lane < CPR && c_ok) {
                const int tile_in_img = (p.tw_log2 >= 0) ? (g.tile_m - g.n_first * p.tiles_per_image) : (g.rem_first / BM);
                const int slot = tile_in_img * WAVES_M + wm;
                float* dst = p.stats + (((long long)g.n_first * p.stats_P + slot) * p.Cout + c) * 2;
#pragma unroll
                for (int e = 0; e < EPO; ++e) {
                    if (c + e < p.Ncols) {
                        dst[2 * e] = st_s[e];
                        dst[2 * e + 1] = st_q[e];
                    }
                }
            }
        }
    }
}

// ---- direct epilogue of the fp32-storage patch-mode kernels (conv3x3_halo_kernel<SPLIT>) ----------------------------------------
// The split product's MFMAs are issued with their operands SWAPPED (D = W X^T instead of X W^T; both operands of a 16x16x32 MFMA have
// the same register layout, so the swap is free): a lane then holds FOUR CONSECUTIVE CHANNELS of ONE pixel per accumulator tile
// (channel = 16 j + 4 (lane >> 4) + r, pixel = 16 i + (lane & 15)), i.e. a 16-byte piece of an NHWC row.  The epilogue stores straight
// from the accumulators: no LDS transpose (2.8 us of VALU per workgroup on the transposed path, tools/debug/halo_stamps.py), no LDS at
// all -- the operand ring is free while it runs.  Per-channel terms (bias + per-sample bias) come as 4 x TN values fetched at entry.
template <int TN>
__device__ __forceinline__ void prefetch_bcol4(const IgemmP& p, int ncols, int col0, int lane, int n_first, f32x4 (&out)[TN]) {
    const int lg = lane >> 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = col0 + j * 16 + 4 * lg;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < ncols) {  // (Cout is a multiple of the 16-byte chunk: a quad is inside or outside as a whole)
            if (p.bias && p.bias_mode == 1) v = *reinterpret_cast<const f32x4*>(p.bias + c);
            if (p.cbias) v += *reinterpret_cast<const f32x4*>(p.cbias + (long long)n_first * p.cbias_stride + c);
        }
        out[j] = v;
    }
}

// row exponents of the split weights for the same quads, four bytes per quad (d_j <= 100): one register per channel tile through the K loop
template <int TN>
__device__ __forceinline__ void prefetch_wexp4(const IgemmP& p, int ncols, int col0, int wrow0, int lane, int (&out)[TN]) {
    const int lg = lane >> 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = col0 + j * 16 + 4 * lg;
        int pk = 0;
        if (p.w_rexp && c < ncols) {
            const i32x4 e = *reinterpret_cast<const i32x4*>(p.w_rexp + wrow0 + c);
            pk = e[0] | (e[1] << 8) | (e[2] << 16) | (e[3] << 24);
        }
        out[j] = pk;
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__device__ __forceinline__ void halo_epilogue_direct(const IgemmP& p, const TileGeom& g, f32x4 (&acc)[BM / WAVES_M / 16][BN / WAVES_N / 16], int wave,
                                                     int lane, int n0, const f32x4 (&bq)[BN / WAVES_N / 16], float alpha,
                                                     const int (&we)[BN / WAVES_N / 16]) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lp = lane & 15, lg = lane >> 4;
    float* const y = reinterpret_cast<float*>(p.y);
    const float* const res = reinterpret_cast<const float*>(p.res);
    const bool want_stats = p.stats != nullptr;
    const int c0 = n0 + wn * WN + 4 * lg;
    // channel tile by channel tile (16 channels: this lane's quad c0 + 16 j .. + 3 of the TM pixels 16 i + lp), so that only ONE tile's
    // statistics accumulators and two tiles' residuals are live next to the accumulators (all TN at once: 76 spilled registers)
    long long off[TM];
    bool rok[TM];  // (8-wide maps: the right half of the tile lies behind the map; every other patch-mode tile is whole)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int nrel, ho, wo;
        rok[i] = decode_row<BM>(p, g, wm * WM + i * 16 + lp, nrel, ho, wo);
        off[i] = rok[i] ? (((long long)(g.n_first + nrel) * p.Ho + ho) * p.Wo + wo) * p.Cout : 0;
    }
    const int slot = ((p.tw_log2 >= 0) ? (g.tile_m - g.n_first * p.tiles_per_image) : (g.rem_first / BM)) * WAVES_M + wm;
    // pre-split output (the qkv conv in front of the fused attention): [8 x fp16 hi | 8 x fp16 lo] per 8 channels of s_n * y; the lanes
    // lg (even: channels 0-3 of the group) and lg + 1 (4-7) exchange halves: the even lane stores the 16 hi bytes, the odd one the lo bytes
    float ps_scale = 0.0f;
    if (p.y_ps_bound) ps_scale = ab_scale_of(ab_wave_bound(p.y_ps_bound, g.n_first), EOD_AB_KMIN_ATTN).s;
    f32x4 rv[2][TM];
    if (res && c0 < p.Ncols) {
#pragma unroll
        for (int i = 0; i < TM; ++i) rv[0][i] = *reinterpret_cast<const f32x4*>(res + off[i] + c0);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = c0 + j * 16;
        if (res && j + 1 < TN && c + 16 < p.Ncols) {  // the next tile's residual is in flight while this tile is stored
#pragma unroll
            for (int i = 0; i < TM; ++i) rv[(j + 1) & 1][i] = *reinterpret_cast<const f32x4*>(res + off[i] + c + 16);
        }
        if (c < p.Ncols) {
            f32x4 ss = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
            // alpha (operand scales of the tensor / image) times the column's 2^-d_j (row scale of the split weights): exact powers of two
            const f32x4 am = {ldexpf(alpha, -(we[j] & 255)), ldexpf(alpha, -((we[j] >> 8) & 255)), ldexpf(alpha, -((we[j] >> 16) & 255)),
                              ldexpf(alpha, -((we[j] >> 24) & 255))};
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                f32x4 v = acc[i][j] * am + bq[j];
                if (res) v += rv[j & 1][i];
                if (!rok[i]) {  // never stored, not part of the statistics
                    v = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (ps_scale != 0.0f) {
                    typedef int i32x2 __attribute__((ext_vector_type(2)));
                    half4 h4, l4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sv = v[e] * ps_scale;
                        const half_t hi = (half_t)sv;
                        h4[e] = hi;
                        l4[e] = (half_t)(sv - (float)hi);
                    }
                    const i32x2 hb = __builtin_bit_cast(i32x2, h4), lb = __builtin_bit_cast(i32x2, l4);
                    const bool odd = (lg & 1) != 0;
                    const int s0 = odd ? hb[0] : lb[0], s1 = odd ? hb[1] : lb[1];   // what the partner needs: my hi (I am odd) / my lo (even)
                    const int r0 = __shfl_xor(s0, 16), r1 = __shfl_xor(s1, 16);
                    const i32x4 w = odd ? i32x4{r0, r1, lb[0], lb[1]} : i32x4{hb[0], hb[1], r0, r1};
                    if (rok[i]) *reinterpret_cast<i32x4*>(y + off[i] + (c & ~7) + (odd ? 4 : 0)) = w;
                } else if (rok[i]) {
                    *reinterpret_cast<f32x4*>(y + off[i] + c) = v;
                }
                ss += v;
                sq += v * v;
            }
            if (want_stats) {
                // per-channel sums over the wave's WM pixels: the TM pixel tiles in-lane (above), then the 16 lanes that share lg
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int o = 8; o >= 1; o >>= 1) {
                        ss[r] += __shfl_xor(ss[r], o);
                        sq[r] += __shfl_xor(sq[r], o);
                    }
                }
                if (lp == 0) {
                    float* dst = p.stats + (((long long)g.n_first * p.stats_P + slot) * p.Cout + c) * 2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dst[2 * r] = ss[r];
                        dst[2 * r + 1] = sq[r];
                    }
                }
            }
        }
    }
}

template <typename T, bool CONV, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool SPLIT = false, int MS = 32, bool DIRECT = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void igemm_kernel(const IgemmP p) {
    static_assert(!DIRECT || (CONV && SPLIT && MS == 16 && sizeof(T) == 4 && BN >= 64), "direct epilogue: fp32-storage split conv instances");

    static_assert(!SPLIT || (sizeof(T) == 4 && STAGES == 2 && MS == 16), "split-fp16 product: fp32 storage, 2-stage ring, 16x16x32 MFMAs");
    static_assert(MS == 32 || SPLIT || sizeof(T) == 2, "the 16x16x32 shape exists for the fp16 products only (see conv3x3_halo_kernel)");
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;   // elements per 16-byte chunk
    constexpr int BKB = 128;       // bytes of K per row per K-step
    constexpr int BK = BKB / ES;   // elements per K-step
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / MS, TN = WN / MS;
    constexpr int NW = WAVES_M * WAVES_N;                // waves per workgroup (4 or 8)
    constexpr int GA = BM / 8, GB = BN / 8;              // 8-row groups (one LDS-DMA instruction each)
    constexpr int LA = GA / NW, LB = GB / NW;            // groups per wave
    constexpr int LPW = LA + LB;                         // DMA instructions per wave per K-step
    constexpr int STAGE_A = BM * BKB, STAGE = (BM + BN) * BKB;
    constexpr int EP_LD = WN + 4;                        // fp32 epilogue row stride (floats)
    static_assert(GA % NW == 0 && GB % NW == 0 && (STAGES == 2 || STAGES == 3), "layout");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int tile_m, tile_n;
    map_tile(p, tile_m, tile_n);
    const int n0 = tile_n * BN;
    const TileGeom g = make_geom<CONV, BM>(p, tile_m);
    const int n_first = g.n_first;
    IgemmP pe = p;  // epilogue view of the parameters
    if constexpr (CONV) {
        if (p.splitk > 1) pe.y = p.y + (long long)blockIdx.y * p.M * p.Cout * 4;
    }
    // direct-epilogue instances: the per-column epilogue terms are fetched at entry (their latency passes under the first DMA round
    // trip instead of standing between the last MFMA and the first store)
    constexpr int TNQ = DIRECT ? BN / WAVES_N / 16 : 1;
    f32x4 pre_bq[TNQ];
    int pre_we[TNQ];
    if constexpr (DIRECT) {
        prefetch_bcol4<TNQ>(p, p.Ncols, n0 + (wave % WAVES_N) * (BN / WAVES_N), threadIdx.x & 63, g.n_first, pre_bq);
        prefetch_wexp4<TNQ>(p, p.Ncols, n0 + (wave % WAVES_N) * (BN / WAVES_N), p.w_row0, threadIdx.x & 63, pre_we);
    }
    // split-fp16 product: operand scales of the activation operand(s) from their bound tables (common.h).  conv: one scale per image;
    // a tile normally lies inside one image (always in patch mode) -> wave-uniform as0; a tile that straddles images (maps smaller
    // than or not a multiple of the 128-row tile) keeps a table of its images' scales in LDS: rows are scaled and un-scaled one by one.
    AbScale as0 = {EOD_SPLIT_ASCALE, 1.0f}, bs0 = {EOD_SPLIT_ASCALE, 1.0f};
    const float* rowtab = nullptr;
    int rowtab_n = 0;
    if constexpr (SPLIT) {
        if constexpr (CONV) {
            if (p.a_bound) {
                if (p.tw_log2 >= 0 || g.rem_first + BM <= p.HWd) {
                    as0 = ab_scale_of(ab_wave_bound(p.a_bound, n_first));
                } else {
                    float* tab = reinterpret_cast<float*>(smem + p.ab_tab_off);
                    rowtab_n = min(p.N - n_first, (g.rem_first + BM - 1) / p.HWd + 1);
                    for (int j = wave; j < rowtab_n; j += NW) {
                        const AbScale a = ab_scale_of(ab_wave_bound(p.a_bound, n_first + j));
                        if (lane == 0) {
                            tab[2 * j] = a.s;
                            tab[2 * j + 1] = a.inv;
                        }
                    }
                    __syncthreads();  // (nothing is in flight yet)
                    rowtab = tab;
                }
            }
            // split-K: the partial tiles are stored already un-scaled (exact: powers of two), the reduce pass only sums them
            pe.alpha = p.alpha * p.w_scale[1] * (rowtab ? 1.0f : as0.inv);
        } else {  // GEMM: both operands are activations, split in LDS; image = outer batch index
            const int b0 = (int)blockIdx.y / p.nb1;
            if (p.a_bound) as0 = ab_scale_of(ab_wave_bound(p.a_bound, b0));
            if (p.b_bound) bs0 = ab_scale_of(ab_wave_bound(p.b_bound, b0));
            pe.alpha = p.alpha * (as0.inv * 0.0625f) * (bs0.inv * 0.0625f);
        }
    }
    const long long offA = g.offA, offB = g.offB;

    // ---- per-thread staging slots: group g = wave + NW*i, row = g*8 + (lane>>3), slot = lane&7 ----
    // a_v0 / a_v1: byte offset of this lane's 16-byte chunk for K-step (tap 0, chunk 0) of source 0 / 1, relative
    // to the descriptor base (conv: first image of the tile; gemm: first row of the tile).  Border rows hold a
    // wrapped "negative" value: only taps whose mask bit is set are ever dereferenced.
    const int srow = lane >> 3, sslot = lane & 7;
    unsigned a_v0[LA], a_v1[LA], a_mask[LA];
    int a_bh[LA], a_bw[LA], a_nh[LA];   // upsample only
    int a_chunk[LA];
    float a_s[LA];                      // split mode: operand scale of this lane's row (its image's)
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int row = (wave + NW * i) * 8 + srow;
        a_chunk[i] = sslot ^ ((row >> 1) & 7);
        a_v0[i] = a_v1[i] = 0;
        a_mask[i] = 0;
        a_bh[i] = a_bw[i] = a_nh[i] = 0;
        a_s[i] = as0.s;
        if constexpr (CONV) {
            int nrel, ho, wo;
            const bool ok = decode_row<BM>(p, g, row, nrel, ho, wo);
            if constexpr (SPLIT) {
                if (rowtab) a_s[i] = rowtab[2 * min(nrel, rowtab_n - 1)];
            }
            const int bh = ho * p.stride - p.pad - p.pad_tl, bw = wo * p.stride - p.pad - p.pad_tl;
            a_bh[i] = bh;
            a_bw[i] = bw;
            a_nh[i] = nrel * p.H;
            unsigned mask = 0;
            if (ok) {
                for (int t = 0; t < p.KS * p.KS; ++t) {
                    const int dy = (p.KS == 3) ? t / 3 : 0, dx = (p.KS == 3) ? t - dy * 3 : 0;
                    bool in = (unsigned)(bh + dy) < (unsigned)p.Heff && (unsigned)(bw + dx) < (unsigned)p.Weff;
                    if (p.ups == 2) in = in && (((bh + dy) | (bw + dx)) & 1) == 0;  // zero-insertion: only even positions hold data
                    if (in) mask |= 1u << t;
                }
            }
            a_mask[i] = mask;
            const unsigned pix = (unsigned)((nrel * p.H + bh) * p.W + bw);  // wraps for border rows (see above)
            a_v0[i] = pix * (unsigned)(p.C0 * ES) + (p.tapmajor_log2 >= 0 ? 0 : a_chunk[i] * 16);
            a_v1[i] = pix * (unsigned)(p.C1 * ES) + a_chunk[i] * 16;
        } else {
            const long long m = (long long)tile_m * BM + row;
            a_mask[i] = m < p.M ? 1u : 0u;
            a_v0[i] = (unsigned)(row * (int)p.lda * ES) + a_chunk[i] * 16;
        }
    }
    unsigned b_v[LB];
    int b_chunk[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int row = (wave + NW * i) * 8 + srow;
        b_chunk[i] = sslot ^ ((row >> 1) & 7);
        const bool ok = n0 + row < p.Ncols;
        const int ldb = CONV ? (p.tapmajor_log2 >= 0 ? p.ldk : p.Cin) : (int)p.ldb;
        b_v[i] = ok ? (unsigned)(row * ldb * ES) + b_chunk[i] * 16 : EOD_OOB;
    }

    // ---- descriptors (wave-uniform): base of the tile's first image / first row ----
    __amdgpu_buffer_rsrc_t rsA0, rsA1, rsB;
    if constexpr (CONV) {
        rsA0 = make_rsrc(p.a0 + (long long)n_first * p.H * p.W * p.C0 * ES);
        rsA1 = make_rsrc(p.a1 ? p.a1 + (long long)n_first * p.H * p.W * p.C1 * ES : p.a0);
        // (row pitch of the packed weights: Cin, or the padded [tap][C0] row of the thin-input first conv -- with Cin here the SECOND
        //  N-tile of a tap-major conv, i.e. model_channels > 128, read its weights from the wrong rows: found by the fuzz hunt, round 3)
        rsB = make_rsrc(p.b + (long long)n0 * (p.tapmajor_log2 >= 0 ? p.ldk : p.Cin) * ES);
    } else {
        rsA0 = make_rsrc(p.a0 + (offA + (long long)tile_m * BM * p.lda) * ES);
        rsA1 = rsA0;
        rsB = make_rsrc(p.b + (offB + (long long)n0 * p.ldb) * ES);
    }

    // ---- scalar K-step state: (src, cc, tap) counters, no division in the loop ----
    int st_tap = 0, st_cc = 0, st_src = 0;
    // split-K (conv, small M): this workgroup handles K-steps [kt_begin, kt_end) and writes a raw fp32 partial tile
    int kt_begin = 0, kt_end = p.KT;
    if constexpr (CONV) {
        if (p.splitk > 1) {
            const int per = (p.KT + p.splitk - 1) / p.splitk;
            kt_begin = (int)blockIdx.y * per;
            kt_end = min(p.KT, kt_begin + per);
            const int cl = kt_begin / p.taps;  // linear channel chunk
            st_tap = kt_begin - cl * p.taps;
            st_src = cl >= p.kc0 ? 1 : 0;
            st_cc = st_src ? cl - p.kc0 : cl;
        }
    }
    const int tapstride = p.Cout * p.Cin * ES;  // bytes between two taps of the packed weights

    // K-step DMA = LA + LB instructions per wave.  They are issued one or two at a time BETWEEN the MFMA sub-steps
    // of the current K-step (an LDS-DMA instruction holds the wave's issue port for ~60-180 cycles; behind a group
    // of MFMAs that time overlaps matrix-pipe execution instead of delaying it).
    struct StepState {
        unsigned tapbytes, soffA, soffB, tapbit;
        int cw, kin, dy, dx, src;
        bool ktail;
    } ss;
    auto prep_step = [&]() {
        if constexpr (CONV) {
            if (p.tapmajor_log2 >= 0) {  // K-step = 8 chunks of the [tap][C0] axis; the tap is a per-lane quantity (issue_a)
                ss.src = 0;
                ss.cw = p.C0;
                ss.kin = st_cc * 8;  // first logical chunk of this K-step
                ss.soffA = 0;
                ss.soffB = (unsigned)(st_cc * BKB);
                ss.ktail = false;
                ss.tapbytes = 0;
                ss.tapbit = 0;
                ss.dy = ss.dx = 0;
                ++st_cc;
                return;
            }
            ss.src = st_src;
            ss.cw = st_src ? p.C1 : p.C0;                         // channels of the current source
            ss.kin = st_cc * BK;                                  // first channel of this chunk inside the source
            const int tap = p.par ? (int)((p.taplist >> (4 * st_tap)) & 15u) : st_tap;  // parity mode: the st_tap-th tap of the class
            ss.dy = (p.KS == 3) ? (tap * 11) >> 5 : 0;            // tap / 3 for tap < 9
            ss.dx = (p.KS == 3) ? tap - ss.dy * 3 : 0;
            // the tap displacement goes into the per-lane offset (border rows hold a wrapped negative base that only
            // becomes a valid in-window offset after this add); the channel chunk goes into the SGPR offset
            ss.tapbytes = (unsigned)((ss.dy * p.W + ss.dx) * ss.cw * ES);
            ss.soffA = (unsigned)(ss.kin * ES);
            ss.soffB = (unsigned)(tap * tapstride + ((st_src ? p.C0 : 0) + ss.kin) * ES);
            ss.ktail = ss.kin + BK > ss.cw;                       // uniform: only the last chunk of a source
            ss.tapbit = 1u << tap;
            // advance (tap inner, chunk, source outer)
            if (++st_tap == p.taps) {
                st_tap = 0;
                if (++st_cc == (st_src ? p.kc1 : p.kc0)) {
                    st_cc = 0;
                    ++st_src;
                }
            }
        } else {
            ss.kin = st_cc * BK;
            ss.soffA = ss.soffB = (unsigned)(ss.kin * ES);
            ss.ktail = ss.kin + BK > p.K;
            ss.cw = p.K;
            ++st_cc;
        }
    };
    auto issue_a = [&](int i, char* sbase) {
        unsigned v;
        bool ok;
        if constexpr (CONV) {
            if (p.tapmajor_log2 >= 0) {
                // logical chunk c of the row = (tap, 16-byte sub-chunk of the tap's C0 channels): one gather per lane
                const int c = ss.kin + a_chunk[i];
                const int tap = c >> p.tapmajor_log2, sub = c - (tap << p.tapmajor_log2);
                const int dy = (tap * 11) >> 5, dx = tap - dy * 3;
                v = a_v0[i] + (unsigned)((dy * p.W + dx) * p.C0 * ES + sub * 16);
                ok = tap < 9 && ((a_mask[i] >> tap) & 1u);
                v = ok ? v : EOD_OOB;
                blds16(rsA0, v, 0u, sbase + (wave + NW * i) * 1024);
                return;
            }
            if (p.ups) {
                const unsigned pix = (unsigned)((a_nh[i] + ((a_bh[i] + ss.dy) >> 1)) * p.W + ((a_bw[i] + ss.dx) >> 1));
                v = pix * (unsigned)(ss.cw * ES) + a_chunk[i] * 16;
            } else {
                v = (ss.src ? a_v1[i] : a_v0[i]) + ss.tapbytes;
            }
            ok = (a_mask[i] & ss.tapbit) != 0;
        } else {
            v = a_v0[i];
            ok = a_mask[i] != 0;
        }
        if (ss.ktail) ok = ok && (ss.kin + a_chunk[i] * EPC < ss.cw);
        v = ok ? v : EOD_OOB;
        if (CONV && ss.src)
            blds16(rsA1, v, ss.soffA, sbase + (wave + NW * i) * 1024);
        else
            blds16(rsA0, v, ss.soffA, sbase + (wave + NW * i) * 1024);
    };
    auto issue_b = [&](int i, char* sbase) {
        unsigned v = b_v[i];
        if (ss.ktail) v = (ss.kin + ((SPLIT && CONV) ? (b_chunk[i] >> 1) * 8 : b_chunk[i] * EPC) < ss.cw) ? v : EOD_OOB;
        blds16(rsB, v, ss.soffB, sbase + STAGE_A + (wave + NW * i) * 1024);
    };
    auto issue_loads = [&](int stage) {  // all DMA instructions of one K-step
        char* sbase = smem + stage * STAGE;
        prep_step();
#pragma unroll
        for (int i = 0; i < LA; ++i) issue_a(i, sbase);
#pragma unroll
        for (int i = 0; i < LB; ++i) issue_b(i, sbase);
    };

    typename AccLayout<MS>::vec acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < AccLayout<MS>::R; ++r) acc[i][j][r] = 0.0f;

    // ---- fragment read offsets: row (wm*WM + i*MS + lr), chunk (2s + lh) ^ ((row>>1)&7) ----
    const int lr = lane & (MS - 1), lh = MS == 32 ? lane >> 5 : lane >> 4;
    const int sw = (lr >> 1) & 7;  // (row>>1)&7 == (lr>>1)&7: the wave / tile row offsets are multiples of 16
    int coff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) coff[s] = ((2 * s + lh) ^ sw) * 16;
    const int a_rd = (wm * WM + lr) * BKB;
    const int b_rd = STAGE_A + (wn * WN + lr) * BKB;

    // ---- main loop: STAGES-deep LDS ring, ONE raw barrier per K-step ----
    //   iteration t:  wait until this wave's DMA for step t has landed (counted vmcnt: the DMA of the next
    //                 STAGES-2 steps stays in flight)  ->  s_barrier (every wave's part of step t landed, and every
    //                 wave finished reading step t-1)  ->  issue the DMA of step t+STAGES-1 into the stage that
    //                 step t-1 just vacated  ->  ds_read + MFMA on stage t.
    // __syncthreads() is avoided on purpose: with LDS-DMA in flight it would drain vmcnt to 0 (guide 5.4).
    const int KT = kt_end - kt_begin;  // (may be <= 0 for a trailing split: the tile then stays zero)
#pragma unroll
    for (int ps = 0; ps < STAGES - 1; ++ps)
        if (ps < KT) issue_loads(ps);

    for (int kt = 0; kt < KT; ++kt) {
        if (STAGES == 3 && kt + 1 < KT) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (SPLIT) {
            // this wave's A pieces of step kt have landed: rewrite each fp32 chunk pair in place as [8 x hi | 8 x lo] (x s), see
            // the fp32x3 note at the top; masked lanes were zero-filled by the DMA and stay zero.  Weights arrive pre-split -- and so
            // does A when its producer wrote it that way (a_ps: the normalising pass in front of qkv, the attention in front of
            // proj_out): the LDS-DMA then delivers the finished image and this K-step has no VALU work at all.
            char* sa = smem + (kt % STAGES) * STAGE;
            if (!(CONV && p.a_ps)) {
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                char* ptr = sa + (wave + NW * i) * 1024 + lane * 16;
                const f32x4 f = *reinterpret_cast<const f32x4*>(ptr);
                *reinterpret_cast<i32x4*>(ptr) = split_pair_exchange_scaled(f, a_s[i], (a_chunk[i] & 1) != 0);
            }
            }
            if constexpr (!CONV) {  // GEMM (attention with wide heads): the B operand is an fp32 activation too
#pragma unroll
                for (int i = 0; i < LB; ++i) {
                    char* ptr = sa + STAGE_A + (wave + NW * i) * 1024 + lane * 16;
                    const f32x4 f = *reinterpret_cast<const f32x4*>(ptr);
                    *reinterpret_cast<i32x4*>(ptr) = split_pair_exchange_scaled(f, bs0.s, (b_chunk[i] & 1) != 0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the rewritten pieces are in LDS before the barrier publishes them
        }
        __builtin_amdgcn_s_barrier();
        // (issuing the DMA instructions one by one between the MFMA sub-steps was measured: no gain, -5..10 %)
        // (split mode, 1x1 / stride-2 convs, round 2: a 3-stage ring does not help either -- 128x128 / 4 waves / 96 KiB, one workgroup per
        //  CU: -30 %; 256x128 / 8 waves / 144 KiB: +-1 % -- so DMA latency is not what holds these HBM-streaming shapes at 2.9 TB/s)
        if (kt + STAGES - 1 < KT) issue_loads((kt + STAGES - 1) % STAGES);
        const char* sb = smem + (kt % STAGES) * STAGE;
        if constexpr (MS == 16 && SPLIT) {
            // one 16x16x32 sub-step per K-step; lane quarter lh supplies chunk pair {0, 3, 1, 2}[lh] (see conv3x3_halo_kernel)
            const int ch = 2 * ((0x2130 >> (4 * lh)) & 3);
            const int ohi = (ch ^ sw) * 16, olo = ((ch + 1) ^ sw) * 16;
            i32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) al[i] = *reinterpret_cast<const i32x4*>(sb + a_rd + i * MS * BKB + olo);
#pragma unroll
            for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const i32x4*>(sb + b_rd + j * MS * BKB + ohi);
#pragma unroll
            for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const i32x4*>(sb + a_rd + i * MS * BKB + ohi);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[j]), __builtin_bit_cast(half8, al[i]), acc[i][j], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, al[i]), __builtin_bit_cast(half8, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) bl[j] = *reinterpret_cast<const i32x4*>(sb + b_rd + j * MS * BKB + olo);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bl[j]), __builtin_bit_cast(half8, ah[i]), acc[i][j], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah[i]), __builtin_bit_cast(half8, bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[j]), __builtin_bit_cast(half8, ah[i]), acc[i][j], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah[i]), __builtin_bit_cast(half8, bh[j]), acc[i][j], 0, 0, 0);
            continue;
        } else if constexpr (MS == 16) {
            // fp16: two 16x16x32 sub-steps per K-step, lane quarter lh of sub-step s reads chunk 4 s + lh
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int o = ((4 * s + lh) ^ sw) * 16;
                i32x4 fa16[TM], fb16[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa16[i] = *reinterpret_cast<const i32x4*>(sb + a_rd + i * MS * BKB + o);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb16[j] = *reinterpret_cast<const i32x4*>(sb + b_rd + j * MS * BKB + o);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fa16[i]), __builtin_bit_cast(half8, fb16[j]), acc[i][j], 0, 0, 0);
            }
            continue;
        }
        if constexpr (MS == 32) {
        // fragments are read one sub-step ahead of their MFMAs and the {ds_read group, MFMA group} order is pinned
        // (see conv3x3_halo_kernel): +5 % over hipcc's own read-then-wait schedule
        i32x4 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const i32x4*>(sb + a_rd + i * 32 * BKB + coff[0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const i32x4*>(sb + b_rd + j * 32 * BKB + coff[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cur = s & 1, nx = cur ^ 1;
            if (s < 3) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nx][i] = *reinterpret_cast<const i32x4*>(sb + a_rd + i * 32 * BKB + coff[s + 1]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[nx][j] = *reinterpret_cast<const i32x4*>(sb + b_rd + j * 32 * BKB + coff[s + 1]);
                __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(fa[cur][i], fb[cur][j], acc[i][j]);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN * (sizeof(T) == 4 ? 4 : 1), 0);
        }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): my fragment reads are done ...
    __builtin_amdgcn_s_barrier();        // ... and so are everybody else's: the ring can be reused by the epilogue

    if constexpr (DIRECT) {
        // one image per tile, no split-K, NHWC output (the launcher checked): stores straight from the (transposed) accumulators
        halo_epilogue_direct<BM, BN, WAVES_M, WAVES_N>(pe, g, acc, wave, lane, n0, pre_bq, pe.alpha, pre_we);
    } else if constexpr (sizeof(T) == 4) {
        igemm_epilogue<T, CONV, BM, BN, WAVES_M, WAVES_N, true, MS>(pe, g, acc, smem, wave, lane, n0, rowtab, rowtab_n);
    } else {
        if (CONV ? p.splitk > 1 : p.c_f32 != 0)
            igemm_epilogue<T, CONV, BM, BN, WAVES_M, WAVES_N, true, MS>(pe, g, acc, smem, wave, lane, n0);
        else
            igemm_epilogue<T, CONV, BM, BN, WAVES_M, WAVES_N, false, MS>(pe, g, acc, smem, wave, lane, n0);
    }
}

// =============================================================================================
// conv3x3_halo_kernel: 3x3 / stride 1 / pad 1 convolution (the ResBlock convs = ~85 % of the step's FLOPs).
//
// The generic kernel above fetches every input line 9 times (once per filter tap); ablation on MI355X showed that this
// A-operand gather is what limits it (858 -> 1062 TFLOP/s with the A DMA removed).  Here a workgroup stages the
// (8+2) x (16+2) pixel HALO PATCH of its 8x16 output tile ONCE per 64-channel chunk (180 rows x 128 B, double
// buffered) and the 9 taps read shifted windows of that patch from LDS: A traffic drops ~7x, the per-tap DMA is only
// the 128 x 128 B weight tile (L2 resident, shared by all workgroups).
//   * patch rows are swizzled like GEMM rows but keyed by the patch COLUMN: chunk ^ ((px>>1)&7) for patch row (py, px); a fragment
//     read covers 16 consecutive columns, so the ds_read_b128 lane groups hit distinct 16-byte slots as in the GEMM tiles, and a tap
//     shift (dy, dx) changes the key only through dx (three address registers per lane in the 16x16x32 paths, see `acur`).
//   * DMA schedule per K-step (chunk cc, tap t), per wave: [4 weight-tile pieces for step+1][1 patch piece of chunk cc+1
//     (taps 0..5)].  vmcnt retires in order, so the wait for step s's weights is vmcnt(1) when a patch piece was
//     issued after them and vmcnt(0) otherwise: every patch piece gets two full K-steps to land.
//   * everything else (MFMA tiling, buffer-descriptor OOB zero fill, concat sources, epilogue) is shared with igemm.
// Requirements (checked by the launcher, otherwise the generic kernel runs): ksize 3, stride 1, pad 1, Wo % 16 == 0 (or Wo == 8, half tiles),
// Ho % 8 == 0 (with or without the virtual nearest-2x upsampling of the input).
// =============================================================================================
// UPS = true: the conv input is the nearest-2x upsampling of x (Upsample.conv, unet_openai.py:236-241).  The 8x16
// output tile then reads only a (8/2+2) x (16/2+2) = 6 x 10 patch of the STORED half-resolution tensor: output pixel
// (u, v) + tap reads patch row ((u+1)>>1, (v+1)>>1) -- the 2x image is never materialised and A traffic drops ~28x.
// MS = MFMA shape of the fp16 products: 32 = v_mfma_f32_32x32x16_f16 (2x2 tiles per wave), 16 = v_mfma_f32_16x16x32_f16 (4x4 tiles).
// Same flops, same LDS bytes and reads per K-step; under the chip's power limit the 16x16x32 form sustains ~1.14x the issued
// FLOP/s (tools/probe/mfma_shape.hip: 1.76 vs 1.54 PFLOP/s for this wave tile with every operand re-read from LDS).
// SKIP = true (16x16x32 instances): the ResBlock's 1x1 skip_connection conv (unet_openai.py:352, 385: `skip_connection(x) + h`) rides in
// the same accumulators.  After the last 3x3 tap the operand ring is reused as a plain two-stage GEMM ring (128 pixel rows x 128 B of the
// block input | 128 x 128 B of the 1x1 weight per K-step) and the K loop continues over the input's channels: the skip tensor is never
// written to or read back from HBM (2 x M x Cout elements per block), and its MFMA work runs at this kernel's rate instead of in an
// HBM-bound launch of its own.
// per-column epilogue terms of a patch-mode tile (one image): bias[col] + cbias[image][col], for igemm_epilogue's pre_bcol
template <int TN, int MS>
__device__ __forceinline__ void prefetch_bcol(const IgemmP& p, int ncols, int col0, int lane, int n_first, float (&out)[TN]) {
    const int lr = AccLayout<MS>::col(lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + j * MS + lr;
        const bool ok = col < ncols;
        float v = (ok && p.bias && p.bias_mode == 1) ? p.bias[col] : 0.0f;
        if (ok && p.cbias) v += p.cbias[(long long)n_first * p.cbias_stride + col];
        out[j] = v;
    }
}

template <typename T, int BN, int WAVES_M, int WAVES_N, bool UPS, int BSTAGES, bool GN, bool SPLIT = false, int MS = 32, bool SKIP = false, bool STREAM = false,
          bool KSPLIT = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, WAVES_N == 4 ? 1 : 2) void conv3x3_halo_kernel(const IgemmP p) {
    // KSPLIT: the K slices of a split conv (p.splitk workgroups per tile, gridDim.y) are instances of their own.  As a run-time mode of
    // every instance (round 4's first form) the chunk loop's bounds, the second epilogue of the fp16 kernels and the workspace pointer cost
    // the UNSPLIT launches 2-6 % (fp16, 128 -> 128 at 256 x 256: 0.399 -> 0.419 ms, same box) -- found by bisecting the libraries of the
    // round's commits on one box.
    static_assert(!KSPLIT || (!STREAM && !UPS && BN == 128), "K slices: the 128-column 4-wave instances");
    static_assert(!SKIP || (MS == 16 && !UPS && WAVES_M == 2), "fused skip conv: 16x16x32 instances of the 8x16 tile");
    static_assert(!SPLIT || (sizeof(T) == 4 && MS == 16), "the split-fp16 product is a mode of fp32 storage, on 16x16x32 MFMAs");
    static_assert(MS == 32 || SPLIT || sizeof(T) == 2, "the 16x16x32 shape exists for the fp16 products only");
    constexpr bool XF = GN || SPLIT;  // the staged patch pieces are rewritten in place by the wave that DMA'd them
    // fp32-storage split instances (not the 32-column NCHW head): swapped MFMA operands + stores straight from the accumulators
    constexpr bool DIRECT = SPLIT && MS == 16 && BN >= 64;
    // STREAM: a workgroup runs p.tpw consecutive pixel tiles of one image as ONE stream of chunks: the last chunk of tile k stages
    // chunk 0 of tile k + 1 (patch pieces, their normalise / split pass, the first weight tile) exactly like any other next chunk, so
    // only the first tile of a workgroup pays the prologue (DMA round trip + rewrite of the whole patch with the matrix pipe idle:
    // 5.9 of a 48 us workgroup life on the 36-step layers, tools/debug/halo_stamps.py).  Needs an epilogue that leaves the operand
    // ring alone (DIRECT); the fused skip phase reuses the ring and stays single-tile.
    static_assert(!STREAM || (DIRECT && !SKIP && !UPS), "streaming instances: direct epilogue, no fused skip phase");
    constexpr int NW = WAVES_M * WAVES_N;      // 4 waves: 8x16 tile, 8 waves: 16x16 tile
    // WAVES_N = 4 (BN = 256): the 8x16 pixel tile with EIGHT waves, 2 x 4, each still 64 x 64 -- one workgroup covers the two N-tiles of a
    // 256-column conv, so the patch is fetched from HBM and normalised / split ONCE for both (one 8-wave workgroup per CU instead of two
    // 4-wave ones: the same waves per SIMD): +3.5 ... 10 % on the 256- and 512-column convs.  (The mirror image for 128-column convs -- a
    // 16 x 16 pixel tile, 4 x 2 waves, halo 1.27x instead of 1.41x and one weight tile per 256 pixels -- measured -1 ... +2 % (round 4, fp16
    // storage, where the weight DMA is the larger share of a K-step: -1 ... +3 %), and the
    // same 2 x 4 form of conv_up4_halo_kernel +0.4 %: neither is used; a three-stage weight ring on this instance: -2 %.)
    constexpr int BM = WAVES_N == 4 ? 64 * WAVES_M : 32 * NW, TH = BM / 16, TW = 16;
    constexpr int PH = UPS ? TH / 2 + 2 : TH + 2, PW = UPS ? TW / 2 + 2 : TW + 2, PR = PH * PW;  // 180 (60) patch rows
    constexpr int PG = (PR + 7) / 8;                                             // 23 (8) DMA groups of 8 rows
    constexpr int LAH = (PG + NW - 1) / NW;                                      // patch pieces per wave
    constexpr int ES = sizeof(T), EPC = 16 / ES, BKB = 128, BK = BKB / ES;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / MS, TN = WN / MS;
    constexpr int GB = BN / 8, LB = GB / NW;
    constexpr int ABUF = PG * 1024, BSTAGE = BN * BKB;
    static_assert(GB % NW == 0 && LAH <= 6 && BSTAGES == 2, "layout");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;              // [2][ABUF]
    char* const sB = smem + 2 * ABUF;   // [BSTAGES][BSTAGE]
    char* const sS = sB + BSTAGES * BSTAGE;  // GN only: [2][1024] scale/shift of the 64 channels of a chunk
    EOD_STAMP_AT(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int tile_m, tile_n;
    map_tile(p, tile_m, tile_n);
    int ntile = 1;  // STREAM: tiles tile_m .. tile_m + ntile - 1 (the launcher keeps a run inside one image)
    if constexpr (STREAM) {
        tile_m *= p.tpw;
        ntile = min(p.tpw, p.tiles_m - tile_m);
    }
    const int n0 = p.n_base + tile_n * BN;
    TileGeom g = make_geom<true, BM>(p, tile_m);  // patch mode: ty0, tx0, n_first (of the tile whose K loop runs)
    // epilogue operands fetched NOW (their latency passes under the prologue's DMA; older than every DMA, so the counted vmcnt waits of
    // the loop are unaffected): per-column bias terms and the weights' scale
    float pre_bcol[DIRECT ? 1 : TN];
    f32x4 pre_bq[DIRECT ? TN : 1];
    int pre_we[DIRECT ? TN : 1];
    if constexpr (DIRECT) {
        prefetch_bcol4<TN>(p, p.Ncols, n0 + wn * WN, lane, g.n_first, pre_bq);
        prefetch_wexp4<TN>(p, p.Ncols, n0 + wn * WN, p.w_row0, lane, pre_we);
    } else {
        prefetch_bcol<TN, MS>(p, p.Ncols, n0 + wn * WN, lane, g.n_first, pre_bcol);
    }
    float wsc1 = 1.0f;
    if constexpr (SPLIT) wsc1 = p.w_scale[1];
    // split-fp16 product: power-of-two operand scale of this tile's image from the bound table(s) (wave-uniform, common.h); the fused
    // skip conv shares the accumulators, so the launch runs on the smaller of the two tensors' scales
    AbScale asc = {EOD_SPLIT_ASCALE, 1.0f};
    const float silu_k = p.gn_silu ? -1.44269504088896341f : 0.0f, silu_c = p.gn_silu ? 0.0f : -__builtin_inff();  // (see xf_finish)
    if constexpr (SPLIT) {
        if (p.a_bound) asc = ab_scale_of(ab_wave_bound(p.a_bound, g.n_first));
        if constexpr (SKIP) {
            if (p.skip_bound) {
                const AbScale k = ab_scale_of(ab_wave_bound(p.skip_bound, g.n_first));
                if (k.s < asc.s) asc = k;
            }
        }
    }

    // ---- patch pieces of this wave: group gi = wave + 4*i, patch row = gi*8 + (lane>>3), slot = lane&7 ----
    const int srow = lane >> 3, sslot = lane & 7;
    // PATCH SWIZZLE: the 16-byte chunk c of patch row (py, px) lives in slot c ^ key(px), key(px) = (px >> 1) & 7 -- a function of the
    // patch COLUMN only.  A fragment read covers 16 consecutive columns of one patch row, so the conflict-free property is the one of
    // the (row >> 1) & 7 key of the GEMM tiles, and a tap shift by (dy, dx) changes the key only through dx: the fragment addresses of a
    // lane are three registers (one per dx) plus compile-time offsets (see `acur` below) instead of ~40 VALU instructions per tap.
    // Register diet (the GN + split instance sits at the 256-VGPR limit of two workgroups per CU): the chunk of a lane's slot is three
    // bits per piece, packed into one register; a piece keeps its pixel index only (the byte offset of either concat source is one
    // multiply at issue time, six times per chunk); the weight rows keep the (row >> 1) & 7 key, one value for all pieces of a lane
    // ((NW*i*8) >> 1 is a multiple of 8).
    unsigned ppix[LAH];
    const int bchunk0 = sslot ^ ((4 * wave + (srow >> 1)) & 7);
    unsigned pck = 0;     // 3 bits per piece: chunk held by this lane's slot
    unsigned pvalid = 0;  // bit i: this lane's row of piece i lies inside the image (zero padding must stay zero)
#pragma unroll
    for (int i = 0; i < LAH; ++i) {
        const int prow = (wave + NW * i) * 8 + srow;
        const int px = prow - (prow / PW) * PW;
        pck |= (unsigned)(sslot ^ ((px >> 1) & 7)) << (3 * i);
    }
    // pixel index / validity of this lane's patch rows for the tile at (ty0, tx0) (STREAM: called again for the next tile of the run)
    auto set_tile_pieces = [&](const TileGeom& tg) {
        pvalid = 0;
#pragma unroll
        for (int i = 0; i < LAH; ++i) {
            const int prow = (wave + NW * i) * 8 + srow;
            const int py = prow / PW, px = prow - py * PW;
            const int hi = (UPS ? tg.ty0 / 2 : tg.ty0) - 1 + py, wi = (UPS ? tg.tx0 / 2 : tg.tx0) - 1 + px;
            const bool ok = (wave + NW * i) < PG && prow < PR && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            if (ok) pvalid |= 1u << i;
            ppix[i] = (unsigned)(hi * p.W + wi);
        }
    };
    set_tile_pieces(g);
    auto pchunk_of = [&](int i) { return (int)((pck >> (3 * i)) & 7u); };
    unsigned b_v[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int row = (wave + NW * i) * 8 + srow;
        b_v[i] = (n0 + row < p.Ncols) ? (unsigned)(row * p.Cin * ES) + bchunk0 * 16 : EOD_OOB;
    }
    const __amdgpu_buffer_rsrc_t rsA0 = make_rsrc(p.a0 + (long long)g.n_first * p.H * p.W * p.C0 * ES);
    const __amdgpu_buffer_rsrc_t rsA1 = make_rsrc(p.a1 ? p.a1 + (long long)g.n_first * p.H * p.W * p.C1 * ES : p.a0);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.b + (long long)n0 * p.Cin * ES);
    const int tapstride = p.Cout * p.Cin * ES;
    // GN: per-(image, channel) {scale, shift} of the fused GroupNorm(+FiLM), fp32 pairs; exact num_records so that the
    // 64-channel chunk read of a channel tail returns zeros instead of touching memory behind the table
    __amdgpu_buffer_rsrc_t rsS = rsB;
    if constexpr (GN)
        rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gn_ss) + (long long)g.n_first * p.Cin * 2, 0, p.Cin * 8, 0x00020000);

    // chunk state (wave-uniform): source, channel offset inside the source, tail flag, weight K offset (bytes)
    struct Chunk {
        int src, kin, cw;
        unsigned bk;
        bool ktail;
    };
    auto chunk_of = [&](int cc) {
        Chunk c;
        c.src = cc >= p.kc0 ? 1 : 0;
        const int ci = c.src ? cc - p.kc0 : cc;
        c.kin = ci * BK;
        c.cw = c.src ? p.C1 : p.C0;
        c.bk = (unsigned)(((c.src ? p.C0 : 0) + c.kin) * ES);
        c.ktail = c.kin + BK > c.cw;
        return c;
    };
    auto issue_patch_piece = [&](int i, const Chunk& c, char* abuf) {
        const int pc = pchunk_of(i);
        unsigned v = ((pvalid >> i) & 1u) ? ppix[i] * (unsigned)(c.cw * ES) + pc * 16 : EOD_OOB;
        if (__builtin_expect(c.ktail, 0)) v = (c.kin + pc * EPC < c.cw) ? v : EOD_OOB;
        if (c.src)
            blds16(rsA1, v, (unsigned)(c.kin * ES), abuf + (wave + NW * i) * 1024);
        else
            blds16(rsA0, v, (unsigned)(c.kin * ES), abuf + (wave + NW * i) * 1024);
    };
    // GN: wave 0 stages the chunk's 64 x {scale, shift} (512 B) next to the patch; every wave then normalises the patch
    // pieces IT has DMA'd (its own vmcnt tells it when they have landed): x -> silu(x*scale + shift), in place in LDS,
    // once per element (the 9 taps read the normalised patch).  Rows outside the image (conv zero padding) and masked
    // channel tails are left untouched (zero).
    auto issue_ss = [&](const Chunk& c, char* ssbuf) {
        // the whole offset goes through the range-checked per-lane part (exact num_records: a channel tail reads zeros)
        const unsigned v = lane < 32 ? (unsigned)(lane * 16 + ((c.src ? p.C0 : 0) + c.kin) * 8) : EOD_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (lds_void*)ssbuf, 16, v, 0, 0, 0);
    };
    // the rewrite of a piece in two parts: its LDS reads (the piece's 16 bytes of this lane + the scale / shift entries of its channels),
    // and the arithmetic + write-back
    struct XfPre {
        i32x4 raw;
        f32x4 q[GN ? (ES == 2 ? 4 : 2) : 1];
    };
    auto xf_load = [&](int i, char* abuf, const char* ssbuf, XfPre& pre) {
        const int pc = pchunk_of(i);
        pre.raw = *reinterpret_cast<const i32x4*>(abuf + (wave + NW * i) * 1024 + lane * 16);
        if constexpr (GN) {
            const float* sp = reinterpret_cast<const float*>(ssbuf) + pc * EPC * 2;
#pragma unroll
            for (int k = 0; k < (ES == 2 ? 4 : 2); ++k) pre.q[k] = *reinterpret_cast<const f32x4*>(sp + 4 * k);
        }
    };
    auto xf_finish = [&](int i, const Chunk& c, char* abuf, const XfPre& pre) {
        const int pc = pchunk_of(i);
        const bool ok = ((pvalid >> i) & 1u) && (!c.ktail || (c.kin + pc * EPC < c.cw));
        char* ptr = abuf + (wave + NW * i) * 1024 + lane * 16;
        const i32x4 raw = pre.raw;
        i32x4 outv = raw;
        if constexpr (GN) {
            if constexpr (ES == 2) {
                const half8 h = __builtin_bit_cast(half8, raw);
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = (float)h[e] * pre.q[e >> 1][(e & 1) * 2] + pre.q[e >> 1][(e & 1) * 2 + 1];
                    v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(v, silu_k, silu_c)));  // (select-free SiLU, see below)
                    o[e] = (half_t)v;
                }
                outv = __builtin_bit_cast(i32x4, o);
            } else {
                const f32x4 f = __builtin_bit_cast(f32x4, raw);
                f32x4 o;
                const f32x4 q0 = pre.q[0], q1 = pre.q[GN && ES != 2 ? 1 : 0];
                const float sc[4] = {q0[0], q0[2], q1[0], q1[2]}, sh[4] = {q0[1], q0[3], q1[1], q1[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = f[e] * sc[e] + sh[e];
                    if constexpr (SPLIT) {
                        // x * sigmoid(x) as v * rcp(1 + 2^(k v + c)) with (k, c) = (-log2 e, 0), or (0, -inf) without a SiLU: the exponential
                        // is then 0 and the factor exactly 1 -- the same bits as the branch-free select it replaces, four selects fewer
                        // (v_exp_f32 / v_rcp_f32, ~2 ulp: far inside the 2^-22 of the split product that consumes it)
                        v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(v, silu_k, silu_c)));
                    } else {
                        if (p.gn_silu) v = silu_f<SPLIT>(v);   // (the exact-fp32 mode keeps the IEEE expf / divide form)
                    }
                    o[e] = v;
                }
                outv = __builtin_bit_cast(i32x4, o);
            }
            if constexpr (!SPLIT) {
                if (!ok) outv = raw;  // conv zero padding / masked channel tail: stays zero
            }
        }
        if constexpr (SPLIT) {
            // fp32 chunk -> its half of the pair's [8 x hi | 8 x lo] image; both lanes of a pair take part.  Conv zero padding and masked
            // channel tails (their lanes read zeros, which a GroupNorm shift would move) go through scale 0: they stay zero
            const float sl = (GN && !ok) ? 0.0f : asc.s;
            *reinterpret_cast<i32x4*>(ptr) = split_pair_exchange_scaled(__builtin_bit_cast(f32x4, outv), sl, (pc & 1) != 0);
        } else {
            if (ok) *reinterpret_cast<i32x4*>(ptr) = outv;
        }
    };
    auto transform_piece = [&](int i, const Chunk& c, char* abuf, const char* ssbuf) {
        XfPre pre;
        xf_load(i, abuf, ssbuf, pre);
        xf_finish(i, c, abuf, pre);
    };
    // (the K-tail mask of a lane's weight offsets is a property of the CHUNK: bve holds the current chunk's, rebuilt once per chunk, so
    //  that the eight taps that stay inside it issue their weight DMA without a select per piece)
    unsigned bve[LB];
    auto weight_offsets = [&](const Chunk& c, unsigned (&out)[LB]) {
        const bool kok = !c.ktail || (c.kin + (SPLIT ? (bchunk0 >> 1) * 8 : bchunk0 * EPC) < c.cw);
#pragma unroll
        for (int i = 0; i < LB; ++i) out[i] = kok ? b_v[i] : EOD_OOB;
    };
    auto issue_weights_at = [&](int tap, const Chunk& c, const unsigned (&off)[LB], char* bst) {
        const unsigned soff = (unsigned)(tap * tapstride) + c.bk;
#pragma unroll
        for (int i = 0; i < LB; ++i) blds16(rsB, off[i], soff, bst + (wave + NW * i) * 1024);
    };
    auto issue_weights = [&](int tap, const Chunk& c, char* bst) {
        unsigned off[LB];
        weight_offsets(c, off);
        issue_weights_at(tap, c, off, bst);
    };

    typename AccLayout<MS>::vec acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < AccLayout<MS>::R; ++r) acc[i][j][r] = 0.0f;

    // ---- fragment addressing: lane (lr = row inside an MS-row block, lh = which 8-k slice of the MFMA's K it supplies) ----
    const int lr = lane & (MS - 1), lh = MS == 32 ? lane >> 5 : lane >> 4;
    int prow0[TM], pyo[TM], pxo[TM];  // patch row of this lane's output pixel for tap (0,0) / its tile coordinates
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * MS + lr;
        pyo[i] = r >> 4;
        pxo[i] = r & 15;
        prow0[i] = pyo[i] * PW + pxo[i];
    }
    const int bsw = (lr >> 1) & 7;
    int bcoff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bcoff[s] = ((2 * s + lh) ^ bsw) * 16;
    const int b_rd = (wn * WN + lr) * BKB;
    // 16x16x32 paths: a lane's pixel of MFMA row block i is column lr of tile row wm*(WM/16) + i, so for tap (dy, dx) its fragment sits at
    //   [first tile row of the wave, column (lr + dx)] + (i + dy) patch rows        (UPS: column (lr + dx + 1) >> 1, (i + dy + 1) >> 1 rows)
    // and the swizzle key depends on dx alone: acur[v][dx] = LDS offset of that fragment in the CURRENT patch buffer for i = dy = 0
    // (v = 0 / 1: [8 x hi] / [8 x lo] chunk of the lane's pair in split mode, sub-step 0 / 1 in fp16 storage); the (i, dy) part is a
    // compile-time offset of the ds_read, and the buffer flip at a chunk boundary is one add per register.
    int acur[2][3] = {{0, 0, 0}, {0, 0, 0}};
    auto aimm = [](int i, int dy) { return (UPS ? ((i + dy + 1) >> 1) : (i + dy)) * PW * BKB; };
    if constexpr (MS == 16) {
        constexpr int LROW = WM / 16;  // tile rows per wave (even)
        static_assert(LROW % 2 == 0, "the UPS row split needs an even first tile row per wave");
        const int c0 = SPLIT ? 2 * ((0x2130 >> (4 * lh)) & 3) : lh, c1 = SPLIT ? c0 + 1 : 4 + lh;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int pxc = UPS ? ((lr + dx + 1) >> 1) : lr + dx;
            const int key = (pxc >> 1) & 7;
            const int rowb = ((UPS ? (wm * LROW) >> 1 : wm * LROW) * PW + pxc) * BKB;
            acur[0][dx] = rowb + ((c0 ^ key) << 4);
            acur[1][dx] = rowb + ((c1 ^ key) << 4);
        }
    }

    const int KC = p.kc0 + p.kc1;
    // split-K (maps with too few pixel tiles to fill the chip): gridDim.y workgroups share a tile, each takes a run of channel chunks (all
    // nine taps of a chunk stay together) and writes a raw fp32 partial tile; the fused skip phase rides in the last one
    int kc_begin = 0, kc_end = KC;
    if constexpr (KSPLIT) {
        kc_begin = (int)blockIdx.y * p.splitk_per;
        kc_end = (int)blockIdx.y == p.splitk - 1 ? KC : kc_begin + p.splitk_per;
    }
    const int NSTEP = (kc_end - kc_begin) * 9;
    const int TSTEP = NSTEP * ntile;  // K-steps of the whole run of tiles
    // ---- prologue: whole patch of the first chunk + weights of the first step ----
    {
        const Chunk c0 = chunk_of(kc_begin);
#pragma unroll
        for (int i = 0; i < LAH; ++i)
            if ((wave + NW * i) < PG) issue_patch_piece(i, c0, sA);
        if constexpr (GN) {
            if (wave == 0) issue_ss(c0, sS);
        }
        issue_weights(0, c0, sB);
        if constexpr (XF) {
            // chunk 0: everything has to land before the first tap anyway; normalise / split the own pieces now
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (GN) __builtin_amdgcn_s_barrier();  // wave 0's scale/shift table is visible
#pragma unroll
            for (int i = 0; i < LAH; ++i)
                if ((wave + NW * i) < PG) transform_piece(i, c0, sA, sS);
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): my LDS writes are done before the loop's first barrier
        }
    }
    // DMA issued AFTER the weights of the step we are about to wait for may stay in flight (vmcnt retires in order):
    //   pp1 : patch pieces (and wave 0's scale / shift table) issued one step ago, behind that step's weights
    EOD_STAMP_AT(1);
    int pp1 = 0;
    int step = 0;   // K-step of the run (weight stage = step & 1)
    int qpar = 0;   // patch buffer (and scale / shift table) of the current chunk
    for (int k = 0; k < ntile; ++k) {
    for (int cc = kc_begin; cc < kc_end; ++cc) {
        const Chunk cur = chunk_of(cc);
        // the chunk behind this one in the stream: the tile's next one, or chunk 0 of the run's next tile (a stream is never split in K)
        const bool next_tile = STREAM && cc + 1 == kc_end && k + 1 < ntile;
        const bool has_next = cc + 1 < kc_end || next_tile;
        const Chunk nxt = chunk_of(cc + 1 < kc_end ? cc + 1 : (next_tile ? 0 : cc));
        if constexpr (STREAM) {
            if (next_tile) set_tile_pieces(make_geom<true, BM>(p, tile_m + k + 1));  // this chunk's taps stage the NEXT tile's patch
        }
        const char* abuf = sA + qpar * ABUF;
        char* abuf_next = sA + (qpar ^ 1) * ABUF;
        weight_offsets(cur, bve);
#pragma unroll
        for (int t = 0; t < 9; ++t, ++step) {
            // (STREAM: the first step of a later tile was waited for and fenced in front of the previous tile's epilogue, see below)
            EOD_TSTAMP_AT(4 * t + 0);
            if (!(STREAM && t == 0 && cc == kc_begin && k > 0)) {
                if (pp1 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (pp1 == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                if constexpr (XF) __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): in-place normalisation / split writes are done
                __builtin_amdgcn_s_barrier();
            }
            EOD_TSTAMP_AT(4 * t + 1);
            // DMA for the next step: weights first, then (taps 0..LAH-1) one piece of the next chunk's patch
            if (step + 1 < TSTEP) {
                if (t < 8) issue_weights_at(t + 1, cur, bve, sB + ((step + 1) & 1) * BSTAGE);
                else issue_weights(0, nxt, sB + ((step + 1) & 1) * BSTAGE);
            }
            pp1 = 0;
            if (t < LAH && has_next && (wave + NW * t) < PG) {
                issue_patch_piece(t, nxt, abuf_next);
                pp1 = 1;
            }
            if constexpr (GN) {
                if (t == 0 && has_next && wave == 0) {
                    issue_ss(nxt, sS + (qpar ^ 1) * 1024);
                    ++pp1;
                }
            }
            EOD_TSTAMP_AT(4 * t + 2);
            // ---- MFMAs of tap t: A fragments = patch rows shifted by (dy, dx) ----
            const int dy = t / 3, dx = t - dy * 3;
            const char* bst = sB + (step & 1) * BSTAGE;
            int arow[TM], asw[TM];  // (32x32x16 paths; the 16x16x32 ones read through acur)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int pxc = UPS ? ((pxo[i] + dx + 1) >> 1) : pxo[i] + dx;   // patch column of this lane's pixel for tap (dy, dx)
                const int prow = UPS ? ((pyo[i] + dy + 1) >> 1) * PW + pxc : prow0[i] + dy * PW + dx;
                arow[i] = prow * BKB;
                asw[i] = (pxc >> 1) & 7;
            }
            if constexpr (MS == 16 && SPLIT) {
                // the K-step's 32 k are ONE 16x16x32 sub-step: lane quarter lh supplies chunk pair pi(lh) = {0, 3, 1, 2}[lh] (the
                // permutation keeps every ds_read_b128 lane group on 16 distinct 16-byte slots; A and B use the same one, so the
                // contraction is unchanged).  [8 x hi] at chunk 2 pi, [8 x lo] behind it; 3 x 16 MFMAs, smallest terms first.
                const int ch = 2 * ((0x2130 >> (4 * lh)) & 3);
                i32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) al[i] = *reinterpret_cast<const i32x4*>(sA + acur[1][dx] + aimm(i, dy));
#pragma unroll
                for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const i32x4*>(bst + b_rd + j * MS * BKB + ((ch ^ bsw) << 4));
#pragma unroll
                for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const i32x4*>(sA + acur[0][dx] + aimm(i, dy));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[j]), __builtin_bit_cast(half8, al[i]), acc[i][j], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, al[i]), __builtin_bit_cast(half8, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j) bl[j] = *reinterpret_cast<const i32x4*>(bst + b_rd + j * MS * BKB + (((ch + 1) ^ bsw) << 4));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bl[j]), __builtin_bit_cast(half8, ah[i]), acc[i][j], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah[i]), __builtin_bit_cast(half8, bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[j]), __builtin_bit_cast(half8, ah[i]), acc[i][j], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah[i]), __builtin_bit_cast(half8, bh[j]), acc[i][j], 0, 0, 0);
            } else if constexpr (MS == 16) {
                // fp16 storage: the K-step's 64 k are two 16x16x32 sub-steps; lane quarter lh of sub-step s reads chunk 4 s + lh
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int ch = 4 * s + lh;
                    i32x4 fa16[TM], fb16[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa16[i] = *reinterpret_cast<const i32x4*>(sA + acur[s][dx] + aimm(i, dy));
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb16[j] = *reinterpret_cast<const i32x4*>(bst + b_rd + j * MS * BKB + ((ch ^ bsw) << 4));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fa16[i]), __builtin_bit_cast(half8, fb16[j]), acc[i][j], 0, 0, 0);
                }
            } else {
            // fragments are read ONE sub-step ahead of the MFMAs that consume them (two register sets), and the order
            // {4 ds_read of s+1, 4 MFMA of s} is pinned: left to itself hipcc reads each fragment right before its MFMA
            // and waits lgkmcnt(0) ~6 times per K-step, exposing the LDS latency every time.
            i32x4 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[0][i] = *reinterpret_cast<const i32x4*>(abuf + arow[i] + (((0 + lh) ^ asw[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const i32x4*>(bst + b_rd + j * 32 * BKB + bcoff[0]);
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int cur = s & 1, nx = cur ^ 1;
                if (s < 3) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[nx][i] = *reinterpret_cast<const i32x4*>(abuf + arow[i] + (((2 * (s + 1) + lh) ^ asw[i]) << 4));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fb[nx][j] = *reinterpret_cast<const i32x4*>(bst + b_rd + j * 32 * BKB + bcoff[s + 1]);
                    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(fa[cur][i], fb[cur][j], acc[i][j]);
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN * (sizeof(T) == 4 ? 4 : 1), 0);
            }
            }
            EOD_TSTAMP_AT(4 * t + 3);
            if constexpr (XF) {
                // piece (t-2) of the NEXT chunk was issued two steps ago and is covered by this step's vmcnt wait;
                // the scale/shift table (wave 0, tap 0) became visible with this step's barrier (t >= 2).
                // (spreading these ~110 VALU ops into the MFMA gaps with sched_group_barrier was measured: 3 % SLOWER)
                // (round 4: its LDS reads issued in FRONT of the step's MFMAs instead, +12 live registers: single layers -1 %, step +-0)
                if (t >= 2 && t - 2 < LAH && has_next && (wave + NW * (t - 2)) < PG)
                    transform_piece(t - 2, nxt, abuf_next, sS + (qpar ^ 1) * 1024);
            }
        }
        EOD_TSTAMP_AT(36);
#ifdef EOD_TSTAMP
        if (GN && wave == 1 && cc == 1 && k == 0 && blockIdx.x < 2048) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane < 37) g_eod_tstamp[blockIdx.x][lane] = reinterpret_cast<const unsigned long long*>(sA + ABUF + PR * 128)[lane];
        }
#endif
        if constexpr (MS == 16) {  // the next chunk reads the other patch buffer
            const int flip = qpar ? -ABUF : ABUF;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                acur[0][dx] += flip;
                acur[1][dx] += flip;
            }
        }
        qpar ^= 1;
    }
    if constexpr (SKIP) {
      if (!KSPLIT || (int)blockIdx.y == p.splitk - 1) {
        // ---- 1x1 skip conv over the block input: GEMM-layout ring in the same LDS, K-step = 32 / 64 channels of one source ----
        constexpr int LA = BM / 8 / NW;                    // 8-row pieces of the pixel tile per wave (4)
        constexpr int STG_A = BM * BKB, STG = STG_A + BSTAGE;
        static_assert(2 * STG <= 2 * ABUF + BSTAGES * BSTAGE, "the skip phase reuses the operand ring");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();  // every wave is done with the 3x3 loop's buffers
        const int SC = p.SC0 + p.SC1;
        const __amdgpu_buffer_rsrc_t rsX0 = make_rsrc(p.sx0 + (long long)g.n_first * p.H * p.W * p.SC0 * ES);
        const __amdgpu_buffer_rsrc_t rsX1 = make_rsrc(p.sx1 ? p.sx1 + (long long)g.n_first * p.H * p.W * p.SC1 * ES : p.sx0);
        const __amdgpu_buffer_rsrc_t rsW2 = make_rsrc(p.b2 + (long long)n0 * SC * ES);
        // row r of the tile = pixel (ty0 + r / 16, tx0 + r % 16) -- always inside the image; rows use the (row >> 1) & 7 swizzle, which
        // for this lane's pieces is bchunk0 (the weight rows' value)
        unsigned xpix[LA], wv[LB];
        unsigned xok = 0;  // bit i: the pixel exists (8-wide maps: the right half of the tile lies behind the map)
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int row = (wave + NW * i) * 8 + srow;
            xpix[i] = (unsigned)((g.ty0 + (row >> 4)) * p.W + g.tx0 + (row & 15));
            if (g.tx0 + (row & 15) < p.W) xok |= 1u << i;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int row = (wave + NW * i) * 8 + srow;
            wv[i] = (n0 + row < p.Ncols) ? (unsigned)(row * SC * ES) + bchunk0 * 16 : EOD_OOB;
        }
        auto issue_skip = [&](int kt, char* stg) {
            const int src = kt >= p.skc0 ? 1 : 0;
            const int kin = (src ? kt - p.skc0 : kt) * BK, cw = src ? p.SC1 : p.SC0;
            const bool ktail = kin + BK > cw;
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                unsigned v = ((xok >> i) & 1u) ? xpix[i] * (unsigned)(cw * ES) + bchunk0 * 16 : EOD_OOB;
                if (ktail) v = (kin + bchunk0 * EPC < cw) ? v : EOD_OOB;
                if (src)
                    blds16(rsX1, v, (unsigned)(kin * ES), stg + (wave + NW * i) * 1024);
                else
                    blds16(rsX0, v, (unsigned)(kin * ES), stg + (wave + NW * i) * 1024);
            }
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                unsigned v = wv[i];
                if (ktail) v = (kin + (SPLIT ? (bchunk0 >> 1) * 8 : bchunk0 * EPC) < cw) ? v : EOD_OOB;
                blds16(rsW2, v, (unsigned)(((src ? p.SC0 : 0) + kin) * ES), stg + STG_A + (wave + NW * i) * 1024);
            }
        };
        const int KS = p.skc0 + p.skc1;
        const int a_rd = (wm * WM + lr) * BKB;
        const int b_rd2 = STG_A + (wn * WN + lr) * BKB;
        if (KS > 0) issue_skip(0, smem);
        for (int kt = 0; kt < KS; ++kt) {
            char* stg = smem + (kt & 1) * STG;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (SPLIT) {  // this wave's pixel pieces have landed: fp32 chunk pairs -> [8 x hi | 8 x lo] (x 16) in place
#pragma unroll
                for (int i = 0; i < LA; ++i) {
                    char* ptr = stg + (wave + NW * i) * 1024 + lane * 16;
                    const f32x4 f = *reinterpret_cast<const f32x4*>(ptr);
                    *reinterpret_cast<i32x4*>(ptr) = split_pair_exchange_scaled(f, asc.s, (bchunk0 & 1) != 0);
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
            }
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < KS) issue_skip(kt + 1, smem + ((kt + 1) & 1) * STG);
            if constexpr (SPLIT) {
                const int ch = 2 * ((0x2130 >> (4 * lh)) & 3);
                const int ohi = (ch ^ bsw) * 16, olo = ((ch + 1) ^ bsw) * 16;
                i32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) al[i] = *reinterpret_cast<const i32x4*>(stg + a_rd + i * MS * BKB + olo);
#pragma unroll
                for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const i32x4*>(stg + b_rd2 + j * MS * BKB + ohi);
#pragma unroll
                for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const i32x4*>(stg + a_rd + i * MS * BKB + ohi);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[j]), __builtin_bit_cast(half8, al[i]), acc[i][j], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, al[i]), __builtin_bit_cast(half8, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j) bl[j] = *reinterpret_cast<const i32x4*>(stg + b_rd2 + j * MS * BKB + olo);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bl[j]), __builtin_bit_cast(half8, ah[i]), acc[i][j], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah[i]), __builtin_bit_cast(half8, bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[j]), __builtin_bit_cast(half8, ah[i]), acc[i][j], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah[i]), __builtin_bit_cast(half8, bh[j]), acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int o = ((4 * s2 + lh) ^ bsw) * 16;
                    i32x4 fa16[TM], fb16[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa16[i] = *reinterpret_cast<const i32x4*>(stg + a_rd + i * MS * BKB + o);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb16[j] = *reinterpret_cast<const i32x4*>(stg + b_rd2 + j * MS * BKB + o);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fa16[i]), __builtin_bit_cast(half8, fb16[j]), acc[i][j], 0, 0, 0);
                }
            }
        }
      }
    }
    if constexpr (STREAM) {
        // The direct epilogue leaves LDS alone.  Before it, the first barrier of the NEXT tile: this wave's DMA for that tile's first step
        // has landed (the weights issued at the last tap; nothing younger is in flight) and every wave is done with the last tap's reads
        // -- so the epilogue's loads and stores are never waited for by a counted vmcnt of the loop before they are a K-step old.
        if (k + 1 < ntile) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
    } else {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();        // every wave is done with the operand buffers: reuse them for the epilogue
    }
    EOD_STAMP_AT(2);
    IgemmP pe = p;
    if constexpr (KSPLIT) pe.y = p.y + (long long)blockIdx.y * p.M * p.Cout * 4;  // raw fp32 partial tile of this K slice (the launcher cleared bias / residual / statistics)
    if constexpr (SPLIT) {
        pe.alpha = p.alpha * wsc1 * asc.inv;  // undo the weight and activation scales (exact powers of two)
        if constexpr (DIRECT) halo_epilogue_direct<BM, BN, WAVES_M, WAVES_N>(pe, g, acc, wave, lane, n0, pre_bq, pe.alpha, pre_we);
        else igemm_epilogue<T, true, BM, BN, WAVES_M, WAVES_N, true, MS>(pe, g, acc, smem, wave, lane, n0, nullptr, 0, pre_bcol);
    } else if constexpr (sizeof(T) == 2) {
        igemm_epilogue<T, true, BM, BN, WAVES_M, WAVES_N, KSPLIT, MS>(pe, g, acc, smem, wave, lane, n0, nullptr, 0, pre_bcol);  // (K slices: fp32 partial tiles)
    } else {
        igemm_epilogue<T, true, BM, BN, WAVES_M, WAVES_N, sizeof(T) == 4, MS>(pe, g, acc, smem, wave, lane, n0, nullptr, 0, pre_bcol);
    }
    if constexpr (STREAM) {
        if (k + 1 < ntile) {
            g = make_geom<true, BM>(p, tile_m + k + 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < AccLayout<MS>::R; ++r) acc[i][j][r] = 0.0f;
        }
    }
    }
    EOD_STAMP_AT(3);
}

// =============================================================================================
// conv_up4_halo_kernel: 3x3 / pad 1 conv of the NEAREST-2x upsampling of x (Upsample.conv, unet_openai.py:236-241) in 4/9 of the MACs.
//
// Output pixel (2i+p, 2j+q) reads upsampled rows 2i+p-1 .. 2i+p+1, i.e. stored rows {i-1, i, i} (p = 0) or {i, i, i+1} (p = 1), and the
// same for columns: every output PARITY CLASS (p, q) is a 2x2-tap conv of the stored (H x W) map whose weights are sums of the original
// taps (rows: p = 0 -> [w0 | w1+w2], p = 1 -> [w0+w1 | w2]).  The host packs those sums as ONE weight tensor with 4*Cout output columns,
// column block `cls` = 2p+q holding its 2x2 kernel in the 3x3 tap slots (dy' in {p, p+1}, dx' in {q, q+1}; the other five slots are never
// read), in the usual [tap][4*Cout][Cin] format (split fp16 pairs for fp32 storage).  A workgroup = one 8x16 tile of STORED positions x
// one class x 128 output channels: the ordinary (8+2) x (16+2) halo patch of the stored tensor, FOUR taps per 32/64-channel chunk instead
// of nine, and the epilogue scatters to the (2i+p, 2j+q) positions of the (2H x 2W) output (decode_row, parity mode).
// Schedule per chunk (4 K-steps): step s issues the weight tile of step s+1 and (s < 2) patch pieces 3s .. 3s+2 of the NEXT chunk; a
// piece is split in place (fp32 storage) two steps after it was issued, behind that step's MFMAs.  16x16x32 MFMAs only.
// =============================================================================================
// BWD = true: the BACKWARD-DATA of that conv with the same machinery (training).  dX[m][n] = sum over the four classes of a 2x2-tap conv
// of G_pq[i][j] = dY[2i+p][2j+q] (the class's stride-2 view of the output gradient) with the transposed class kernels: class (p, q)
// uses tap slots dy' in {1-p, 2-p}, dx' in {1-q, 2-q}.  The K loop runs over (channel chunk of dY, class): every (chunk, class) pair is
// a "chunk" of the forward schedule with its own patch (gathered at pixel stride 2) and four taps, all accumulating into ONE dense
// (H x W) output tile -- no (2H x 2W) intermediate, no 2x2 sum pool.  Weights: eod_pack_conv_weight_dgrad of the forward class-kernel
// tensor [4*Cy][Cx][3][3] = [slot][Cx rows][4*Cy], class block `cls` at K offset cls*Cy (the flip of that packing maps the forward slots
// {p, p+1} to {2-p, 1-p}, exactly the ones above).  p.H, p.W, p.Ho, p.Wo = stored (output) map; dY is (2H x 2W) with p.C0 channels.
template <typename T, bool SPLIT, bool BWD = false>
__global__ __launch_bounds__(256, 2) void conv_up4_halo_kernel(const IgemmP p) {
    static_assert(!SPLIT || sizeof(T) == 4, "the split-fp16 product is a mode of fp32 storage");
    static_assert(SPLIT || sizeof(T) == 2, "fp16 storage or split fp32");
    constexpr int NW = 4, WAVES_M = 2, WAVES_N = 2, BN = 128, BM = 128, MS = 16;
    constexpr int PH = 10, PW = 18, PR = PH * PW, PG = (PR + 7) / 8, LAH = (PG + NW - 1) / NW;  // 180 rows, 23 groups, 6 pieces per wave
    constexpr int ES = sizeof(T), EPC = 16 / ES, BKB = 128, BK = BKB / ES;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / MS, TN = WN / MS;
    constexpr int LB = (BN / 8) / NW;
    constexpr int ABUF = PG * 1024, BSTAGE = BN * BKB;
    static_assert(LAH == 6, "schedule: three patch pieces per K-step in steps 0 and 1");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;             // [2][ABUF]
    char* const sB = smem + 2 * ABUF;  // [2][BSTAGE]

    constexpr bool DIRECT = SPLIT && !BWD;  // swapped MFMA operands + stores straight from the accumulators (see halo_epilogue_direct)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int tile_m, tile_nn;
    map_tile(p, tile_m, tile_nn);                 // tile_nn enumerates (class, 128-column tile): the four classes of a tile run together
    const int tpc = BWD ? p.tiles_n : p.tiles_n >> 2;  // column tiles per class
    const int cls = BWD ? 0 : tile_nn / tpc, n0 = (tile_nn - cls * tpc) * BN;
    const int par_y = cls >> 1, par_x = cls & 1;        // (forward: the workgroup's class; backward: classes rotate inside the K loop)
    const TileGeom g = make_geom<true, BM>(p, tile_m);  // patch mode on the STORED map: ty0, tx0, n_first
    float pre_bcol[DIRECT ? 1 : TN];  // epilogue operands fetched at entry (see conv3x3_halo_kernel)
    f32x4 pre_bq[DIRECT ? TN : 1];
    int pre_we[DIRECT ? TN : 1];
    if constexpr (DIRECT) {
        prefetch_bcol4<TN>(p, p.Cout, n0 + wn * WN, lane, g.n_first, pre_bq);
        prefetch_wexp4<TN>(p, p.Cout, n0 + wn * WN, cls * p.Cout, lane, pre_we);  // (class kernels: weight row = class * Cout + column)
    } else {
        prefetch_bcol<TN, MS>(p, p.Cout, n0 + wn * WN, lane, g.n_first, pre_bcol);
    }
    float wsc1 = 1.0f;
    if constexpr (SPLIT) wsc1 = p.w_scale[1];
    AbScale asc = {EOD_SPLIT_ASCALE, 1.0f};             // split-fp16 product: operand scale of this tile's image (see conv3x3_halo_kernel)
    if constexpr (SPLIT) {
        if (p.a_bound) asc = ab_scale_of(ab_wave_bound(p.a_bound, g.n_first));
    }

    const int srow = lane >> 3, sslot = lane & 7;
    unsigned ppix[LAH];
    const int bchunk0 = sslot ^ ((4 * wave + (srow >> 1)) & 7);
    unsigned pck = 0, pvalid = 0;
#pragma unroll
    for (int i = 0; i < LAH; ++i) {
        const int prow = (wave + NW * i) * 8 + srow;
        const int py = prow / PW, px = prow - py * PW;
        const int hi = g.ty0 - 1 + py, wi = g.tx0 - 1 + px;
        const bool ok = (wave + NW * i) < PG && prow < PR && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        if (ok) pvalid |= 1u << i;
        ppix[i] = BWD ? (unsigned)(4 * hi * p.W + 2 * wi)   // pixel (2 hi, 2 wi) of the (2H x 2W) gradient; the class adds (p*2W + q)
                      : (unsigned)(hi * p.W + wi);
        pck |= (unsigned)(sslot ^ ((px >> 1) & 7)) << (3 * i);
    }
    auto pchunk_of = [&](int i) { return (int)((pck >> (3 * i)) & 7u); };
    unsigned b_v[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int row = (wave + NW * i) * 8 + srow;
        b_v[i] = (n0 + row < p.Cout) ? (unsigned)(row * (BWD ? 4 * p.C0 : p.Cin) * ES) + bchunk0 * 16 : EOD_OOB;
    }
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.a0 + (long long)g.n_first * p.H * p.W * p.C0 * ES * (BWD ? 4 : 1));
    // forward: [slot][4*Cout rows][Cin];  backward: [slot][Cout (= Cx) rows][4*C0 (= 4*Cy)], class block at K offset cls*C0
    const int bpitch = BWD ? 4 * p.C0 : p.Cin;
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.b + ((long long)cls * p.Cout + n0) * bpitch * ES);
    const int tapstride = (BWD ? p.Cout * bpitch : 4 * p.Cout * p.Cin) * ES;

    struct Chunk {
        int kin, cls;  // first channel of the chunk; backward: class of this (chunk, class) pair
        bool ktail;
    };
    auto chunk_of = [&](int cc) {
        Chunk c;
        c.cls = BWD ? (cc & 3) : cls;
        c.kin = (BWD ? (cc >> 2) : cc) * BK;
        c.ktail = c.kin + BK > p.C0;
        return c;
    };
    auto issue_patch_piece = [&](int i, const Chunk& c, char* abuf) {
        const int pc = pchunk_of(i);
        unsigned v = ((pvalid >> i) & 1u) ? ppix[i] * (unsigned)(p.C0 * ES) + pc * 16 : EOD_OOB;
        if (c.ktail) v = (c.kin + pc * EPC < p.C0) ? v : EOD_OOB;
        // backward: the class's pixel offset (p*2W + q) rides in the scalar offset
        const unsigned soff = (unsigned)(c.kin * ES) + (BWD ? (unsigned)(((c.cls >> 1) * 2 * p.W + (c.cls & 1)) * p.C0 * ES) : 0u);
        blds16(rsA, v, soff, abuf + (wave + NW * i) * 1024);
    };
    auto split_piece = [&](int i, char* abuf) {  // SPLIT: fp32 chunk -> its half of the pair's [8 x hi | 8 x lo] image (zeros stay zeros)
        char* ptr = abuf + (wave + NW * i) * 1024 + lane * 16;
        const f32x4 f = *reinterpret_cast<const f32x4*>(ptr);
        *reinterpret_cast<i32x4*>(ptr) = split_pair_exchange_scaled(f, asc.s, (pchunk_of(i) & 1) != 0);
    };
    // tap slot of K-step s (a = s >> 1, b = s & 1) of the 3x3 frame: forward (par_y + a, par_x + b); backward (1 - p + a, 1 - q + b)
    auto issue_weights = [&](int s, const Chunk& c, char* bst) {
        const int oy = BWD ? 1 - (c.cls >> 1) : par_y, ox = BWD ? 1 - (c.cls & 1) : par_x;
        const int tap = (oy + (s >> 1)) * 3 + ox + (s & 1);
        const unsigned soff = (unsigned)(tap * tapstride) + (unsigned)((c.kin + (BWD ? c.cls * p.C0 : 0)) * ES);
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            unsigned v = b_v[i];
            if (c.ktail) v = (c.kin + (SPLIT ? (bchunk0 >> 1) * 8 : bchunk0 * EPC) < p.C0) ? v : EOD_OOB;
            blds16(rsB, v, soff, bst + (wave + NW * i) * 1024);
        }
    };
    auto wait_vm = [](int n) {  // s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        }
    };

    typename AccLayout<MS>::vec acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < AccLayout<MS>::R; ++r) acc[i][j][r] = 0.0f;

    // fragment addresses (see conv3x3_halo_kernel): acur[v][b] for the two column taps b of the class, rows via compile-time offsets
    const int lr = lane & 15, lh = lane >> 4;
    const int c0 = SPLIT ? 2 * ((0x2130 >> (4 * lh)) & 3) : lh, c1 = SPLIT ? c0 + 1 : 4 + lh;
    int acur[2][2];
    auto set_acur = [&](int oy, int ox, int aoff) {  // first patch row / column offset of the class, patch buffer offset
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int pxc = lr + ox + b;
            const int key = (pxc >> 1) & 7;
            const int rowb = ((wm * (WM / 16) + oy) * PW + pxc) * BKB + aoff;
            acur[0][b] = rowb + ((c0 ^ key) << 4);
            acur[1][b] = rowb + ((c1 ^ key) << 4);
        }
    };
    set_acur(BWD ? 1 : par_y, BWD ? 1 : par_x, 0);  // (backward: class 0 = (0, 0) -> offsets (1, 1))
    const int bsw = (lr >> 1) & 7;
    const int b_rd = (wn * WN + lr) * BKB;
    const int boff0 = b_rd + ((c0 ^ bsw) << 4), boff1 = b_rd + ((c1 ^ bsw) << 4);

    const int KC = BWD ? 4 * p.kc0 : p.kc0, NSTEP = KC * 4;  // backward: (chunk, class) pairs
    {  // prologue: whole patch of chunk 0 + weights of step 0
        const Chunk ch0 = chunk_of(0);
#pragma unroll
        for (int i = 0; i < LAH; ++i)
            if ((wave + NW * i) < PG) issue_patch_piece(i, ch0, sA);
        issue_weights(0, ch0, sB);
        if constexpr (SPLIT) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < LAH; ++i)
                if ((wave + NW * i) < PG) split_piece(i, sA);
        }
    }
    int np_prev = 0;  // patch pieces this wave issued in the previous K-step (they sit behind that step's weight DMA in the vmcnt order)
    int step = 0;
    for (int cc = 0; cc < KC; ++cc) {
        const Chunk cur = chunk_of(cc);
        const bool has_next = cc + 1 < KC;
        const Chunk nxt = chunk_of(has_next ? cc + 1 : cc);
        char* abuf_next = sA + ((cc + 1) & 1) * ABUF;
#pragma unroll
        for (int s = 0; s < 4; ++s, ++step) {
            wait_vm(np_prev);  // this step's weights have landed (only the pieces issued after them may still be in flight)
            if constexpr (SPLIT) __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): my in-place splits are written
            __builtin_amdgcn_s_barrier();
            const bool w_next = step + 1 < NSTEP;
            if (w_next) issue_weights((s + 1) & 3, s == 3 ? nxt : cur, sB + ((step + 1) & 1) * BSTAGE);
            int np = 0;
            if (s < 2 && has_next) {  // three patch pieces of the next chunk in each of steps 0 and 1: every piece gets two full steps to land
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if ((wave + NW * (3 * s + k)) < PG) {
                        issue_patch_piece(3 * s + k, nxt, abuf_next);
                        ++np;
                    }
            }
            // ---- MFMAs of tap (a, b) ----
            const int a = s >> 1, b = s & 1;
            const char* bst = sB + (step & 1) * BSTAGE;
            if constexpr (SPLIT) {
                i32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) al[i] = *reinterpret_cast<const i32x4*>(sA + acur[1][b] + (i + a) * PW * BKB);
#pragma unroll
                for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const i32x4*>(bst + boff0 + j * MS * BKB);
#pragma unroll
                for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const i32x4*>(sA + acur[0][b] + (i + a) * PW * BKB);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[j]), __builtin_bit_cast(half8, al[i]), acc[i][j], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, al[i]), __builtin_bit_cast(half8, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j) bl[j] = *reinterpret_cast<const i32x4*>(bst + boff1 + j * MS * BKB);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bl[j]), __builtin_bit_cast(half8, ah[i]), acc[i][j], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah[i]), __builtin_bit_cast(half8, bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = DIRECT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, bh[j]), __builtin_bit_cast(half8, ah[i]), acc[i][j], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah[i]), __builtin_bit_cast(half8, bh[j]), acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int v = 0; v < 2; ++v) {  // fp16 storage: two 32-k sub-steps per 64-channel chunk
                    i32x4 fa16[TM], fb16[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa16[i] = *reinterpret_cast<const i32x4*>(sA + acur[v][b] + (i + a) * PW * BKB);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb16[j] = *reinterpret_cast<const i32x4*>(bst + (v ? boff1 : boff0) + j * MS * BKB);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fa16[i]), __builtin_bit_cast(half8, fb16[j]), acc[i][j], 0, 0, 0);
                }
            }
            if constexpr (SPLIT) {
                // the pieces issued TWO steps ago (3(s-2) ..): the wait at the top of this step (for this step's weights, which were issued
                // after them) has already proven that they landed
                if (s >= 2 && has_next) {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if ((wave + NW * (3 * (s - 2) + k)) < PG) split_piece(3 * (s - 2) + k, abuf_next);
                }
            }
            np_prev = np;
        }
        if constexpr (BWD) {  // next (chunk, class) pair: its class offsets, the other patch buffer
            const int ncl = (cc + 1) & 3;
            set_acur(1 - (ncl >> 1), 1 - (ncl & 1), ((cc + 1) & 1) * ABUF);
        } else {              // the next chunk reads the other patch buffer
            const int flip = (cc & 1) ? -ABUF : ABUF;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acur[0][b] += flip;
                acur[1][b] += flip;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();  // every wave is done with the operand buffers: reuse them for the epilogue
    // epilogue on the (2H x 2W) output: rows of the tile are positions (i, j) of the stored map, pixel (2i + par_y, 2j + par_x)
    IgemmP pe = p;
    pe.Ncols = p.Cout;
    TileGeom ge = g;
    if constexpr (!BWD) {
        pe.par = 1; pe.par_y = par_y; pe.par_x = par_x;
        pe.Wd = p.W;  // decode_row masks the tile columns behind an 8-wide STORED map (before the parity scaling)
        pe.tiles_per_image = p.tiles_pi * 4;  // statistics slots: (tile of the stored map, class)
        ge.tile_m = g.n_first * pe.tiles_per_image + (tile_m - g.n_first * p.tiles_pi) * 4 + cls;
    }
    if constexpr (SPLIT) pe.alpha = p.alpha * wsc1 * asc.inv;
    if constexpr (!BWD) pe.w_row0 = cls * p.Cout;
    if constexpr (DIRECT) halo_epilogue_direct<BM, BN, WAVES_M, WAVES_N>(pe, ge, acc, wave, lane, n0, pre_bq, pe.alpha, pre_we);
    else igemm_epilogue<T, true, BM, BN, WAVES_M, WAVES_N, sizeof(T) == 4, MS>(pe, ge, acc, smem, wave, lane, n0, nullptr, 0, pre_bcol);
}

template <typename T, bool SPLIT, bool BWD = false> static int launch_up4(IgemmP& p, hipStream_t st) {
    constexpr int BK = 128 / (int)sizeof(T);
    const size_t ring = 2 * (size_t)(23 * 1024) + 2 * (size_t)128 * 128;
    const size_t epi = 4 * (size_t)64 * (64 + 4) * sizeof(float);
    const size_t lds = ring > epi ? ring : epi;
    auto kern = conv_up4_halo_kernel<T, SPLIT, BWD>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    p.kc0 = (p.C0 + BK - 1) / BK;
    p.kc1 = 0;
    p.KT = p.kc0 * 4;
    p.tiles_n = (BWD ? 1 : 4) * ((p.Cout + 127) / 128);  // forward: (class, column tile)
    p.tw_log2 = 4;                            // 8 x 16 tiles of the STORED map
    p.th = 8;
    p.tiles_pw = (p.W + 15) / 16;
    p.tiles_pi = p.tiles_pw * (p.H / 8);
    p.tiles_m = p.tiles_pi * p.N;
    const long long nblk = (long long)p.tiles_m * p.tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        eod_set_error("conv_up4: bad grid %lld", nblk);
        return EOD_EINVAL;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, p);
    EOD_CHECK_LAUNCH("conv_up4_halo");
    return EOD_OK;
}

// =============================================================================================
// conv_head_kernel: the UNet's output head, GroupNorm -> SiLU -> 3x3 conv to <= 16 channels, NCHW fp32 output (unet_openai.py:738-743).
// 2 x 128 x 9 x Cout MACs per pixel against 512 bytes of input: HBM-bound, and in the halo kernel's frame (one barrier per tap for a
// 12-MFMA step) bound by synchronisation instead (0.41 ms for 537 MB of input).  Here a tile's tap loop has NO barrier: the weights
// of a whole chunk (9 taps x 16 output rows) live in REGISTERS, loaded from L2 one chunk ahead; the patch ring is three deep (chunk cc
// is read, cc+1 is normalised / split by the waves that fetched it, cc+2 is in flight), so there is ONE barrier per chunk.
// Workgroup = 8 x 16 pixel tile, 4 waves x 32 pixels; MFMA 16x16x32 with the WEIGHTS as the A operand: D[row = channel][col = pixel],
// so a lane of the first quarter holds the <= 4 output channels of one pixel and a plane row of 16 pixels is one 64-byte store.
// =============================================================================================
#define HEAD_MAX_C 384  // input channels of the head (the whole scale / shift table sits in LDS)
template <typename T, bool SPLIT>
__global__ __launch_bounds__(256, 2) void conv_head_kernel(const IgemmP p) {
    static_assert(SPLIT ? sizeof(T) == 4 : sizeof(T) == 2, "fp16 storage or split fp32");
    constexpr int NW = 4, BM = 128, MS = 16;
    constexpr int PH = 10, PW = 18, PR = PH * PW, LAH = ((PR + 7) / 8 + NW - 1) / NW, PG = LAH * NW;  // 180 rows in 24 groups of 8 (the last one
    // is padding: every wave issues exactly LAH = 6 pieces per chunk, which keeps the counted vmcnt waits exact for the compiler too)
    constexpr int ES = sizeof(T), EPC = 16 / ES, BKB = 128, BK = BKB / ES;
    constexpr int ABUF = PG * 1024;
    constexpr int NSUB = SPLIT ? 1 : 2;  // 32-k MFMA sub-steps per chunk row (fp16: 64 channels)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;              // [3][ABUF]
    char* const sS = smem + 3 * ABUF;   // [HEAD_MAX_C][2] scale / shift of every input channel of this image

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a workgroup streams a RUN of p.tpw consecutive tiles of one image through one chunk ring (launch_head: tpw divides the tiles of an
    // image): the ring never drains between tiles -- the chunk two ahead is in flight and the one ahead is being normalised whichever
    // tile they belong to -- so the first-fetch latency and the table load are paid once per run, not once per 4 chunks
    int run, tile_n;
    map_tile(p, run, tile_n);  // (p.tiles_m = number of runs)
    const int tile0 = run * p.tpw, ntile = p.tpw;
    const TileGeom g = make_geom<true, BM>(p, tile0);  // n_first: the image of the whole run
    AbScale asc = {EOD_SPLIT_ASCALE, 1.0f};  // split-fp16 product: operand scale of this tile's image (see conv3x3_halo_kernel)
    const float silu_k = p.gn_silu ? -1.44269504088896341f : 0.0f, silu_c = p.gn_silu ? 0.0f : -__builtin_inff();
    if constexpr (SPLIT) {
        if (p.a_bound) asc = ab_scale_of(ab_wave_bound(p.a_bound, g.n_first));
    }

    const int srow = lane >> 3, sslot = lane & 7;
    // patch pieces of this lane: the chunk slot of a piece is a property of the patch position (the same for every tile), source pixel
    // and validity belong to a tile: one set for the tile whose chunks are being ISSUED, one for the tile being TRANSFORMED
    struct PatchGeom {
        unsigned ppix[LAH];
        unsigned pvalid;
    };
    unsigned pck = 0;
#pragma unroll
    for (int i = 0; i < LAH; ++i) {
        const int prow = (wave + NW * i) * 8 + srow;
        const int px = prow % PW;
        pck |= (unsigned)(sslot ^ ((px >> 1) & 7)) << (3 * i);
    }
    auto patch_geom = [&](int k, PatchGeom& o) {  // tile k of the run (k >= ntile: past the end, nothing valid)
        const TileGeom gk = make_geom<true, BM>(p, tile0 + (k < ntile ? k : 0));
        o.pvalid = 0;
#pragma unroll
        for (int i = 0; i < LAH; ++i) {
            const int prow = (wave + NW * i) * 8 + srow;
            const int py = prow / PW, px = prow - py * PW;
            const int hi = gk.ty0 - 1 + py, wi = gk.tx0 - 1 + px;
            const bool ok = k < ntile && prow < PR && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            if (ok) o.pvalid |= 1u << i;
            o.ppix[i] = (unsigned)(hi * p.W + wi);
        }
    };
    PatchGeom gI, gT;
    patch_geom(0, gI);
    gT = gI;
    auto pchunk_of = [&](int i) { return (int)((pck >> (3 * i)) & 7u); };
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.a0 + (long long)g.n_first * p.H * p.W * p.C0 * ES);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gn_ss) + (long long)g.n_first * p.C0 * 2, 0, p.C0 * 8, 0x00020000);
    const int KC = p.kc0;

    // the stream of chunks: q = tile * KC + chunk, ring slot q % 3.  issue / transform / weights each walk it once, in order: three
    // (tile, chunk, slot) counters instead of divisions
    struct Cursor {
        int k, cc, slot;
    };
    auto advance = [&](Cursor& c) {
        c.slot = c.slot == 2 ? 0 : c.slot + 1;
        if (++c.cc == KC) {
            c.cc = 0;
            ++c.k;
        }
    };
    Cursor cI = {0, 0, 0}, cT = {0, 0, 0};
    auto issue_next = [&]() {  // this wave's patch pieces of the next chunk of the stream into its ring slot (past the end: out of range)
        char* abuf = sA + cI.slot * ABUF;
        const int kin = cI.cc * BK;
        const bool ktail = kin + BK > p.C0;
#pragma unroll
        for (int i = 0; i < LAH; ++i) {
            const int pc = pchunk_of(i);
            unsigned v = ((gI.pvalid >> i) & 1u) ? gI.ppix[i] * (unsigned)(p.C0 * ES) + pc * 16 : EOD_OOB;
            if (ktail) v = (kin + pc * EPC < p.C0) ? v : EOD_OOB;
            blds16(rsA, v, (unsigned)(kin * ES), abuf + (wave + NW * i) * 1024);
        }
        advance(cI);
        if (cI.cc == 0) patch_geom(cI.k, gI);  // the next tile's pixels
    };
    auto transform_next = [&]() {  // x -> silu(x * scale + shift) [-> fp16 pair image], this wave's pieces of the next chunk, in place
        char* abuf = sA + cT.slot * ABUF;
        const int kin = cT.cc * BK;
        const char* ssbuf = sS + kin * 8;
        const bool ktail = kin + BK > p.C0;
        const unsigned pvalid = gT.pvalid;
#pragma unroll
        for (int i = 0; i < LAH; ++i) {
            if ((wave + NW * i) * 8 >= PR) continue;  // the padding group
            const int pc = pchunk_of(i);
            const bool ok = ((pvalid >> i) & 1u) && (!ktail || (kin + pc * EPC < p.C0));
            char* ptr = abuf + (wave + NW * i) * 1024 + lane * 16;
            const i32x4 raw = *reinterpret_cast<const i32x4*>(ptr);
            const float* sp = reinterpret_cast<const float*>(ssbuf) + pc * EPC * 2;
            if constexpr (SPLIT) {
                const f32x4 f = __builtin_bit_cast(f32x4, raw);
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(sp), q1 = *reinterpret_cast<const f32x4*>(sp + 4);
                const float sc[4] = {q0[0], q0[2], q1[0], q1[2]}, sh[4] = {q0[1], q0[3], q1[1], q1[3]};
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = f[e] * sc[e] + sh[e];
                    o[e] = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(v, silu_k, silu_c)));  // (select-free SiLU: see conv3x3_halo_kernel)
                }
                // (conv zero padding / masked channel tail stay zero: scale 0)
                *reinterpret_cast<i32x4*>(ptr) = split_pair_exchange_scaled(o, ok ? asc.s : 0.0f, (pc & 1) != 0);
            } else {
                const half8 h = __builtin_bit_cast(half8, raw);
                half8 o;
                f32x4 q[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = *reinterpret_cast<const f32x4*>(sp + 4 * k);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = (float)h[e] * q[e >> 1][(e & 1) * 2] + q[e >> 1][(e & 1) * 2 + 1];
                    v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(v, silu_k, silu_c)));  // (select-free SiLU)
                    o[e] = (half_t)v;
                }
                if (ok) *reinterpret_cast<i32x4*>(ptr) = __builtin_bit_cast(i32x4, o);
            }
        }
        advance(cT);
        if (cT.cc == 0) patch_geom(cT.k, gT);
    };
    // weights of one chunk in registers: [tap][sub-step / (hi, lo)]; lane = (output row lr, k-quarter lh).  Packed [tap][Cout][Cin]
    // rows; split storage: 32-byte [8 x hi | 8 x lo] pairs, quarter lh takes pair {0, 3, 1, 2}[lh] like the patch reads.
    const int lr = lane & 15, lh = lane >> 4;
    const int c0 = SPLIT ? 2 * ((0x2130 >> (4 * lh)) & 3) : lh, c1 = SPLIT ? c0 + 1 : 4 + lh;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.b), 0, 9 * p.Cout * p.Cin * ES, 0x00020000);
    const unsigned wrow = lr < p.Cout ? (unsigned)(lr * p.Cin * ES) : EOD_OOB;  // rows past Cout: zeros
    const int tapbytes = p.Cout * p.Cin * ES;
    auto load_weights = [&](int cc, i32x4 (&w)[9][2]) {
        const int kin = cc * BK;
        const unsigned v0 = (kin + c0 * EPC < p.C0) ? wrow + c0 * 16 : EOD_OOB;  // (a split pair lies wholly inside or outside: C0 % 8 == 0)
        const unsigned v1 = (kin + c1 * EPC < p.C0) ? wrow + c1 * 16 : EOD_OOB;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            w[t][0] = __builtin_amdgcn_raw_buffer_load_b128(rsW, v0, (unsigned)(t * tapbytes + kin * ES), 0);
            w[t][1] = __builtin_amdgcn_raw_buffer_load_b128(rsW, v1, (unsigned)(t * tapbytes + kin * ES), 0);
        }
    };

    // fragment addresses of the patch (see conv3x3_halo_kernel): this wave's tile rows 2*wave + i, column lr + dx
    int abase[2][3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int pxc = lr + dx;
        const int key = (pxc >> 1) & 7;
        const int rowb = ((wave * 2) * PW + pxc) * BKB;
        abase[0][dx] = rowb + ((c0 ^ key) << 4);
        abase[1][dx] = rowb + ((c1 ^ key) << 4);
    }
    f32x4 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.0f;

    // VMEM per iteration: the LAH pieces of chunk cc + 2 and the 18 weight loads of chunk cc + 1 -- exactly that in every iteration
    // (past the last chunk all of it out of range: zeros into a ring slot nobody reads), in whichever order the scheduler puts them
    // (it sinks the weight loads below the tap loop and reloads each register after its last use: no second register set).  The
    // rewrite of chunk cc + 1 needs ITS pieces, issued one iteration earlier: vmcnt(LAH + 18) keeps this iteration's issues in flight.
    // Issues and waits are unconditional and the wait is the builtin: the compiler's own wait insertion is path-insensitive, and a
    // conditional issue or an inline-asm wait makes it guard every MFMA of the tap loop with a vmcnt of its own (seen in the ISA).
    constexpr int KEEP = LAH + 18;
    static_assert(KEEP < 64, "vmcnt is 6 bits");
    auto wait_keep_pieces = [&]() { __builtin_amdgcn_s_waitcnt(0x0f70 | (KEEP & 15) | ((KEEP >> 4) << 14)); };
    i32x4 wreg[2][9][2];
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < HEAD_MAX_C / 128; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (lds_void*)(sS + j * 1024), 16, (unsigned)(lane * 16 + j * 1024), 0, 0, 0);
    }
    // per output channel: {scale (operand scales x the row scale of the split weights), bias}, through LDS -- fetched by 16 lanes before
    // any DMA is in flight and read back into registers behind the first barrier, so that no global load of the epilogue is pending
    // inside the loop (the compiler would guard its first use with a vmcnt(0) in every iteration)
    float* const sE = reinterpret_cast<float*>(sS + HEAD_MAX_C * 8);
    if (tid < 16) {
        const float alpha = SPLIT ? p.alpha * p.w_scale[1] * asc.inv : p.alpha;
        const bool cok = tid < p.Cout;
        sE[2 * tid] = (SPLIT && p.w_rexp && cok) ? ldexpf(alpha, -p.w_rexp[tid]) : alpha;
        sE[2 * tid + 1] = (p.bias && cok) ? p.bias[tid] : 0.0f;
    }
    issue_next();
    load_weights(0, wreg[0]);
    issue_next();
    wait_keep_pieces();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // (the 16 lanes' LDS writes)
    __builtin_amdgcn_s_barrier();  // wave 0's tables are visible
    float am[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        am[r] = sE[2 * (4 * lh + r)];
        bv[r] = sE[2 * (4 * lh + r) + 1];
    }
    transform_next();
    const int Q = ntile * KC;
    Cursor cR = {0, 0, 0};  // the chunk being read
    int q = 0;
    // one chunk of the stream: reads `wc`, fetches the next chunk's weights into `wn` (the loop alternates the two register sets:
    // rotating one set through the other cost 36 moves per chunk)
    auto step = [&](auto cur_c) {
        constexpr int CUR = decltype(cur_c)::value;
        i32x4 (&wc)[9][2] = wreg[CUR];
        i32x4 (&wn)[9][2] = wreg[CUR ^ 1];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): my in-place rewrites of chunk q are in LDS
        __builtin_amdgcn_s_barrier();        // chunk q is complete for every wave, and ring slot (q + 2) % 3 is no longer read
        load_weights(cR.cc + 1 == KC ? 0 : cR.cc + 1, wn);
        issue_next();
        const char* abuf = sA + cR.slot * ABUF;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const i32x4 a0 = *reinterpret_cast<const i32x4*>(abuf + abase[0][dx] + (i + dy) * PW * BKB);
                const i32x4 a1 = *reinterpret_cast<const i32x4*>(abuf + abase[1][dx] + (i + dy) * PW * BKB);
                if constexpr (SPLIT) {  // a0 = hi, a1 = lo of the pixels; wc[t][0] = hi, [1] = lo of the weights; smallest terms first
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wc[t][0]), __builtin_bit_cast(half8, a1), acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wc[t][1]), __builtin_bit_cast(half8, a0), acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wc[t][0]), __builtin_bit_cast(half8, a0), acc[i], 0, 0, 0);
                } else {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wc[t][0]), __builtin_bit_cast(half8, a0), acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wc[t][1]), __builtin_bit_cast(half8, a1), acc[i], 0, 0, 0);
                }
            }
        }
        wait_keep_pieces();
        if (q + 1 < Q) transform_next();
        if (cR.cc + 1 == KC) {
            // ---- a tile is complete: rows of D = output channels 4 lh + r, column = pixel lr of tile row 2 wave + i: 64-byte runs per plane
            // row.  (The stores are younger than every piece a later counted wait is for: they can only make it wait longer.)
            const TileGeom gk = make_geom<true, BM>(p, tile0 + cR.k);
            float* yb = reinterpret_cast<float*>(p.y) + (long long)g.n_first * p.Cout * p.HoWo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * lh + r;
                if (co < p.Cout) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        yb[(long long)co * p.HoWo + (long long)(gk.ty0 + wave * 2 + i) * p.Wo + gk.tx0 + lr] = acc[i][r] * am[r] + bv[r];
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] = 0.0f;
        }
        advance(cR);
        ++q;
    };
    while (q < Q) {
        step(std::integral_constant<int, 0>{});
        if constexpr (SPLIT) {
            if (q < Q) step(std::integral_constant<int, 1>{});
        } else {  // (the fp16 instance keeps the rotation: with two copies of the loop body its register arrays end up in scratch)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                wreg[0][t][0] = wreg[1][t][0];
                wreg[0][t][1] = wreg[1][t][1];
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // the out-of-range pieces of the last two iterations: nothing in flight at exit
}

// =============================================================================================
// conv_first_x3_kernel: the UNet's FIRST conv in the fp32x3 mode -- 3x3 / stride 1 / pad 1 over an image of <= 8 (padded) channels
// (unet_openai.py:609, input_blocks.0.0; tap-major weights [Cout][ldk], k = tap * CP + c, eod_pack_conv_weight_tapmajor_split).
// 9 x CP x Cout MACs per pixel against 4 x Cout bytes of output: HBM-bound on the WRITE side.  On the generic kernel a workgroup lived
// ~17 us for two K-steps (gathered LDS-DMA of 16 + 16 pieces per step, the in-LDS split, two barriers per step, 64 KiB of LDS: two
// workgroups per CU and nothing to hide the first fetch behind): 0.27 ms for 537 MB = 2.0 TB/s.  Here the 10 x 18 pixel patch of an
// 8 x 16 tile is ONE 16- or 32-byte load per thread, split on the way into 3-6 KiB of LDS ([CP x hi | CP x lo] per pixel), the
// weights go from L2 straight into registers, the k-groups of a pixel are plain LDS reads at tap offsets (k-group = two taps of 4
// channels, or one tap of 8), ONE barrier per workgroup, and three workgroups fit a CU (160 registers).  Swapped MFMA operands and the direct
// epilogue of the halo kernel (halo_epilogue_direct: 16-byte stores from the accumulators, bias, GroupNorm partial sums in the same
// two slots per tile).
// =============================================================================================
template <int CP>
__global__ __launch_bounds__(256, 3) void conv_first_x3_kernel(const IgemmP p) {
    static_assert(CP == 4 || CP == 8, "one or two 16-byte chunks of fp32 channels per pixel");
    constexpr int BM = 128, BN = 128, TM = 4, TN = 4;
    constexpr int PW = 18, PR = 10 * PW, PB = CP * 4;        // LDS bytes per patch pixel: [CP x fp16 hi | CP x fp16 lo]
    constexpr int KG = (9 * CP + 7) / 8, STEPS = (KG + 3) / 4;  // 8-element k-groups that hold taps; MFMA steps of 4 groups
    constexpr int ZOFF = PR * PB;                            // 32 zero bytes behind the patch: what the k-groups past tap 8 read
    __shared__ __attribute__((aligned(16))) char sP[PR * PB + 32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n;
    map_tile(p, tile_m, tile_n);
    const int n0 = tile_n * BN;
    const TileGeom g = make_geom<true, BM>(p, tile_m);
    f32x4 pre_bq[TN];
    int pre_we[TN];
    prefetch_bcol4<TN>(p, p.Ncols, n0 + wn * 64, lane, g.n_first, pre_bq);
    prefetch_wexp4<TN>(p, p.Ncols, n0 + wn * 64, 0, lane, pre_we);
    const float wsc1 = p.w_scale[1];
    AbScale asc = {EOD_SPLIT_ASCALE, 1.0f};
    if (p.a_bound) asc = ab_scale_of(ab_wave_bound(p.a_bound, g.n_first));

    // ---- the patch: one pixel per thread, scaled and split on the way into LDS (zeros outside the image = the conv's padding) ----
    if (tid < PR) {
        const int py = tid / PW, px = tid - py * PW;
        const int hi = g.ty0 - 1 + py, wi = g.tx0 - 1 + px;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) {
            const float* src = reinterpret_cast<const float*>(p.a0) + (((long long)g.n_first * p.H + hi) * p.W + wi) * CP;
            a = *reinterpret_cast<const f32x4*>(src);
            if constexpr (CP == 8) c = *reinterpret_cast<const f32x4*>(src + 4);
        }
        int ha[2], la[2];
        split4_scaled(a, asc.s, ha, la);
        if constexpr (CP == 4) {
            *reinterpret_cast<i32x4*>(sP + tid * PB) = i32x4{ha[0], ha[1], la[0], la[1]};
        } else {
            int hc[2], lc[2];
            split4_scaled(c, asc.s, hc, lc);
            *reinterpret_cast<i32x4*>(sP + tid * PB) = i32x4{ha[0], ha[1], hc[0], hc[1]};
            *reinterpret_cast<i32x4*>(sP + tid * PB + 16) = i32x4{la[0], la[1], lc[0], lc[1]};
        }
    } else if (tid < PR + 2) {
        *reinterpret_cast<i32x4*>(sP + ZOFF + (tid - PR) * 16) = i32x4{0, 0, 0, 0};
    }
    __syncthreads();

    const int lp = lane & 15, kgl = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.b), 0, p.Cout * p.ldk * 4, 0x00020000);
    typename AccLayout<16>::vec acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // patch address of (tile row 4 wm, column lp) shifted by a tap
    auto tap_addr = [&](int tap) {
        const int dy = (tap * 11) >> 5, dx = tap - dy * 3;  // tap / 3, tap % 3 for tap < 9
        return ((wm * 4 + dy) * PW + lp + dx) * PB;
    };
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int gq = 4 * s + kgl;  // this lane's k-group: 8 consecutive k = tap * CP + c
        i32x4 wh[TN], wl[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n0 + wn * 64 + j * 16 + lp;
            const unsigned off = (co < p.Ncols && gq * 8 < p.ldk) ? (unsigned)(co * p.ldk * 4 + gq * 32) : EOD_OOB;
            wh[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0);
            wl[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 16, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            i32x4 xh, xl;
            if constexpr (CP == 8) {
                const int a = gq < 9 ? tap_addr(gq) + i * PW * PB : ZOFF;
                xh = *reinterpret_cast<const i32x4*>(sP + a);
                xl = *reinterpret_cast<const i32x4*>(sP + a + 16);
            } else {
                const int a0 = 2 * gq < 9 ? tap_addr(2 * gq) + i * PW * PB : ZOFF;
                const int a1 = 2 * gq + 1 < 9 ? tap_addr(2 * gq + 1) + i * PW * PB : ZOFF;
                const i32x4 p0 = *reinterpret_cast<const i32x4*>(sP + a0), p1 = *reinterpret_cast<const i32x4*>(sP + a1);
                xh = i32x4{p0[0], p0[1], p1[0], p1[1]};
                xl = i32x4{p0[2], p0[3], p1[2], p1[3]};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {  // D = W X^T, smallest terms first (see conv3x3_halo_kernel)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wh[j]), __builtin_bit_cast(half8, xl), acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wl[j]), __builtin_bit_cast(half8, xh), acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wh[j]), __builtin_bit_cast(half8, xh), acc[i][j], 0, 0, 0);
            }
        }
    }
    IgemmP pe = p;
    pe.alpha = p.alpha * wsc1 * asc.inv;  // undo the weight and activation scales (exact powers of two)
    halo_epilogue_direct<BM, BN, 2, 2>(pe, g, acc, wave, lane, n0, pre_bq, pe.alpha, pre_we);
}

template <int CP> static int launch_first(IgemmP& p, hipStream_t st) {
    p.tiles_n = (p.Ncols + 127) / 128;
    p.tw_log2 = 4;
    p.th = 8;
    p.tiles_pw = p.Wo / 16;
    p.tiles_pi = p.tiles_pw * (p.Ho / 8);
    p.tiles_m = p.tiles_pi * p.N;
    p.tiles_per_image = p.tiles_pi;
    const long long nblk = (long long)p.tiles_m * p.tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        eod_set_error("conv_first: bad grid %lld", nblk);
        return EOD_EINVAL;
    }
    hipLaunchKernelGGL(conv_first_x3_kernel<CP>, dim3((unsigned)nblk), dim3(256), 0, st, p);
    EOD_CHECK_LAUNCH("conv_first");
    return EOD_OK;
}

template <typename T, bool SPLIT> static int launch_head(IgemmP& p, hipStream_t st) {
    constexpr int BK = 128 / (int)sizeof(T);
    const size_t lds = 3 * (size_t)(24 * 1024) + HEAD_MAX_C * 8 + 128;
    auto kern = conv_head_kernel<T, SPLIT>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    p.kc0 = (p.C0 + BK - 1) / BK;
    p.kc1 = 0;
    p.tiles_n = 1;
    p.tw_log2 = 4;
    p.th = 8;
    p.tiles_pw = p.Wo / 16;
    p.tiles_pi = p.tiles_pw * (p.Ho / 8);
    // runs of tpw consecutive tiles of one image per workgroup (the kernel's chunk stream): the longest run that still leaves every CU its
    // two workgroups.  A tile's result does not depend on the run it is computed in, so the choice may depend on the batch.
    int tpw = p.tpw;  // (the caller's request: the head_tpw option; 0 = choose)
    if (tpw <= 0) {
        tpw = 16;
        while (tpw > 1 && (p.tiles_pi % tpw || (long long)p.tiles_pi * p.N / tpw < 512)) tpw >>= 1;
    }
    while (tpw > 1 && p.tiles_pi % tpw) --tpw;
    p.tpw = tpw;
    p.tiles_m = p.tiles_pi / tpw * p.N;  // (runs)
    if (p.tiles_m <= 0) {
        eod_set_error("conv_head: bad grid");
        return EOD_EINVAL;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)p.tiles_m), dim3(256), lds, st, p);
    EOD_CHECK_LAUNCH("conv_head");
    return EOD_OK;
}

// split-K second pass: y[m][c] = alpha * sum_z ws[z][m][c] + bias[c] + cbias[n(m)][c] + res[m][c]   (fixed z order)
template <typename T>
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int S, long long M, int Cout, int HoWo, float alpha,
                                     const float* __restrict__ bias, const float* __restrict__ cbias, long long cbias_stride,
                                     const T* __restrict__ res, T* __restrict__ y, const float* __restrict__ w_scale) {
    if (w_scale) alpha *= w_scale[1];  // split-fp16 product: undo the operand scales
    const long long total4 = M * Cout / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long e = i * 4, m = e / Cout;
        const int c = (int)(e - m * Cout);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < S; ++z) a += *reinterpret_cast<const f32x4*>(ws + (long long)z * M * Cout + e);
        const long long n = m / HoWo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = a[k] * alpha;
            if (bias) v += bias[c + k];
            if (cbias) v += cbias[n * cbias_stride + c + k];
            if (res) v += (float)res[e + k];
            y[e + k] = (T)v;
        }
    }
}

// the same second pass with the next GroupNorm's per-channel partial sums of the STORED values on the way (one slot per image):
// grid (ceil(Cout / 64), N), block 256 = 16 chunk columns (4 channels each) x 16 pixel lanes; the S partial tiles of a pixel are
// fetched together (independent loads), summed in the fixed z order
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* __restrict__ ws, int S, long long M, int Cout, int HoWo, float alpha,
                                                                  const float* __restrict__ bias, const float* __restrict__ cbias,
                                                                  long long cbias_stride, const T* __restrict__ res, T* __restrict__ y,
                                                                  float* __restrict__ stats) {
    __shared__ float red[16][16][8];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, n = blockIdx.y;
    const int c = (blockIdx.x * 16 + tx) * 4;
    float ss[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < Cout) {
        float b4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) b4[k] = (bias ? bias[c + k] : 0.0f) + (cbias ? cbias[(long long)n * cbias_stride + c + k] : 0.0f);
        const long long zs = M * Cout;
        for (int pix = ty; pix < HoWo; pix += 16) {
            const long long e = ((long long)n * HoWo + pix) * Cout + c;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            int z = 0;
            for (; z + 4 <= S; z += 4) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(ws + (long long)z * zs + e), v1 = *reinterpret_cast<const f32x4*>(ws + (long long)(z + 1) * zs + e);
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(ws + (long long)(z + 2) * zs + e), v3 = *reinterpret_cast<const f32x4*>(ws + (long long)(z + 3) * zs + e);
                a += v0; a += v1; a += v2; a += v3;
            }
            for (; z < S; ++z) a += *reinterpret_cast<const f32x4*>(ws + (long long)z * zs + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = a[k] * alpha + b4[k];
                if (res) v += (float)res[e + k];
                const T o = (T)v;
                y[e + k] = o;
                const float x = (float)o;
                ss[k] += x;
                sq[k] += x * x;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[ty][tx][2 * k] = ss[k];
        red[ty][tx][2 * k + 1] = sq[k];
    }
    __syncthreads();
    if (ty == 0 && c < Cout) {
        float* dst = stats + ((long long)n * Cout + c) * 2;  // [N][1 slot][Cout][2]
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += red[r][tx][k];
            dst[k] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
#include <stdlib.h>
#include <string.h>
// Kernel-selection options (each has ONE measured-best default; the losing arm stays reachable for same-box A/B runs and for the
// per-switch parity tests, through the environment at first use or eod_set_option at any time):
//   skip_fuse         EOD_SKIP_FUSE=0          ResBlock 1x1 skip convs as launches of their own instead of inside out_layers' conv
//   head              EOD_HEAD=0               the output head on the 32-column halo instance instead of conv_head_kernel
//   halo_bn256        EOD_HALO_BN256=0         256- / 512-column convs as two 4-wave workgroups per pixel tile instead of one 8-wave one
//   gn_fuse_max_cout  EOD_GN_FUSE_MAX_COUT=n   widest conv that takes its input GroupNorm in its patch staging (-1: the defaults)
//   halo_tpw          EOD_HALO_TPW=n           pixel tiles per workgroup of the streaming halo instances (1: off = the default; 0: chosen
//                                              per launch).  Measured (round 4): single layers 3-8 % faster at 4-8 tiles per run, the
//                                              whole 256 x 256 step 0.4 % SLOWER than off (same-box A/B, three interleaved runs)
//   halo_splitk       EOD_HALO_SPLITK=0        3x3 convs on maps with fewer than two workgroups per CU unsplit in K (64-column tiles instead)
//   first             EOD_FIRST=0              the fp32x3 first conv (tap-major weights) on the generic kernel instead of conv_first_x3_kernel
//   head_tpw          EOD_HEAD_TPW=n           pixel tiles per workgroup of conv_head_kernel's chunk stream (0: chosen per launch = the default)
// (Round 2's EOD_IGEMM_CFG / EOD_MFMA_SHAPE / EOD_HALO_SPLIT_N / EOD_CONV_PARITY arms were measured slower and are gone: the fp16
// products run on v_mfma_f32_16x16x32_f16, 384-column convs as 256 + 128, zero-insertion convs as four parity-class launches.)
enum { OPT_SKIP_FUSE, OPT_HEAD, OPT_HALO_BN256, OPT_GN_FUSE_MAX_COUT, OPT_HALO_TPW, OPT_HALO_SPLITK, OPT_FIRST, OPT_HEAD_TPW, OPT_COUNT };
static const char* const g_opt_name[OPT_COUNT] = {"skip_fuse", "head", "halo_bn256", "gn_fuse_max_cout", "halo_tpw", "halo_splitk", "first", "head_tpw"};
static const char* const g_opt_env[OPT_COUNT] = {"EOD_SKIP_FUSE", "EOD_HEAD", "EOD_HALO_BN256", "EOD_GN_FUSE_MAX_COUT", "EOD_HALO_TPW", "EOD_HALO_SPLITK",
                                                 "EOD_FIRST", "EOD_HEAD_TPW"};
static int g_opt[OPT_COUNT] = {1, 1, 1, -1, 1, 1, 1, 0};
static bool g_opt_init = false;
static int opt(int k) {
    if (!g_opt_init) {
        for (int i = 0; i < OPT_COUNT; ++i) {
            const char* e = getenv(g_opt_env[i]);
            if (e) g_opt[i] = atoi(e);
        }
        g_opt_init = true;
    }
    return g_opt[k];
}
extern "C" int eod_set_option(const char* name, int value) {
    (void)opt(0);
    for (int i = 0; name && i < OPT_COUNT; ++i)
        if (!strcmp(name, g_opt_name[i])) {
            const int prev = g_opt[i];
            g_opt[i] = value;
            return prev;
        }
    eod_set_error("set_option: unknown option '%s'", name ? name : "(null)");
    return EOD_EINVAL;
}
extern "C" int eod_get_option(const char* name) {
    (void)opt(0);
    for (int i = 0; name && i < OPT_COUNT; ++i)
        if (!strcmp(name, g_opt_name[i])) return g_opt[i];
    eod_set_error("get_option: unknown option '%s'", name ? name : "(null)");
    return EOD_EINVAL;
}

template <typename T, bool CONV, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool SPLIT = false, int MS = 32, bool DIRECT = false>
static int launch_cfg(IgemmP& p, int batch, hipStream_t st) {
    constexpr int BK = 128 / (int)sizeof(T);
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, NW = WAVES_M * WAVES_N;
    const size_t ring = STAGES * (size_t)(BM + BN) * 128;
    const size_t epi = NW * (size_t)WM * (WN + 4) * sizeof(float);
    size_t lds = ring > epi ? ring : epi;
    if (SPLIT && CONV) {  // per-image operand scales of a tile that straddles images: at most BM + 1 images of {s, 16/s}
        lds = (lds + 15) & ~(size_t)15;
        p.ab_tab_off = (int)lds;
        lds += (BM + 2) * 2 * sizeof(float);
    }
    auto kern = igemm_kernel<T, CONV, BM, BN, WAVES_M, WAVES_N, STAGES, SPLIT, MS, DIRECT>;
    static bool attr_done = false;  // >64 KiB dynamic LDS needs the opt-in attribute once per kernel
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    if (CONV) {
        p.kc0 = (p.C0 + BK - 1) / BK;
        p.kc1 = (p.C1 + BK - 1) / BK;
        p.KT = (p.kc0 + p.kc1) * p.taps;
        if (p.tapmajor_log2 >= 0) p.KT = (9 * p.C0 + BK - 1) / BK;
    } else {
        p.KT = (p.K + BK - 1) / BK;
    }
    p.tiles_n = (p.Ncols + BN - 1) / BN;
    p.tw_log2 = -1;
    if (CONV) {
        // patch mode: TW = min(16, Wo) if it is a power of two dividing Wo and TH = BM/TW divides Ho
        int tw = 16;
        while (tw > p.Wd) tw >>= 1;
        if (tw >= 4 && p.Wd % tw == 0 && (BM % tw) == 0 && p.Hd % (BM / tw) == 0) {
            int l = 0;
            while ((1 << l) < tw) ++l;
            p.tw_log2 = l;
            p.th = BM / tw;
            p.tiles_pw = p.Wd / tw;
            p.tiles_pi = p.tiles_pw * (p.Hd / p.th);
            p.tiles_m = p.tiles_pi * p.N;
        } else {
            p.tiles_m = (int)((p.M + BM - 1) / BM);
        }
    } else {
        p.tiles_m = (int)((p.M + BM - 1) / BM);
    }
    const long long nblk = (long long)p.tiles_m * p.tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        eod_set_error("igemm: bad grid %lld", nblk);
        return EOD_EINVAL;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)batch), dim3(64 * NW), lds, st, p);
    EOD_CHECK_LAUNCH("igemm");
    return EOD_OK;
}

template <typename T, int BN, int WAVES_M, int WAVES_N, bool UPS, int BSTAGES, bool GN, bool SPLIT = false, int MS = 32, bool SKIP = false, bool STREAM = false,
          bool KSPLIT = false>
static int launch_halo(IgemmP& p, hipStream_t st) {
    if constexpr (!KSPLIT && !STREAM && !UPS && BN == 128 && WAVES_N == 2) {
        if (p.splitk > 1) return launch_halo<T, BN, WAVES_M, WAVES_N, UPS, BSTAGES, GN, SPLIT, MS, SKIP, false, true>(p, st);  // the K-slice instance
    }
    if constexpr (!KSPLIT) {
        if (p.splitk > 1) {
            eod_set_error("conv_halo: K slices exist for the 128-column 4-wave instances only");
            return EOD_EINVAL;
        }
    }
    constexpr int BK = 128 / (int)sizeof(T);
    constexpr int NW = WAVES_M * WAVES_N, BM = WAVES_N == 4 ? 64 * WAVES_M : 32 * NW, TH = BM / 16;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int PR = UPS ? (TH / 2 + 2) * 10 : (TH + 2) * 18, PG = (PR + 7) / 8;
    const size_t ring = 2 * (size_t)(PG * 1024) + BSTAGES * (size_t)BN * 128 + (GN ? 2048 : 0);
    const size_t epi = NW * (size_t)WM * (WN + 4) * sizeof(float);
    const size_t lds = ring > epi ? ring : epi;
    auto kern = conv3x3_halo_kernel<T, BN, WAVES_M, WAVES_N, UPS, BSTAGES, GN, SPLIT, MS, SKIP, STREAM, KSPLIT>;
    if constexpr (SKIP) {
        p.skc0 = (p.SC0 + BK - 1) / BK;
        p.skc1 = (p.SC1 + BK - 1) / BK;
    }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    p.kc0 = (p.C0 + BK - 1) / BK;
    p.kc1 = (p.C1 + BK - 1) / BK;
    p.KT = (p.kc0 + p.kc1) * 9;
    p.tiles_n = (p.Ncols - p.n_base + BN - 1) / BN;
    p.tw_log2 = 4;  // TH x 16 pixel patches
    p.th = TH;
    p.tiles_pw = (p.Wo + 15) / 16;
    p.tiles_pi = p.tiles_pw * (p.Ho / TH);
    p.tiles_m = p.tiles_pi * p.N;
    p.tiles_per_image = p.tiles_pi;  // (statistics slots: tile of the image x WAVES_M)
    // streaming instances (the kernel's STREAM): runs of tpw consecutive pixel tiles of one image per workgroup.  Which workgroup
    // computes a tile never changes its result (same K order, same MFMAs, its own statistics slot), so the choice may depend on the batch.
    p.tpw = 1;
    if constexpr (SPLIT && MS == 16 && BN >= 128 && !SKIP && !UPS) {
        const long long slots = 256LL * (NW == 4 ? 2 : 1);  // co-resident workgroups of the chip
        int want = opt(OPT_HALO_TPW);
        if (want <= 0) {  // at least two rounds of workgroups stay (measured on 256 x 256 and 128 x 128 maps: 4 ... 8 tiles per run tie)
            want = 1;
            while (want < 8 && (long long)p.tiles_m * p.tiles_n / (2 * want) >= 2 * slots) want *= 2;
        }
        while (want > 1 && p.tiles_pi % want) --want;
        if (p.splitk > 1) want = 1;
        if constexpr (!STREAM) {
            if (want > 1) return launch_halo<T, BN, WAVES_M, WAVES_N, UPS, BSTAGES, GN, SPLIT, MS, SKIP, true>(p, st);
        } else {
            p.tpw = want;
        }
    }
    const long long nblk = (long long)(p.tiles_m / p.tpw) * p.tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        eod_set_error("conv_halo: bad grid %lld", nblk);
        return EOD_EINVAL;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)(p.splitk > 1 ? p.splitk : 1)), dim3(64 * NW), lds, st, p);
    EOD_CHECK_LAUNCH("conv3x3_halo");
    return EOD_OK;
}

template <typename T, bool CONV> static int launch_T(IgemmP& p, int batch, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {  // fp16 products on v_mfma_f32_16x16x32_f16
        if (p.Ncols <= 32) return launch_cfg<T, CONV, 128, 32, 4, 1, 2, false, 16>(p, batch, st);
        if (p.Ncols <= 64) return launch_cfg<T, CONV, 128, 64, 4, 1, 2, false, 16>(p, batch, st);
        return launch_cfg<T, CONV, 128, 128, 2, 2, 2, false, 16>(p, batch, st);
    } else {                         // exact fp32: v_mfma_f32_32x32x2_f32
        if (p.Ncols <= 32) return launch_cfg<T, CONV, 128, 32, 4, 1, 2>(p, batch, st);
        if (p.Ncols <= 64) return launch_cfg<T, CONV, 128, 64, 4, 1, 2>(p, batch, st);
        // (a 256x128 tile / 8 waves / 3-stage ring measured 0-5 % SLOWER than two co-resident 128x128 workgroups per CU)
        return launch_cfg<T, CONV, 128, 128, 2, 2, 2>(p, batch, st);
    }
}

// fp32 conv as the split-fp16 product on the generic kernel (1x1, stride 2, ragged maps)
static int launch_conv_split(IgemmP& p, int batch, hipStream_t st) {
    if (p.Ncols <= 32) return launch_cfg<float, true, 128, 32, 4, 1, 2, true, 16>(p, batch, st);
    if (p.Ncols <= 64) return launch_cfg<float, true, 128, 64, 4, 1, 2, true, 16>(p, batch, st);
    // 256 columns per 8-wave workgroup: the pixel rows of a K-step are fetched and split once for two N-tiles (these launches are
    // bound by that in-place split: four pieces per wave against 48 MFMAs) -- the qkv / proj 1x1 convs of the attention blocks
    // whole tiles inside one image, no split-K, NHWC output: the instances with swapped MFMA operands and the direct epilogue
    // (halo_epilogue_direct: no LDS transpose; fp32 or pre-split output)
    const bool direct = p.splitk <= 1 && !p.out_nchw && p.HWd % 128 == 0 && p.M % 128 == 0 && !p.par && p.Ncols % 4 == 0;
    if (p.Ncols % 256 == 0 && batch == 1 && ((p.M + 127) / 128) * (p.Ncols / 256) >= 256 && opt(OPT_HALO_BN256))
        return direct ? launch_cfg<float, true, 128, 256, 2, 4, 2, true, 16, true>(p, batch, st)
                      : launch_cfg<float, true, 128, 256, 2, 4, 2, true, 16>(p, batch, st);
    return direct ? launch_cfg<float, true, 128, 128, 2, 2, 2, true, 16, true>(p, batch, st)
                  : launch_cfg<float, true, 128, 128, 2, 2, 2, true, 16>(p, batch, st);
}

// row length (elements) of tap-major packed weights: 9*C0 rounded up to whole 128-byte K-steps
extern "C" int eod_conv_tapmajor_ldk(int C0, int dtype) {
    const int bk = 128 / (dtype == EOD_F16 ? 2 : 4);
    return (9 * C0 + bk - 1) / bk * bk;
}

// which kernel configuration a conv descriptor gets (shared by the launcher and eod_conv_stats_slots)
static bool conv_uses_halo(const eod_conv_desc* d, int Ho, int Wo) {
    // wide convs (128-column tiles) and the narrow NCHW-fp32 head conv (32-column tiles, 4x1 waves)
    const bool shape_ok = ((d->Cout > 64 && !d->out_nchw_f32) || (d->Cout <= 32 && d->out_nchw_f32 && !d->upsample)) && d->upsample != 2;
    // (8-wide maps of the wide convs too: one image row per tile row, the right half of the 8 x 16 tile masked -- half the MFMA rows idle, on
    //  maps whose launches are latency-bound; with the K slices of conv_splitk they leave the generic kernel's 16-way split + gather)
    const bool w_ok = Wo % 16 == 0 || (Wo == 8 && d->Cout > 64 && !d->out_nchw_f32 && !d->upsample);
    return d->ksize == 3 && d->stride == 1 && d->pad == 1 && !d->pad_tl && w_ok && Ho % 8 == 0 && shape_ok && !d->w_tapmajor;
}
static int conv_waves_m(const eod_conv_desc* d, bool halo) { return (halo || d->Cout > 64) ? 2 : 4; }
static int conv_bm(const eod_conv_desc* d, bool halo) { return 128; }

// 1 if this conv can run as the split-fp16 product (fp32 storage, weights packed by eod_pack_conv_weight_split): whole chunk
// pairs (8 channels) per source; every kernel variant has it except the thin-input (tap-major) first conv
static bool conv_split_ok(const eod_conv_desc* d, int Ho, int Wo) {
    // (thin-input first conv: the K axis is [tap][C0] flattened and padded to whole K-steps, so its 8-k pairs always exist)
    return d->dtype == EOD_F32 && (d->w_tapmajor || (d->C0 % 8 == 0 && d->C1 % 8 == 0)) && d->upsample != 2;
}
// zero-insertion upsampling through the four parity-class launches (see eod_conv2d_igemm); other geometries keep the single full-grid
// launch that multiplies the inserted zeros
static bool conv_parity_ok(const eod_conv_desc* d, int Ho, int Wo) {
    return d->upsample == 2 && d->ksize == 3 && d->stride == 1 && d->pad == 1 && !d->pad_tl && !d->out_nchw_f32 && !d->stats &&
           !d->w_tapmajor && !d->w_split && Ho % 2 == 0 && Wo % 2 == 0;
}
// upsample = 3: nearest-2x upsampling + 3x3 conv as four 2x2-tap parity classes with pre-summed weights (conv_up4_halo_kernel)
static bool conv_up4_ok(const eod_conv_desc* d) {
    const bool store_ok = d->dtype == EOD_F16 || (d->dtype == EOD_F32 && d->w_split);
    return d->upsample == 3 && d->ksize == 3 && d->stride == 1 && d->pad == 1 && !d->pad_tl && d->C1 == 0 && !d->x2 && d->Cout > 64 &&
           d->Cout % 8 == 0 && d->C0 % 8 == 0 && (d->W % 16 == 0 || d->W == 8) && d->H % 8 == 0 && !d->out_nchw_f32 && !d->w_tapmajor &&
           !d->gn_scale_shift && store_ok;  // (8-wide stored maps: the right half of the 8 x 16 tile masked, like conv3x3_halo_kernel's)
}
extern "C" int eod_conv_up4_ok(const eod_conv_desc* d) { return d && conv_up4_ok(d) ? 1 : 0; }
// upsample = 4: backward-data of the parity-class upsample conv (conv_up4_halo_kernel<BWD>): x = dY [N][H][W][C0] on the (2H' x 2W') grid,
// y = dX [N][Ho = H/2][Wo = W/2][Cout], w = eod_pack_conv_weight_dgrad of the forward class-kernel tensor ([slot][Cout][4*C0])
static bool conv_up4_bwd_ok(const eod_conv_desc* d) {
    return d->upsample == 4 && d->dtype == EOD_F16 && d->ksize == 3 && d->stride == 1 && d->pad == 1 && !d->pad_tl && d->C1 == 0 && !d->x2 &&
           d->Cout > 64 && d->Cout % 8 == 0 && d->C0 % 8 == 0 && d->H % 2 == 0 && d->W % 2 == 0 && d->Ho == d->H / 2 && d->Wo == d->W / 2 &&
           d->Wo % 16 == 0 && d->Ho % 8 == 0 && !d->out_nchw_f32 && !d->w_tapmajor && !d->gn_scale_shift && !d->stats && !d->w_split;
}
extern "C" int eod_conv_up4_bwd_ok(const eod_conv_desc* d) { return d && conv_up4_bwd_ok(d) ? 1 : 0; }
static int conv_up4_bwd(const eod_conv_desc* d, void* stream) {
    EOD_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C0 > 0 && d->Cout > 0 && d->x && d->w && d->y, "conv (upsample 4): bad args");
    EOD_REQUIRE(conv_up4_bwd_ok(d), "conv: upsample = 4 needs a geometry for which eod_conv_up4_bwd_ok(d) == 1");
    EOD_REQUIRE(eod_aligned16(d->x) && eod_aligned16(d->w), "conv: 16-byte alignment");
    EOD_REQUIRE((long long)d->H * d->W * d->C0 * 2 < 0x7fffffffLL && 9LL * d->Cout * 4 * d->C0 * 2 < 0x7fffffffLL, "conv (upsample 4): operand exceeds the 2 GiB window");
    IgemmP p = {};
    p.a0 = (const char*)d->x; p.b = (const char*)d->w; p.bias = d->bias; p.bias_mode = d->bias ? 1 : 0;
    p.cbias = d->cbias; p.cbias_stride = d->cbias_stride; p.res = (const char*)d->res; p.y = (char*)d->y;
    p.N = d->N; p.H = d->Ho; p.W = d->Wo;  // the kernel works on the STORED (output) map; the gradient is (2H x 2W)
    p.C0 = d->C0; p.C1 = 0; p.Cin = d->C0; p.Cout = d->Cout; p.KS = 3; p.stride = 1; p.pad = 1;
    p.Ho = d->Ho; p.Wo = d->Wo; p.HoWo = d->Ho * d->Wo; p.Heff = d->Ho; p.Weff = d->Wo;
    p.Hd = d->Ho; p.Wd = d->Wo; p.HWd = d->Ho * d->Wo;
    p.M = (long long)d->N * d->Ho * d->Wo; p.Ncols = d->Cout; p.taps = 9; p.alpha = d->alpha; p.nb1 = 1; p.tapmajor_log2 = -1;
    return launch_up4<half_t, false, true>(p, (hipStream_t)stream);
}
// the UNet's output head on conv_head_kernel (EOD_HEAD=0: the 32-column halo instance, A/B)
// the fp32x3 first conv (thin input, tap-major split weights) on conv_first_x3_kernel: 8 x 16 pixel tiles, 128-column workgroups
static bool conv_first_ok(const eod_conv_desc* d, int Ho, int Wo) {
    return opt(OPT_FIRST) != 0 && d->w_tapmajor && d->w_split && d->dtype == EOD_F32 && d->ksize == 3 && d->stride == 1 && d->pad == 1 && !d->pad_tl &&
           !d->upsample && d->C1 == 0 && !d->x2 && (d->C0 == 4 || d->C0 == 8) && Ho % 8 == 0 && Wo % 16 == 0 && d->Cout % 4 == 0 && d->Cout > 64 && !d->out_nchw_f32 &&
           !d->gn_scale_shift && !d->skip_x && !d->x_presplit && !d->y_presplit_bound;
}
static bool conv_head_ok(const eod_conv_desc* d, bool halo_ok) {
    const bool on = opt(OPT_HEAD) != 0;
    const bool store_ok = d->dtype == EOD_F16 || (d->dtype == EOD_F32 && d->w_split);
    return on && halo_ok && d->out_nchw_f32 && d->Cout <= 16 && d->gn_scale_shift && d->C1 == 0 && !d->x2 && !d->upsample && !d->res && !d->cbias &&
           !d->stats && store_ok && d->C0 % 8 == 0 && d->C0 <= HEAD_MAX_C;
}
// 256-column convs with a fused GroupNorm on the 8-wave instance that shares one patch between the two N-tiles (EOD_HALO_BN256=0: off, A/B).
// (fp16 storage WITHOUT a fused GroupNorm -- the training step's convs -- measured on it in round 4: 128 x 128 maps -3 %, 32 x 32 +2 %: not used)
static int conv_splitk(const eod_conv_desc* d, int Ho, int Wo, bool halo);
static bool halo_bn256(const eod_conv_desc* d) {
    const bool on = opt(OPT_HALO_BN256) != 0 && conv_splitk(d, d->H, d->W, true) <= 1;
    // only where it still fills the chip: one 8-wave workgroup occupies a CU, so fewer than 256 of them leave CUs idle (32 x 32 maps at
    // batch 8: 128 workgroups, measured -20 %; the choice never changes a result: same K order, same MFMAs)
    const long long wgs = (long long)d->N * (d->H / 8) * ((d->W + 15) / 16) * (d->Cout / 256);
    return on && d->Cout % 256 == 0 && !d->upsample && wgs >= 256;
}
// few pixel tiles (16 x 16 maps and smaller at batch 16): 64-column N-tiles double the workgroup count of a launch that cannot fill the
// chip (a 384-column conv on a 16 x 16 map at batch 16 has 32 x 3 = 96 workgroups of 128 columns).  Decided from the PER-IMAGE geometry
// at the nominal batch of 16, never from the actual batch -- and the choice never changes a result (same K order, same MFMA tiles).
static bool halo_bn64(const eod_conv_desc* d) {
    const long long wgs = 16LL * (d->H / 8) * ((d->W + 15) / 16) * ((d->Cout + 127) / 128);
    return d->Cout > 64 && !d->upsample && wgs < 256 && conv_splitk(d, d->H, d->W, true) <= 1;
}
// 384, 640, ... columns: all but the last 128 on the 8-wave form, as a launch of its own
static bool halo_bn256_plus128(const eod_conv_desc* d) {
    const bool on = opt(OPT_HALO_BN256) != 0 && conv_splitk(d, d->H, d->W, true) <= 1;
    const long long wgs = (long long)d->N * (d->H / 8) * ((d->W + 15) / 16) * ((d->Cout - 128) / 256);
    return on && d->Cout > 256 && d->Cout % 256 == 128 && !d->upsample && wgs >= 256;
}
// ResBlock 1x1 skip conv fused behind the 3x3 K loop (conv3x3_halo_kernel<SKIP>; EOD_SKIP_FUSE=0: off, A/B)
static bool conv_skip_geom_ok(const eod_conv_desc* d) {
    const bool on = opt(OPT_SKIP_FUSE) != 0;
    if (!on || !d || d->ksize != 3 || d->stride != 1 || d->pad != 1 || d->upsample || d->pad_tl || d->C1 != 0 || d->x2 || d->res ||
        d->out_nchw_f32 || d->w_tapmajor || d->Cout <= 64 || d->Cout % 8 || d->C0 % 8)
        return false;
    if (d->skip_C0 <= 0 || d->skip_C0 % 8 || d->skip_C1 < 0 || d->skip_C1 % 8) return false;
    const int Ho = d->H, Wo = d->W;
    if (!conv_uses_halo(d, Ho, Wo)) return false;
    return d->dtype == EOD_F16 || (d->dtype == EOD_F32 && d->w_split);
}
extern "C" int eod_conv_skip_ok(const eod_conv_desc* d) { return conv_skip_geom_ok(d) ? 1 : 0; }
extern "C" int eod_conv_split_ok(const eod_conv_desc* d) {
    if (!d) return 0;
    const int Heff = d->H * (d->upsample ? 2 : 1), Weff = d->W * (d->upsample ? 2 : 1);
    const int Ho = (Heff + d->pad_tl + 2 * d->pad - d->ksize) / d->stride + 1;
    const int Wo = (Weff + d->pad_tl + 2 * d->pad - d->ksize) / d->stride + 1;
    return conv_split_ok(d, Ho, Wo) ? 1 : 0;
}

// 1 if eod_conv2d_igemm can apply GroupNorm(+SiLU) to the conv INPUT on the fly (gn_scale_shift) for this geometry.
// The fused form re-normalises the halo patch once per N-tile (Cout / 128 times), so in fp16 storage it only pays while the conv has
// few N-tiles (measured on MI355X: Cout <= 256, the separate apply pass wins beyond); in fp32 storage (split product) it pays at
// every width the UNet has.
extern "C" int eod_conv_gn_fusable(const eod_conv_desc* d) {
    if (!d || d->upsample) return 0;
    const int Ho = (d->H + d->pad_tl + 2 * d->pad - d->ksize) / d->stride + 1;
    const int Wo = (d->W + d->pad_tl + 2 * d->pad - d->ksize) / d->stride + 1;
    const int env_cout = opt(OPT_GN_FUSE_MAX_COUT);
    // measured (same-box A/B of bench.py): fp32 storage with the split product pays for the fusion at every width (a separate pass
    // moves 8 bytes per element; 384 / 512-wide layers: step -0.2 ms), fp16 storage up to 256 output channels
    const int max_cout = env_cout >= 0 ? env_cout : (d->w_split ? 512 : 256);
    if (d->Cout > max_cout) return 0;
    return conv_uses_halo(d, Ho, Wo) ? 1 : 0;
}

// split-K factor of a conv that the generic kernel would run with too few workgroups to fill the chip (small maps)
// K slices of a split halo conv: `s` slices of `per` channel chunks, the last one takes the rest + the fused skip phase.  The largest
// s <= smax whose longest slice stays within 25 % of the mean (K-steps, the skip phase's included); per of that plan is returned.
static int halo_split_plan(const eod_conv_desc* d, int smax, int* per_out) {
    const int bk = 128 / eod_esize(d->dtype);
    const int kc = (d->C0 + bk - 1) / bk + (d->C1 + bk - 1) / bk;
    const int sk = d->skip_x ? (d->skip_C0 + bk - 1) / bk + (d->skip_C1 + bk - 1) / bk : 0;  // K-steps of the skip phase
    const int total = kc * 9 + sk;
    for (int s = smax; s >= 2; --s) {
        for (int per = (total + 9 * s - 1) / (9 * s); per >= 1 && per >= total / (9 * s); --per) {
            if (per * (s - 1) >= kc) continue;  // the last slice keeps at least one chunk
            const int last = (kc - per * (s - 1)) * 9 + sk, longest = last > per * 9 ? last : per * 9;
            if ((long long)longest * s * 100 <= (long long)total * 125) {
                if (per_out) *per_out = per;
                return s;
            }
        }
    }
    return 1;
}
static int conv_splitk(const eod_conv_desc* d, int Ho, int Wo, bool halo) {
    // the factor must NOT depend on the batch size: the K summation order of a sample has to be the same whether it is
    // computed alone or inside a larger batch (bit-exact batch-sharding invariance), so a nominal batch of 16 is used
    const int bk = 128 / eod_esize(d->dtype);
    if (halo) {
        // halo-patch kernel: 128-column tiles of 8 x 16 pixels; fewer than one workgroup per CU (16 x 16 maps at batch 16) -> the channel
        // chunks are split over gridDim.y workgroups (whole chunks: the nine taps of a chunk share its staged patch).  32 x 32 maps
        // (256 workgroups) measured SLOWER split in two: the reduce pass costs more than the second workgroup per CU gains.
        if (!opt(OPT_HALO_SPLITK) || d->out_nchw_f32 || d->upsample || d->Cout <= 64 || d->Cout % 4) return 1;
        const long long wgs = 16LL * (Ho / 8) * ((Wo + 15) / 16) * ((d->Cout + 127) / 128);
        const int kc = (d->C0 + bk - 1) / bk + (d->C1 + bk - 1) / bk;
        if (wgs >= 256 || kc < 4) return 1;
        int s = (int)((512 + wgs - 1) / wgs);
        if (s > kc / 2) s = kc / 2;  // at least two chunks (18 K-steps) per slice
        if (s > 8) s = 8;
        return halo_split_plan(d, s, nullptr);
    }
    if (d->out_nchw_f32 || d->Cout % 4 || d->w_tapmajor) return 1;
    const long long M = 16LL * Ho * Wo;
    const int bn = d->Cout <= 32 ? 32 : (d->Cout <= 64 ? 64 : 128);
    const long long tiles = ((M + 127) / 128) * ((d->Cout + bn - 1) / bn);
    const int kt = ((d->C0 + bk - 1) / bk + (d->C1 + bk - 1) / bk) * d->ksize * d->ksize;
    if (tiles >= 128 || kt < 8) return 1;
    int s = (int)(512 / tiles);  // two workgroups per CU (same-box A/B on the 64 x 64 configuration: 3.59 -> 3.47 ms per step against 256 / tiles)
    if (s > kt / 4) s = kt / 4;
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}

// bytes of caller-provided fp32 workspace eod_conv2d_igemm needs for this descriptor (0 = none)
extern "C" int64_t eod_conv_workspace_size(const eod_conv_desc* d) {
    if (!d || d->upsample == 4) return 0;
    const int Heff = d->H * (d->upsample ? 2 : 1), Weff = d->W * (d->upsample ? 2 : 1);
    const int Ho = (Heff + d->pad_tl + 2 * d->pad - d->ksize) / d->stride + 1;
    const int Wo = (Weff + d->pad_tl + 2 * d->pad - d->ksize) / d->stride + 1;
    const int s = conv_splitk(d, Ho, Wo, conv_uses_halo(d, Ho, Wo));
    return s > 1 ? (int64_t)s * d->N * Ho * Wo * d->Cout * 4 : 0;
}

extern "C" int eod_conv_stats_slots(const eod_conv_desc* d) {
    if (!d || d->out_nchw_f32 || d->upsample == 4) return 0;
    if (d->Cout % (16 / eod_esize(d->dtype))) return 0;  // statistics are accumulated on full 16-byte output chunks only
    const int Heff = d->H * (d->upsample ? 2 : 1), Weff = d->W * (d->upsample ? 2 : 1);
    const int Ho = (Heff + d->pad_tl + 2 * d->pad - d->ksize) / d->stride + 1;
    const int Wo = (Weff + d->pad_tl + 2 * d->pad - d->ksize) / d->stride + 1;
    if (d->upsample == 3) return (d->H / 8) * ((d->W + 15) / 16) * 8;  // parity-class form: (8 x 16 tile of the STORED map, class, wave row)
    const bool halo = conv_uses_halo(d, Ho, Wo);
    if (conv_splitk(d, Ho, Wo, halo) > 1) return 1;  // split-K: the reduce pass takes the sums, one slot per image
    if (halo) return (Ho / 8) * ((Wo + 15) / 16) * 2;  // 8 x 16 pixel tiles (8-wide maps: half of each tile masked), two wave rows each
    const int bm = conv_bm(d, halo);
    if ((Ho * Wo) % bm != 0) return 0;  // tiles must not straddle images
    return (Ho * Wo / bm) * conv_waves_m(d, halo);
}

// second pass of a split-K conv: sum the gridDim.y raw fp32 partial tiles of the workspace in their fixed order, apply alpha / bias /
// per-sample bias / residual, store y (and the next GroupNorm's partial sums: one slot per image)
static int splitk_finish(const eod_conv_desc* d, const IgemmP& p, int splitk, hipStream_t st) {
    if (d->stats) {  // reduce + the next GroupNorm's partial sums (one slot per image)
        const dim3 grid((unsigned)((p.Cout + 63) / 64), (unsigned)d->N);
        if (d->dtype == EOD_F16)
            hipLaunchKernelGGL(splitk_reduce_stats_kernel<half_t>, grid, dim3(256), 0, st, (const float*)d->workspace, splitk, p.M, p.Cout, p.HoWo, d->alpha, d->bias, d->cbias, (long long)d->cbias_stride, (const half_t*)d->res, (half_t*)d->y, d->stats);
        else
            hipLaunchKernelGGL(splitk_reduce_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)d->workspace, splitk, p.M, p.Cout, p.HoWo, d->alpha, d->bias, d->cbias, (long long)d->cbias_stride, (const float*)d->res, (float*)d->y, d->stats);
        EOD_CHECK_LAUNCH("splitk_reduce_stats");
        return EOD_OK;
    }
    const long long total4 = p.M * p.Cout / 4;
    const unsigned blocks = (unsigned)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
    if (d->dtype == EOD_F16)
        hipLaunchKernelGGL(splitk_reduce_kernel<half_t>, dim3(blocks), dim3(256), 0, st, (const float*)d->workspace, splitk, p.M, p.Cout, p.HoWo, d->alpha, d->bias, d->cbias, (long long)d->cbias_stride, (const half_t*)d->res, (half_t*)d->y, (const float*)nullptr);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)d->workspace, splitk, p.M, p.Cout, p.HoWo, d->alpha, d->bias, d->cbias, (long long)d->cbias_stride, (const float*)d->res, (float*)d->y, (const float*)nullptr);  // (split-fp16 partial tiles arrive un-scaled)
    EOD_CHECK_LAUNCH("splitk_reduce");
    return EOD_OK;
}

extern "C" int eod_conv2d_igemm(const eod_conv_desc* d, void* stream) {
    EOD_REQUIRE(d, "conv: null desc");
    EOD_REQUIRE(d->dtype == EOD_F32 || d->dtype == EOD_F16, "conv: bad dtype %d", d->dtype);
    const int es = eod_esize(d->dtype), epc = 16 / es;
    EOD_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv: ksize %d", d->ksize);
    EOD_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d", d->stride);
    EOD_REQUIRE(d->upsample >= 0 && d->upsample <= 4, "conv: upsample %d", d->upsample);
    if (d->upsample == 4) return conv_up4_bwd(d, stream);
    EOD_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C0 > 0 && d->C1 >= 0 && d->Cout > 0, "conv: bad dims");
    EOD_REQUIRE(d->C0 % epc == 0 && d->C1 % epc == 0, "conv: C0=%d C1=%d must be multiples of %d", d->C0, d->C1, epc);
    EOD_REQUIRE(d->x && d->w && d->y, "conv: null pointer");
    EOD_REQUIRE((d->C1 == 0) == (d->x2 == nullptr), "conv: x2/C1 mismatch");
    EOD_REQUIRE(eod_aligned16(d->x) && eod_aligned16(d->w) && (!d->x2 || eod_aligned16(d->x2)), "conv: 16-byte alignment");
    const int Heff = d->H * (d->upsample ? 2 : 1), Weff = d->W * (d->upsample ? 2 : 1);
    const int Ho = (Heff + d->pad_tl + 2 * d->pad - d->ksize) / d->stride + 1;
    const int Wo = (Weff + d->pad_tl + 2 * d->pad - d->ksize) / d->stride + 1;
    EOD_REQUIRE(Ho == d->Ho && Wo == d->Wo, "conv: Ho/Wo mismatch: got %dx%d, geometry gives %dx%d", d->Ho, d->Wo, Ho, Wo);
    EOD_REQUIRE(!(d->out_nchw_f32 && d->res), "conv: residual not supported with NCHW output");
    // per-lane source offsets are 32-bit pixel indices relative to the tile's first image; a 128-row tile spans
    // at most 129 images (1x1 maps), so bound the pixel index range conservatively
    // per-lane source offsets are 32-bit BYTE offsets inside a 2 GiB window that starts at the tile's first image; a
    // 128-row tile touches at most 2 images once Ho*Wo >= 128 and at most 129 otherwise
    {
        const long long img_bytes = (long long)d->H * d->W * (d->C0 > d->C1 ? d->C0 : d->C1) * es;
        // (maps that split into whole 256-row tiles never put two images into one tile: the window then only has to hold ONE image)
        const long long span = ((long long)Ho * Wo % 256 == 0) ? 1 : ((long long)Ho * Wo >= 128) ? 2 : 130;
        EOD_REQUIRE(img_bytes * span < 0x7fffffffLL, "conv: one image (%lld bytes) is too large for the 2 GiB tile window", img_bytes);
        EOD_REQUIRE((long long)d->ksize * d->ksize * d->Cout * (d->C0 + d->C1) * es < 0x7fffffffLL, "conv: weights exceed the 2 GiB window");
    }
    IgemmP p = {};
    p.a0 = (const char*)d->x;
    p.a1 = (const char*)d->x2;
    p.b = (const char*)d->w;
    p.bias = d->bias;
    p.bias_mode = d->bias ? 1 : 0;
    p.cbias = d->cbias;
    p.cbias_stride = d->cbias_stride;
    p.res = (const char*)d->res;
    p.y = (char*)d->y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C0 = d->C0; p.C1 = d->C1; p.Cin = d->C0 + d->C1; p.Cout = d->Cout;
    p.KS = d->ksize; p.stride = d->stride; p.pad = d->pad; p.ups = d->upsample; p.pad_tl = d->pad_tl;
    p.Ho = Ho; p.Wo = Wo; p.HoWo = Ho * Wo; p.Heff = Heff; p.Weff = Weff;
    p.Hd = Ho; p.Wd = Wo; p.HWd = Ho * Wo;
    p.M = (long long)d->N * Ho * Wo;
    p.Ncols = d->Cout;
    p.taps = d->ksize * d->ksize;
    p.out_nchw = d->out_nchw_f32;
    p.alpha = d->alpha;
    p.nb1 = 1;
    p.tapmajor_log2 = -1;
    if (d->w_tapmajor) {
        // thin-input mode (first conv: 3 / 7 / 13 image channels padded to one or two 16-byte chunks): the K axis is
        // [tap][C0] flattened, so the conv takes ceil(9*C0/BK) K-steps instead of 9 mostly-zero ones
        const int cpt = d->C0 / epc;
        EOD_REQUIRE(d->ksize == 3 && d->C1 == 0 && !d->upsample && !d->pad_tl && !d->gn_scale_shift && (cpt == 1 || cpt == 2 || cpt == 4),
                    "conv: w_tapmajor needs a 3x3 conv of a single source with C0 in {1,2,4} x %d channels", epc);
        p.tapmajor_log2 = cpt == 1 ? 0 : cpt == 2 ? 1 : 2;
        p.ldk = eod_conv_tapmajor_ldk(d->C0, d->dtype);
    }
    hipStream_t st = (hipStream_t)stream;
    // 3x3 / stride 1 / pad 1 on maps that tile into 8x16 patches: halo-patch kernel
    const bool halo_ok = conv_uses_halo(d, Ho, Wo);
    if (d->stats) {
        const int slots = eod_conv_stats_slots(d);
        EOD_REQUIRE(slots > 0 && slots == d->stats_slots, "conv: stats_slots=%d but this geometry provides %d", d->stats_slots, slots);
        p.stats = d->stats;
        p.stats_P = slots;
        p.tiles_per_image = Ho * Wo / conv_bm(d, halo_ok);
    }
    EOD_REQUIRE(!d->w_split || (conv_split_ok(d, Ho, Wo) && d->w_scale),
                "conv: w_split needs a geometry for which eod_conv_split_ok(d) == 1 and the w_scale of eod_pack_conv_weight_split");
    p.w_scale = d->w_split ? d->w_scale : nullptr;
    p.w_rexp = d->w_split ? reinterpret_cast<const int*>(d->w_scale) + 4 : nullptr;  // (EOD_WSCALE_ROWS of csrc/misc.hip)
    p.w_row0 = 0;
    p.a_bound = d->w_split ? d->a_bound : nullptr;
    EOD_REQUIRE(!d->x_presplit || (d->w_split && d->a_bound && !halo_ok && !d->upsample && !d->w_tapmajor && !d->gn_scale_shift),
                "conv: x_presplit needs w_split with the producer's bound table, on the generic kernel (1x1 / stride-2 / small maps)");
    p.a_ps = d->x_presplit;
    EOD_REQUIRE(!d->y_presplit_bound || (d->w_split && !halo_ok && !d->out_nchw_f32 && !d->stats && d->Cout % 8 == 0 && (Ho * Wo) % 128 == 0 &&
                                         conv_splitk(d, Ho, Wo, false) <= 1),
                "conv: y_presplit_bound needs w_split on the generic kernel, Cout %% 8 == 0, whole 128-row tiles per image, no statistics");
    p.y_ps_bound = d->y_presplit_bound;
    p.skip_bound = (d->w_split && d->skip_x) ? d->skip_bound : nullptr;
    if (d->upsample == 3) {
        EOD_REQUIRE(conv_up4_ok(d), "conv: upsample = 3 (parity-class form of the nearest-2x conv) needs a geometry for which eod_conv_up4_ok(d) == 1");
        return d->dtype == EOD_F16 ? launch_up4<half_t, false>(p, st) : launch_up4<float, true>(p, st);
    }
    if (conv_first_ok(d, Ho, Wo)) return d->C0 == 4 ? launch_first<4>(p, st) : launch_first<8>(p, st);
    if (conv_head_ok(d, halo_ok)) {  // output head: GroupNorm + SiLU fused, <= 16 channels, NCHW fp32 (conv_head_kernel)
        p.gn_ss = d->gn_scale_shift;
        p.gn_silu = d->gn_silu;
        p.tpw = opt(OPT_HEAD_TPW);
        return d->dtype == EOD_F16 ? launch_head<half_t, false>(p, st) : launch_head<float, true>(p, st);
    }
    // ---- halo-patch kernels (3x3 / stride 1 / pad 1 on maps that tile into 8 x 16 patches), with or without the fused skip conv ----
    constexpr int NOT_HALO = 1 << 20;
    auto run_halo = [&](IgemmP& p) -> int {
        if (d->skip_x) {  // ResBlock 1x1 skip conv fused behind the 3x3 K loop
            EOD_REQUIRE(conv_skip_geom_ok(d) && halo_ok && d->skip_w, "conv: skip_x needs a geometry for which eod_conv_skip_ok(d) == 1, and skip_w");
            EOD_REQUIRE((d->skip_C1 > 0) == (d->skip_x2 != nullptr), "conv: skip_x2 / skip_C1 mismatch");
            EOD_REQUIRE(eod_aligned16(d->skip_x) && eod_aligned16(d->skip_x2) && eod_aligned16(d->skip_w), "conv: 16-byte alignment (skip operands)");
            {
                const int es = d->dtype == EOD_F16 ? 2 : 4;
                const long long smax = d->skip_C0 > d->skip_C1 ? d->skip_C0 : d->skip_C1;
                EOD_REQUIRE((long long)d->H * d->W * smax * es < 0x7fffffffLL && (long long)d->Cout * (d->skip_C0 + d->skip_C1) * es < 0x7fffffffLL,
                            "conv: skip operand exceeds the 2 GiB window");
            }
            p.sx0 = (const char*)d->skip_x;
            p.sx1 = (const char*)d->skip_x2;
            p.b2 = (const char*)d->skip_w;
            p.SC0 = d->skip_C0;
            p.SC1 = d->skip_C1;
            p.gn_ss = d->gn_scale_shift;
            p.gn_silu = d->gn_silu;
            if (halo_bn64(d)) {
                if (d->w_split) return d->gn_scale_shift ? launch_halo<float, 64, 2, 2, false, 2, true, true, 16, true>(p, st)
                                                         : launch_halo<float, 64, 2, 2, false, 2, false, true, 16, true>(p, st);
                return d->gn_scale_shift ? launch_halo<half_t, 64, 2, 2, false, 2, true, false, 16, true>(p, st)
                                         : launch_halo<half_t, 64, 2, 2, false, 2, false, false, 16, true>(p, st);
            }
            if (d->w_split && halo_bn256(d))
                return d->gn_scale_shift ? launch_halo<float, 256, 2, 4, false, 2, true, true, 16, true>(p, st)
                                         : launch_halo<float, 256, 2, 4, false, 2, false, true, 16, true>(p, st);
            if (d->w_split && d->gn_scale_shift && halo_bn256_plus128(d)) {  // 256 columns on the 8-wave form, the last 128 on the 4-wave one
                IgemmP q = p;
                q.Ncols = d->Cout - 128;
                const int rc = launch_halo<float, 256, 2, 4, false, 2, true, true, 16, true>(q, st);
                if (rc != EOD_OK) return rc;
                p.n_base = d->Cout - 128;
                return launch_halo<float, 128, 2, 2, false, 2, true, true, 16, true>(p, st);
            }
            if (d->w_split) return d->gn_scale_shift ? launch_halo<float, 128, 2, 2, false, 2, true, true, 16, true>(p, st)
                                                     : launch_halo<float, 128, 2, 2, false, 2, false, true, 16, true>(p, st);
            if (d->gn_scale_shift && halo_bn256(d)) return launch_halo<half_t, 256, 2, 4, false, 2, true, false, 16, true>(p, st);
            return d->gn_scale_shift ? launch_halo<half_t, 128, 2, 2, false, 2, true, false, 16, true>(p, st)
                                     : launch_halo<half_t, 128, 2, 2, false, 2, false, false, 16, true>(p, st);
        }
        if (halo_ok && d->w_split) {
            // fp32 storage, three fp16 MFMAs per product (weights pre-split and pre-scaled, activations split in LDS)
            if (d->gn_scale_shift) {
                EOD_REQUIRE(!d->upsample, "conv: fused input GroupNorm is not available together with upsample");
                p.gn_ss = d->gn_scale_shift;
                p.gn_silu = d->gn_silu;
                if (d->Cout <= 32) return launch_halo<float, 32, 4, 1, false, 2, true, true, 16>(p, st);
                if (halo_bn64(d)) return launch_halo<float, 64, 2, 2, false, 2, true, true, 16>(p, st);
                if (halo_bn256(d)) return launch_halo<float, 256, 2, 4, false, 2, true, true, 16>(p, st);
                if (halo_bn256_plus128(d)) {
                    IgemmP q = p;
                    q.Ncols = d->Cout - 128;
                    const int rc = launch_halo<float, 256, 2, 4, false, 2, true, true, 16>(q, st);
                    if (rc != EOD_OK) return rc;
                    p.n_base = d->Cout - 128;
                    return launch_halo<float, 128, 2, 2, false, 2, true, true, 16>(p, st);
                }
                return launch_halo<float, 128, 2, 2, false, 2, true, true, 16>(p, st);
            }
            if (d->Cout <= 32) return launch_halo<float, 32, 4, 1, false, 2, false, true, 16>(p, st);
            if (d->upsample) return launch_halo<float, 128, 2, 2, true, 2, false, true, 16>(p, st);
            if (halo_bn64(d)) return launch_halo<float, 64, 2, 2, false, 2, false, true, 16>(p, st);
            return launch_halo<float, 128, 2, 2, false, 2, false, true, 16>(p, st);
        }
        if (halo_ok && d->dtype == EOD_F16) {
            if (d->gn_scale_shift) {
                EOD_REQUIRE(!d->upsample, "conv: fused input GroupNorm is not available together with upsample");
                p.gn_ss = d->gn_scale_shift;
                p.gn_silu = d->gn_silu;
                if (d->Cout <= 32) return launch_halo<half_t, 32, 4, 1, false, 2, true, false, 16>(p, st);
                if (halo_bn64(d)) return launch_halo<half_t, 64, 2, 2, false, 2, true, false, 16>(p, st);
                if (halo_bn256(d)) return launch_halo<half_t, 256, 2, 4, false, 2, true, false, 16>(p, st);
                return launch_halo<half_t, 128, 2, 2, false, 2, true, false, 16>(p, st);
            }
            if (d->Cout <= 32) return launch_halo<half_t, 32, 4, 1, false, 2, false, false, 16>(p, st);
            if (d->upsample) return launch_halo<half_t, 128, 2, 2, true, 2, false, false, 16>(p, st);
            if (halo_bn64(d)) return launch_halo<half_t, 64, 2, 2, false, 2, false, false, 16>(p, st);
            return launch_halo<half_t, 128, 2, 2, false, 2, false, false, 16>(p, st);
        }
        if (halo_ok) {  // exact fp32 (v_mfma_f32_32x32x2_f32 on the same byte-oriented LDS image)
            if (d->gn_scale_shift) {  // GroupNorm(+SiLU) of the input fused into the patch staging
                EOD_REQUIRE(!d->upsample, "conv: fused input GroupNorm is not available together with upsample");
                p.gn_ss = d->gn_scale_shift;
                p.gn_silu = d->gn_silu;
                return d->Cout <= 32 ? launch_halo<float, 32, 4, 1, false, 2, true>(p, st) : launch_halo<float, 128, 2, 2, false, 2, true>(p, st);
            }
            if (d->Cout <= 32) return launch_halo<float, 32, 4, 1, false, 2, false>(p, st);  // head conv (out_nchw_f32)
            if (d->upsample) return launch_halo<float, 128, 2, 2, true, 2, false>(p, st);
            return launch_halo<float, 128, 2, 2, false, 2, false>(p, st);
        }
        return NOT_HALO;
    };
    if (halo_ok) {
        const int hsk = conv_splitk(d, Ho, Wo, true);
        if (hsk > 1) {
            // too few pixel tiles to fill the chip: K slices over gridDim.y workgroups (fp32 partial tiles in the caller's workspace),
            // then the deterministic reduce + bias / residual (+ statistics) pass
            EOD_REQUIRE(d->workspace && d->workspace_bytes >= eod_conv_workspace_size(d), "conv: workspace of %lld bytes required (eod_conv_workspace_size)", (long long)eod_conv_workspace_size(d));
            IgemmP q = p;
            q.stats = nullptr;
            q.splitk = hsk;
            (void)halo_split_plan(d, hsk, &q.splitk_per);
            q.y = (char*)d->workspace;
            q.bias = nullptr; q.bias_mode = 0; q.cbias = nullptr; q.res = nullptr; q.alpha = 1.0f;
            const int rc = run_halo(q);
            if (rc != EOD_OK) return rc == NOT_HALO ? EOD_EINVAL : rc;
            return splitk_finish(d, p, hsk, st);
        }
    }
    {
        const int rc = run_halo(p);
        if (rc != NOT_HALO) return rc;
    }
    EOD_REQUIRE(!d->gn_scale_shift, "conv: fused input GroupNorm needs the halo-patch kernel (ask eod_conv_gn_fusable first)");
    const int splitk = conv_splitk(d, Ho, Wo, false);
    if (splitk > 1) {
        // small maps: too few output tiles to fill 256 CUs -> split the K loop over gridDim.y workgroups (fp32 partial
        // tiles in the caller's workspace), then one deterministic reduce + bias/residual pass
        EOD_REQUIRE(d->workspace && d->workspace_bytes >= eod_conv_workspace_size(d), "conv: workspace of %lld bytes required (eod_conv_workspace_size)", (long long)eod_conv_workspace_size(d));
        IgemmP q = p;
        q.stats = nullptr;
        q.splitk = splitk;
        q.y = (char*)d->workspace;
        q.bias = nullptr; q.bias_mode = 0; q.cbias = nullptr; q.res = nullptr; q.alpha = 1.0f;
        const int rc = d->w_split ? launch_conv_split(q, splitk, st)
                                  : d->dtype == EOD_F16 ? launch_T<half_t, true>(q, splitk, st) : launch_T<float, true>(q, splitk, st);
        if (rc != EOD_OK) return rc;
        return splitk_finish(d, p, splitk, st);
    }
    if (d->w_split) return launch_conv_split(p, 1, st);
    if (conv_parity_ok(d, Ho, Wo)) {
        // zero-insertion upsampling (backward-data of a stride-2 conv): 3/4 of the (output position, tap) pairs meet an inserted zero,
        // and WHICH taps do depends only on the parity of the output position -> four launches, one per parity class, each over the
        // quarter grid with its 1 / 2 / 2 / 4 live taps (9 tap-visits per 4 outputs instead of 36)
        p.par = 1;
        p.Hd = Ho / 2; p.Wd = Wo / 2; p.HWd = p.Hd * p.Wd;
        p.M = (long long)d->N * p.HWd;
        for (int cls = 0; cls < 4; ++cls) {
            IgemmP q = p;
            q.par_y = cls >> 1; q.par_x = cls & 1;
            // output row ho reads upsampled row ho - 1 + dy, stored iff even: ho even -> dy = 1, ho odd -> dy in {0, 2} (same for x)
            const int ndy = q.par_y ? 2 : 1, ndx = q.par_x ? 2 : 1;
            const int dys[2] = {q.par_y ? 0 : 1, 2}, dxs[2] = {q.par_x ? 0 : 1, 2};
            q.taps = 0; q.taplist = 0;
            for (int a = 0; a < ndy; ++a)
                for (int b = 0; b < ndx; ++b) q.taplist |= (unsigned)(dys[a] * 3 + dxs[b]) << (4 * q.taps++);
            const int rc = d->dtype == EOD_F16 ? launch_T<half_t, true>(q, 1, st) : launch_T<float, true>(q, 1, st);
            if (rc != EOD_OK) return rc;
        }
        return EOD_OK;
    }
    return d->dtype == EOD_F16 ? launch_T<half_t, true>(p, 1, st) : launch_T<float, true>(p, 1, st);
}

extern "C" int eod_gemm_nt(const eod_gemm_desc* d, void* stream) {
    EOD_REQUIRE(d, "gemm: null desc");
    EOD_REQUIRE(d->dtype == EOD_F32 || d->dtype == EOD_F16, "gemm: bad dtype %d", d->dtype);
    const int es = eod_esize(d->dtype), epc = 16 / es;
    EOD_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->nb0 > 0 && d->nb1 > 0, "gemm: bad dims");
    EOD_REQUIRE(d->K % epc == 0, "gemm: K=%d must be a multiple of %d", d->K, epc);
    EOD_REQUIRE(d->lda % epc == 0 && d->ldb % epc == 0, "gemm: lda/ldb must be multiples of %d", epc);
    EOD_REQUIRE(d->sa0 % epc == 0 && d->sa1 % epc == 0 && d->sb0 % epc == 0 && d->sb1 % epc == 0, "gemm: batch strides of a/b must be multiples of %d", epc);
    EOD_REQUIRE(d->a && d->b && d->c, "gemm: null pointer");
    EOD_REQUIRE(eod_aligned16(d->a) && eod_aligned16(d->b), "gemm: a/b must be 16-byte aligned");
    EOD_REQUIRE((long long)d->nb0 * d->nb1 <= 65535, "gemm: batch too large");
    EOD_REQUIRE(128 * d->lda * es + (long long)d->K * es < 0x7fffffffLL && 128 * d->ldb * es + (long long)d->K * es < 0x7fffffffLL,
                "gemm: leading dimension too large for the 2 GiB tile window");
    IgemmP p = {};
    p.tapmajor_log2 = -1;
    p.a0 = (const char*)d->a;
    p.b = (const char*)d->b;
    p.bias = d->bias;
    p.bias_mode = d->bias ? d->bias_mode : 0;
    EOD_REQUIRE(d->bias_mode >= 0 && d->bias_mode <= 4 && (d->bias_mode != 3 || (d->bias && d->res)) && (d->bias_mode != 4 || (d->bias && !d->res)),
                "gemm: bias_mode %d (mode 3 needs bias and res, mode 4 bias and no res)", d->bias_mode);
    p.res = (const char*)d->res;
    p.y = (char*)d->c;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
    p.sa0 = d->sa0; p.sa1 = d->sa1; p.sb0 = d->sb0; p.sb1 = d->sb1; p.sc0 = d->sc0; p.sc1 = d->sc1;
    p.nb1 = d->nb1;
    p.c_f32 = d->c_f32;
    p.M = d->M; p.Ncols = d->N; p.K = d->K;
    p.taps = 1;
    p.alpha = d->alpha;
    hipStream_t st = (hipStream_t)stream;
    const int batch = d->nb0 * d->nb1;
    if (d->x3) {  // fp32 operands, every product as three fp16 MFMAs on operands split in LDS (the fp32x3 mode's attention GEMMs)
        EOD_REQUIRE(d->dtype == EOD_F32 && d->K % 8 == 0, "gemm: x3 needs fp32 operands and K %% 8 == 0");
        p.a_bound = d->a_bound;
        p.b_bound = d->b_bound;
        if (p.Ncols <= 32) return launch_cfg<float, false, 128, 32, 4, 1, 2, true, 16>(p, batch, st);
        if (p.Ncols <= 64) return launch_cfg<float, false, 128, 64, 4, 1, 2, true, 16>(p, batch, st);
        return launch_cfg<float, false, 128, 128, 2, 2, 2, true, 16>(p, batch, st);
    }
    return d->dtype == EOD_F16 ? launch_T<half_t, false>(p, batch, st) : launch_T<float, false>(p, batch, st);
}
