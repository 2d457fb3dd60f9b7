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
//    Both consume the SAME 16-byte-chunk LDS image: lane (r = lane&31, h = lane>>5) reads chunk 2s+h of
//    row r for k-substep s, so the kernel body is byte-oriented and the dtype only appears in mma().
//  * global -> registers -> LDS staging, double-buffered, one barrier per K-step; next tile's global
//    loads are issued before the MFMAs of the current one (guide T14).  LDS rows are padded by 16 B
//    (row stride 80 / 144 B), which makes every ds_read_b128 lane group bank-conflict-free.
//  * K loop order is channel-chunk OUTER, filter tap INNER: the 9 taps of one channel chunk re-read the
//    same (BM + halo) x BK input bytes, which stay in the CU's 32 KiB L1.
//  * M tiles are TH x TW pixel patches when the feature map allows it (halo reuse in L1), linear runs
//    of BM pixels otherwise (ragged shapes: 28x28, 7x7, 3x3...).
//  * workgroup -> tile mapping is XCD-aware (bijective remap, guide T1): the N-tiles of one M-tile and
//    neighbouring M-tiles land on the same XCD's L2.
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
    float alpha;
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

template <typename T> __device__ __forceinline__ float ld_elem(const char* p, long long idx) {
    return (float)reinterpret_cast<const T*>(p)[idx];
}

template <typename T, bool CONV, int BM, int BN, int BKB, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmP p) {
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;      // elements per 16-byte chunk
    constexpr int BK = BKB / ES;      // elements per K-step
    constexpr int CH = BKB / 16;      // chunks per row per K-step
    constexpr int ROWB = BKB + 16;    // padded LDS row stride
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int LA = BM * CH / 256, LB = (BN * CH + 255) / 256;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    static_assert(BM * CH % 256 == 0, "A staging");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + 2 * BM * ROWB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- XCD-aware tile mapping (bijective) ----
    int tile_m, tile_n;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        tile_n = swz % p.tiles_n;
        tile_m = swz / p.tiles_n;
    }
    const int n0 = tile_n * BN;

    // gemm-mode batch offsets (elements)
    long long offA = 0, offB = 0, offC = 0;
    if constexpr (!CONV) {
        const int z = blockIdx.y;
        const int b0 = z / p.nb1, b1 = z % p.nb1;
        offA = b0 * p.sa0 + b1 * p.sa1;
        offB = b0 * p.sb0 + b1 * p.sb1;
        offC = b0 * p.sc0 + b1 * p.sc1;
    }

    auto decode_row = [&](int r, int& n, int& ho, int& wo, long long& m) -> bool {
        if (p.tw_log2 >= 0) {
            n = tile_m / p.tiles_pi;
            const int t = tile_m - n * p.tiles_pi;
            const int ty = t / p.tiles_pw, tx = t - ty * p.tiles_pw;
            ho = ty * p.th + (r >> p.tw_log2);
            wo = (tx << p.tw_log2) + (r & ((1 << p.tw_log2) - 1));
            m = ((long long)n * p.Ho + ho) * p.Wo + wo;
            return n < p.N;
        }
        m = (long long)tile_m * BM + r;
        n = (int)(m / p.HoWo);
        const int rem = (int)(m - (long long)n * p.HoWo);
        ho = rem / p.Wo;
        wo = rem - ho * p.Wo;
        return m < p.M;
    };

    // ---- per-thread staging slots ----
    int a_n[LA], a_bh[LA], a_bw[LA];
    bool a_ok[LA];
    long long a_row[LA];  // gemm: row element offset
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int c = tid + i * 256, row = c / CH;
        if constexpr (CONV) {
            int n, ho, wo;
            long long m;
            a_ok[i] = decode_row(row, n, ho, wo, m);
            a_n[i] = n;
            a_bh[i] = ho * p.stride - p.pad - p.pad_tl;
            a_bw[i] = wo * p.stride - p.pad - p.pad_tl;
        } else {
            const long long m = (long long)tile_m * BM + row;
            a_ok[i] = m < p.M;
            a_row[i] = offA + m * p.lda;
        }
    }
    bool b_ok[LB];
    int b_co[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int c = tid + i * 256, row = c / CH;
        b_co[i] = n0 + row;
        b_ok[i] = (row < BN) && (b_co[i] < p.Ncols);
    }

    i32x4 ra[LA], rb[LB];
    const i32x4 zero4 = {0, 0, 0, 0};

    auto load_tile = [&](int kt) {
        int tap = 0, c0;
        if constexpr (CONV) {
            const int cc = kt / p.taps;
            tap = kt - cc * p.taps;
            c0 = cc * BK;
        } else {
            c0 = kt * BK;
        }
        int dy = 0, dx = 0;
        if constexpr (CONV) {
            if (p.KS == 3) {
                dy = tap / 3;
                dx = tap - dy * 3;
            }
        }
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int c = tid + i * 256, ch = c % CH;
            const int k = c0 + ch * EPC;
            // always issue the load (from a safe address when masked) and select afterwards: a
            // branch around each load would serialise them behind per-load vmcnt(0) waits
            const char* src = p.a0;
            bool ok;
            if constexpr (CONV) {
                int hi = a_bh[i] + dy, wi = a_bw[i] + dx;
                ok = a_ok[i] && (k < p.Cin) && ((unsigned)hi < (unsigned)p.Heff) && ((unsigned)wi < (unsigned)p.Weff);
                if (p.ups) {
                    hi >>= 1;
                    wi >>= 1;
                }
                const long long pix = ((long long)a_n[i] * p.H + hi) * p.W + wi;
                if (ok) src = (k < p.C0) ? p.a0 + (pix * p.C0 + k) * ES : p.a1 + (pix * p.C1 + (k - p.C0)) * ES;
            } else {
                ok = a_ok[i] && k < p.K;
                if (ok) src = p.a0 + (a_row[i] + k) * ES;
            }
            const i32x4 v = *reinterpret_cast<const i32x4*>(src);
            ra[i] = ok ? v : zero4;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int c = tid + i * 256, ch = c % CH;
            const int k = c0 + ch * EPC;
            const char* src = p.b;
            bool ok;
            if constexpr (CONV) {
                ok = b_ok[i] && k < p.Cin;
                if (ok) src = p.b + (((long long)tap * p.Cout + b_co[i]) * p.Cin + k) * ES;
            } else {
                ok = b_ok[i] && k < p.K;
                if (ok) src = p.b + (offB + (long long)b_co[i] * p.ldb + k) * ES;
            }
            const i32x4 v = *reinterpret_cast<const i32x4*>(src);
            rb[i] = ok ? v : zero4;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int c = tid + i * 256, row = c / CH, ch = c % CH;
            *reinterpret_cast<i32x4*>(sA + buf * BM * ROWB + row * ROWB + ch * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int c = tid + i * 256, row = c / CH, ch = c % CH;
            if (row < BN) *reinterpret_cast<i32x4*>(sB + buf * BN * ROWB + row * ROWB + ch * 16) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int KT = p.KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) load_tile(kt + 1);
        const char* bA = sA + buf * BM * ROWB + (wm * WM + lr) * ROWB + lh * 16;
        const char* bB = sB + buf * BN * ROWB + (wn * WN + lr) * ROWB + lh * 16;
#pragma unroll
        for (int s = 0; s < BKB / 32; ++s) {
            i32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const i32x4*>(bA + i * 32 * ROWB + s * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const i32x4*>(bB + j * 32 * ROWB + s * 32);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
        if (kt + 1 < KT) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane = output column, registers = 16 rows of the 32x32 tile ----
    int col[TN];
    bool cok[TN];
    float bcol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        col[j] = n0 + wn * WN + j * 32 + lr;
        cok[j] = col[j] < p.Ncols;
        bcol[j] = (cok[j] && p.bias && p.bias_mode == 1) ? p.bias[col[j]] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if constexpr (CONV) {
                int n, ho, wo;
                long long m;
                if (!decode_row(row, n, ho, wo, m)) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!cok[j]) continue;
                    float v = acc[i][j][r] * p.alpha + bcol[j];
                    if (p.cbias) v += p.cbias[(long long)n * p.cbias_stride + col[j]];
                    if (p.out_nchw) {
                        reinterpret_cast<float*>(p.y)[((long long)n * p.Cout + col[j]) * p.HoWo + (long long)ho * p.Wo + wo] = v;
                    } else {
                        const long long o = m * p.Cout + col[j];
                        if (p.res) v += ld_elem<T>(p.res, o);
                        reinterpret_cast<T*>(p.y)[o] = (T)v;
                    }
                }
            } else {
                const long long m = (long long)tile_m * BM + row;
                if (m >= p.M) continue;
                const float brow = (p.bias && p.bias_mode == 2) ? p.bias[m] : 0.0f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!cok[j]) continue;
                    float v = acc[i][j][r] * p.alpha + bcol[j] + brow;
                    const long long o = offC + m * p.ldc + col[j];
                    if (p.res) v += p.c_f32 ? reinterpret_cast<const float*>(p.res)[o] : ld_elem<T>(p.res, o);
                    if (p.c_f32)
                        reinterpret_cast<float*>(p.y)[o] = v;
                    else
                        reinterpret_cast<T*>(p.y)[o] = (T)v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
template <typename T, bool CONV, int BM, int BN, int BKB, int WAVES_M, int WAVES_N>
static int launch_cfg(IgemmP& p, int batch, hipStream_t st) {
    constexpr int ROWB = BKB + 16;
    constexpr int BK = BKB / (int)sizeof(T);
    const size_t lds = 2 * (size_t)(BM + BN) * ROWB;
    auto kern = igemm_kernel<T, CONV, BM, BN, BKB, WAVES_M, WAVES_N>;
    static bool attr_done = false;  // >64 KiB dynamic LDS needs the opt-in attribute once per kernel
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int kdim = CONV ? p.Cin : p.K;
    const int kc = (kdim + BK - 1) / BK;
    p.KT = kc * p.taps;
    p.tiles_n = (p.Ncols + BN - 1) / BN;
    if (CONV) {
        // patch mode: TW = min(16, Wo) if it is a power of two dividing Wo and TH = BM/TW divides Ho
        p.tw_log2 = -1;
        int tw = 16;
        while (tw > p.Wo) tw >>= 1;
        if (tw >= 4 && p.Wo % tw == 0 && (BM % tw) == 0 && p.Ho % (BM / tw) == 0) {
            int l = 0;
            while ((1 << l) < tw) ++l;
            p.tw_log2 = l;
            p.th = BM / tw;
            p.tiles_pw = p.Wo / tw;
            p.tiles_pi = p.tiles_pw * (p.Ho / p.th);
            p.tiles_m = p.tiles_pi * p.N;
        } else {
            p.tiles_m = (int)((p.M + BM - 1) / BM);
        }
    } else {
        p.tw_log2 = -1;
        p.tiles_m = (int)((p.M + BM - 1) / BM);
    }
    const long long nblk = (long long)p.tiles_m * p.tiles_n;
    if (nblk <= 0 || nblk > 0x7fffffffLL) {
        eod_set_error("igemm: bad grid %lld", nblk);
        return EOD_EINVAL;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)batch), dim3(256), lds, st, p);
    EOD_CHECK_LAUNCH("igemm");
    return EOD_OK;
}

template <typename T, bool CONV> static int launch_T(IgemmP& p, int batch, hipStream_t st) {
    const int kdim_bytes = (CONV ? p.Cin : p.K) * (int)sizeof(T);
    const bool small_k = kdim_bytes <= 64;
    if (p.Ncols <= 32) {
        return small_k ? launch_cfg<T, CONV, 128, 32, 64, 4, 1>(p, batch, st) : launch_cfg<T, CONV, 128, 32, 128, 4, 1>(p, batch, st);
    } else if (p.Ncols <= 64) {
        return small_k ? launch_cfg<T, CONV, 128, 64, 64, 4, 1>(p, batch, st) : launch_cfg<T, CONV, 128, 64, 128, 4, 1>(p, batch, st);
    }
    return small_k ? launch_cfg<T, CONV, 128, 128, 64, 2, 2>(p, batch, st) : launch_cfg<T, CONV, 128, 128, 128, 2, 2>(p, batch, st);
}

extern "C" int eod_conv2d_igemm(const eod_conv_desc* d, void* stream) {
    EOD_REQUIRE(d, "conv: null desc");
    EOD_REQUIRE(d->dtype == EOD_F32 || d->dtype == EOD_F16, "conv: bad dtype %d", d->dtype);
    const int es = eod_esize(d->dtype), epc = 16 / es;
    EOD_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv: ksize %d", d->ksize);
    EOD_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d", d->stride);
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
    p.M = (long long)d->N * Ho * Wo;
    p.Ncols = d->Cout;
    p.taps = d->ksize * d->ksize;
    p.out_nchw = d->out_nchw_f32;
    p.alpha = d->alpha;
    p.nb1 = 1;
    hipStream_t st = (hipStream_t)stream;
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
    IgemmP p = {};
    p.a0 = (const char*)d->a;
    p.b = (const char*)d->b;
    p.bias = d->bias;
    p.bias_mode = d->bias ? d->bias_mode : 0;
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
    return d->dtype == EOD_F16 ? launch_T<half_t, false>(p, batch, st) : launch_T<float, false>(p, batch, st);
}
