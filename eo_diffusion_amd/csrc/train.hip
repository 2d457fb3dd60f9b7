// Training-path kernels (train.py:109-124 = forward, MSE, backward, AdamW, EMA).
//
// Backward of the UNet blocks is built from the SAME MFMA kernels as the forward:
//   * conv backward-data  = eod_conv2d_igemm on dY with flipped / transposed weights (eod_pack_conv_weight_dgrad);
//     stride-2 convs use the zero-insertion input mode (eod_conv_desc.upsample = 2), convs on a nearest-2x input are
//     followed by a 2x2 sum pool (eod_resample2x mode 2);
//   * conv backward-weights = eod_gemm_nt over the PIXEL axis: dW[tap][co][ci] = sum_p dY[p][co] * X[p + tap][ci].
//     Both operands are needed K(=pixel)-contiguous, so dY and X are first transposed to [channel][pixel]
//     (eod_transpose_gather).  For 3x3 / stride-1 convs only three copies of X exist (dx = -1, 0, +1): every image gets
//     a zero row above and below, so the dy = -1 / +1 taps are the SAME buffer read at a +-W element offset (16-byte
//     aligned when W % 8 == 0) and the conv's zero padding needs no masks.  The K axis is split over gridDim batches
//     (fp32 partial tiles), eod_wgrad_reduce sums them in a fixed order into the OIHW fp32 gradient;
//   * GroupNorm(+SiLU) backward: per-channel partial sums (eod_gn_bwd_partial) -> per-group coefficients
//     (eod_gn_bwd_finalize, double accumulation) -> dx = k1*dz + k2*x + k3 (eod_gn_bwd_apply), dgamma / dbeta from the
//     same partial sums.
// Everything here is HBM-bound elementwise / reduction work: 16-byte accesses, fixed-order sums, no atomics.
#include "common.h"

template <typename T> __device__ __forceinline__ T cvt_to(float v) { return (T)v; }
__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------
// weights for backward-data: OIHW fp32 -> [tap'][ci - ci0][cout_pad], tap' = taps-1-tap (180-degree flip), i.e. the
// packed weights of the conv that maps dY (Cout channels) to dX (the nci input channels starting at ci0)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_conv_w_dgrad_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int taps, int ci0,
                                         int nci, int cout_pad) {
    const long long total = (long long)taps * nci * cout_pad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout_pad);
        const long long r = i / cout_pad;
        const int ci = (int)(r % nci);
        const int tp = (int)(r / nci);
        const int tap = taps - 1 - tp;
        const float v = co < Cout ? w[((long long)co * Cin + ci0 + ci) * taps + tap] : 0.0f;
        dst[i] = cvt_to<T>(v);
    }
}

extern "C" int eod_pack_conv_weight_dgrad(const float* w, void* dst, int dtype, int Cout, int Cin, int ksize, int ci0, int nci,
                                          int cout_pad, void* stream) {
    EOD_REQUIRE(w && dst && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3) && ci0 >= 0 && nci > 0 && ci0 + nci <= Cin && cout_pad >= Cout,
                "pack_conv_weight_dgrad: bad args");
    EOD_REQUIRE(dtype == EOD_F16 || dtype == EOD_F32, "pack_conv_weight_dgrad: bad dtype %d", dtype);
    const int taps = ksize * ksize;
    const long long total = (long long)taps * nci * cout_pad;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(pack_conv_w_dgrad_kernel<half_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (half_t*)dst, Cout, Cin, taps, ci0, nci, cout_pad);
    else
        hipLaunchKernelGGL(pack_conv_w_dgrad_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)dst, Cout, Cin, taps, ci0, nci, cout_pad);
    EOD_CHECK_LAUNCH("pack_conv_weight_dgrad");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// All weight re-packs of a training step in ONE launch.  The optimizer changes every parameter every step, so the forward
// ([tap][Cout][cin_pad]) and backward-data ([taps-1-tap][ci][cout_pad]) packings of every conv are refreshed before each
// forward: ~200 tiny launches (1.5 ms at A0) become one.  The host cuts each job into blocks of EOD_PACK_CHUNK elements once
// (blk_job[b] = job of block b, blk_first[b] = its first destination element).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_jobs_kernel(const eod_pack_job* __restrict__ jobs, const int* __restrict__ blk_job,
                                                        const long long* __restrict__ blk_first) {
    const eod_pack_job j = jobs[blk_job[blockIdx.x]];
    const long long total = (long long)j.taps * (j.kind ? (long long)j.nci * j.cpad : (long long)j.Cout * j.cpad);
    const long long i0 = blk_first[blockIdx.x];
    const float* __restrict__ w = j.w;
    T* __restrict__ dst = reinterpret_cast<T*>(j.dst);
    // (32-bit index arithmetic: a job has < 2^31 destination elements -- checked on the host side of the conv kernels -- and 64-bit
    //  divisions per element would dominate this kernel)
    const unsigned cpad = (unsigned)j.cpad, rows = (unsigned)(j.kind ? j.nci : j.Cout), utotal = (unsigned)total;
#pragma unroll 4
    for (int k = 0; k < EOD_PACK_CHUNK / 256; ++k) {
        const unsigned i = (unsigned)i0 + k * 256 + threadIdx.x;
        if (i >= utotal) break;
        const unsigned r = i / cpad, c = i - r * cpad;   // c: fastest destination index (ci for kind 0, co for kind 1)
        const unsigned t = r / rows, m = r - t * rows;   // m: co (kind 0) / ci (kind 1); t: destination tap slot
        float v;
        if (j.kind == 0) {
            v = (int)c < j.Cin ? w[((long long)m * j.Cin + c) * j.taps + t] : 0.0f;
        } else {
            v = (int)c < j.Cout ? w[((long long)c * j.Cin + j.ci0 + m) * j.taps + (j.taps - 1 - t)] : 0.0f;
        }
        dst[i] = cvt_to<T>(v);
    }
}

extern "C" int eod_pack_jobs(const eod_pack_job* jobs, const int32_t* blk_job, const int64_t* blk_first, int nblocks, int dtype, void* stream) {
    EOD_REQUIRE(jobs && blk_job && blk_first && nblocks > 0, "pack_jobs: bad args");
    EOD_REQUIRE(dtype == EOD_F16 || dtype == EOD_F32, "pack_jobs: bad dtype %d", dtype);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(pack_jobs_kernel<half_t>, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, jobs, blk_job, (const long long*)blk_first);
    else
        hipLaunchKernelGGL(pack_jobs_kernel<float>, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, jobs, blk_job, (const long long*)blk_first);
    EOD_CHECK_LAUNCH("pack_jobs");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// transpose + gather: NHWC [N][H][W][C] -> [C][K],  K = N * (Ho + 2*rp) * Wo (+ nothing else), element
//   dst[c][(n*(Ho+2rp) + ho + rp)*Wo + wo] = src[n][hi][wi][c],  (hi, wi) = (ho*stride - pad + dy, wo*stride - pad + dx),
//   with ups: (hi >> 1, wi >> 1) of the stored half-resolution tensor; zero outside the image and in the rp pad rows.
// 64 x 64 tiles through LDS: 128-byte (fp16) reads along the channels, 128-byte writes along the pixels.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_gather_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int H, int W, int C,
                                                               int Ho, int Wo, int stride, int pad, int dy, int dx, int ups, int rp,
                                                               long long K, long long ld_dst) {
    // (columns K .. ld_dst-1 of every row are written as zeros: the K-split GEMM reads whole K-steps)
    constexpr int EPC = dt<T>::epc;          // elements per 16-byte chunk
    constexpr int TPR = 64 / EPC;            // threads per pixel row of the tile (64 channels)
    constexpr int PPP = 256 / TPR;           // pixels per load pass
    constexpr int LDT = 64 + EPC;            // tile row stride (elements): rows stay 16-byte aligned
    __shared__ __attribute__((aligned(16))) T tile[64 * LDT];
    const long long k0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int Hp = Ho + 2 * rp;
    const int Heff = ups ? 2 * H : H, Weff = ups ? 2 * W : W;
    // load: 16 bytes (EPC channels) per thread
    const int q = tid % TPR;
#pragma unroll
    for (int ps = 0; ps < 64 / PPP; ++ps) {
        const int r = ps * PPP + tid / TPR;
        const long long k = k0 + r;
        i32x4 v = {0, 0, 0, 0};
        const int c = c0 + q * EPC;
        if (k < K && c < C) {
            const int wo = (int)(k % Wo);
            const long long qq = k / Wo;
            const int hp = (int)(qq % Hp), n = (int)(qq / Hp);
            const int ho = hp - rp;
            int hi = ho * stride - pad + dy, wi = wo * stride - pad + dx;
            if (ho >= 0 && ho < Ho && (unsigned)hi < (unsigned)Heff && (unsigned)wi < (unsigned)Weff) {
                if (ups) {
                    hi >>= 1;
                    wi >>= 1;
                }
                v = *reinterpret_cast<const i32x4*>(src + (((long long)n * H + hi) * W + wi) * C + c);  // C % EPC == 0
            }
        }
        *reinterpret_cast<i32x4*>(&tile[r * LDT + q * EPC]) = v;
    }
    __syncthreads();
    // store: EPC consecutive pixels (16 bytes) of one channel per thread; lanes of a wave = 64 consecutive channels
    const int ch = tid & 63;
#pragma unroll
    for (int ps = 0; ps < 64 / (4 * EPC); ++ps) {
        const int px0 = (ps * 4 + (tid >> 6)) * EPC;
        T o[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = tile[(px0 + e) * LDT + ch];
        const long long k = k0 + px0;
        if (c0 + ch < C && k < ld_dst) *reinterpret_cast<i32x4*>(dst + (long long)(c0 + ch) * ld_dst + k) = *reinterpret_cast<const i32x4*>(o);
    }
}

extern "C" int eod_transpose_gather(const void* src, int dtype, int N, int H, int W, int C, void* dst, int64_t ld_dst, int Ho, int Wo,
                                    int stride, int pad, int dy, int dx, int ups, int row_pad, void* stream) {
    EOD_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && (stride == 1 || stride == 2) && row_pad >= 0,
                "transpose_gather: bad args");
    EOD_REQUIRE(dtype == EOD_F16 || dtype == EOD_F32, "transpose_gather: bad dtype %d", dtype);
    const long long K = (long long)N * (Ho + 2 * row_pad) * Wo;
    EOD_REQUIRE(ld_dst >= K, "transpose_gather: ld_dst %lld < K %lld", (long long)ld_dst, K);
    EOD_REQUIRE(C % (16 / eod_esize(dtype)) == 0 && ld_dst % (16 / eod_esize(dtype)) == 0 && eod_aligned16(src) && eod_aligned16(dst),
                "transpose_gather: C and ld_dst must be multiples of one 16-byte chunk, pointers 16-byte aligned");
    const long long kb = (ld_dst + 63) / 64;
    EOD_REQUIRE(kb <= 0x7fffffffLL && (C + 63) / 64 <= 65535, "transpose_gather: grid too large");
    dim3 grid((unsigned)kb, (unsigned)((C + 63) / 64));
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(transpose_gather_kernel<half_t>, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)src, (half_t*)dst, N, H, W, C, Ho, Wo, stride, pad, dy, dx, ups, row_pad, K, (long long)ld_dst);
    else
        hipLaunchKernelGGL(transpose_gather_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, (float*)dst, N, H, W, C, Ho, Wo, stride, pad, dy, dx, ups, row_pad, K, (long long)ld_dst);
    EOD_CHECK_LAUNCH("transpose_gather");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// segmented row sums of a [C][ld] matrix: seg[s][c] = sum_{k in segment s} x[c][k], segments of seg_len elements
// (bias gradient: one segment = everything; timestep-embedding gradient: one segment per image).  grid (C, nseg).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rowsum_kernel(const T* __restrict__ x, long long ld, long long seg_len, float* __restrict__ seg,
                                                     long long seg_ld, float scale) {
    __shared__ float red[256];
    const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const T* row = x + (long long)c * ld + (long long)s * seg_len;
    float a = 0.0f;
    for (long long k = tid; k < seg_len; k += 256) a += (float)row[k];
    red[tid] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) seg[(long long)s * seg_ld + c] = red[0] * scale;
}

extern "C" int eod_rowsum_segments(const void* x, int dtype, int C, int64_t ld, int nseg, int64_t seg_len, float scale, float* seg,
                                   int64_t seg_ld, void* stream) {
    EOD_REQUIRE(x && seg && C > 0 && nseg > 0 && seg_len > 0 && nseg <= 65535 && ld >= (int64_t)nseg * seg_len && seg_ld >= C, "rowsum_segments: bad args");
    dim3 grid((unsigned)C, (unsigned)nseg);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(rowsum_kernel<half_t>, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (long long)ld, (long long)seg_len, seg, (long long)seg_ld, scale);
    else
        hipLaunchKernelGGL(rowsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (long long)ld, (long long)seg_len, seg, (long long)seg_ld, scale);
    EOD_CHECK_LAUNCH("rowsum_segments");
    return EOD_OK;
}

// out[c] = sum_s seg[s][c] (fixed order)
__global__ void colsum_kernel(const float* __restrict__ seg, int S, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.0f;
    for (int s = 0; s < S; ++s) a += seg[(long long)s * C + c];
    out[c] = a;
}

extern "C" int eod_colsum(const float* seg, int S, int C, float* out, void* stream) {
    EOD_REQUIRE(seg && out && S > 0 && C > 0, "colsum: bad args");
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, seg, S, C, out);
    EOD_CHECK_LAUNCH("colsum");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// weight gradient, second pass: dW_oihw[co][ci0 + ci][tap] = scale * sum_s partial[s][tap][co][ci]  (ci < nci <= ldp)
// ---------------------------------------------------------------------------------------------
// One thread = one (co, 4 consecutive ci) quad of ALL taps for a quarter of the splits: 16-byte loads along ci, TAPS independent load
// streams per thread, and the OIHW rows come out as 4 x TAPS contiguous floats.  The four split groups of a quad are combined
// through LDS in a fixed order (deterministic).
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int S, int Cout, int nci, int ldp, int ci0, int Cin,
                                                          float scale, float* __restrict__ dw) {
    __shared__ f32x4 red[3][64][TAPS];
    const int ql = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int nq = (nci + 3) >> 2;
    const long long quad = (long long)blockIdx.x * 64 + ql;
    const bool valid = quad < (long long)Cout * nq;
    const int co = valid ? (int)(quad / nq) : 0;
    const int ci = valid ? (int)(quad - (long long)co * nq) * 4 : 0;
    const int per = (S + 3) >> 2, s0 = grp * per, s1 = min(S, s0 + per);
    const long long tap_stride = (long long)Cout * ldp, s_stride = TAPS * tap_stride;
    const float* base = part + (long long)co * ldp + ci;
    f32x4 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (valid) {
        for (int sp = s0; sp < s1; ++sp) {
            f32x4 v[TAPS];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) v[t] = *reinterpret_cast<const f32x4*>(base + sp * s_stride + t * tap_stride);
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[t] += v[t];
        }
    }
    if (grp > 0) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) red[grp - 1][ql][t] = acc[t];
    }
    __syncthreads();
    if (grp == 0 && valid) {
#pragma unroll
        for (int gq = 0; gq < 3; ++gq)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[t] += red[gq][ql][t];
        float* out = dw + ((long long)co * Cin + ci0 + ci) * TAPS;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ci + e < nci) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t) out[e * TAPS + t] = acc[t][e] * scale;
            }
    }
}

extern "C" int eod_wgrad_reduce(const float* partial, int S, int ksize, int Cout, int nci, int ldp, int ci0, int Cin, float scale,
                                float* dw_oihw, void* stream) {
    EOD_REQUIRE(partial && dw_oihw && S > 0 && (ksize == 1 || ksize == 3 || ksize == 4) && Cout > 0 && nci > 0 && ldp >= nci && ci0 >= 0 && ci0 + nci <= Cin,
                "wgrad_reduce: bad args");  // (ksize 4: the 16 tap planes of the parity-class form, folded by eod_wgrad_up4_map)
    EOD_REQUIRE(ldp % 4 == 0 && eod_aligned16(partial), "wgrad_reduce: the partial tiles need 16-byte rows (ldp %% 4 == 0)");
    const long long quads = (long long)Cout * ((nci + 3) / 4);
    const long long blocks = (quads + 63) / 64;
    EOD_REQUIRE(blocks <= 0x7fffffffLL, "wgrad_reduce: grid too large");
    if (ksize == 4)
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partial, S, Cout, nci, ldp, ci0, Cin, scale, dw_oihw);
    else if (ksize == 3)
        hipLaunchKernelGGL(wgrad_reduce_kernel<9>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partial, S, Cout, nci, ldp, ci0, Cin, scale, dw_oihw);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partial, S, Cout, nci, ldp, ci0, Cin, scale, dw_oihw);
    EOD_CHECK_LAUNCH("wgrad_reduce");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics as (mean, rstd) per (image, group), from the forward's per-channel partial sums
// (same inputs and arithmetic as eod_gn_finalize: double accumulation)
// ---------------------------------------------------------------------------------------------
__global__ void gn_mean_rstd_kernel(const float* __restrict__ part0, int P0, int C0, const float* __restrict__ part1, int P1, int C1,
                                    long long HW, int groups, float eps, float* __restrict__ mr) {
    __shared__ double rs[256], rq[256];
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int Ctot = C0 + C1, cpg = Ctot / groups, c0 = g * cpg;
    double s = 0.0, q = 0.0;
    const int a0 = min(c0, C0), a1 = min(c0 + cpg, C0);
    const int n0c = a1 - a0, n1c = cpg - n0c;
    // {sum, sumsq} pairs as one 8-byte load, four independent loads in flight per thread (the loop of eod_gn_finalize, norm.hip: a
    // 512 x 512 map at batch 2 hands over 2048 slabs per image)
    auto accumulate = [&](const float* __restrict__ part, int P, int Cs, int first, int nc) {
        const float2* base = reinterpret_cast<const float2*>(part) + (long long)n * P * Cs;
        const int total = P * nc;
        int i = tid;
        for (; i + 768 < total; i += 1024) {
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = i + 256 * u, p = k / nc;
                v[u] = base[(long long)p * Cs + first + (k - p * nc)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s += (double)v[u].x;
                q += (double)v[u].y;
            }
        }
        for (; i < total; i += 256) {
            const int p = i / nc;
            const float2 v = base[(long long)p * Cs + first + (i - p * nc)];
            s += (double)v.x;
            q += (double)v.y;
        }
    };
    if (n0c > 0) accumulate(part0, P0, C0, a0, n0c);
    if (n1c > 0) accumulate(part1, P1, C1, max(c0, C0) - C0, n1c);
    rs[tid] = s;
    rq[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            rs[tid] += rs[tid + o];
            rq[tid] += rq[tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double cnt = (double)HW * cpg;
        const double mean = rs[0] / cnt;
        double var = rq[0] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        mr[((long long)n * groups + g) * 2 + 0] = (float)mean;
        mr[((long long)n * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

extern "C" int eod_gn_mean_rstd(const float* part0, int P0, int C0, const float* part1, int P1, int C1, int N, int64_t HW, int groups,
                                float eps, float* mean_rstd, void* stream) {
    EOD_REQUIRE(part0 && mean_rstd && N > 0 && P0 > 0 && C0 > 0 && C1 >= 0 && (C1 == 0 || (part1 && P1 > 0)) && groups > 0 &&
                    (C0 + C1) % groups == 0,
                "gn_mean_rstd: bad args");
    hipLaunchKernelGGL(gn_mean_rstd_kernel, dim3(groups, N), dim3(256), 0, (hipStream_t)stream, part0, P0, C0, part1, P1, C1, (long long)HW, groups, eps, mean_rstd);
    EOD_CHECK_LAUNCH("gn_mean_rstd");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// GroupNorm(+SiLU) backward.  Forward: z = (x - mean)*rstd*gamma + beta, y = silu(z) (or z).
//   dz = dy * silu'(z);  A[n][c] = sum_hw dz,  B[n][c] = sum_hw dz * x   (raw x: no statistics needed in this pass)
// partial: same grid / layout as gn_partial: part[n][p][coff + c][0..1] = (A, B) over the pixels of slab p.
// x is one concat source (C channels at channel offset coff of the Ctot-wide dy / scale-shift tables).
// ---------------------------------------------------------------------------------------------
template <bool FAST> __device__ __forceinline__ float dsilu_f(float z) {
    float sg;
    if constexpr (FAST)
        sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
    else
        sg = 1.0f / (1.0f + expf(-z));
    return sg * (1.0f + z * (1.0f - sg));
}

template <typename T>
__global__ void gn_bwd_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ ss, int HW, int C,
                                      float* __restrict__ part, int P, int Ctot, int coff, int silu) {
    constexpr int EPC = dt<T>::epc;
    constexpr bool FAST = (EPC == 8);
    extern __shared__ float red[];  // [RY][CPP*EPC*2]
    const int CPP = blockDim.x, RY = blockDim.y;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int p = blockIdx.x, n = blockIdx.y;
    const int col0 = blockIdx.z * CPP;          // first chunk column of this channel block (wide layers: several blocks along z)
    const bool col_ok = col0 + tx < C / EPC;
    const int colc = col_ok ? col0 + tx : 0;
    const int per = (HW + P - 1) / P;
    const int p0 = p * per, p1 = col_ok ? min(HW, p0 + per) : 0;
    float sa[EPC], sb[EPC], sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sa[e] = sb[e] = 0.0f;
        sc[e] = ss[((long long)n * Ctot + coff + colc * EPC + e) * 2 + 0];
        sh[e] = ss[((long long)n * Ctot + coff + colc * EPC + e) * 2 + 1];
    }
    const T* xb = x + (long long)n * HW * C + (long long)colc * EPC;
    const T* db = dy + (long long)n * HW * Ctot + coff + (long long)colc * EPC;
    constexpr int U = 2;  // pixels in flight per thread; the accumulation order stays pixel-ascending per thread
    for (int pix = p0 + ty; pix < p1; pix += U * RY) {
        T xv[U][EPC], dv[U][EPC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pu = pix + u * RY;
            if (pu < p1) {
                *reinterpret_cast<i32x4*>(xv[u]) = *reinterpret_cast<const i32x4*>(xb + pu * C);
                *reinterpret_cast<i32x4*>(dv[u]) = *reinterpret_cast<const i32x4*>(db + pu * Ctot);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pix + u * RY >= p1) break;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float xf = (float)xv[u][e];
                float dz = (float)dv[u][e];
                if (silu) dz *= dsilu_f<FAST>(xf * sc[e] + sh[e]);
                sa[e] += dz;
                sb[e] += dz * xf;
            }
        }
    }
    const int W2 = CPP * EPC * 2;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        red[ty * W2 + (tx * EPC + e) * 2 + 0] = sa[e];
        red[ty * W2 + (tx * EPC + e) * 2 + 1] = sb[e];
    }
    __syncthreads();
    const int tid = ty * CPP + tx, nthr = CPP * RY;
    for (int i = tid; i < W2; i += nthr) {
        float a = 0.0f;
        for (int r = 0; r < RY; ++r) a += red[r * W2 + i];
        const int c = col0 * EPC + (i >> 1);
        if (c < C) part[(((long long)n * P + p) * Ctot + coff + c) * 2 + (i & 1)] = a;
    }
}

extern "C" int eod_gn_bwd_partial(const void* x, const void* dy, const float* scale_shift, int dtype, int N, int HW, int C, float* part,
                                  int P, int Ctot, int coff, int silu, void* stream) {
    EOD_REQUIRE(x && dy && scale_shift && part && N > 0 && HW > 0 && C > 0 && P > 0 && P <= HW, "gn_bwd_partial: bad args");
    const int epc = 16 / eod_esize(dtype);
    EOD_REQUIRE(C % epc == 0 && Ctot % epc == 0 && coff % epc == 0 && N <= 65535, "gn_bwd_partial: C=%d Ctot=%d coff=%d unsupported", C, Ctot, coff);
    EOD_REQUIRE(eod_aligned16(x) && eod_aligned16(dy), "gn_bwd_partial: alignment");
    const GnSlab gs = gn_slab(N, HW, C, epc, 1);  // (channel decomposition only: the slab count P is the caller's)
    const int cpp = gs.cpp, ry = gs.ry;
    const size_t lds = (size_t)ry * cpp * epc * 2 * sizeof(float);
    dim3 grid(P, N, gs.nz), block(cpp, ry);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(gn_bwd_partial_kernel<half_t>, grid, block, lds, (hipStream_t)stream, (const half_t*)x, (const half_t*)dy, scale_shift, HW, C, part, P, Ctot, coff, silu);
    else
        hipLaunchKernelGGL(gn_bwd_partial_kernel<float>, grid, block, lds, (hipStream_t)stream, (const float*)x, (const float*)dy, scale_shift, HW, C, part, P, Ctot, coff, silu);
    EOD_CHECK_LAUNCH("gn_bwd_partial");
    return EOD_OK;
}

// finalize: grid (groups, N).  With xh = (x - mean)*rstd and m = cpg*HW:
//   S1 = sum_{c in g} gamma_c A_c,   S2 = sum_{c in g} gamma_c * rstd * (B_c - mean*A_c)
//   dx = rstd*gamma_c*dz - rstd*S1/m - xh*rstd*S2/m  =  k1_c*dz + k2*x + k3
//   k1_c = rstd*gamma_c,  k2 = -rstd^2*S2/m,  k3 = -rstd*S1/m + mean*rstd^2*S2/m        -> coef[n][c][0..2]
// and the per-image parameter-gradient terms  gb[n][c] = (rstd*(B_c - mean*A_c), A_c)  (dgamma, dbeta = sum over n).
__global__ void gn_bwd_finalize_kernel(const float* __restrict__ part, int P, int Ctot, long long HW, int groups,
                                       const float* __restrict__ mr, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ film, long long film_stride, float* __restrict__ dfilm,
                                       long long dfilm_stride, float* __restrict__ coef, float* __restrict__ gb) {
    // FiLM (use_scale_shift_norm, unet_openai.py:377-381): y = z*(1+s) + t with z = xh*gamma + beta and film[n] = [s | t]:
    // the GroupNorm sees the effective gamma_c*(1+s[n][c]); ds = sum dy*z = gamma*Bh + beta*A, dt = A  -> dfilm[n] = [ds | dt]
    __shared__ double r1[256], r2[256];
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = Ctot / groups, c0 = g * cpg;
    const double mean = (double)mr[((long long)n * groups + g) * 2 + 0], rstd = (double)mr[((long long)n * groups + g) * 2 + 1];
    // per-channel slab sums A_c, B_c: the 256 threads are cpg channels x nseg slab segments (a lone thread per channel walking
    // all P slabs took 45 us per launch); partial sums meet in LDS and are combined in a fixed order
    const int nseg = cpg <= 256 ? 256 / cpg : 1;
    {
        const int cl = tid % cpg, sg = tid / cpg;
        double A = 0.0, B = 0.0;
        if (sg < nseg && cpg <= 256) {
            for (int p = sg; p < P; p += nseg) {
                const float* pp = part + (((long long)n * P + p) * Ctot + c0 + cl) * 2;
                A += (double)pp[0];
                B += (double)pp[1];
            }
        }
        r1[tid] = A;
        r2[tid] = B;
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    for (int c = c0 + tid; c < c0 + cpg; c += 256) {
        double A = 0.0, B = 0.0;
        if (cpg <= 256) {
            for (int k = 0; k < nseg; ++k) {
                A += r1[k * cpg + (c - c0)];
                B += r2[k * cpg + (c - c0)];
            }
        } else {
            for (int p = 0; p < P; ++p) {
                const float* pp = part + (((long long)n * P + p) * Ctot + c) * 2;
                A += (double)pp[0];
                B += (double)pp[1];
            }
        }
        const double bh = rstd * (B - mean * A);  // sum dy * xh
        const double fs = film ? 1.0 + (double)film[(long long)n * film_stride + c] : 1.0;
        const double ge = (double)gamma[c] * fs;
        gb[((long long)n * Ctot + c) * 2 + 0] = (float)(bh * fs);
        gb[((long long)n * Ctot + c) * 2 + 1] = (float)(A * fs);
        if (dfilm) {
            dfilm[(long long)n * dfilm_stride + c] = (float)((double)gamma[c] * bh + (double)beta[c] * A);
            dfilm[(long long)n * dfilm_stride + Ctot + c] = (float)A;
        }
        s1 += ge * A;
        s2 += ge * bh;
    }
    __syncthreads();  // (the slab partials in r1 / r2 have been consumed)
    r1[tid] = s1;
    r2[tid] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            r1[tid] += r1[tid + o];
            r2[tid] += r2[tid + o];
        }
        __syncthreads();
    }
    const double m = (double)HW * cpg;
    const double S1 = r1[0], S2 = r2[0];
    const double k2 = -rstd * rstd * S2 / m;
    const double k3 = -rstd * S1 / m + mean * rstd * rstd * S2 / m;
    for (int c = c0 + tid; c < c0 + cpg; c += 256) {
        const double fs = film ? 1.0 + (double)film[(long long)n * film_stride + c] : 1.0;
        coef[((long long)n * Ctot + c) * 3 + 0] = (float)(rstd * (double)gamma[c] * fs);
        coef[((long long)n * Ctot + c) * 3 + 1] = (float)k2;
        coef[((long long)n * Ctot + c) * 3 + 2] = (float)k3;
    }
}

extern "C" int eod_gn_bwd_finalize(const float* part, int P, int Ctot, int N, int64_t HW, int groups, const float* mean_rstd,
                                   const float* gamma, const float* beta, const float* film, int64_t film_stride, float* dfilm,
                                   int64_t dfilm_stride, float* coef, float* gb, void* stream) {
    EOD_REQUIRE(part && mean_rstd && gamma && coef && gb && P > 0 && Ctot > 0 && N > 0 && groups > 0 && Ctot % groups == 0, "gn_bwd_finalize: bad args");
    EOD_REQUIRE(!dfilm || (film && beta), "gn_bwd_finalize: dfilm needs film and beta");
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(groups, N), dim3(256), 0, (hipStream_t)stream, part, P, Ctot, (long long)HW, groups, mean_rstd, gamma, beta,
                       film, (long long)film_stride, dfilm, (long long)dfilm_stride, coef, gb);
    EOD_CHECK_LAUNCH("gn_bwd_finalize");
    return EOD_OK;
}

// dgamma[c] = sum_n gb[n][c][0], dbeta[c] = sum_n gb[n][c][1]   (fixed order)
__global__ void gn_bwd_params_kernel(const float* __restrict__ gb, int N, int Ctot, float scale, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Ctot) return;
    float a = 0.0f, b = 0.0f;
    for (int n = 0; n < N; ++n) {
        a += gb[((long long)n * Ctot + c) * 2 + 0];
        b += gb[((long long)n * Ctot + c) * 2 + 1];
    }
    dgamma[c] = a * scale;
    dbeta[c] = b * scale;
}

extern "C" int eod_gn_bwd_params(const float* gb, int N, int Ctot, float scale, float* dgamma, float* dbeta, void* stream) {
    EOD_REQUIRE(gb && dgamma && dbeta && N > 0 && Ctot > 0, "gn_bwd_params: bad args");
    hipLaunchKernelGGL(gn_bwd_params_kernel, dim3((Ctot + 255) / 256), dim3(256), 0, (hipStream_t)stream, gb, N, Ctot, scale, dgamma, dbeta);
    EOD_CHECK_LAUNCH("gn_bwd_params");
    return EOD_OK;
}

// apply: dx[n][pix][c] = k1*dz + k2*x + k3 (+ add[n][pix][c]);  one concat source per launch (C channels at coff)
// (slab decomposition of common.h: the 5 per-channel coefficients of a thread's chunk stay in registers)
// csum (optional): per-(image, slab, channel) sums of the STORED dx, csum[n][slab][c][0] -- dx is the output gradient dY of the conv
// that produced x, and that conv's bias / timestep-projection gradients are exactly these sums (eod_channel_sums_finish), so the
// backward-weights step needs no pass of its own over dY (eod_gn_partial) when the gradient comes straight out of this kernel.
template <typename T, bool SILU, bool ADD>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ ss,
                                                          const float* __restrict__ coef, const T* __restrict__ add, int HW, int C, int Ctot,
                                                          int coff, T* __restrict__ dx, int per, float* __restrict__ csum) {
    constexpr int EPC = dt<T>::epc, U = 2;
    constexpr bool FAST = (EPC == 8);
    __shared__ float red[256 * EPC];
    const int tx = threadIdx.x, ty = threadIdx.y, RY = blockDim.y;
    const int n = blockIdx.y;
    const int col = blockIdx.z * blockDim.x + tx;  // chunk column (channel blocks along z for wide layers)
    const bool live = col < C / EPC;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.0f;
    if (live) {
    const int p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
    float sc[EPC], sh[EPC], k1[EPC], k2[EPC], k3[EPC];
    const float* sp = ss + ((long long)n * Ctot + coff + col * EPC) * 2;
    const float* kp = coef + ((long long)n * Ctot + coff + col * EPC) * 3;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = sp[2 * e];
        sh[e] = sp[2 * e + 1];
        k1[e] = kp[3 * e];
        k2[e] = kp[3 * e + 1];
        k3[e] = kp[3 * e + 2];
    }
    const long long xoff = (long long)n * HW * C + col * EPC;
    const T* xb = x + xoff;
    const T* ab = ADD ? add + xoff : nullptr;
    T* ob = dx + xoff;
    const T* db = dy + (long long)n * HW * Ctot + coff + col * EPC;
    for (int pix = p0 + ty; pix < p1; pix += U * RY) {
        T xv[U][EPC], dv[U][EPC], av[U][EPC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pu = pix + u * RY;
            if (pu < p1) {
                *reinterpret_cast<i32x4*>(xv[u]) = *reinterpret_cast<const i32x4*>(xb + pu * C);
                *reinterpret_cast<i32x4*>(dv[u]) = *reinterpret_cast<const i32x4*>(db + pu * Ctot);
                if (ADD) *reinterpret_cast<i32x4*>(av[u]) = *reinterpret_cast<const i32x4*>(ab + pu * C);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pu = pix + u * RY;
            if (pu >= p1) break;
            T ov[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float xf = (float)xv[u][e];
                float dz = (float)dv[u][e];
                if (SILU) dz *= dsilu_f<FAST>(xf * sc[e] + sh[e]);
                float v = k1[e] * dz + k2[e] * xf + k3[e];
                if (ADD) v += (float)av[u][e];
                ov[e] = (T)v;
                acc[e] += (float)ov[e];
            }
            *reinterpret_cast<i32x4*>(ob + pu * C) = *reinterpret_cast<const i32x4*>(ov);
        }
    }
    }
    if (csum) {  // fixed-order sum over the block's pixel rows (ty), one slot per (image, slab)
        const int CPP = blockDim.x;
#pragma unroll
        for (int e = 0; e < EPC; ++e) red[(ty * CPP + tx) * EPC + e] = acc[e];
        __syncthreads();
        if (ty == 0 && live) {
            float* dst = csum + (((long long)n * gridDim.x + blockIdx.x) * C + col * EPC) * 2;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float t = 0.0f;
                for (int r = 0; r < RY; ++r) t += red[(r * CPP + tx) * EPC + e];
                dst[2 * e] = t;
            }
        }
    }
}

template <typename T>
static void launch_gn_bwd_apply(const GnSlab& g, int N, bool silu, hipStream_t st, const T* x, const T* dy, const float* ss, const float* coef,
                                const T* add, int HW, int C, int Ctot, int coff, T* dx, float* csum) {
    const dim3 grid(g.P, N, g.nz), block(g.cpp, g.ry);
#define LAUNCH(S, A) hipLaunchKernelGGL((gn_bwd_apply_kernel<T, S, A>), grid, block, 0, st, x, dy, ss, coef, add, HW, C, Ctot, coff, dx, g.per, csum)
    if (silu) {
        if (add) LAUNCH(true, true); else LAUNCH(true, false);
    } else {
        if (add) LAUNCH(false, true); else LAUNCH(false, false);
    }
#undef LAUNCH
}

extern "C" int eod_gn_bwd_apply_slabs(int dtype, int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    return gn_slab(N, HW, C, 16 / eod_esize(dtype), 2).P;
}
extern "C" int eod_gn_bwd_apply(const void* x, const void* dy, const float* scale_shift, const float* coef, const void* add, int dtype, int N,
                                int HW, int C, int Ctot, int coff, int silu, void* dx, float* csum, void* stream) {
    EOD_REQUIRE(x && dy && scale_shift && coef && dx && N > 0 && HW > 0 && C > 0, "gn_bwd_apply: bad args");
    const int epc = 16 / eod_esize(dtype);
    EOD_REQUIRE(C % epc == 0 && Ctot % epc == 0 && coff % epc == 0, "gn_bwd_apply: channel alignment");
    EOD_REQUIRE(N <= 65535, "gn_bwd_apply: N=%d unsupported", N);
    EOD_REQUIRE(eod_aligned16(x) && eod_aligned16(dy) && eod_aligned16(dx) && (!add || eod_aligned16(add)), "gn_bwd_apply: alignment");
    const GnSlab g = gn_slab(N, HW, C, epc, 2);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EOD_F16)
        launch_gn_bwd_apply<half_t>(g, N, silu != 0, st, (const half_t*)x, (const half_t*)dy, scale_shift, coef, (const half_t*)add, HW, C, Ctot, coff, (half_t*)dx, csum);
    else
        launch_gn_bwd_apply<float>(g, N, silu != 0, st, (const float*)x, (const float*)dy, scale_shift, coef, (const float*)add, HW, C, Ctot, coff, (float*)dx, csum);
    EOD_CHECK_LAUNCH("gn_bwd_apply");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// y = a + b (storage dtype, 16-byte chunks): joins two gradient branches when no conv epilogue is at hand
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long long nchunks) {
    constexpr int EPC = dt<T>::epc;
    for (long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x; f < nchunks; f += (long long)gridDim.x * blockDim.x) {
        T av[EPC], bv[EPC], ov[EPC];
        *reinterpret_cast<i32x4*>(av) = *reinterpret_cast<const i32x4*>(a + f * EPC);
        *reinterpret_cast<i32x4*>(bv) = *reinterpret_cast<const i32x4*>(b + f * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) ov[e] = (T)((float)av[e] + (float)bv[e]);
        *reinterpret_cast<i32x4*>(y + f * EPC) = *reinterpret_cast<const i32x4*>(ov);
    }
}

extern "C" int eod_add(const void* a, const void* b, void* y, int dtype, int64_t n, void* stream) {
    const int epc = 16 / eod_esize(dtype);
    EOD_REQUIRE(a && b && y && n > 0 && n % epc == 0 && eod_aligned16(a) && eod_aligned16(b) && eod_aligned16(y), "add: bad args");
    const long long nchunks = n / epc;
    long long blocks = (nchunks + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(add_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const half_t*)a, (const half_t*)b, (half_t*)y, nchunks);
    else
        hipLaunchKernelGGL(add_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)y, nchunks);
    EOD_CHECK_LAUNCH("add");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// Small dense layers of the timestep-embedding MLP (unet_openai.py:597-602, 329-335), fp32, N = batch rows:
//   forward  out[n][j] = sum_k in[n][k] * w[j][k] + b[j]
//   backward dW[j][k] = scale * sum_n dout[n][j] * in[n][k],  db[j] = scale * sum_n dout[n][j],
//            din[n][k] = sum_j dout[n][j] * w[j][k]
// `in` is the ACTIVATED input of the layer (act_in: 0 = as stored, 1 = SiLU(in), 2 = sinusoid of t).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float temb_in(const float* in, const long long* t, const float* freqs, int n, int k, int K, int act_in) {
    if (act_in == 2) {
        const int half = K / 2;
        const float tf = (float)t[n];
        if (k < half) return cosf(tf * freqs[k]);
        if (k < 2 * half) return sinf(tf * freqs[k - half]);
        return 0.0f;
    }
    const float v = in[(long long)n * K + k];
    return act_in == 1 ? silu_f<false>(v) : v;
}

__global__ void linear_bwd_w_kernel(const float* __restrict__ dout, long long ld_dout, const float* __restrict__ in, const long long* __restrict__ t,
                                    const float* __restrict__ freqs, int N, int K, int J, int act_in, float scale, float* __restrict__ dW,
                                    float* __restrict__ db) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)J * K) return;
    const int j = (int)(i / K), k = (int)(i - (long long)j * K);
    float a = 0.0f, b = 0.0f;
    for (int n = 0; n < N; ++n) {
        const float d = dout[(long long)n * ld_dout + j];
        a += d * temb_in(in, t, freqs, n, k, K, act_in);
        b += d;
    }
    dW[i] = a * scale;
    if (k == 0 && db) db[j] = b * scale;
}

// din[n][k] = sum_j dout[n][j] * w[j][k]; optionally multiplied by silu'(pre[n][k]) (the layer's input was SiLU(pre)).
// J is split over gridDim.y (fixed-order partial sums in `din` itself when gridDim.y == 1, else in a scratch that the
// caller provides: here the split results are combined by a second launch, linear_bwd_in_sum_kernel)
__global__ void linear_bwd_in_kernel(const float* __restrict__ dout, long long ld_dout, const float* __restrict__ w, const float* __restrict__ pre,
                                     int N, int K, int J, int jper, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * K) return;
    const int n = (int)(i / K), k = (int)(i - (long long)n * K);
    const int j0 = blockIdx.y * jper, j1 = min(J, j0 + jper);
    float a = 0.0f;
    for (int j = j0; j < j1; ++j) a += dout[(long long)n * ld_dout + j] * w[(long long)j * K + k];
    if (gridDim.y == 1 && pre) a *= dsilu_f<false>(pre[i]);
    out[(long long)blockIdx.y * N * K + i] = a;
}

__global__ void linear_bwd_in_sum_kernel(const float* __restrict__ part, int S, const float* __restrict__ pre, long long NK, float* __restrict__ din) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NK) return;
    float a = 0.0f;
    for (int s = 0; s < S; ++s) a += part[(long long)s * NK + i];
    if (pre) a *= dsilu_f<false>(pre[i]);
    din[i] = a;
}

extern "C" int eod_linear_bwd_small(const float* dout, int64_t ld_dout, const float* in, const int64_t* t, const float* freqs,
                                    const float* w, const float* pre, int N, int K, int J, int act_in, float scale, float* dW, float* db,
                                    float* din, float* scratch, void* stream) {
    EOD_REQUIRE(dout && w && N > 0 && K > 0 && J > 0 && ld_dout >= J && act_in >= 0 && act_in <= 2, "linear_bwd_small: bad args");
    EOD_REQUIRE(act_in == 2 ? (t && freqs) : (in != nullptr || !dW), "linear_bwd_small: missing input");
    hipStream_t st = (hipStream_t)stream;
    if (dW) {
        const long long tot = (long long)J * K;
        hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, dout, (long long)ld_dout, in, (const long long*)t, freqs, N, K, J, act_in, scale, dW, db);
    }
    if (din) {
        const long long tot = (long long)N * K;
        const int S = (scratch && J >= 512) ? 32 : 1;  // split the J loop when a scratch of 32*N*K floats is supplied
        const int jper = (J + S - 1) / S;
        hipLaunchKernelGGL(linear_bwd_in_kernel, dim3((unsigned)((tot + 255) / 256), S), dim3(256), 0, st, dout, (long long)ld_dout, w, pre, N, K, J, jper,
                           S == 1 ? din : scratch);
        if (S > 1)
            hipLaunchKernelGGL(linear_bwd_in_sum_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, scratch, S, pre, tot, din);
    }
    EOD_CHECK_LAUNCH("linear_bwd_small");
    return EOD_OK;
}

// pre-activation of time_embed[0] (needed for SiLU' in the backward; the forward only keeps SiLU(pre1)):
//   pre1[n][j] = sum_k sinusoid(t[n])[k] * w1[j][k] + b1[j]
__global__ void temb_pre1_kernel(const long long* __restrict__ t, const float* __restrict__ freqs, const float* __restrict__ w1,
                                 const float* __restrict__ b1, int N, int D, int E, float* __restrict__ pre1) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * E) return;
    const int n = (int)(i / E), j = (int)(i - (long long)n * E);
    float a = 0.0f;
    for (int k = 0; k < D; ++k) a += temb_in(nullptr, t, freqs, n, k, D, 2) * w1[(long long)j * D + k];
    pre1[i] = a + b1[j];
}

extern "C" int eod_temb_pre1(const int64_t* t, const float* freqs, const float* w1, const float* b1, int N, int D, int E, float* pre1,
                             void* stream) {
    EOD_REQUIRE(t && freqs && w1 && b1 && pre1 && N > 0 && D > 0 && E > 0, "temb_pre1: bad args");
    const long long tot = (long long)N * E;
    hipLaunchKernelGGL(temb_pre1_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long long*)t, freqs, w1, b1, N, D, E, pre1);
    EOD_CHECK_LAUNCH("temb_pre1");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// softmax backward, one wave per row: dS = P * (dP - sum(dP * P)); pad columns [n, ldp) are zeroed (they are K
// positions of the following GEMMs)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const T* __restrict__ P, long long ldp, const float* __restrict__ dP, long long lds,
                                                               T* __restrict__ dS, long long rows, int n) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const T* p = P + r * ldp;
    const float* g = dP + r * lds;
    float acc = 0.0f;
    for (int j = lane; j < n; j += 64) acc += g[j] * (float)p[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    T* o_ = dS + r * ldp;
    for (int j = lane; j < (int)ldp; j += 64) o_[j] = j < n ? (T)((float)p[j] * (g[j] - acc)) : (T)0.0f;
}

// long rows: one workgroup per row, P and dP held in registers (read once), see softmax_row_block_kernel in norm.hip
template <typename T, int V4>
__global__ __launch_bounds__(256) void softmax_bwd_row_block_kernel(const T* __restrict__ P, long long ldp, const float* __restrict__ dP, long long lds,
                                                                    T* __restrict__ dS, int n) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const T* pr = P + row * ldp;
    const float* gr = dP + row * lds;
    float pv[V4][4], gv[V4][4];
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        const int c = (i * 256 + threadIdx.x) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[i][e] = gv[i][e] = 0.0f;
        if (c + 3 < n) {
            T t4[4];
            if constexpr (sizeof(T) == 2)
                *reinterpret_cast<unsigned long long*>(t4) = *reinterpret_cast<const unsigned long long*>(pr + c);
            else
                *reinterpret_cast<f32x4*>(t4) = *reinterpret_cast<const f32x4*>(pr + c);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pv[i][e] = (float)t4[e];
                gv[i][e] = g4[e];
            }
        } else {
            for (int e = 0; e < 4; ++e)
                if (c + e < n) {
                    pv[i][e] = (float)pr[c + e];
                    gv[i][e] = gr[c + e];
                }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += gv[i][e] * pv[i][e];
    }
    acc = wave_sum_t(acc);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = acc;
    __syncthreads();
    acc = (red[0] + red[1]) + (red[2] + red[3]);
    T* o_ = dS + row * ldp;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        const int c = (i * 256 + threadIdx.x) * 4;
        if (c >= ldp) continue;
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (c + e < n) ? (T)(pv[i][e] * (gv[i][e] - acc)) : (T)0.0f;
        if (c + 3 < ldp) {
            if constexpr (sizeof(T) == 2)
                *reinterpret_cast<unsigned long long*>(o_ + c) = *reinterpret_cast<const unsigned long long*>(o);
            else
                *reinterpret_cast<f32x4*>(o_ + c) = *reinterpret_cast<const f32x4*>(o);
        } else {
            for (int e = 0; e < 4 && c + e < ldp; ++e) o_[c + e] = o[e];
        }
    }
}

template <typename T>
static bool launch_softmax_bwd_block(const T* P, long long ldp, const float* dP, long long lds, T* dS, long long rows, int n, hipStream_t st) {
    if (lds % 4 || ldp % 4 || (reinterpret_cast<uintptr_t>(P) & 15) || (reinterpret_cast<uintptr_t>(dP) & 15) || (reinterpret_cast<uintptr_t>(dS) & 15) ||
        ldp > 16384 || n < 1024 || rows > 0x7fffffffLL)
        return false;
    const int need = (int)((ldp + 1023) / 1024);
    const dim3 grid((unsigned)rows), block(256);
    if (need <= 1) hipLaunchKernelGGL((softmax_bwd_row_block_kernel<T, 1>), grid, block, 0, st, P, ldp, dP, lds, dS, n);
    else if (need <= 2) hipLaunchKernelGGL((softmax_bwd_row_block_kernel<T, 2>), grid, block, 0, st, P, ldp, dP, lds, dS, n);
    else if (need <= 4) hipLaunchKernelGGL((softmax_bwd_row_block_kernel<T, 4>), grid, block, 0, st, P, ldp, dP, lds, dS, n);
    else if (need <= 8) hipLaunchKernelGGL((softmax_bwd_row_block_kernel<T, 8>), grid, block, 0, st, P, ldp, dP, lds, dS, n);
    else hipLaunchKernelGGL((softmax_bwd_row_block_kernel<T, 16>), grid, block, 0, st, P, ldp, dP, lds, dS, n);
    return true;
}

extern "C" int eod_softmax_bwd_rows(const void* P, int64_t ldp, const float* dP, int64_t lds, void* dS, int dtype, int64_t rows, int n,
                                    void* stream) {
    EOD_REQUIRE(P && dP && dS && rows > 0 && n > 0 && ldp >= n && lds >= n, "softmax_bwd_rows: bad args");
    if (dtype == EOD_F16 ? launch_softmax_bwd_block<half_t>((const half_t*)P, (long long)ldp, dP, (long long)lds, (half_t*)dS, (long long)rows, n, (hipStream_t)stream)
                         : launch_softmax_bwd_block<float>((const float*)P, (long long)ldp, dP, (long long)lds, (float*)dS, (long long)rows, n, (hipStream_t)stream)) {
        EOD_CHECK_LAUNCH("softmax_bwd_rows");
        return EOD_OK;
    }
    const long long blocks = (rows + 3) / 4;
    EOD_REQUIRE(blocks <= 0x7fffffffLL, "softmax_bwd_rows: too many rows");
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(softmax_bwd_rows_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const half_t*)P, (long long)ldp, dP, (long long)lds, (half_t*)dS, (long long)rows, n);
    else
        hipLaunchKernelGGL(softmax_bwd_rows_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)P, (long long)ldp, dP, (long long)lds, (float*)dS, (long long)rows, n);
    EOD_CHECK_LAUNCH("softmax_bwd_rows");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// Optimizer side of the step (train.py:86,117-122): nn.MSELoss(reduction='mean') with its gradient, torch.optim.AdamW
// (single-tensor algorithm of the pinned PyTorch 1.13: decoupled decay, bias corrections folded into step_size and
// bc2_sqrt on the host), and the EMA of utils.py:56-67 (decay*avg + (1-decay)*param).  Plain fp32 elementwise kernels
// over FLAT buffers (one launch for all parameters); this file is built with -ffp-contract=off so that they are
// bit-exact against oracle/train_ref.py.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target, long long n,
                                                          float grad_scale, float* __restrict__ dpred, float* __restrict__ part) {
    __shared__ float red[256];
    float a = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        a += d * d;
        if (dpred) dpred[i] = d * grad_scale;  // grad_scale = 2/n (mean reduction)
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ void mse_final_kernel(const float* __restrict__ part, int P, float inv_n, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float a = 0.0f;
        for (int i = 0; i < P; ++i) a += part[i];
        loss[0] = a * inv_n;
    }
}

extern "C" int eod_mse_loss(const float* pred, const float* target, int64_t n, float* loss, float* dpred, float* scratch, int scratch_len,
                            void* stream) {
    EOD_REQUIRE(pred && target && loss && scratch && n > 0 && scratch_len >= 1, "mse_loss: bad args");
    int P = scratch_len < 1024 ? scratch_len : 1024;
    if ((long long)P * 256 > n) P = (int)((n + 255) / 256);
    hipLaunchKernelGGL(mse_partial_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, pred, target, (long long)n, 2.0f / (float)n, dpred, scratch);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, P, 1.0f / (float)n, loss);
    EOD_CHECK_LAUNCH("mse_loss");
    return EOD_OK;
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                             float decay_mul, float beta1, float one_m_beta1, float beta2, float one_m_beta2, float bc2_sqrt, float eps,
                             float neg_step_size) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float pi = p[i] * decay_mul;                       // param.mul_(1 - lr * weight_decay)
        const float mi = m[i] * beta1 + gi * one_m_beta1;  // exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
        const float vi = v[i] * beta2 + (one_m_beta2 * gi) * gi;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;    // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        pi = pi + neg_step_size * (mi / denom);            // param.addcdiv_(exp_avg, denom, value=-step_size)
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

extern "C" int eod_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                              double weight_decay, int step, void* stream) {
    EOD_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw_step: bad args");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = lr / bc1;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, (float)(1.0 - lr * weight_decay),
                       (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, (float)(-step_size));
    EOD_CHECK_LAUNCH("adamw_step");
    return EOD_OK;
}

// ---- overflow guard of fp16 training (static loss scale): a gradient that is not finite must not reach the weights.  Two launches,
// no atomics, no host round trip: per-block flags, then one block folds them into state = {found now, skipped steps so far}; the
// guarded AdamW reads state[0] and leaves p / m / v untouched when it is set.
__global__ void nonfinite_blocks_kernel(const float* __restrict__ g, long long n, int* __restrict__ block_flags) {
    int bad = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = g[i];
        bad |= !(fabsf(v) <= 3.402823466e38f);  // inf or NaN
    }
    bad = __any(bad);
    __shared__ int sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) block_flags[blockIdx.x] = sh[0] | sh[1] | sh[2] | sh[3];
}
// state = {this step skipped, steps skipped so far, bits(sqrt(1 - beta2^a)), bits(-lr / (1 - beta1^a))} with a = the number of steps
// APPLIED so far including this one (= calls - skipped): a skipped step never reaches the optimizer (torch.cuda.amp.GradScaler
// semantics), so the bias corrections follow the count of moment updates, computed here on the device (no host round trip)
__global__ void nonfinite_fold_kernel(const int* __restrict__ block_flags, int nblocks, int* __restrict__ state, int calls, double lr, double beta1,
                                      double beta2) {
    int bad = 0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) bad |= block_flags[i];
    bad = __any(bad);
    __shared__ int sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int f = sh[0] | sh[1] | sh[2] | sh[3];
        state[0] = f;
        state[1] += f;
        const int applied = calls - state[1] > 1 ? calls - state[1] : 1;
        state[2] = __float_as_int((float)sqrt(1.0 - pow(beta2, (double)applied)));
        state[3] = __float_as_int((float)(-(lr / (1.0 - pow(beta1, (double)applied)))));
    }
}
__global__ void adamw_guarded_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                                     float decay_mul, float beta1, float one_m_beta1, float beta2, float one_m_beta2, float eps,
                                     const int* __restrict__ state) {
    if (state[0]) return;  // this step's gradients are not finite: skipped (uniform over the grid)
    const float bc2_sqrt = __int_as_float(state[2]), neg_step_size = __int_as_float(state[3]);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float pi = p[i] * decay_mul;
        const float mi = m[i] * beta1 + gi * one_m_beta1;
        const float vi = v[i] * beta2 + (one_m_beta2 * gi) * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi + neg_step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

extern "C" int eod_adamw_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                                      double weight_decay, int step, int* state, int* scratch, int scratch_len, void* stream) {
    EOD_REQUIRE(p && g && m && v && n > 0 && step >= 1 && state && scratch && scratch_len >= 1, "adamw_step_guarded: bad args");
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    int cb = (int)(blocks < scratch_len ? blocks : scratch_len);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nonfinite_blocks_kernel, dim3((unsigned)cb), dim3(256), 0, st, g, (long long)n, scratch);
    hipLaunchKernelGGL(nonfinite_fold_kernel, dim3(1), dim3(256), 0, st, (const int*)scratch, cb, state, step, lr, beta1, beta2);
    hipLaunchKernelGGL(adamw_guarded_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, (long long)n, (float)(1.0 - lr * weight_decay),
                       (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (const int*)state);
    EOD_CHECK_LAUNCH("adamw_step_guarded");
    return EOD_OK;
}

__global__ void ema_kernel(float* __restrict__ avg, const float* __restrict__ p, long long n, float decay, float one_m_decay) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        avg[i] = decay * avg[i] + one_m_decay * p[i];
}

extern "C" int eod_ema_update(float* avg, const float* p, int64_t n, double decay, void* stream) {
    EOD_REQUIRE(avg && p && n > 0, "ema_update: bad args");
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, avg, p, (long long)n, (float)decay, (float)(1.0 - decay));
    EOD_CHECK_LAUNCH("ema_update");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// nn.Embedding backward (label_emb, unet_openai.py:604-605,764-766): dW[c][e] = scale * sum_{n: y[n] == c} dout[n][e]
// (fixed order over n: deterministic, no atomics)
// ---------------------------------------------------------------------------------------------
__global__ void embedding_bwd_kernel(const float* __restrict__ dout, const long long* __restrict__ y, int N, int E, int classes, float scale,
                                     float* __restrict__ dW) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)classes * E) return;
    const int c = (int)(i / E), e = (int)(i - (long long)c * E);
    float a = 0.0f;
    for (int n = 0; n < N; ++n)
        if (y[n] == c) a += dout[(long long)n * E + e];
    dW[i] = a * scale;
}

extern "C" int eod_embedding_bwd(const float* dout, const int64_t* y, int N, int E, int classes, float scale, float* dW, void* stream) {
    EOD_REQUIRE(dout && y && dW && N > 0 && E > 0 && classes > 0, "embedding_bwd: bad args");
    const long long tot = (long long)classes * E;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, (const long long*)y, N, E, classes, scale, dW);
    EOD_CHECK_LAUNCH("embedding_bwd");
    return EOD_OK;
}

// =============================================================================================
// conv3x3_wgrad_kernel (fp16): backward-weights of a 3x3 / stride-1 / pad-1 conv WITHOUT transposed copies in HBM.
//   dW[ky][kx][co][ci] = sum_{n,h,w} dY[n][h][w][co] * X[n][h+ky-1][w+kx-1][ci]
// A workgroup owns a 128 (co) x 128 (ci) tile for ONE ky and all three kx, and walks a range of 64-pixel strips
// (64 consecutive w of one image row).  Per strip it stages, pixel-major exactly as the tensors lie in HBM (LDS-DMA, 256-byte
// rows = 128 channels):  dY[n][h][w0 .. w0+63][co tile]  and  X[n][h+ky-1][w0-1 .. w0+64][ci tile]  (zero outside the image);
// the three kx taps are the same X strip read at row offsets 0 / 1 / 2.  The MFMA operands need the PIXEL axis as K, i.e. a
// transposed view of those tiles: ds_read_b64_tr_b16 delivers it (each 16-lane group reads a 4-pixel x 16-channel block and
// every lane receives one channel's 4 pixels).  Rows are XOR-swizzled at 16-byte granularity, swz(row) = ((row&3)<<2)|((row>>2)&3)
// (MI355X guide, T10 layout (b)), applied on the SOURCE side of the DMA.
// Arithmetic intensity: 3 x 128 x 128 x 64 MACs per 32.5 KiB staged = 193 FLOP/B (vs 64 for the generic NT GEMM over
// transposed copies), and the transposes themselves disappear.
// Output: fp32 partial tiles partial[split][ky*3+kx][co][ci] (same layout as the GEMM path -> eod_wgrad_reduce).
// Requirements (checked on the host): fp16, Wo % 64 == 0, channel counts multiples of 8.
// MFMA shape: 32x32x16.  The 16x16x32 shape that pays in the forward kernels was tried here (round 2) and LOST 17 % on the whole
// training step: the 192 accumulator registers leave 64 for everything else, four A + two B fragments and their addresses do not
// fit (38 spilled VGPRs; 9 with the sub-step loop rolled, which then serialises the fragment loads behind the MFMAs).
// =============================================================================================
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) void lds_void_t;

struct WgradP {
    const char* dy;   // [N][Ho][Wo][Cy] fp16
    const char* x;    // [N][H][W][Cx] fp16 (ups: the conv input is its nearest-2x upsampling)
    float* partial;   // [S][9][Cout][ldp]
    int N, H, W, Cx, Ho, Wo, Cy, Cout, ups, ldp, S;
    int tiles_co, tiles_ci, strips_w, strips_total, strips_per;
};

__device__ __forceinline__ int wg_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// WS = image-row width covered by a strip (64: one 64-pixel segment of a row; 32 / 16: 2 / 4 whole rows of a 32- / 16-wide map).
// Pixel k of the strip lies in image row r = k / WS; its X row for tap kx is staged at  k + 16*r + kx  (every image row gets its own
// left / right halo, and the 16-row pitch keeps swz(row) invariant under the k -> k + 16 steps of the MFMA sub-steps).
// CLS = true (p.ups == 2): the conv input is the nearest-2x upsampling of X and the weight gradient is taken in the parity-class form of
// csrc/igemm.hip (conv_up4_halo_kernel): for class (p, q) and row tap a the workgroup correlates the stride-2 view G_pq[i][j] =
// dY[2i+p][2j+q] with X rows i - 1 + p + a, the two column taps b being X row offsets q + b:
//   dW'_pq[a][b][co][ci] = sum_{n,i,j} G_pq[i][j][co] * X[i-1+p+a][j-1+q+b][ci]        (16 tap products per stored position, not 36)
// strips walk the STORED (H x W) map; partial[split][((2p+q)*2 + a)*2 + b][co][ci]; eod_wgrad_up4_map folds the 16 back into the 9.
// MODE 2 (p.ups == 3): STRIDE-2 conv (Downsample.op).  dW[ky][kx] = sum dY[oh][ow] X[2oh+ky-1][2ow+kx-1]: strips walk dY, the X row is
// gathered at pixel stride 2 in one of two column phases per workgroup -- odd columns 2(w0+j)-1 serve kx = 0 and kx = 2 (row offsets 0
// and 1), even columns 2(w0+j) serve kx = 1 -- so the workgroup index enumerates (ky, phase) and the planes land in the ordinary
// partial[split][ky*3+kx] layout (no transposed copies, no GEMM path).
template <int WS, int MODE = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const WgradP p) {
    constexpr bool CLS = MODE == 1, S2 = MODE == 2;
    constexpr int RPS = 64 / WS, RPITCH = WS + 16;                    // image rows per strip, staged-row pitch of an image row
    constexpr int A_ROWS = 64, X_ROWS = ((RPS - 1) * RPITCH + WS + 2 + 3) / 4 * 4, ROWB = 256, XG = X_ROWS / 4;
    constexpr int A_BYTES = A_ROWS * ROWB, X_BYTES = X_ROWS * ROWB, STAGE = A_BYTES + X_BYTES;  // 16 KiB + 17 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][STAGE]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int b = blockIdx.x;
    const int tile_ci = b % p.tiles_ci; b /= p.tiles_ci;
    const int tile_co = b % p.tiles_co; b /= p.tiles_co;
    constexpr int NKY = CLS ? 8 : S2 ? 6 : 3, NKX = (CLS || S2) ? 2 : 3;
    const int kyi = b % NKY;
    const int split = b / NKY;
    const int cp = CLS ? kyi >> 2 : 0, cq = CLS ? (kyi >> 1) & 1 : 0;   // class (p, q)
    const int ky = CLS ? cp + (kyi & 1) : S2 ? kyi >> 1 : kyi;          // X row of the strip's image row h: h + ky - 1
    const int phase = S2 ? kyi & 1 : 0;                                 // stride 2: 0 = odd columns (kx 0 / 2), 1 = even columns (kx 1)
    const int co0 = tile_co * 128, ci0 = tile_ci * 128;
    const int s_begin = split * p.strips_per, s_end = min(p.strips_total, s_begin + p.strips_per);
    const int Heff = (MODE == 0 && p.ups) ? 2 * p.H : p.H, Weff = (MODE == 0 && p.ups) ? 2 * p.W : p.W;
    const int Hs = CLS ? p.H : p.Ho, Ws = CLS ? p.W : p.Wo;             // the map the strips walk

    // ---- DMA slots: one instruction = 4 tile rows x 16 chunks; this lane: row (4*grp + lane/16), LDS slot lane%16 ----
    const int drow = lane >> 4, dslot = lane & 15;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.dy), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, 0x7fffffff, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    auto issue_strip = [&](int strip, int stage) {
        // strip -> (n, h, w0): 64 consecutive pixels of one image in raster order
        const int spi = Hs * Ws / 64;  // strips per image
        const int n = strip / spi;
        const int pix0 = (strip - n * spi) * 64;
        const int h = pix0 / Ws, w0 = pix0 - h * Ws;
        char* sa = smem + stage * STAGE;
        char* sx = sa + A_BYTES;
        // dY: groups wave, wave+4, wave+8, wave+12 (4 rows each)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave + 4 * i) * 4 + drow;
            const int chunk = dslot ^ wg_swz(row);
            long long pix = (long long)n * p.Ho * p.Wo + pix0 + row;
            if constexpr (CLS) {  // stored position (h + row / WS, w0 + row % WS) -> pixel (2 i + p, 2 j + q) of the (2H x 2W) gradient
                const int ir = h + row / WS, jc = w0 + row % WS;
                pix = ((long long)n * p.Ho + 2 * ir + cp) * p.Wo + 2 * jc + cq;
            }
            const int c = co0 + chunk * 8;
            const unsigned v = c < p.Cy ? (unsigned)((pix * p.Cy + c) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_void_t*)(sa + (wave + 4 * i) * 1024), 16, v, 0, 0, 0);
        }
        // X: XG groups of 4 staged rows; staged row -> (image row r of the strip, column w0 - 1 + rr)
#pragma unroll
        for (int i = 0; i < (XG + 3) / 4; ++i) {
            const int grp = wave + 4 * i;
            if (grp < XG) {
                const int row = grp * 4 + drow;
                const int chunk = dslot ^ wg_swz(row);
                const int r = row / RPITCH, rr = row - r * RPITCH;
                const int hh = S2 ? 2 * (h + r) + ky - 1 : h + r + ky - 1;
                const int ww = S2 ? 2 * (w0 + rr) - 1 + phase : w0 - 1 + rr;
                const int c = ci0 + chunk * 8;
                const bool ok = r < RPS && rr < WS + 2 && (unsigned)hh < (unsigned)Heff && (unsigned)ww < (unsigned)Weff && c < p.Cx;
                int hs = hh, ws = ww;
                if (MODE == 0 && p.ups) {
                    hs >>= 1;
                    ws >>= 1;
                }
                const long long pix = ((long long)n * p.H + hs) * p.W + ws;
                const unsigned v = ok ? (unsigned)((pix * p.Cx + c) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void_t*)(sx + grp * 1024), 16, v, 0, 0, 0);
            }
        }
    };

    // ---- transposed-read addresses (bytes inside a stage), see the header comment ----
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    auto tr_addr = [&](int row, int col) {  // col = channel inside the 128-wide tile (multiple of 4)
        return row * ROWB + (((col >> 3) ^ wg_swz(row)) << 4) + ((col >> 2) & 1) * 8;
    };
    int a_ad[2][2];     // [j][i]  : dY tile, K rows (g>>1)*8 + j*4 + q, M cols wm*64 + i*32 + (g&1)*16 + 4*pp
    int b_ad[3][2][2];  // [kx][j][i2] : X tile, rows + kx
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (g >> 1) * 8 + j * 4 + q;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a_ad[j][i] = tr_addr(row, wm * 64 + i * 32 + (g & 1) * 16 + 4 * pp);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) b_ad[kx][j][i] = A_BYTES + tr_addr(row + kx + cq, wn * 64 + i * 32 + (g & 1) * 16 + 4 * pp);
        }
    }

    f32x16 acc[3][2][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.0f;

    if (s_begin < s_end) issue_strip(s_begin, 0);
    for (int s = s_begin; s < s_end; ++s) {
        const int stage = (s - s_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 1 < s_end) issue_strip(s + 1, stage ^ 1);
        const char* sb = smem + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {  // 16 pixels per MFMA; (row + 16) keeps swz(row): +4096 bytes per sub-step
            const int xoff = (ks + (ks * 16) / WS) * 4096;  // X rows: + 16 staged rows per image row of the strip
            half8 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(sb + a_ad[0][i] + ks * 4096));
                const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(sb + a_ad[1][i] + ks * 4096));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    fa[i][e] = (half_t)lo[e];
                    fa[i][4 + e] = (half_t)hi[e];
                }
            }
#pragma unroll
            for (int kx = 0; kx < NKX; ++kx) {
                half8 fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(sb + b_ad[kx][0][i] + xoff));
                    const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(sb + b_ad[kx][1][i] + xoff));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        fb[i][e] = (half_t)lo[e];
                        fb[i][4 + e] = (half_t)hi[e];
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[kx][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[kx][i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: C layout (lane&31 = ci column, regs = co rows) -> fp32 partial tile ----
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int kx = 0; kx < NKX; ++kx) {
        if (S2 && phase == 1 && kx == 1) break;  // the even phase carries one tap
        const int plane = CLS ? kyi * 2 + kx : S2 ? ky * 3 + (phase ? 1 : 2 * kx) : ky * 3 + kx;
        float* base = p.partial + ((long long)split * (CLS ? 16 : 9) + plane) * p.Cout * p.ldp;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ci = ci0 + wn * 64 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (co < p.Cout && ci < p.Cx) base[(long long)co * p.ldp + ci] = acc[kx][i][j][r];
                }
            }
    }
}

extern "C" int eod_conv3x3_wgrad(const void* dy, const void* x, int dtype, int N, int H, int W, int Cx, int Ho, int Wo, int Cy, int Cout,
                                 int ups, float* partial, int ldp, int S, void* stream) {
    EOD_REQUIRE(dy && x && partial && N > 0 && H > 0 && W > 0 && Cx > 0 && Ho > 0 && Wo > 0 && Cy > 0 && Cout > 0 && S > 0 && ldp >= Cx,
                "conv3x3_wgrad: bad args");
    EOD_REQUIRE(dtype == EOD_F16, "conv3x3_wgrad: fp16 only (the transposed LDS read is a 16-bit instruction)");
    EOD_REQUIRE(ups >= 0 && ups <= 3, "conv3x3_wgrad: ups %d", ups);
    const bool cls = ups == 2;  // parity-class form: strips walk the stored (H x W) map, 16 tap planes per split
    const bool s2 = ups == 3;   // stride-2 conv: X is (H x W) = (2 Ho x 2 Wo), gathered at pixel stride 2
    const int Hs = cls ? H : Ho, Ws = cls ? W : Wo;
    const int ws = Ws % 64 == 0 ? 64 : Ws;
    EOD_REQUIRE((ws == 64 || ws == 32 || ws == 16) && (Hs * Ws) % 64 == 0 && Cx % 8 == 0 && Cy % 8 == 0 && Cout <= Cy,
                "conv3x3_wgrad: needs a map width %% 64 == 0 (or 32 / 16 with H*W %% 64 == 0) and channel counts that are multiples of 8");
    EOD_REQUIRE(s2 ? (H == 2 * Ho && W == 2 * Wo) : (Ho == (ups ? 2 * H : H) && Wo == (ups ? 2 * W : W)),
                "conv3x3_wgrad: stride-1 / pad-1 geometry (or, ups = 3, a stride-2 conv of an even map) expected");
    EOD_REQUIRE(eod_aligned16(dy) && eod_aligned16(x), "conv3x3_wgrad: 16-byte alignment");
    EOD_REQUIRE((long long)N * Ho * Wo * Cy * 2 < 0x7fffffffLL && (long long)N * H * W * Cx * 2 < 0x7fffffffLL, "conv3x3_wgrad: tensors exceed the 2 GiB buffer window");
    WgradP p;
    p.dy = (const char*)dy; p.x = (const char*)x; p.partial = partial;
    p.N = N; p.H = H; p.W = W; p.Cx = Cx; p.Ho = Ho; p.Wo = Wo; p.Cy = Cy; p.Cout = Cout; p.ups = ups; p.ldp = ldp; p.S = S;
    p.tiles_co = (Cout + 127) / 128;
    p.tiles_ci = (Cx + 127) / 128;
    p.strips_w = 0;
    p.strips_total = N * (Hs * Ws / 64);
    p.strips_per = (p.strips_total + S - 1) / S;
    const long long grid = (long long)p.tiles_co * p.tiles_ci * (cls ? 8 : s2 ? 6 : 3) * S;
    EOD_REQUIRE(grid <= 0x7fffffffLL, "conv3x3_wgrad: grid too large");
    const int rps = 64 / ws, xrows = ((rps - 1) * (ws + 16) + ws + 2 + 3) / 4 * 4;
    const size_t lds = 2 * (size_t)(64 * 256 + xrows * 256);
    const int mode = cls ? 1 : s2 ? 2 : 0;
    void (*kern)(const WgradP) =
        mode == 1 ? (ws == 64 ? conv3x3_wgrad_kernel<64, 1> : ws == 32 ? conv3x3_wgrad_kernel<32, 1> : conv3x3_wgrad_kernel<16, 1>)
      : mode == 2 ? (ws == 64 ? conv3x3_wgrad_kernel<64, 2> : ws == 32 ? conv3x3_wgrad_kernel<32, 2> : conv3x3_wgrad_kernel<16, 2>)
                  : (ws == 64 ? conv3x3_wgrad_kernel<64, 0> : ws == 32 ? conv3x3_wgrad_kernel<32, 0> : conv3x3_wgrad_kernel<16, 0>);
    static bool attr_done[3][3] = {{false, false, false}, {false, false, false}, {false, false, false}};
    const int vi = ws == 64 ? 0 : ws == 32 ? 1 : 2;
    if (!attr_done[mode][vi]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done[mode][vi] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, p);
    EOD_CHECK_LAUNCH("conv3x3_wgrad");
    return EOD_OK;
}

// dW[co][ci][ky][kx] (+)= sum over the classes of dW'[co][ci][(2p+q)*4 + a(p,ky)*2 + b(q,kx)]: the transpose of the tap sums that form
// the class kernels (rows: class 0 puts w0 in slot a = 0 and w1 + w2 in a = 1, class 1 puts w0 + w1 in a = 0 and w2 in a = 1)
__global__ void wgrad_up4_map_kernel(const float* __restrict__ t16, long long n, float* __restrict__ dw) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(v + 4 * k) = *reinterpret_cast<const f32x4*>(t16 + i * 16 + 4 * k);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float acc = 0.0f;
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    const int p = pq >> 1, q = pq & 1;
                    const int a = p ? (ky == 2) : (ky != 0), b = q ? (kx == 2) : (kx != 0);
                    acc += v[pq * 4 + a * 2 + b];
                }
                dw[i * 9 + ky * 3 + kx] = acc;
            }
    }
}
extern "C" int eod_wgrad_up4_map(const float* t16, int Cout, int Cin, float* dw_oihw, void* stream) {
    EOD_REQUIRE(t16 && dw_oihw && Cout > 0 && Cin > 0 && eod_aligned16(t16), "wgrad_up4_map: bad args");
    const long long n = (long long)Cout * Cin;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(wgrad_up4_map_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t16, n, dw_oihw);
    EOD_CHECK_LAUNCH("wgrad_up4_map");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// bias / timestep-projection gradients straight from the NHWC gradient (no transposed copy): eod_gn_partial already
// produces per-(image, slab, channel) sums  part[n][p][c][0];  this reduces them:
//   dbias[c] = scale * sum_{n,p} part[n][p][c][0],   demb[n][c] = sum_p part[n][p][c][0]   (fixed order)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void channel_sums_finish_kernel(const float* __restrict__ part, int N, int P, int C, int cvalid, float scale,
                                                                   float* __restrict__ pern, float* __restrict__ demb, long long demb_ld) {
    // grid (64-channel blocks, N): one image per block row.  1024 threads = 64 channels x 16 slab segments: every thread sums its
    // contiguous share of the P slabs with FOUR loads in flight (four accumulators, combined in a fixed order), the 16 partial sums of
    // a channel are then combined in a FIXED order (deterministic); per-image totals go to pern[n][c] (scaled) for the bias gradient.
    // (A batch of 2 at 512 x 512 hands over 2048 slabs per image to FOUR workgroups: with 8 segments and one load in flight at a time this
    //  launch took 40-100 us, 5 % of that training step.)
    constexpr int SEG = 16;
    __shared__ float seg[SEG][64];
    const int cl = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
    const int per = (P + SEG - 1) / SEG, p0 = sg * per, p1 = min(P, p0 + per);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (c < cvalid) {
        const float* base = part + ((long long)n * P * C + c) * 2;
        const long long st = (long long)C * 2;
        int p = p0;
        for (; p + 3 < p1; p += 4) {
            const float v0 = base[(long long)p * st], v1 = base[(long long)(p + 1) * st], v2 = base[(long long)(p + 2) * st], v3 = base[(long long)(p + 3) * st];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; p < p1; ++p) a0 += base[(long long)p * st];
    }
    seg[sg][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sg == 0 && c < cvalid) {
        float t = seg[0][cl];
#pragma unroll
        for (int k = 1; k < SEG; ++k) t += seg[k][cl];
        if (demb) demb[(long long)n * demb_ld + c] = t;
        if (pern) pern[(long long)n * cvalid + c] = t * scale;
    }
}

extern "C" int eod_channel_sums_finish(const float* part, int N, int P, int C, int cvalid, float scale, float* dbias, float* demb,
                                       int64_t demb_ld, float* scratch, void* stream) {
    EOD_REQUIRE(part && N > 0 && P > 0 && C > 0 && cvalid > 0 && cvalid <= C && (dbias || demb) && N <= 65535, "channel_sums_finish: bad args");
    EOD_REQUIRE(!dbias || scratch, "channel_sums_finish: the bias gradient needs a scratch of N*cvalid floats");
    hipLaunchKernelGGL(channel_sums_finish_kernel, dim3((cvalid + 63) / 64, N), dim3(1024), 0, (hipStream_t)stream, part, N, P, C, cvalid, scale,
                       dbias ? scratch : nullptr, demb, (long long)demb_ld);
    if (dbias) hipLaunchKernelGGL(colsum_kernel, dim3((cvalid + 255) / 256), dim3(256), 0, (hipStream_t)stream, scratch, N, cvalid, dbias);
    EOD_CHECK_LAUNCH("channel_sums_finish");
    return EOD_OK;
}

// =============================================================================================
// gemm_tn_kernel (fp16): C[m][n] = alpha * sum_k A[k][m] * B[k][n]  with BOTH operands stored K-major (row k holds the m / n
// values contiguously) -- the layout of the attention backward's  dV = P^T dO  and  dK = dS^T Q  (P, dS are [query][key],
// dO / Q are [query][channel]): K = queries.  Same machinery as conv3x3_wgrad_kernel: 64-row K strips staged row-major by
// LDS-DMA (256-byte rows = 128 columns), operand fragments through ds_read_b64_tr_b16, so the T x T matrices are never
// transposed in HBM.  128 x 128 tile per workgroup (columns beyond N are zero-filled by the buffer bounds check), two-level
// batch strides, fp16 output.
// =============================================================================================
struct GemmTnP {
    const char* a;
    const char* b;
    char* c;
    long long lda, ldb, ldc, sa0, sa1, sb0, sb1, sc0, sc1;  // elements
    int M, N, K, nb1, tiles_m, tiles_n;
    int k_total;  // > 0: batch level 0 walks K-ranges of ONE operand pair (split-K): batch b0 owns rows [b0*K, min(k_total, (b0+1)*K))
    float alpha;
};

// F32OUT: fp32 C (the split-K partial tiles of the 1x1 backward-weights, eod_conv1x1_wgrad), alpha not applied
template <bool F32OUT>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const GemmTnP p) {
    constexpr int ROWS = 64, ROWB = 256, TILE = ROWS * ROWB, STAGE = 2 * TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int bidx = blockIdx.x;
    const int tile_n = bidx % p.tiles_n; bidx /= p.tiles_n;
    const int tile_m = bidx % p.tiles_m;
    const int z = bidx / p.tiles_m;
    const int b0 = z / p.nb1, b1 = z - b0 * p.nb1;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const char* A = p.a + (b0 * p.sa0 + b1 * p.sa1) * 2;
    const char* B = p.b + (b0 * p.sb0 + b1 * p.sb1) * 2;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(B), 0, 0x7fffffff, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int K = p.k_total > 0 ? max(0, min(p.K, p.k_total - b0 * p.K)) : p.K;
    const int drow = lane >> 4, dslot = lane & 15;
    auto issue = [&](int strip, int stage) {
        char* sa = smem + stage * STAGE;
        char* sb = sa + TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave + 4 * i) * 4 + drow;
            const int chunk = dslot ^ wg_swz(row);
            const long long k = (long long)strip * 64 + row;
            const int cm = m0 + chunk * 8, cn = n0 + chunk * 8;
            const unsigned va = (k < K && cm < p.M) ? (unsigned)((k * p.lda + cm) * 2) : OOB;
            const unsigned vb = (k < K && cn < p.N) ? (unsigned)((k * p.ldb + cn) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(sa + (wave + 4 * i) * 1024), 16, va, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(sb + (wave + 4 * i) * 1024), 16, vb, 0, 0, 0);
        }
    };
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    auto tr_addr = [&](int row, int col) { return row * ROWB + (((col >> 3) ^ wg_swz(row)) << 4) + ((col >> 2) & 1) * 8; };
    int a_ad[2][2], b_ad[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (g >> 1) * 8 + j * 4 + q;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a_ad[j][i] = tr_addr(row, wm * 64 + i * 32 + (g & 1) * 16 + 4 * pp);
            b_ad[j][i] = TILE + tr_addr(row, wn * 64 + i * 32 + (g & 1) * 16 + 4 * pp);
        }
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int strips = (K + 63) / 64;
    if (strips > 0) issue(0, 0);
    for (int s = 0; s < strips; ++s) {
        const int stage = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 1 < strips) issue(s + 1, stage ^ 1);
        const char* sb = smem + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            half8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const fp16x4_t alo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(sb + a_ad[0][i] + ks * 4096));
                const fp16x4_t ahi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(sb + a_ad[1][i] + ks * 4096));
                const fp16x4_t blo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(sb + b_ad[0][i] + ks * 4096));
                const fp16x4_t bhi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(sb + b_ad[1][i] + ks * 4096));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    fa[i][e] = (half_t)alo[e];
                    fa[i][4 + e] = (half_t)ahi[e];
                    fb[i][e] = (half_t)blo[e];
                    fb[i][4 + e] = (half_t)bhi[e];
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    half_t* C = reinterpret_cast<half_t*>(p.c) + (F32OUT ? 2 : 1) * (b0 * p.sc0 + b1 * p.sc1);
    float* Cf = reinterpret_cast<float*>(C);
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M && n < p.N) {
                    if constexpr (F32OUT)
                        Cf[(long long)m * p.ldc + n] = acc[i][j][r];
                    else
                        C[(long long)m * p.ldc + n] = (half_t)(acc[i][j][r] * p.alpha);
                }
            }
        }
}

extern "C" int eod_gemm_tn(const void* a, int64_t lda, const void* b, int64_t ldb, void* c, int64_t ldc, int dtype, int M, int N, int K,
                           float alpha, int nb0, int nb1, int64_t sa0, int64_t sa1, int64_t sb0, int64_t sb1, int64_t sc0, int64_t sc1,
                           void* stream) {
    EOD_REQUIRE(a && b && c && M > 0 && N > 0 && K > 0 && nb0 > 0 && nb1 > 0, "gemm_tn: bad args");
    EOD_REQUIRE(dtype == EOD_F16, "gemm_tn: fp16 only (transposed LDS reads are 16-bit)");
    EOD_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && M % 8 == 0 && N % 8 == 0 && sa0 % 8 == 0 && sa1 % 8 == 0 && sb0 % 8 == 0 && sb1 % 8 == 0 &&
                    eod_aligned16(a) && eod_aligned16(b),
                "gemm_tn: leading dimensions, M, N and batch strides must be multiples of 8 elements, pointers 16-byte aligned");
    EOD_REQUIRE((long long)K * lda * 2 < 0x7fffffffLL && (long long)K * ldb * 2 < 0x7fffffffLL, "gemm_tn: one batch operand exceeds the 2 GiB window");
    GemmTnP p;
    p.a = (const char*)a; p.b = (const char*)b; p.c = (char*)c;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.sa0 = sa0; p.sa1 = sa1; p.sb0 = sb0; p.sb1 = sb1; p.sc0 = sc0; p.sc1 = sc1;
    p.M = M; p.N = N; p.K = K; p.nb1 = nb1; p.alpha = alpha; p.k_total = 0;
    p.tiles_m = (M + 127) / 128;
    p.tiles_n = (N + 127) / 128;
    const long long grid = (long long)p.tiles_m * p.tiles_n * nb0 * nb1;
    EOD_REQUIRE(grid <= 0x7fffffffLL, "gemm_tn: grid too large");
    const size_t lds = 2 * 2 * 64 * 256;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, p);
    EOD_CHECK_LAUNCH("gemm_tn");
    return EOD_OK;
}

// backward-weights of a 1x1 / stride-1 conv (the skip connections and the attention projections) with the same machinery:
//   dW[co][ci] = sum_pix dY[pix][co] * X[pix][ci]      (both tensors pixel-major as stored: no transposed copies in HBM)
// split over S pixel ranges; fp32 partial tiles partial[split][co][ldp] (the layout eod_wgrad_reduce takes with ksize 1).
extern "C" int eod_conv1x1_wgrad(const void* dy, const void* x, int dtype, int64_t npix, int Cx, int Cy, int Cout, float* partial, int ldp,
                                 int S, void* stream) {
    EOD_REQUIRE(dy && x && partial && npix > 0 && Cx > 0 && Cy > 0 && Cout > 0 && Cout <= Cy && S > 0 && ldp >= Cx, "conv1x1_wgrad: bad args");
    EOD_REQUIRE(dtype == EOD_F16, "conv1x1_wgrad: fp16 only (transposed LDS reads are 16-bit)");
    EOD_REQUIRE(Cx % 8 == 0 && Cy % 8 == 0 && eod_aligned16(dy) && eod_aligned16(x), "conv1x1_wgrad: channel counts must be multiples of 8, pointers 16-byte aligned");
    EOD_REQUIRE(npix * Cy * 2 < 0x7fffffffLL && npix * Cx * 2 < 0x7fffffffLL, "conv1x1_wgrad: tensors exceed the 2 GiB buffer window");
    const long long strips = (npix + 63) / 64;
    const long long per = (strips + S - 1) / S;
    GemmTnP p;
    p.a = (const char*)dy; p.b = (const char*)x; p.c = (char*)partial;
    p.lda = Cy; p.ldb = Cx; p.ldc = ldp;
    p.K = (int)(per * 64); p.k_total = (int)npix;
    p.sa0 = (long long)p.K * Cy; p.sb0 = (long long)p.K * Cx; p.sc0 = (long long)Cout * ldp;
    p.sa1 = p.sb1 = p.sc1 = 0;
    p.M = Cout; p.N = Cx; p.nb1 = 1; p.alpha = 1.0f;
    p.tiles_m = (Cout + 127) / 128;
    p.tiles_n = (Cx + 127) / 128;
    const long long grid = (long long)p.tiles_m * p.tiles_n * S;
    EOD_REQUIRE(grid <= 0x7fffffffLL, "conv1x1_wgrad: grid too large");
    const size_t lds = 2 * 2 * 64 * 256;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm_tn_kernel<true>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, p);
    EOD_CHECK_LAUNCH("conv1x1_wgrad");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// D[i0][i1][i2] = sum_{j<d} a[off + j] * b[off + j],  off = i0*s0 + i1*s1 + i2*s2: per-(image, head, query) dot product of
// the attention output and its gradient (= rowsum(dP * P), the softmax-backward row term, without forming dP)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void rowdot_kernel(const T* __restrict__ a, const T* __restrict__ b, long long n0, long long n1, long long n2, long long s0, long long s1,
                              long long s2, int d, float* __restrict__ out) {
    const long long total = n0 * n1 * n2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long i2 = i % n2, r = i / n2;
        const long long i1 = r % n1, i0 = r / n1;
        const long long off = i0 * s0 + i1 * s1 + i2 * s2;
        float acc = 0.0f;
        for (int j = 0; j < d; ++j) acc += (float)a[off + j] * (float)b[off + j];
        out[i] = acc;
    }
}

extern "C" int eod_rowdot(const void* a, const void* b, int dtype, int64_t n0, int64_t n1, int64_t n2, int64_t s0, int64_t s1, int64_t s2, int d,
                          float* out, void* stream) {
    EOD_REQUIRE(a && b && out && n0 > 0 && n1 > 0 && n2 > 0 && d > 0, "rowdot: bad args");
    const long long total = (long long)n0 * n1 * n2;
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(rowdot_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const half_t*)a, (const half_t*)b, (long long)n0, (long long)n1, (long long)n2, (long long)s0, (long long)s1, (long long)s2, d, out);
    else
        hipLaunchKernelGGL(rowdot_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (long long)n0, (long long)n1, (long long)n2, (long long)s0, (long long)s1, (long long)s2, d, out);
    EOD_CHECK_LAUNCH("rowdot");
    return EOD_OK;
}

// x *= s (fp32, in place): puts the softmax-backward row term D on the scale of the scaled dS (see training.py: _attn_bwd)
__global__ void scale_f32_kernel(float* __restrict__ x, long long n, float s) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= s;
}
extern "C" int eod_scale_f32(float* x, int64_t n, float s, void* stream) {
    EOD_REQUIRE(x && n > 0, "scale_f32: bad args");
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n, s);
    EOD_CHECK_LAUNCH("scale_f32");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// nn.Dropout (ResBlock.out_layers[2], unet_openai.py:339): y = x * keep / (1 - p), keep ~ Bernoulli(1 - p).
// The mask is a pure function of (seed, layer, step, element) through Philox4x32-10 -- the backward re-derives it from the
// same key (same kernel applied to the gradient), nothing is stored.  One Philox call = 4 elements.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void tr_philox4(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, unsigned keep_thr, float scale, unsigned long long seed,
                               unsigned layer, unsigned step) {
    const long long quads = (n + 3) / 4;
    for (long long qd = (long long)blockIdx.x * blockDim.x + threadIdx.x; qd < quads; qd += (long long)gridDim.x * blockDim.x) {
        unsigned r[4];
        tr_philox4((unsigned)qd, (unsigned)(qd >> 32), layer, step, (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long e = qd * 4 + j;
            if (e < n) y[e] = (T)(r[j] < keep_thr ? (float)x[e] * scale : 0.0f);
        }
    }
}

extern "C" int eod_dropout(const void* x, void* y, int dtype, int64_t n, float p, uint64_t seed, uint32_t layer, uint32_t step, void* stream) {
    EOD_REQUIRE(x && y && n > 0 && p >= 0.0f && p < 1.0f, "dropout: bad args (0 <= p < 1)");
    const double keep = 1.0 - (double)p;
    const unsigned thr = keep >= 1.0 ? 0xFFFFFFFFu : (unsigned)(keep * 4294967296.0);
    long long blocks = ((n + 3) / 4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(dropout_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)y, (long long)n, thr, (float)(1.0 / keep), (unsigned long long)seed, layer, step);
    else
        hipLaunchKernelGGL(dropout_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, (long long)n, thr, (float)(1.0 / keep), (unsigned long long)seed, layer, step);
    EOD_CHECK_LAUNCH("dropout");
    return EOD_OK;
}
