// Small HBM-bound helpers: weight packing, API-edge layout conversion, timestep-embedding MLP,
// 2x resampling for resblock_updown.
#include "common.h"

template <typename T> __device__ __forceinline__ T cvt(float v) { return (T)v; }

// ---------------------------------------------------------------------------------------------
// OIHW fp32 -> [tap][Cout][cin_pad] (zero padded input channels)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_conv_w_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int taps, int cin_pad) {
    const long long total = (long long)taps * Cout * cin_pad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin_pad);
        const long long r = i / cin_pad;
        const int co = (int)(r % Cout);
        const int tap = (int)(r / Cout);
        const float v = ci < Cin ? w[((long long)co * Cin + ci) * taps + tap] : 0.0f;
        dst[i] = cvt<T>(v);
    }
}

extern "C" int eod_pack_conv_weight(const float* w, void* dst, int dtype, int Cout, int Cin, int ksize, int cin_pad,
                                    void* stream) {
    EOD_REQUIRE(w && dst && Cout > 0 && Cin > 0 && cin_pad >= Cin && (ksize == 1 || ksize == 3), "pack_conv_weight: bad args");
    const int taps = ksize * ksize;
    const long long total = (long long)taps * Cout * cin_pad;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(pack_conv_w_kernel<half_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (half_t*)dst, Cout, Cin, taps, cin_pad);
    else
        hipLaunchKernelGGL(pack_conv_w_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)dst, Cout, Cin, taps, cin_pad);
    EOD_CHECK_LAUNCH("pack_conv_weight");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// split-fp16 weights of the "fp32x3" mode (csrc/igemm.hip): OIHW fp32 -> [tap][Cout][cin_pad] at 4 bytes per element, every group
// of 8 input channels stored as [8 x hi | 8 x lo] fp16 with hi = fp16(s_j*w), lo = fp16(s_j*w - hi); s_j = s * 2^d_j (row part below), s = 2^k per tensor such that
// max|w|*s lies in (2^12, 2^13].  scale[0] = s, scale[1] = 1 / (s * 16) (what the conv epilogue multiplies by: weight scale and the
// activation scale of 16).  Two launches, no host round trip: the scale is read from device memory by the pack kernel and by the conv.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void absmax_scale_kernel(const float* __restrict__ w, long long n, float* __restrict__ scale,
                                                            const float* __restrict__ w2 = nullptr, long long n2 = 0) {
    __shared__ float red[16];
    float m = 0.0f;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(w[i]));
    for (long long i = threadIdx.x; i < n2; i += blockDim.x) m = fmaxf(m, fabsf(w2[i]));  // (a second tensor sharing the scale)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
        float s = 1.0f;
        if (m > 0.0f && m < 3.0e38f) {
            int e;
            frexpf(m, &e);           // m = f * 2^e, f in [0.5, 1)  ->  m * 2^(13 - e) in [2^12, 2^13)
            s = ldexpf(1.0f, 13 - e);
        }
        scale[0] = s;
        scale[1] = 1.0f / (s * 16.0f);
    }
}

// Per-ROW part of the scale (round 3): output row j (one output channel: Cin * taps consecutive OIHW values) is packed under
// s_j = s * 2^d_j with d_j = E - e_j >= 0 the distance of the row's binary exponent e_j from the tensor's E, so that EVERY row's largest
// value lies in (2^12, 2^13] and every row keeps its own 22 bits -- with the tensor scale alone a row 2^18 times smaller than the
// largest one starts to lose bits (fp16 subnormals) and an output channel (a whole GroupNorm group, once renormalised) can be off by
// 1e-4 ... 1 where the reference's fp32 product is exact.  d_j (int32, clamped to min(100, E + 109), 0 for an all-zero row) is stored at
// scale[EOD_WSCALE_ROWS + j]; the conv epilogues multiply column j by 2^-d_j on top of scale[1].  One wave per row.
constexpr int EOD_WSCALE_ROWS = 4;  // float slots in front of the per-row exponents (keeps them 16-byte aligned)
__global__ __launch_bounds__(256) void row_exp_kernel(const float* __restrict__ w, long long row_len, const float* __restrict__ w2,
                                                      long long row_len2, int rows, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= rows) return;
    float m = 0.0f;
    for (long long i = lane; i < row_len; i += 64) m = fmaxf(m, fabsf(w[(long long)j * row_len + i]));
    for (long long i = lane; i < row_len2; i += 64) m = fmaxf(m, fabsf(w2[(long long)j * row_len2 + i]));  // (a second tensor sharing the scale)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) {
        int d = 0;
        if (m > 0.0f && m < 3.0e38f) {
            int e;
            frexpf(m, &e);
            const int E = 13 - ilogbf(scale[0]);  // scale[0] = 2^(13 - E)
            // d <= E + 109 keeps the row's pack scale 2^(13 - E + d) finite and the epilogue's 2^(E - 17 - d) a normal fp32 number when
            // the whole tensor is tiny (max |w| < 2^-9: E < -9); rows further down lose bits gradually, as fp32 itself does there
            d = min(max(E - e, 0), min(100, max(E + 109, 0)));
        }
        reinterpret_cast<int*>(scale)[EOD_WSCALE_ROWS + j] = d;
    }
}

__global__ void pack_conv_w_split_kernel(const float* __restrict__ w, half_t* __restrict__ dst, const float* __restrict__ scale, int Cout,
                                         int Cin, int taps, int cin_pad) {
    const float s = scale[0];
    const int* rexp = reinterpret_cast<const int*>(scale) + EOD_WSCALE_ROWS;
    const long long groups = (long long)taps * Cout * (cin_pad / 8);
    for (long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(gi % (cin_pad / 8));
        const long long r = gi / (cin_pad / 8);
        const int co = (int)(r % Cout), tap = (int)(r / Cout);
        const float sj = ldexpf(s, rexp[co]);
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = cg * 8 + j;
            const float v = ci < Cin ? w[((long long)co * Cin + ci) * taps + tap] * sj : 0.0f;
            hi[j] = (half_t)v;
            lo[j] = (half_t)(v - (float)hi[j]);
        }
        half8* o = reinterpret_cast<half8*>(dst + gi * 16);
        o[0] = hi;
        o[1] = lo;
    }
}

extern "C" int eod_pack_conv_weight_split(const float* w, void* dst, float* scale, int Cout, int Cin, int ksize, int cin_pad, void* stream) {
    EOD_REQUIRE(w && dst && scale && Cout > 0 && Cin > 0 && cin_pad >= Cin && cin_pad % 8 == 0 && (ksize == 1 || ksize == 3),
                "pack_conv_weight_split: bad args (cin_pad must be a multiple of 8)");
    EOD_REQUIRE(eod_aligned16(dst), "pack_conv_weight_split: dst must be 16-byte aligned");
    const int taps = ksize * ksize;
    hipLaunchKernelGGL(absmax_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (long long)Cout * Cin * taps, scale);
    hipLaunchKernelGGL(row_exp_kernel, dim3((Cout + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, (long long)Cin * taps, nullptr, 0LL, Cout, scale);
    const long long groups = (long long)taps * Cout * (cin_pad / 8);
    const unsigned blocks = (unsigned)((groups + 255) / 256 > 4096 ? 4096 : (groups + 255) / 256);
    hipLaunchKernelGGL(pack_conv_w_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (half_t*)dst, scale, Cout, Cin, taps, cin_pad);
    EOD_CHECK_LAUNCH("pack_conv_weight_split");
    return EOD_OK;
}

extern "C" int eod_pack_conv_weight_split_pair(const float* w, void* dst, const float* w2, void* dst2, float* scale, int Cout, int Cin,
                                               int ksize, int cin_pad, int Cin2, void* stream) {
    EOD_REQUIRE(w && dst && w2 && dst2 && scale && Cout > 0 && Cin > 0 && Cin2 > 0 && cin_pad >= Cin && cin_pad % 8 == 0 && Cin2 % 8 == 0 &&
                    (ksize == 1 || ksize == 3),
                "pack_conv_weight_split_pair: bad args (cin_pad and Cin2 must be multiples of 8)");
    EOD_REQUIRE(eod_aligned16(dst) && eod_aligned16(dst2), "pack_conv_weight_split_pair: dst must be 16-byte aligned");
    const int taps = ksize * ksize;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(absmax_scale_kernel, dim3(1), dim3(1024), 0, st, w, (long long)Cout * Cin * taps, scale, w2, (long long)Cout * Cin2);
    hipLaunchKernelGGL(row_exp_kernel, dim3((Cout + 3) / 4), dim3(256), 0, st, w, (long long)Cin * taps, w2, (long long)Cin2, Cout, scale);
    const long long g1 = (long long)taps * Cout * (cin_pad / 8), g2 = (long long)Cout * (Cin2 / 8);
    const unsigned b1 = (unsigned)((g1 + 255) / 256 > 4096 ? 4096 : (g1 + 255) / 256), b2 = (unsigned)((g2 + 255) / 256 > 4096 ? 4096 : (g2 + 255) / 256);
    hipLaunchKernelGGL(pack_conv_w_split_kernel, dim3(b1), dim3(256), 0, st, w, (half_t*)dst, scale, Cout, Cin, taps, cin_pad);
    hipLaunchKernelGGL(pack_conv_w_split_kernel, dim3(b2), dim3(256), 0, st, w2, (half_t*)dst2, scale, Cout, Cin2, 1, Cin2);
    EOD_CHECK_LAUNCH("pack_conv_weight_split_pair");
    return EOD_OK;
}

// class kernels of the parity-class form of a conv over a nearest-2x upsampling (eod_conv_up4_ok): wc [4*Cout][Cin][3][3] from
// w [Cout][Cin][3][3]; row block 2p+q holds class (p, q)'s 2x2 kernel in tap slots dy' in {p, p+1}, dx' in {q, q+1}: per axis
// class 0 -> slots (w0 | w1+w2 | 0), class 1 -> (0 | w0+w1 | w2).  One thread per (co, ci).
__global__ void up4_weight_sums_kernel(const float* __restrict__ w, float* __restrict__ wc, int Cout, int Cin) {
    const long long n = (long long)Cout * Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t / 3][t % 3] = w[i * 9 + t];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float r[3][3];  // rows summed for class p
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                r[0][x] = p ? 0.0f : v[0][x];
                r[1][x] = p ? v[0][x] + v[1][x] : v[1][x] + v[2][x];
                r[2][x] = p ? v[2][x] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float* o = wc + ((long long)(2 * p + q) * n + i) * 9;
#pragma unroll
                for (int y = 0; y < 3; ++y) {
                    o[y * 3 + 0] = q ? 0.0f : r[y][0];
                    o[y * 3 + 1] = q ? r[y][0] + r[y][1] : r[y][1] + r[y][2];
                    o[y * 3 + 2] = q ? r[y][2] : 0.0f;
                }
            }
        }
    }
}
extern "C" int eod_conv_up4_weights(const float* w_oihw, float* wc, int Cout, int Cin, void* stream) {
    EOD_REQUIRE(w_oihw && wc && Cout > 0 && Cin > 0, "conv_up4_weights: bad args");
    const long long n = (long long)Cout * Cin;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(up4_weight_sums_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, wc, Cout, Cin);
    EOD_CHECK_LAUNCH("conv_up4_weights");
    return EOD_OK;
}

// OIHW fp32 -> [Cout][ldk], k = tap*cin_pad + c (thin-input first conv, eod_conv_desc.w_tapmajor)
template <typename T>
__global__ void pack_conv_w_tapmajor_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int cin_pad, int ldk) {
    const long long total = (long long)Cout * ldk;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % ldk), co = (int)(i / ldk);
        const int tap = k / cin_pad, ci = k - tap * cin_pad;
        const float v = (tap < 9 && ci < Cin) ? w[((long long)co * Cin + ci) * 9 + tap] : 0.0f;
        dst[i] = cvt<T>(v);
    }
}

// the same in the split-fp16 format of the fp32x3 product (every 8 consecutive k = [8 x fp16 hi | 8 x fp16 lo] of s_j*w, s_j = s * 2^d_j as above)
__global__ void pack_conv_w_tapmajor_split_kernel(const float* __restrict__ w, half_t* __restrict__ dst, const float* __restrict__ scale,
                                                  int Cout, int Cin, int cin_pad, int ldk) {
    const float s = scale[0];
    const int* rexp = reinterpret_cast<const int*>(scale) + EOD_WSCALE_ROWS;
    const long long groups = (long long)Cout * (ldk / 8);
    for (long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (long long)gridDim.x * blockDim.x) {
        const int kg = (int)(gi % (ldk / 8)), co = (int)(gi / (ldk / 8));
        const float sj = ldexpf(s, rexp[co]);
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kg * 8 + j, tap = k / cin_pad, ci = k - tap * cin_pad;
            const float v = (tap < 9 && ci < Cin) ? w[((long long)co * Cin + ci) * 9 + tap] * sj : 0.0f;
            hi[j] = (half_t)v;
            lo[j] = (half_t)(v - (float)hi[j]);
        }
        half8* o = reinterpret_cast<half8*>(dst + gi * 16);
        o[0] = hi;
        o[1] = lo;
    }
}

extern "C" int eod_conv_tapmajor_ldk(int C0, int dtype);
extern "C" int eod_pack_conv_weight_tapmajor(const float* w, void* dst, int dtype, int Cout, int Cin, int cin_pad, void* stream) {
    EOD_REQUIRE(w && dst && Cout > 0 && Cin > 0 && cin_pad >= Cin, "pack_conv_weight_tapmajor: bad args");
    EOD_REQUIRE(dtype == EOD_F16 || dtype == EOD_F32, "pack_conv_weight_tapmajor: bad dtype %d", dtype);
    const int ldk = eod_conv_tapmajor_ldk(cin_pad, dtype);
    const long long total = (long long)Cout * ldk;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(pack_conv_w_tapmajor_kernel<half_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (half_t*)dst, Cout, Cin, cin_pad, ldk);
    else
        hipLaunchKernelGGL(pack_conv_w_tapmajor_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)dst, Cout, Cin, cin_pad, ldk);
    EOD_CHECK_LAUNCH("pack_conv_weight_tapmajor");
    return EOD_OK;
}

extern "C" int eod_pack_conv_weight_tapmajor_split(const float* w, void* dst, float* scale, int Cout, int Cin, int cin_pad, void* stream) {
    EOD_REQUIRE(w && dst && scale && Cout > 0 && Cin > 0 && cin_pad >= Cin && eod_aligned16(dst), "pack_conv_weight_tapmajor_split: bad args");
    const int ldk = eod_conv_tapmajor_ldk(cin_pad, EOD_F32);
    hipLaunchKernelGGL(absmax_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (long long)Cout * Cin * 9, scale);
    hipLaunchKernelGGL(row_exp_kernel, dim3((Cout + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, (long long)Cin * 9, nullptr, 0LL, Cout, scale);
    const long long groups = (long long)Cout * (ldk / 8);
    const unsigned blocks = (unsigned)((groups + 255) / 256 > 4096 ? 4096 : (groups + 255) / 256);
    hipLaunchKernelGGL(pack_conv_w_tapmajor_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (half_t*)dst, scale, Cout, Cin, cin_pad, ldk);
    EOD_CHECK_LAUNCH("pack_conv_weight_tapmajor_split");
    return EOD_OK;
}

template <typename T>
__global__ void pack_rows_kernel(const float* __restrict__ src, long long ld_src, const int* __restrict__ row_map,
                                 T* __restrict__ dst, long long ld_dst, int rows, int cols) {
    const long long total = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        const int sr = row_map ? row_map[r] : r;  // a negative source row produces a zero row (padding)
        dst[(long long)r * ld_dst + c] = cvt<T>(sr >= 0 ? src[(long long)sr * ld_src + c] : 0.0f);
    }
}

extern "C" int eod_pack_rows(const float* src, int64_t ld_src, const int32_t* row_map, void* dst, int64_t ld_dst, int dtype,
                             int rows, int cols, void* stream) {
    EOD_REQUIRE(src && dst && rows > 0 && cols > 0, "pack_rows: bad args");
    const long long total = (long long)rows * cols;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(pack_rows_kernel<half_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (long long)ld_src, row_map, (half_t*)dst, (long long)ld_dst, rows, cols);
    else
        hipLaunchKernelGGL(pack_rows_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (long long)ld_src, row_map, (float*)dst, (long long)ld_dst, rows, cols);
    EOD_CHECK_LAUNCH("pack_rows");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// NCHW fp32 (two concatenated sources) -> NHWC storage dtype, channels zero-padded to c_pad.
// One thread per (n, pixel): reads are coalesced along W per channel plane, writes are 16-byte chunks.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ s0, int C0, const float* __restrict__ s1, int C1, T* __restrict__ dst,
                                    int N, long long HW, int c_pad) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW, pix = i - n * HW;
        T* d = dst + i * c_pad;
        for (int c = 0; c < c_pad; ++c) {
            float v = 0.0f;
            if (c < C0)
                v = s0[(n * C0 + c) * HW + pix];
            else if (c < C0 + C1)
                v = s1[(n * C1 + (c - C0)) * HW + pix];
            d[c] = cvt<T>(v);
        }
    }
}

extern "C" int eod_nchw_to_nhwc(const float* src0, int C0, const float* src1, int C1, void* dst, int dtype, int N, int H, int W,
                                int c_pad, void* stream) {
    EOD_REQUIRE(src0 && dst && C0 > 0 && C1 >= 0 && (C1 == 0 || src1) && c_pad >= C0 + C1, "nchw_to_nhwc: bad args");
    const long long total = (long long)N * H * W;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<half_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src0, C0, src1, C1, (half_t*)dst, N, (long long)H * W, c_pad);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src0, C0, src1, C1, (float*)dst, N, (long long)H * W, c_pad);
    EOD_CHECK_LAUNCH("nchw_to_nhwc");
    return EOD_OK;
}

// NHWC storage -> NCHW fp32 through a 32x32 LDS transpose tile (pixels x channels)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, long long HW, int C) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const long long p0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const long long pix = p0 + r;
        const int c = c0 + tx;
        tile[r][tx] = (pix < HW && c < C) ? (float)src[((long long)n * HW + pix) * C + c] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const long long pix = p0 + tx;
        if (pix < HW && c < C) dst[((long long)n * C + c) * HW + pix] = tile[tx][r];
    }
}

extern "C" int eod_nhwc_to_nchw(const void* src, int dtype, float* dst, int N, int H, int W, int C, void* stream) {
    EOD_REQUIRE(src && dst && N > 0 && C > 0, "nhwc_to_nchw: bad args");
    const long long HW = (long long)H * W;
    dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N), block(32, 8);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<half_t>, grid, block, 0, (hipStream_t)stream, (const half_t*)src, dst, HW, C);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)src, dst, HW, C);
    EOD_CHECK_LAUNCH("nhwc_to_nchw");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// 2x resampling on NHWC (resblock_updown, unet_openai.py:320-325): mode 0 avg-pool 2x2, 1 nearest 2x,
// 2 = 2x2 SUM pool (backward of nearest 2x), 3 = nearest 2x times 0.25 (backward of the average pool)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void resample2x_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int mode, int pad_tl) {
    // pad_tl: mode 1 = zero row / column on top / left of the 2x image (3x3 -> 7x7 hack); mode 2 = the backward of that (the sums
    // skip the first row / column of the input); mode 3 = the average pool dropped a last odd row / column: one zero row / column
    // at the BOTTOM / RIGHT of the output
    // mode 4 = crop: drops the last row (pad_tl & 1) and / or column (pad_tl & 2) -- the backward-data conv of a stride-2 conv on
    // an odd map is computed on the even (H+1) x (W+1) grid
    const bool up = mode == 1 || mode == 3;
    int Ho = up ? 2 * H + pad_tl : (H - (mode == 2 ? pad_tl : 0)) / 2, Wo = up ? 2 * W + pad_tl : (W - (mode == 2 ? pad_tl : 0)) / 2;
    if (mode == 4) {
        Ho = H - (pad_tl & 1);
        Wo = W - ((pad_tl >> 1) & 1);
    }
    const long long total = (long long)N * Ho * Wo * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int wo = (int)(r % Wo);
        r /= Wo;
        const int ho = (int)(r % Ho);
        const long long n = r / Ho;
        float v;
        if (mode == 4) {
            v = (float)x[((n * H + ho) * W + wo) * C + c];
        } else if (up) {
            const int hi = mode == 1 ? ho - pad_tl : ho, wi = mode == 1 ? wo - pad_tl : wo;
            v = (hi >= 0 && wi >= 0 && hi < 2 * H && wi < 2 * W) ? (float)x[((n * H + (hi >> 1)) * W + (wi >> 1)) * C + c] : 0.0f;
            if (mode == 3) v *= 0.25f;
        } else {
            const int o = mode == 2 ? pad_tl : 0;
            const T* b = x + ((n * H + 2 * ho + o) * W + 2 * wo + o) * C + c;
            // same summation order as ATen's avg_pool2d (row-major window), then * 0.25
            v = ((float)b[0] + (float)b[C] + (float)b[(long long)W * C] + (float)b[(long long)W * C + C]) * (mode == 2 ? 1.0f : 0.25f);
        }
        y[i] = cvt<T>(v);
    }
}

extern "C" int eod_resample2x(const void* x, int dtype, int N, int H, int W, int C, int mode, int pad_tl, void* y, void* stream) {
    EOD_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, "resample2x: bad args");
    EOD_REQUIRE(mode >= 0 && mode <= 4, "resample2x: mode %d", mode);
    const bool up = mode == 1 || mode == 3;
    EOD_REQUIRE(up || (H >= 2 && W >= 2), "resample2x: avg-pool needs at least 2x2 (%dx%d)", H, W);  // odd dims: floor, like ATen
    const int o2 = mode == 2 ? pad_tl : 0;
    long long total = (long long)N * (up ? 2 * H + pad_tl : (H - o2) / 2) * (up ? 2 * W + pad_tl : (W - o2) / 2) * C;
    if (mode == 4) total = (long long)N * (H - (pad_tl & 1)) * (W - ((pad_tl >> 1) & 1)) * C;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(resample2x_kernel<half_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)y, N, H, W, C, mode, pad_tl);
    else
        hipLaunchKernelGGL(resample2x_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, N, H, W, C, mode, pad_tl);
    EOD_CHECK_LAUNCH("resample2x");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// timestep embedding MLP (fp32).  One wave per output column j; lanes stride over k (coalesced
// weight-row reads), loop over the N samples; wave reduction.  MODE 0: input = sinusoid(t) built in
// LDS, output silu(.) ; MODE 1: plain ; MODE 2: input silu(.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int MODE>
__global__ void temb_linear_kernel(const float* __restrict__ in, const long long* __restrict__ t, const float* __restrict__ freqs,
                                   const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ label_emb,
                                   const long long* __restrict__ y, float* __restrict__ out, float* __restrict__ out2, int N, int K,
                                   int J, int t_f32 = 0) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= J) return;
    const float* wr = w + (long long)j * K;
    for (int n = 0; n < N; ++n) {
        float acc = 0.0f;
        if (MODE == 0) {
            // timestep_embedding (unet_openai.py:91-99): [cos(t*f) | sin(t*f)] (+ one zero if K is odd)
            const int half = K / 2;
            const float tf = t_f32 ? reinterpret_cast<const float*>(t)[n] : (float)t[n];
            for (int k = lane; k < K; k += 64) {
                float e = 0.0f;
                if (k < half)
                    e = cosf(mul_rn(tf, freqs[k]));
                else if (k < 2 * half)
                    e = sinf(mul_rn(tf, freqs[k - half]));
                acc += e * wr[k];
            }
        } else {
            const float* ir = in + (long long)n * K;
            for (int k = lane; k < K; k += 64) {
                float e = ir[k];
                if (MODE == 2) e = silu_f<false>(e);
                acc += e * wr[k];
            }
        }
        acc = wsum(acc);
        if (lane == 0) {
            float v = acc + b[j];
            if (MODE == 0) v = silu_f<false>(v);
            if (MODE == 1 && label_emb) v += label_emb[y[n] * J + j];
            out[(long long)n * J + j] = v;
        }
    }
}

// Stage 1 of the embedding MLP with the sinusoid table built ONCE per block in LDS (the generic kernel above would
// re-evaluate the N x D precise sin/cos for every output column: ~100 us of pure range reduction at E = 512).
// Block = 4 waves x OPW output columns; summation order per column identical to temb_linear_kernel<0>.
constexpr int TEMB_NB = 16;   // images per LDS table pass
constexpr int TEMB_OPW = 4;   // output columns per wave
constexpr int TEMB_KI = 8;    // K <= 64 * TEMB_KI for the table kernel
// STAGE 1: table = sinusoid(t), epilogue SiLU (time_embed[0:2]).  STAGE 2: table = h1, epilogue + label_emb[y]
// (time_embed[2], :597-605).  STAGE 3: table = SiLU(emb), plain epilogue (every ResBlock's emb_layers = SiLU -> Linear,
// concatenated along J).  Each lane keeps TEMB_NB independent accumulators (one per image), so the loop is not a
// chain of dependent load -> reduce -> store round trips per image.
template <int STAGE>
__global__ __launch_bounds__(256) void temb_table_linear_kernel(const float* __restrict__ in, const long long* __restrict__ t,
                                                                const float* __restrict__ freqs, const float* __restrict__ w,
                                                                const float* __restrict__ b, const float* __restrict__ label_emb,
                                                                const long long* __restrict__ y, float* __restrict__ out, int N,
                                                                int K, int J, int t_f32 = 0) {
    extern __shared__ float tab[];  // [TEMB_NB][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = K / 2;
    for (int n0 = 0; n0 < N; n0 += TEMB_NB) {
        const int nb = min(TEMB_NB, N - n0);
        __syncthreads();
        for (int i = threadIdx.x; i < TEMB_NB * K; i += blockDim.x) {
            const int n = i / K, k = i - n * K;
            float e = 0.0f;
            if (n < nb) {
                if (STAGE == 1) {
                    const float tf = t_f32 ? reinterpret_cast<const float*>(t)[n0 + n] : (float)t[n0 + n];
                    if (k < half)
                        e = cosf(mul_rn(tf, freqs[k]));
                    else if (k < 2 * half)
                        e = sinf(mul_rn(tf, freqs[k - half]));
                } else {
                    e = in[(long long)(n0 + n) * K + k];
                    if (STAGE == 3) e = silu_f<false>(e);
                }
            }
            tab[i] = e;
        }
        __syncthreads();
        // all weight values this wave needs (TEMB_OPW rows x K/64 per lane) are fetched up front: one memory round trip
        // instead of one per k iteration
        float wv[TEMB_OPW][TEMB_KI];
#pragma unroll
        for (int o = 0; o < TEMB_OPW; ++o) {
            const int j = (blockIdx.x * 4 + wave) * TEMB_OPW + o;
#pragma unroll
            for (int i = 0; i < TEMB_KI; ++i) {
                const int k = lane + 64 * i;
                wv[o][i] = (j < J && k < K) ? w[(long long)j * K + k] : 0.0f;
            }
        }
#pragma unroll
        for (int o = 0; o < TEMB_OPW; ++o) {
            const int j = (blockIdx.x * 4 + wave) * TEMB_OPW + o;
            if (j >= J) break;  // wave-uniform
            float acc[TEMB_NB];
#pragma unroll
            for (int n = 0; n < TEMB_NB; ++n) acc[n] = 0.0f;
#pragma unroll
            for (int i = 0; i < TEMB_KI; ++i) {
                const int k = lane + 64 * i;
                if (k < K) {
#pragma unroll
                    for (int n = 0; n < TEMB_NB; ++n) acc[n] += tab[n * K + k] * wv[o][i];
                }
            }
            float mine = 0.0f;  // lane n keeps image n's sum
#pragma unroll
            for (int n = 0; n < TEMB_NB; ++n) {
                const float r = wsum(acc[n]);
                if (lane == n) mine = r;
            }
            if (lane < nb) {
                float v = mine + b[j];
                if (STAGE == 1) v = silu_f<false>(v);
                if (STAGE == 2 && label_emb) v += label_emb[y[n0 + lane] * J + j];
                out[(long long)(n0 + lane) * J + j] = v;
            }
        }
    }
}

// standalone timestep_embedding (unet_openai.py:81-99): out[n] = [cos(t_n f) | sin(t_n f)] (+ one zero column when dim is odd)
__global__ void timestep_embedding_kernel(const float* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out, int N,
                                          int dim) {
    const int half = dim / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)N * dim; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / dim), k = (int)(i - (long long)n * dim);
        float e = 0.0f;
        if (k < half)
            e = cosf(mul_rn(t[n], freqs[k]));  // (rounded product: see mul_rn)
        else if (k < 2 * half)
            e = sinf(mul_rn(t[n], freqs[k - half]));
        out[i] = e;
    }
}

extern "C" int eod_timestep_embedding(const float* t, const float* freqs, float* out, int N, int dim, void* stream) {
    EOD_REQUIRE(t && out && N > 0 && dim > 0 && (freqs || dim < 2), "timestep_embedding: bad args");
    const long long total = (long long)N * dim;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, freqs, out, N, dim);
    EOD_CHECK_LAUNCH("timestep_embedding");
    return EOD_OK;
}

extern "C" int eod_time_embed(const eod_temb_desc* d, void* stream) {
    EOD_REQUIRE(d && d->emb, "time_embed: null pointer");
    EOD_REQUIRE(d->N > 0 && d->E > 0 && d->J >= 0, "time_embed: bad dims");
    hipStream_t st = (hipStream_t)stream;
    if (d->w1) {  // stages 1-2: sinusoid -> Linear -> SiLU -> Linear (+label_emb).  w1 == NULL: `emb` is an input
        EOD_REQUIRE(d->t && d->freqs && d->b1 && d->w2 && d->b2 && d->h1 && d->D > 0, "time_embed: null pointer");
        EOD_REQUIRE((d->label_emb == nullptr) == (d->y == nullptr), "time_embed: label_emb / y mismatch");
        const unsigned be = (unsigned)((d->E + 3) / 4);
        if (d->D <= 64 * TEMB_KI) {
            const unsigned b1 = (unsigned)((d->E + 4 * TEMB_OPW - 1) / (4 * TEMB_OPW));
            hipLaunchKernelGGL(temb_table_linear_kernel<1>, dim3(b1), dim3(256), (size_t)TEMB_NB * d->D * sizeof(float), st, nullptr, (const long long*)d->t, d->freqs, d->w1, d->b1, nullptr, nullptr, d->h1, d->N, d->D, d->E, d->t_f32);
        } else {
            hipLaunchKernelGGL(temb_linear_kernel<0>, dim3(be), dim3(256), 0, st, nullptr, (const long long*)d->t, d->freqs, d->w1, d->b1, nullptr, nullptr, d->h1, nullptr, d->N, d->D, d->E, d->t_f32);
        }
        if (d->E <= 64 * TEMB_KI) {
            const unsigned b2 = (unsigned)((d->E + 4 * TEMB_OPW - 1) / (4 * TEMB_OPW));
            hipLaunchKernelGGL(temb_table_linear_kernel<2>, dim3(b2), dim3(256), (size_t)TEMB_NB * d->E * sizeof(float), st, d->h1, nullptr, nullptr, d->w2, d->b2, d->label_emb, (const long long*)d->y, d->emb, d->N, d->E, d->E);
        } else {
            hipLaunchKernelGGL(temb_linear_kernel<1>, dim3(be), dim3(256), 0, st, d->h1, nullptr, nullptr, d->w2, d->b2, d->label_emb, (const long long*)d->y, d->emb, nullptr, d->N, d->E, d->E);
        }
    }
    if (d->J > 0) {
        EOD_REQUIRE(d->wcat && d->bcat && d->out, "time_embed: null wcat/bcat/out");
        const unsigned bj = (unsigned)((d->J + 3) / 4);
        if (d->E <= 64 * TEMB_KI) {
            const unsigned b3 = (unsigned)((d->J + 4 * TEMB_OPW - 1) / (4 * TEMB_OPW));
            hipLaunchKernelGGL(temb_table_linear_kernel<3>, dim3(b3), dim3(256), (size_t)TEMB_NB * d->E * sizeof(float), st, d->emb, nullptr, nullptr, d->wcat, d->bcat, nullptr, nullptr, d->out, d->N, d->E, d->J);
        } else {
            hipLaunchKernelGGL(temb_linear_kernel<2>, dim3(bj), dim3(256), 0, st, d->emb, nullptr, nullptr, d->wcat, d->bcat, nullptr, nullptr, d->out, nullptr, d->N, d->E, d->J);
        }
    }
    EOD_CHECK_LAUNCH("time_embed");
    return EOD_OK;
}
