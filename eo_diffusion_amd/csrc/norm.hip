// GroupNorm32 (+SiLU) and row softmax on channels-last activations.  HBM-bound kernels: 16-byte
// coalesced accesses, wavefront/LDS reductions, deterministic two-level sums (no atomics).
//
// GroupNorm over a *virtual concat* (th.cat([h, hs.pop()]) -> in_layers GN, unet_openai.py:773,313):
// group boundaries straddle the seam between the two tensors (e.g. 512|384 channels -> 28/group), so
// statistics are first reduced per CHANNEL (eod_gn_partial, one launch per source, writing into its
// channel range), then per GROUP across both sources (eod_gn_finalize).
#include "common.h"

// ---------------------------------------------------------------------------------------------
// partial: grid (P, N), block (CPP, RY)  with CPP = C / EPC 16-byte chunks per pixel (<= 256).
// part[n][p][coff + c][0..1] = (sum, sumsq) over the pixels of chunk p.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void gn_partial_kernel(const T* __restrict__ x, int HW, int C, float* __restrict__ part, int P, int Ctot,
                                  int coff) {
    constexpr int EPC = dt<T>::epc;
    extern __shared__ float red[];  // [RY][CPP*EPC*2]
    const int CPP = blockDim.x, RY = blockDim.y;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int p = blockIdx.x, n = blockIdx.y;
    const int col0 = blockIdx.z * CPP;          // first chunk column of this channel block
    const bool col_ok = col0 + tx < C / EPC;    // (the last block of a wide layer may be partial)
    const int per = (HW + P - 1) / P;
    const int p0 = p * per, p1 = col_ok ? min(HW, p0 + per) : 0;
    float s[EPC], q[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = q[e] = 0.0f;
    const T* base = x + (long long)n * HW * C + (long long)(col0 + tx) * EPC;
    for (int pix = p0 + ty; pix < p1; pix += RY) {
        const i32x4 raw = *reinterpret_cast<const i32x4*>(base + (long long)pix * C);
        if constexpr (EPC == 8) {
            const half8 h = __builtin_bit_cast(half8, raw);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (float)h[e];
                s[e] += v;
                q[e] += v * v;
            }
        } else {
            const f32x4 f = __builtin_bit_cast(f32x4, raw);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[e] += f[e];
                q[e] += f[e] * f[e];
            }
        }
    }
    const int W2 = CPP * EPC * 2;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        red[ty * W2 + (tx * EPC + e) * 2 + 0] = s[e];
        red[ty * W2 + (tx * EPC + e) * 2 + 1] = q[e];
    }
    __syncthreads();
    // fixed-order sum over ty -> deterministic
    const int tid = ty * CPP + tx, nthr = CPP * RY;
    for (int i = tid; i < W2; i += nthr) {
        float a = 0.0f;
        for (int r = 0; r < RY; ++r) a += red[r * W2 + i];
        const int c = col0 * EPC + (i >> 1);
        if (c < C) part[(((long long)n * P + p) * Ctot + coff + c) * 2 + (i & 1)] = a;
    }
}

extern "C" int eod_gn_partial(const void* x, int dtype, int N, int HW, int C, float* part, int P, int Ctot, int coff,
                              void* stream) {
    EOD_REQUIRE(x && part && N > 0 && HW > 0 && C > 0 && P > 0 && P <= HW, "gn_partial: bad args");
    const int epc = 16 / eod_esize(dtype);
    EOD_REQUIRE(C % epc == 0 && N <= 65535, "gn_partial: C=%d / N=%d unsupported", C, N);
    EOD_REQUIRE(eod_aligned16(x), "gn_partial: alignment");
    const GnSlab gs = gn_slab(N, HW, C, epc, 1);  // (only the channel decomposition is used: the slab count P is the caller's)
    const int cpp = gs.cpp, ry = gs.ry;
    const size_t lds = (size_t)ry * cpp * epc * 2 * sizeof(float);
    dim3 grid(P, N, gs.nz), block(cpp, ry);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(gn_partial_kernel<half_t>, grid, block, lds, (hipStream_t)stream, (const half_t*)x, HW, C, part, P, Ctot, coff);
    else
        hipLaunchKernelGGL(gn_partial_kernel<float>, grid, block, lds, (hipStream_t)stream, (const float*)x, HW, C, part, P, Ctot, coff);
    EOD_CHECK_LAUNCH("gn_partial");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// finalize: grid (groups, N), block 256.  mean/var of group g of sample n over P chunks x cpg channels
// (double accumulation), then scale/shift per channel:
//     y = x*scale + shift,  scale = rstd*gamma,  shift = beta - mean*rstd*gamma
// FiLM (use_scale_shift_norm, unet_openai.py:377-381): y' = y*(1+s) + t with film[n] = [s(0..C) | t(0..C)].
// ---------------------------------------------------------------------------------------------
// Bound tables (common.h, EOD_AB = 32 entries per image; groups <= 32): entry g of image n receives
//   ab_raw : sqrt(max over the group's (slot, channel) partial sums of squares)  >= max|x| over the group's channels
//   ab_norm: max_c|scale_c| * that + max_c|shift_c|                                >= max|x*scale + shift| (and of its SiLU)
// for the consumers that split the raw tensor (skip / resampling convs) and the normalised one (the conv behind the GroupNorm).
__global__ void gn_finalize_kernel(const float* __restrict__ part0, int P0, int C0, const float* __restrict__ part1, int P1,
                                   int C1, long long HW, int groups, float eps, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const float* __restrict__ film, long long film_stride,
                                   float* __restrict__ ss, float* __restrict__ ab_raw, float* __restrict__ ab_norm) {
    __shared__ double rs[256], rq[256];
    __shared__ float rm[256];
    float qmax = 0.0f;  // largest partial sum of squares of the group (a NaN sum is dropped by fmaxf, but it also poisons q and with it
                        // every scale / shift below: NaN in, NaN out)
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int Ctot = C0 + C1;
    const int cpg = Ctot / groups;
    const int c0 = g * cpg;
    double s = 0.0, q = 0.0;
    // channels of this group living in source 0 / source 1 (a group may straddle the concat seam)
    const int a0 = min(c0, C0), a1 = min(c0 + cpg, C0);  // [a0, a1) in source 0
    const int n0c = a1 - a0, n1c = cpg - n0c;
    // {sum, sumsq} pairs as one 8-byte load, four independent loads in flight per thread (a 256 x 256 map hands over 1024 slots per
    // image: 16 MB per launch, and the dependent scalar loads of the plain loop took 15 us for it)
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
                qmax = fmaxf(qmax, v[u].y);
            }
        }
        for (; i < total; i += 256) {
            const int p = i / nc;
            const float2 v = base[(long long)p * Cs + first + (i - p * nc)];
            s += (double)v.x;
            q += (double)v.y;
            qmax = fmaxf(qmax, v.y);
        }
    };
    if (n0c > 0) accumulate(part0, P0, C0, a0, n0c);
    if (n1c > 0) accumulate(part1, P1, C1, max(c0, C0) - C0, n1c);  // (first channel inside source 1)
    rs[tid] = s;
    rq[tid] = q;
    rm[tid] = qmax;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            rs[tid] += rs[tid + o];
            rq[tid] += rq[tid + o];
            rm[tid] = fmaxf(rm[tid], rm[tid + o]);
        }
        __syncthreads();
    }
    const double cnt = (double)HW * cpg;
    const double mean = rs[0] / cnt;
    double var = rq[0] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    const float xmax = sqrtf(rm[0]);  // (a sum of squares that overflowed gives inf: the consumers then fall back to the smallest scale)
    float scmax = 0.0f, shmax = 0.0f;
    for (int c = c0 + tid; c < c0 + cpg; c += 256) {
        float sc = rstd * gamma[c];
        float sh = beta[c] - meanf * sc;
        if (film) {
            const float fs = 1.0f + film[(long long)n * film_stride + c];
            const float ft = film[(long long)n * film_stride + Ctot + c];
            sc = sc * fs;
            sh = sh * fs + ft;
        }
        ss[((long long)n * Ctot + c) * 2 + 0] = sc;
        ss[((long long)n * Ctot + c) * 2 + 1] = sh;
        scmax = fmaxf(scmax, fabsf(sc));
        shmax = fmaxf(shmax, fabsf(sh));
    }
    if (ab_raw || ab_norm) {
        __syncthreads();  // (rm[0] has been read by everybody)
        rm[tid] = scmax;
        reinterpret_cast<float*>(rs)[tid] = shmax;
        __syncthreads();
        if (tid == 0) {
            const int nt = cpg < 256 ? cpg : 256;
            for (int i = 1; i < nt; ++i) {
                scmax = fmaxf(scmax, rm[i]);
                shmax = fmaxf(shmax, reinterpret_cast<float*>(rs)[i]);
            }
            if (ab_raw) ab_raw[(long long)n * EOD_AB + g] = xmax;
            if (ab_norm) ab_norm[(long long)n * EOD_AB + g] = scmax * xmax + shmax;
        }
        // entries [groups, EOD_AB) of the tables are never written: the caller zero-fills them once (groups < 32 only)
    }
}

extern "C" int eod_gn_finalize(const float* part0, int P0, int C0, const float* part1, int P1, int C1, int N, int64_t HW,
                               int groups, float eps, const float* gamma, const float* beta, const float* film,
                               int64_t film_stride, float* scale_shift, float* ab_raw, float* ab_norm, void* stream) {
    EOD_REQUIRE(part0 && gamma && beta && scale_shift, "gn_finalize: null pointer");
    EOD_REQUIRE(N > 0 && P0 > 0 && C0 > 0 && C1 >= 0 && (C1 == 0 || (part1 && P1 > 0)) && groups > 0 && (C0 + C1) % groups == 0,
                "gn_finalize: C0=%d C1=%d groups=%d", C0, C1, groups);
    EOD_REQUIRE((!ab_raw && !ab_norm) || groups <= EOD_AB, "gn_finalize: bound tables hold %d entries per image, got %d groups", EOD_AB, groups);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, (hipStream_t)stream, part0, P0, C0, part1, P1, C1,
                       (long long)HW, groups, eps, gamma, beta, film, (long long)film_stride, scale_shift, ab_raw, ab_norm);
    EOD_CHECK_LAUNCH("gn_finalize");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// eod_act_bound: the bound table ab[N][EOD_AB] (common.h) of a tensor that no GroupNorm takes statistics of right before its
// split-fp16 consumer (the inputs of Downsample.op / Upsample.conv unet_openai.py:262-264,227, the image in front of the first conv
// :609, q / k / v of the attention :476-480).  grid (EOD_AB, N): block (j, n) covers the j-th 1/32 of image n's data.
//   parts mode : from the {sum, sumsq} slots a conv epilogue / eod_gn_partial wrote ([N][P][C][2], one or two concat sources):
//                entry = sqrt(max sumsq) -- an upper bound of max|x| that costs no pass over the tensor
//   direct mode: entry = max|x| over the image's elements, exact for every finite fp32 value (NaN / inf -> inf)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bound_parts_kernel(const float* __restrict__ part0, long long L0, const float* __restrict__ part1,
                                                              long long L1, float* __restrict__ ab) {
    __shared__ float red[4];
    const int j = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    float m = 0.0f;
    auto scan = [&](const float* __restrict__ part, long long L) {  // L = P * C {sum, sumsq} pairs per image
        const float2* base = reinterpret_cast<const float2*>(part) + (long long)n * L;
        const long long i0 = L * j / EOD_AB, i1 = L * (j + 1) / EOD_AB;
        for (long long i = i0 + tid; i < i1; i += 256) {
            const float q = base[i].y;
            m = fmaxf(m, (q == q) ? q : __uint_as_float(0x7f800000u));
        }
    };
    scan(part0, L0);
    if (part1) scan(part1, L1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) ab[(long long)n * EOD_AB + j] = sqrtf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

template <typename T>
__global__ __launch_bounds__(256) void act_bound_direct_kernel(const T* __restrict__ x, long long per_image, float* __restrict__ ab, int accumulate) {
    constexpr int EPC = dt<T>::epc;
    __shared__ float red[4];
    const int j = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const long long chunks = per_image / EPC;  // (per_image % EPC == 0: checked by the launcher)
    const long long i0 = chunks * j / EOD_AB, i1 = chunks * (j + 1) / EOD_AB;
    const i32x4* base = reinterpret_cast<const i32x4*>(x + (long long)n * per_image);
    unsigned mb = 0;  // max over |x| as an ordered bit pattern: for non-negative floats the integer order IS the float order, and NaN sorts above inf
    for (long long i = i0 + tid; i < i1; i += 256) {
        const i32x4 raw = base[i];
        if constexpr (EPC == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) mb = max(mb, (unsigned)raw[e] & 0x7fffffffu);
        } else {
            const half8 h = __builtin_bit_cast(half8, raw);
#pragma unroll
            for (int e = 0; e < 8; ++e) mb = max(mb, __float_as_uint((float)h[e]) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
    if ((tid & 63) == 0) red[tid >> 6] = __uint_as_float(mb);
    __syncthreads();
    if (tid == 0) {
        unsigned r = accumulate ? __float_as_uint(ab[(long long)n * EOD_AB + j]) : 0u;  // (a second source of a virtual concat)
        for (int w = 0; w < 4; ++w) r = max(r, __float_as_uint(red[w]));
        ab[(long long)n * EOD_AB + j] = __uint_as_float(r > 0x7f800000u ? 0x7f800000u : r);  // NaN -> inf
    }
}

extern "C" int eod_act_bound(const void* x, int dtype, int N, int64_t per_image, const float* part0, int P0, int C0, const float* part1,
                             int P1, int C1, float* ab, int accumulate, void* stream) {
    EOD_REQUIRE(ab && N > 0 && N <= 65535, "act_bound: bad args");
    const dim3 grid(EOD_AB, N), block(256);
    if (part0) {
        EOD_REQUIRE(P0 > 0 && C0 > 0 && (!part1 || (P1 > 0 && C1 > 0)) && !accumulate, "act_bound: bad partial-sum geometry");
        hipLaunchKernelGGL(act_bound_parts_kernel, grid, block, 0, (hipStream_t)stream, part0, (long long)P0 * C0, part1, (long long)P1 * C1, ab);
    } else {
        EOD_REQUIRE(x && per_image > 0 && (dtype == EOD_F32 || dtype == EOD_F16), "act_bound: bad args");
        EOD_REQUIRE(per_image % (16 / eod_esize(dtype)) == 0 && eod_aligned16(x), "act_bound: 16-byte chunks per image required");
        if (dtype == EOD_F16)
            hipLaunchKernelGGL(act_bound_direct_kernel<half_t>, grid, block, 0, (hipStream_t)stream, (const half_t*)x, (long long)per_image, ab, accumulate);
        else
            hipLaunchKernelGGL(act_bound_direct_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, (long long)per_image, ab, accumulate);
    }
    EOD_CHECK_LAUNCH("act_bound");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// apply: y[n][pix][coff + c] = act(x[n][pix][c] * scale[n][coff+c] + shift[n][coff+c])
// slab decomposition of common.h (gn_slab); y rows are Ctot wide (materialises the concat, normalised).
// ---------------------------------------------------------------------------------------------
// SPLIT (fp32 storage): y is written PRE-SPLIT for a split-fp16 consumer (eod_conv_desc.x_presplit): the two threads that own the
// 16-byte chunks of one 8-channel group (lanes l, l ^ 1: blockDim.x is even) exchange halves through DPP, exactly like the in-LDS
// rewrite of the conv kernels (igemm.hip: split_pair_exchange), scaled by the image's power-of-two s from the bound table.
template <typename T, bool SILU, bool SPLIT = false>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int HW, int C, const float* __restrict__ ss, int Ctot,
                                                      int coff, T* __restrict__ y, int per, const float* __restrict__ ab = nullptr) {
    constexpr int EPC = dt<T>::epc, U = 4;
    constexpr bool FAST = (EPC == 8);
    const int tx = threadIdx.x, ty = threadIdx.y, RY = blockDim.y;
    const int n = blockIdx.y;
    float osc = 1.0f;
    if constexpr (SPLIT) {
        // the image's table maximum, read by every thread itself (a block-uniform address: scalar loads).  NOT a wave reduction: the
        // block is (C / EPC) x ry threads, and its last wave is partial whenever that is no multiple of 64 (C = 288: 72 x 3 = 216) -- a
        // 24-lane wave then missed the entries 24..31 and scaled ITS pixels by another power of two than the consumer un-scales by
        // (found by tests/test_gpu_fuzz_archs.py case 83: 1e-2 on a whole UNet whenever the maximum sat in one of those entries)
        float b = 0.0f;
#pragma unroll
        for (int j = 0; j < EOD_AB; ++j) {
            const float v = ab[(long long)n * EOD_AB + j];
            b = fmaxf(b, (v == v) ? v : __uint_as_float(0x7f800000u));
        }
        osc = ab_scale_of(b).s;
    }
    const int col = blockIdx.z * blockDim.x + tx;  // chunk column of this thread (channel blocks along z for wide layers)
    if (col >= C / EPC) return;
    const int p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
    float sc[EPC], sh[EPC];
    const float* sp = ss + ((long long)n * Ctot + coff + col * EPC) * 2;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = sp[2 * e];
        sh[e] = sp[2 * e + 1];
    }
    const T* xb = x + (long long)n * HW * C + col * EPC;
    T* yb = y + (long long)n * HW * Ctot + coff + col * EPC;
    for (int pix = p0 + ty; pix < p1; pix += U * RY) {
        i32x4 raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (pix + u * RY < p1) raw[u] = *reinterpret_cast<const i32x4*>(xb + (long long)(pix + u * RY) * C);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pix + u * RY >= p1) break;
            i32x4 outv;
            if constexpr (EPC == 8) {
                const half8 h = __builtin_bit_cast(half8, raw[u]);
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = (float)h[e] * sc[e] + sh[e];
                    if (SILU) v = silu_f<FAST>(v);
                    o[e] = (half_t)v;
                }
                outv = __builtin_bit_cast(i32x4, o);
            } else {
                const f32x4 fv = __builtin_bit_cast(f32x4, raw[u]);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fv[e] * sc[e] + sh[e];
                    if (SILU) v = silu_f<FAST>(v);
                    o[e] = v;
                }
                outv = __builtin_bit_cast(i32x4, o);
                if constexpr (SPLIT) {
                    half4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = o[e] * osc;
                        h[e] = (half_t)v;
                        l[e] = (half_t)(v - (float)h[e]);
                    }
                    typedef int i32x2 __attribute__((ext_vector_type(2)));
                    const i32x2 hb = __builtin_bit_cast(i32x2, h), lb = __builtin_bit_cast(i32x2, l);
                    const bool odd = (col & 1) != 0;  // the even chunk of a pair keeps [hi_even | hi_odd], the odd one [lo_even | lo_odd]
                    const int s0 = odd ? hb[0] : lb[0], s1 = odd ? hb[1] : lb[1];
                    const int r0 = __builtin_amdgcn_mov_dpp(s0, 0xB1, 0xF, 0xF, true), r1 = __builtin_amdgcn_mov_dpp(s1, 0xB1, 0xF, 0xF, true);
                    outv = odd ? i32x4{r0, r1, lb[0], lb[1]} : i32x4{hb[0], hb[1], r0, r1};
                }
            }
            *reinterpret_cast<i32x4*>(yb + (long long)(pix + u * RY) * Ctot) = outv;
        }
    }
}

extern "C" int eod_gn_apply(const void* x, int dtype, int N, int HW, int C, const float* scale_shift, int Ctot, int coff,
                            int silu, void* y, const float* split_bound, void* stream) {
    EOD_REQUIRE(x && y && scale_shift && N > 0 && HW > 0 && C > 0, "gn_apply: bad args");
    const int epc = 16 / eod_esize(dtype);
    EOD_REQUIRE(C % epc == 0 && Ctot % epc == 0 && coff % epc == 0, "gn_apply: channel alignment");
    EOD_REQUIRE(N <= 65535, "gn_apply: N=%d unsupported", N);
    EOD_REQUIRE(eod_aligned16(x) && eod_aligned16(y), "gn_apply: alignment");
    GnSlab g = gn_slab(N, HW, C, epc, 4);
    hipStream_t st = (hipStream_t)stream;
    if (split_bound) {
        EOD_REQUIRE(dtype == EOD_F32 && C % 8 == 0 && Ctot % 8 == 0 && coff % 8 == 0, "gn_apply: split_out needs fp32 storage and whole 8-channel groups");
        if (g.cpp & 1) {  // the two chunks of an 8-channel group must sit in neighbouring lanes
            g.cpp += 1;
            g.ry = 256 / g.cpp > 0 ? 256 / g.cpp : 1;
            const int quantum = g.ry * 4;
            g.per = (g.per + quantum - 1) / quantum * quantum;
            g.P = (HW + g.per - 1) / g.per;
        }
        const dim3 grid(g.P, N, g.nz), block(g.cpp, g.ry);
        if (silu) hipLaunchKernelGGL((gn_apply_kernel<float, true, true>), grid, block, 0, st, (const float*)x, HW, C, scale_shift, Ctot, coff, (float*)y, g.per, split_bound);
        else hipLaunchKernelGGL((gn_apply_kernel<float, false, true>), grid, block, 0, st, (const float*)x, HW, C, scale_shift, Ctot, coff, (float*)y, g.per, split_bound);
        EOD_CHECK_LAUNCH("gn_apply");
        return EOD_OK;
    }
    const dim3 grid(g.P, N, g.nz), block(g.cpp, g.ry);
#define LAUNCH(T, S) hipLaunchKernelGGL((gn_apply_kernel<T, S>), grid, block, 0, st, (const T*)x, HW, C, scale_shift, Ctot, coff, (T*)y, g.per)
    if (dtype == EOD_F16) {
        if (silu) LAUNCH(half_t, true); else LAUNCH(half_t, false);
    } else {
        if (silu) LAUNCH(float, true); else LAUNCH(float, false);
    }
#undef LAUNCH
    EOD_CHECK_LAUNCH("gn_apply");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// row softmax (th.softmax(weight.float(), dim=-1), unet_openai.py:479/513): one wave per row, fp32 math,
// output in the storage dtype, columns [n, ldp) zero-filled (K padding of the P.V GEMM).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename T>
__global__ void softmax_rows_kernel(const float* __restrict__ s, long long lds, T* __restrict__ p, long long ldp, long long rows,
                                    int n) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = s + row * lds;
    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, sr[i]);
    m = wave_max(m);
    float sum = 0.0f;
    for (int i = lane; i < n; i += 64) sum += expf(sr[i] - m);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    T* pr = p + row * ldp;
    for (int i = lane; i < (int)ldp; i += 64) pr[i] = (i < n) ? (T)(expf(sr[i] - m) * inv) : (T)0.0f;
}

// Long rows (attention over thousands of keys): ONE workgroup per row, the whole row lives in registers (V4 float4 per thread,
// 256 threads), so the scores are read once, exponentiated once and written once (the wave-per-row kernel above makes three
// passes over the row and evaluates exp twice per element).  fp16 output uses the fast exp (the result is rounded to 11 bits).
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    v = is_max ? wave_max(v) : wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    const float a = red[0], b = red[1], c = red[2], d = red[3];
    return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

template <typename T, int V4>
__global__ __launch_bounds__(256) void softmax_row_block_kernel(const float* __restrict__ s, long long lds, T* __restrict__ p, long long ldp, int n) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const float* sr = s + row * lds;
    f32x4 v[V4];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        const int c = (i * 256 + threadIdx.x) * 4;
        if (c + 3 < n) {
            v[i] = *reinterpret_cast<const f32x4*>(sr + c);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = (c + e < n) ? sr[c + e] : -INFINITY;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, v[i][e]);
    }
    m = block_reduce(m, red, true);
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < V4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[i][e] - m;
            const float ex = sizeof(T) == 2 ? __expf(x) : expf(x);  // exp(-inf) = 0 for the masked tail
            v[i][e] = ex;
            sum += ex;
        }
    sum = block_reduce(sum, red, false);
    const float inv = 1.0f / sum;
    T* pr = p + row * ldp;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        const int c = (i * 256 + threadIdx.x) * 4;
        if (c >= ldp) continue;
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (c + e < n) ? (T)(v[i][e] * inv) : (T)0.0f;
        if (c + 3 < ldp) {
            if constexpr (sizeof(T) == 2)
                *reinterpret_cast<unsigned long long*>(pr + c) = *reinterpret_cast<const unsigned long long*>(o);
            else
                *reinterpret_cast<f32x4*>(pr + c) = *reinterpret_cast<const f32x4*>(o);
        } else {
            for (int e = 0; e < 4 && c + e < ldp; ++e) pr[c + e] = o[e];
        }
    }
}

template <typename T>
static bool launch_softmax_block(const float* s, long long lds, T* p, long long ldp, long long rows, int n, hipStream_t st) {
    // needs aligned rows (16-byte loads, 8- / 16-byte stores) and a row that fits V4 <= 16 float4 per thread
    if (lds % 4 || ldp % 4 || (reinterpret_cast<uintptr_t>(s) & 15) || (reinterpret_cast<uintptr_t>(p) & 15) || ldp > 16384 || n < 1024 || rows > 0x7fffffffLL)
        return false;
    const int need = (int)((ldp + 1023) / 1024);
    const dim3 grid((unsigned)rows), block(256);
    if (need <= 1) hipLaunchKernelGGL((softmax_row_block_kernel<T, 1>), grid, block, 0, st, s, lds, p, ldp, n);
    else if (need <= 2) hipLaunchKernelGGL((softmax_row_block_kernel<T, 2>), grid, block, 0, st, s, lds, p, ldp, n);
    else if (need <= 4) hipLaunchKernelGGL((softmax_row_block_kernel<T, 4>), grid, block, 0, st, s, lds, p, ldp, n);
    else if (need <= 8) hipLaunchKernelGGL((softmax_row_block_kernel<T, 8>), grid, block, 0, st, s, lds, p, ldp, n);
    else hipLaunchKernelGGL((softmax_row_block_kernel<T, 16>), grid, block, 0, st, s, lds, p, ldp, n);
    return true;
}

extern "C" int eod_softmax_rows(const float* s, int64_t lds, void* p, int64_t ldp, int dtype, int64_t rows, int n,
                                void* stream) {
    EOD_REQUIRE(s && p && rows > 0 && n > 0 && ldp >= n && lds >= n, "softmax: bad args");
    if (dtype == EOD_F16 ? launch_softmax_block<half_t>(s, (long long)lds, (half_t*)p, (long long)ldp, (long long)rows, n, (hipStream_t)stream)
                         : launch_softmax_block<float>(s, (long long)lds, (float*)p, (long long)ldp, (long long)rows, n, (hipStream_t)stream)) {
        EOD_CHECK_LAUNCH("softmax_rows");
        return EOD_OK;
    }
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (dtype == EOD_F16)
        hipLaunchKernelGGL(softmax_rows_kernel<half_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, (long long)lds, (half_t*)p, (long long)ldp, (long long)rows, n);
    else
        hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, (long long)lds, (float*)p, (long long)ldp, (long long)rows, n);
    EOD_CHECK_LAUNCH("softmax_rows");
    return EOD_OK;
}

// ---------------------------------------------------------------------------------------------
// A-PRIORI bound of a linear layer's output: |y_r| = |sum_c w[r][c] x_c + b_r| <= (max_r sum_c |w[r][c]|) * max|x| + max|b|.
// A producer that writes its output pre-split needs the scale BEFORE it has seen its own values: eod_weight_l1max condenses the
// weight into coef = {max row L1 norm, max|bias|} once per plan (pack time), eod_bound_affine turns the input's table into the
// output's every step (N x 32 multiply-adds).  The qkv projection in front of the fused attention uses it (unet_openai.py:414).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weight_l1max_kernel(const float* __restrict__ w, int rows, int cols, const float* __restrict__ bias,
                                                           float* __restrict__ coef) {
    __shared__ float red[4], best;
    const int tid = threadIdx.x;
    if (tid == 0) best = 0.0f;
    float bm = 0.0f;
    for (int r = 0; r < rows; ++r) {  // (one block: a few hundred thousand weights, once per plan)
        float a = 0.0f;
        for (int c = tid; c < cols; c += 256) a += fabsf(w[(long long)r * cols + c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = a;
        __syncthreads();
        if (tid == 0) best = fmaxf(best, (red[0] + red[1]) + (red[2] + red[3]));
    }
    if (bias)
        for (int r = tid; r < rows; r += 256) bm = fmaxf(bm, fabsf(bias[r]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bm = fmaxf(bm, __shfl_xor(bm, o));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = bm;
    __syncthreads();
    if (tid == 0) {
        coef[0] = best * 1.0001f;  // (the fp32 sums above are rounded: keep the bound a bound)
        coef[1] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
}
extern "C" int eod_weight_l1max(const float* w, int rows, int cols, const float* bias, float* coef, void* stream) {
    EOD_REQUIRE(w && coef && rows > 0 && cols > 0, "weight_l1max: bad args");
    hipLaunchKernelGGL(weight_l1max_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w, rows, cols, bias, coef);
    EOD_CHECK_LAUNCH("weight_l1max");
    return EOD_OK;
}
__global__ void bound_affine_kernel(const float* __restrict__ ab_in, const float* __restrict__ coef, float* __restrict__ ab_out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ab_out[i] = ab_in[i] * coef[0] + coef[1];
}
extern "C" int eod_bound_affine(const float* ab_in, const float* coef, float* ab_out, int N, void* stream) {
    EOD_REQUIRE(ab_in && coef && ab_out && N > 0, "bound_affine: bad args");
    const int n = N * EOD_AB;
    hipLaunchKernelGGL(bound_affine_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ab_in, coef, ab_out, n);
    EOD_CHECK_LAUNCH("bound_affine");
    return EOD_OK;
}
