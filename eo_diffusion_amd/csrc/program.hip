// Error reporting + the native program executor.
//
// A "program" is a flat array of tagged descriptors (eod_op) built once per (model, shape, precision)
// by the host language.  eod_program_run walks it and enqueues every kernel on one HIP stream without
// returning to the host between launches and without any host synchronisation, so a whole UNet forward
// is a single FFI call (and can be captured into a hipGraph by the caller).
//
// eod_small_desc field use per op kind:
//   GN_PARTIAL : p[0]=x p[1]=part            i = {dtype, N, HW, C, P, Ctot, coff}
//   GN_FINALIZE: p[0]=part0 p[1]=gamma p[2]=beta p[3]=film p[4]=scale_shift p[5]=part1 p[6]=ab_raw p[7]=ab_norm   l = {HW, film_stride}
//                i = {N, P0, C0, groups, P1, C1}    f = {eps}
//   ACT_BOUND  : p[0]=x p[1]=part0 p[2]=part1 p[3]=ab     l = {per_image}   i = {dtype, N, P0, C0, P1, C1, accumulate}
//   ATTN_NAT   : p[0]=qkv p[1]=out p[2]=lse p[3]=qkv_bound   i = {dtype, N, T, C, heads, d, q_off, k_off, v_off, head_stride}   l = {flags}
//   BOUND_AFFINE: p[0]=ab_in p[1]=coef p[2]=ab_out   i = {N}
//   GN_APPLY   : p[0]=x p[1]=scale_shift p[2]=y p[3]=split_bound   i = {dtype, N, HW, C, Ctot, coff, silu}
//   SOFTMAX    : p[0]=s p[1]=p               l = {lds, ldp, rows}   i = {dtype, n}
//   TO_NHWC    : p[0]=src0 p[1]=src1 p[2]=dst     i = {C0, C1, dtype, N, H, W, c_pad}
//   TO_NCHW    : p[0]=src p[1]=dst           i = {dtype, N, H, W, C}
//   POOL       : p[0]=x p[1]=y               i = {dtype, N, H, W, C, mode, pad_tl}
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void eod_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* eod_last_error(void) { return g_err; }
extern "C" int eod_version(void) { return EOD_ABI_VERSION; }
extern "C" int eod_struct_size(int kind) {
    switch (kind) {
        case 1: return (int)sizeof(eod_conv_desc);
        case 2: return (int)sizeof(eod_gemm_desc);
        case 3: return (int)sizeof(eod_temb_desc);
        case 4: return (int)sizeof(eod_small_desc);
        case 5: return (int)sizeof(eod_op);
        case 6: return (int)sizeof(eod_attn_desc);
        default: return -1;
    }
}

// ---- per-op HIP-event timer (measurement only; used by bench.py for the roofline numbers) ----
struct eod_timer {
    int n_ops, max_runs, runs;
    hipEvent_t* ev;        // [max_runs][n_ops][2]
    unsigned char* mask;   // [n_ops] 1 = bracket this op with events (default: all)
};

extern "C" void* eod_timer_create(int n_ops, int max_runs) {
    if (n_ops <= 0 || max_runs <= 0) return nullptr;
    eod_timer* t = new eod_timer{n_ops, max_runs, 0, nullptr, nullptr};
    const size_t n = (size_t)n_ops * max_runs * 2;
    t->ev = new hipEvent_t[n];
    t->mask = new unsigned char[n_ops];
    memset(t->mask, 1, n_ops);
    for (size_t i = 0; i < n; ++i)
        if (hipEventCreate(&t->ev[i]) != hipSuccess) return nullptr;
    return t;
}

extern "C" void eod_timer_destroy(void* h) {
    eod_timer* t = (eod_timer*)h;
    if (!t) return;
    const size_t n = (size_t)t->n_ops * t->max_runs * 2;
    for (size_t i = 0; i < n; ++i) (void)hipEventDestroy(t->ev[i]);
    delete[] t->ev;
    delete[] t->mask;
    delete t;
}

// every event pair costs a few microseconds of stream idle time; restrict the bracketing to the ops of interest
extern "C" int eod_timer_set_mask(void* h, const unsigned char* mask, int n_ops) {
    eod_timer* t = (eod_timer*)h;
    EOD_REQUIRE(t && mask && n_ops == t->n_ops, "timer_set_mask: bad args");
    memcpy(t->mask, mask, n_ops);
    return EOD_OK;
}

// after the stream has been synchronised: ms[k] = SUM over recorded runs of op k's duration; returns #runs
extern "C" int eod_timer_read(void* h, float* ms) {
    eod_timer* t = (eod_timer*)h;
    EOD_REQUIRE(t && ms, "timer_read: bad args");
    for (int k = 0; k < t->n_ops; ++k) ms[k] = 0.0f;
    for (int r = 0; r < t->runs; ++r)
        for (int k = 0; k < t->n_ops; ++k) {
            if (!t->mask[k]) continue;
            float e = 0.0f;
            hipEvent_t* p = t->ev + ((size_t)r * t->n_ops + k) * 2;
            if (hipEventElapsedTime(&e, p[0], p[1]) != hipSuccess) {
                eod_set_error("timer_read: hipEventElapsedTime failed (run %d op %d)", r, k);
                return EOD_ELAUNCH;
            }
            ms[k] += e;
        }
    const int runs = t->runs;
    t->runs = 0;
    return runs;
}

static int run_impl(const eod_op* ops, int n_ops, void* stream, eod_timer* tm) {
    EOD_REQUIRE(ops && n_ops >= 0, "program_run: bad args");
    hipEvent_t* ev = nullptr;
    if (tm) {
        EOD_REQUIRE(tm->n_ops == n_ops, "program_run_timed: timer was created for %d ops, program has %d", tm->n_ops, n_ops);
        EOD_REQUIRE(tm->runs < tm->max_runs, "program_run_timed: timer full (%d runs)", tm->max_runs);
        ev = tm->ev + (size_t)tm->runs * n_ops * 2;
        tm->runs++;
    }
    for (int k = 0; k < n_ops; ++k) {
        const eod_op& o = ops[k];
        const bool timed = ev && tm->mask[k];
        if (timed) (void)hipEventRecord(ev[2 * k], (hipStream_t)stream);
        const eod_small_desc& s = o.u.small;
        int rc;
        switch (o.kind) {
            case EOD_OP_CONV: rc = eod_conv2d_igemm(&o.u.conv, stream); break;
            case EOD_OP_GEMM: rc = eod_gemm_nt(&o.u.gemm, stream); break;
            case EOD_OP_TEMB: rc = eod_time_embed(&o.u.temb, stream); break;
            case EOD_OP_ATTN: rc = eod_attention_fwd(&o.u.attn, stream); break;
            case EOD_OP_ATTN_NAT:
                rc = eod_attention_fwd_nat(s.p[0], (void*)s.p[1], (float*)s.p[2], s.i[0], s.i[1], s.i[2], s.i[3], s.i[4], s.i[5], s.i[6], s.i[7], s.i[8],
                                           s.i[9], (const float*)s.p[3], (int)s.l[0], stream);
                break;
            case EOD_OP_BOUND_AFFINE:
                rc = eod_bound_affine((const float*)s.p[0], (const float*)s.p[1], (float*)s.p[2], s.i[0], stream);
                break;
            case EOD_OP_ACT_BOUND:
                rc = eod_act_bound(s.p[0], s.i[0], s.i[1], s.l[0], (const float*)s.p[1], s.i[2], s.i[3], (const float*)s.p[2], s.i[4], s.i[5],
                                   (float*)s.p[3], s.i[6], stream);
                break;
            case EOD_OP_DROPOUT:
                rc = eod_dropout(s.p[0], (void*)s.p[1], s.i[0], s.l[0], s.f[0], (uint64_t)s.l[1], (uint32_t)s.i[1], (uint32_t)s.i[2], stream);
                break;
            case EOD_OP_TRANSPOSE:
                rc = eod_transpose_gather(s.p[0], s.i[0], s.i[1], s.i[2], s.i[3], s.i[4], (void*)s.p[1], s.l[0], s.i[5], s.i[6], s.i[7], s.i[8],
                                          s.i[9], (int)s.l[1], (int)s.l[2], (int)s.l[3], stream);
                break;
            case EOD_OP_GN_PARTIAL:
                rc = eod_gn_partial(s.p[0], s.i[0], s.i[1], s.i[2], s.i[3], (float*)s.p[1], s.i[4], s.i[5], s.i[6], stream);
                break;
            case EOD_OP_GN_FINALIZE:
                rc = eod_gn_finalize((const float*)s.p[0], s.i[1], s.i[2], (const float*)s.p[5], s.i[4], s.i[5], s.i[0], s.l[0],
                                     s.i[3], s.f[0], (const float*)s.p[1], (const float*)s.p[2], (const float*)s.p[3], s.l[1],
                                     (float*)s.p[4], (float*)s.p[6], (float*)s.p[7], stream);
                break;
            case EOD_OP_GN_APPLY:
                rc = eod_gn_apply(s.p[0], s.i[0], s.i[1], s.i[2], s.i[3], (const float*)s.p[1], s.i[4], s.i[5], s.i[6], (void*)s.p[2],
                                  (const float*)s.p[3], stream);
                break;
            case EOD_OP_SOFTMAX:
                rc = eod_softmax_rows((const float*)s.p[0], s.l[0], (void*)s.p[1], s.l[1], s.i[0], s.l[2], s.i[1], stream);
                break;
            case EOD_OP_TO_NHWC:
                rc = eod_nchw_to_nhwc((const float*)s.p[0], s.i[0], (const float*)s.p[1], s.i[1], (void*)s.p[2], s.i[2], s.i[3], s.i[4], s.i[5], s.i[6], stream);
                break;
            case EOD_OP_TO_NCHW:
                rc = eod_nhwc_to_nchw(s.p[0], s.i[0], (float*)s.p[1], s.i[1], s.i[2], s.i[3], s.i[4], stream);
                break;
            case EOD_OP_POOL:
                rc = eod_resample2x(s.p[0], s.i[0], s.i[1], s.i[2], s.i[3], s.i[4], s.i[5], s.i[6], (void*)s.p[1], stream);
                break;
            default:
                eod_set_error("program_run: op %d has unknown kind %d", k, o.kind);
                return EOD_EINVAL;
        }
        if (timed) (void)hipEventRecord(ev[2 * k + 1], (hipStream_t)stream);
        if (rc != EOD_OK) {
            char tmp[400];
            strncpy(tmp, g_err, sizeof(tmp) - 1);
            tmp[sizeof(tmp) - 1] = 0;
            eod_set_error("program_run: op %d (kind %d): %s", k, o.kind, tmp);
            return rc;
        }
    }
    return EOD_OK;
}

extern "C" int eod_program_run(const eod_op* ops, int n_ops, void* stream) { return run_impl(ops, n_ops, stream, nullptr); }
extern "C" int eod_program_run_timed(const eod_op* ops, int n_ops, void* stream, void* timer) {
    EOD_REQUIRE(timer, "program_run_timed: null timer");
    return run_impl(ops, n_ops, stream, (eod_timer*)timer);
}
