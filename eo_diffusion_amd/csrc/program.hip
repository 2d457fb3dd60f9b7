// Error reporting + the native program executor.
//
// A "program" is a flat array of tagged descriptors (eod_op) built once per (model, shape, precision)
// by the host language.  eod_program_run walks it and enqueues every kernel on one HIP stream without
// returning to the host between launches and without any host synchronisation, so a whole UNet forward
// is a single FFI call (and can be captured into a hipGraph by the caller).
//
// eod_small_desc field use per op kind:
//   GN_PARTIAL : p[0]=x p[1]=part            i = {dtype, N, HW, C, P, Ctot, coff}
//   GN_FINALIZE: p[0]=part p[1]=gamma p[2]=beta p[3]=film p[4]=scale_shift   l = {HW, film_stride}
//                i = {N, P, Ctot, groups}    f = {eps}
//   GN_APPLY   : p[0]=x p[1]=scale_shift p[2]=y   i = {dtype, N, HW, C, Ctot, coff, silu}
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
extern "C" int eod_version(void) { return 100; }
extern "C" int eod_struct_size(int kind) {
    switch (kind) {
        case 1: return (int)sizeof(eod_conv_desc);
        case 2: return (int)sizeof(eod_gemm_desc);
        case 3: return (int)sizeof(eod_temb_desc);
        case 4: return (int)sizeof(eod_small_desc);
        case 5: return (int)sizeof(eod_op);
        default: return -1;
    }
}

extern "C" int eod_program_run(const eod_op* ops, int n_ops, void* stream) {
    EOD_REQUIRE(ops && n_ops >= 0, "program_run: bad args");
    for (int k = 0; k < n_ops; ++k) {
        const eod_op& o = ops[k];
        const eod_small_desc& s = o.u.small;
        int rc;
        switch (o.kind) {
            case EOD_OP_CONV: rc = eod_conv2d_igemm(&o.u.conv, stream); break;
            case EOD_OP_GEMM: rc = eod_gemm_nt(&o.u.gemm, stream); break;
            case EOD_OP_TEMB: rc = eod_time_embed(&o.u.temb, stream); break;
            case EOD_OP_GN_PARTIAL:
                rc = eod_gn_partial(s.p[0], s.i[0], s.i[1], s.i[2], s.i[3], (float*)s.p[1], s.i[4], s.i[5], s.i[6], stream);
                break;
            case EOD_OP_GN_FINALIZE:
                rc = eod_gn_finalize((const float*)s.p[0], s.i[0], s.i[1], s.i[2], s.l[0], s.i[3], s.f[0], (const float*)s.p[1],
                                     (const float*)s.p[2], (const float*)s.p[3], s.l[1], (float*)s.p[4], stream);
                break;
            case EOD_OP_GN_APPLY:
                rc = eod_gn_apply(s.p[0], s.i[0], s.i[1], s.i[2], s.i[3], (const float*)s.p[1], s.i[4], s.i[5], s.i[6], (void*)s.p[2], stream);
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
