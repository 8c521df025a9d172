// api.hip -- error plumbing and library identification for libgear_hip.so
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void gear_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gear_last_error(void) { return g_err; }
extern "C" int gear_abi_version(void) { return 5; }

// ---- run-time options ------------------------------------------------------------------------------------------
// A handful of switches select an alternative (always exact) code path; the tests use them to cover the paths a normal
// input does not reach.  They are read from the environment ONCE, when the library is first used, and can be set through
// gear_set_option(); the launch paths read the cached table (no getenv() per call).
GearOptions& gear_options() {
    static GearOptions o = [] {
        GearOptions v{};
        auto flag = [](const char* n) { const char* e = getenv(n); return e && *e && *e != '0'; };
        v.attn_generic = flag("GEAR_ATTN_GENERIC");
        v.lowrank_generic = flag("GEAR_LOWRANK_GENERIC");
        v.rows_hist_only = flag("GEAR_ROWS_HIST_ONLY");
        v.rows_v1 = flag("GEAR_ROWS_V1");
        v.rows_wg_only = flag("GEAR_ROWS_WG_ONLY");
        v.kfused_generic = flag("GEAR_KFUSED_GENERIC");
        v.kselect_slow = flag("GEAR_KSELECT_SLOW");
        v.kfused_no_tr = flag("GEAR_KFUSED_NO_TR");
        auto ival = [](const char* n) { const char* e = getenv(n); return (e && *e) ? atoi(e) : 0; };
        v.gram_fused = ival("GEAR_GRAM_FUSED");
        v.rows_masked = ival("GEAR_ROWS_MASKED");
        v.gram_nstg = ival("GEAR_GRAM_NSTG");
        v.decomp_general = ival("GEAR_DECOMP_GENERAL");
        v.attn_gqa_group = ival("GEAR_ATTN_GQA_GROUP");
        v.attn_win_chunk = ival("GEAR_ATTN_WIN_CHUNK");
        v.attn_keep_chunk_index = ival("GEAR_ATTN_KEEP_CHUNK_INDEX");
        v.kfused_nslab = ival("GEAR_KFUSED_NSLAB");
        v.kfused_one = ival("GEAR_KFUSED_ONE");
        v.kfused_main = ival("GEAR_KFUSED_MAIN");
        v.kfused_eout = ival("GEAR_KFUSED_EOUT");
        v.decomp_rpb = ival("GEAR_DECOMP_RPB");
        v.attn_fold = ival("GEAR_ATTN_FOLD");
        v.attn_mfma = ival("GEAR_ATTN_MFMA");
        return v;
    }();
    return o;
}

extern "C" int gear_set_option(const char* name, int value) {
    GearOptions& o = gear_options();
    const struct { const char* n; int* p; } tab[] = {
        {"attn_generic", &o.attn_generic},     {"lowrank_generic", &o.lowrank_generic}, {"rows_hist_only", &o.rows_hist_only},
        {"rows_v1", &o.rows_v1}, {"rows_masked", &o.rows_masked}, {"rows_wg_only", &o.rows_wg_only},               {"kfused_generic", &o.kfused_generic},   {"kselect_slow", &o.kselect_slow},
        {"kfused_no_tr", &o.kfused_no_tr}, {"gram_fused", &o.gram_fused}, {"gram_nstg", &o.gram_nstg},
        {"decomp_general", &o.decomp_general}, {"attn_gqa_group", &o.attn_gqa_group}, {"attn_win_chunk", &o.attn_win_chunk}, {"attn_keep_chunk_index", &o.attn_keep_chunk_index}, {"kfused_nslab", &o.kfused_nslab}, {"kfused_one", &o.kfused_one}, {"kfused_main", &o.kfused_main}, {"kfused_eout", &o.kfused_eout}, {"decomp_rpb", &o.decomp_rpb}, {"attn_fold", &o.attn_fold}, {"attn_mfma", &o.attn_mfma},
    };
    for (const auto& t : tab)
        if (name && !strcmp(name, t.n)) { *t.p = value; return 0; }
    gear_set_error("gear_set_option: unknown option '%s'", name ? name : "(null)");
    return -1;
}
