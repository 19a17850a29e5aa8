// Error plumbing and descriptor validation for libmtc.
#include <stdarg.h>
#include <string.h>
#include "common.hpp"

namespace mtc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_desc(const mtc_desc* d, const char* who) {
    if (!d) { set_error("%s: null descriptor", who); return -1; }
    if (d->B <= 0 || d->D <= 0 || d->A_s <= 0 || d->A_t <= 0) {
        set_error("%s: B, D, A_s, A_t must be positive (got %d %d %d %d)", who, d->B, d->D, d->A_s, d->A_t);
        return -1;
    }
    if ((d->A_s & 7) || (d->A_t & 7) || (d->D & 7)) {
        set_error("%s: A_s, A_t and D must be multiples of 8 (got %d %d %d)", who, d->A_s, d->A_t, d->D);
        return -1;
    }
    if (d->A_t > 8192 || d->A_s > 8192) { set_error("%s: activation width > 8192 unsupported", who); return -1; }
    if (d->dtype != MTC_F32 && d->dtype != MTC_BF16) { set_error("%s: bad dtype %d", who, d->dtype); return -1; }
    if (d->G < 1 || d->G > MTC_MAX_PREFIX) { set_error("%s: G must be in 1..%d", who, MTC_MAX_PREFIX); return -1; }
    int prev = 0;
    for (int i = 0; i < d->G; ++i) {
        if (d->prefix[i] < prev) { set_error("%s: prefix sizes must be non-decreasing", who); return -1; }
        prev = d->prefix[i];
    }
    if (d->prefix[d->G - 1] != d->D) { set_error("%s: last prefix (%d) must equal D (%d)", who, prev, d->D); return -1; }
    if ((uint64_t)d->B * (uint64_t)d->D >= (1ull << 40)) { set_error("%s: B*D too large", who); return -1; }
    // the row gathers address whole matrices through 32-bit buffer offsets (rowops.hpp)
    const uint64_t a_max = (uint64_t)(d->A_s > d->A_t ? d->A_s : d->A_t), es = d->dtype == MTC_BF16 ? 2 : 4;
    if ((uint64_t)d->D * a_max * es >= (1ull << 31) || (uint64_t)d->B * a_max * 4ull >= (1ull << 31)) {
        set_error("%s: a weight or batch matrix of 2 GiB or more is unsupported (D*A*elem_size, B*A*4)", who);
        return -1;
    }
    return 0;
}

int check_launch(const char* who) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return -2; }
    return 0;
}

}  // namespace mtc

extern "C" const char* mtc_last_error(void) { return mtc::g_err; }
extern "C" int mtc_version(void) { return MTC_VERSION; }
