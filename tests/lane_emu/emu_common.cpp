// emu_common.cpp -- TEST INFRASTRUCTURE: what the emulated kernel sources expect from the rest of the library
#include "medt_common.h"

namespace medt {
alignas(16) float smem[160 * 1024 / 4];          // the workgroup's LDS (extern __shared__ in the kernels)
static char g_err[256];
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int launch_status(const char*) { return MEDT_OK; }
bool abl_skip(const char*) { return false; }
}  // namespace medt

static const bool lds_registered = (lane_emu::set_lds(medt::smem, sizeof(medt::smem)), true);

extern "C" void emu_set_order(int mode, unsigned long long seed) { lane_emu::set_order(mode, seed); }
extern "C" const char* emu_last_error() { return medt::g_err; }
