// emu_common.cpp -- TEST INFRASTRUCTURE: the arrays that play the LDS, the emulator's switches
#include "medt_common.h"

namespace medt {
// the workgroup's LDS: the kernels' `extern __shared__` arrays (a launch uses one of them)
alignas(16) float smem[160 * 1024 / 4];
alignas(16) float wl[160 * 1024 / 4];            // conv.hip, pointwise.hip
alignas(16) float sm[160 * 1024 / 4];            // elementwise.hip
namespace fast_f32 { alignas(16) float smem[160 * 1024 / 4]; }       // axial_fast.hip, compiled twice
namespace fast_bf16 { alignas(16) float smem[160 * 1024 / 4]; }
}  // namespace medt

static const bool lds_registered = (lane_emu::add_lds(medt::smem, sizeof(medt::smem)), lane_emu::add_lds(medt::wl, sizeof(medt::wl)),
                                    lane_emu::add_lds(medt::sm, sizeof(medt::sm)),
                                    lane_emu::add_lds(medt::fast_f32::smem, sizeof(medt::fast_f32::smem)),
                                    lane_emu::add_lds(medt::fast_bf16::smem, sizeof(medt::fast_bf16::smem)), true);

extern "C" void emu_set_order(int mode, unsigned long long seed) { lane_emu::set_order(mode, seed); }
extern "C" const char* emu_last_error() { return medt_last_error(); }
