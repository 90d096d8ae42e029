// axial_bwd.hip keeps its kernels in an unnamed namespace: their `extern __shared__` arrays are members of THAT namespace, so the
// arrays that play the LDS are defined in the same translation unit (g++ ... -include emu_axial_bwd_pre.h -x c++ axial_bwd.hip)
#pragma once
#include "lane_emu.h"
namespace medt { namespace {
alignas(16) float smem[160 * 1024 / 4];
alignas(16) float pgs[160 * 1024 / 4];
alignas(16) float lds[160 * 1024 / 4];
const bool lds_registered = (lane_emu::add_lds(smem, sizeof(smem)), lane_emu::add_lds(pgs, sizeof(pgs)),
                             lane_emu::add_lds(lds, sizeof(lds)), true);
} }
