// emu_block.cpp -- TEST INFRASTRUCTURE: the one-launch block kernels' own source (csrc/block_small.hip) compiled for the CPU
// lane emulator, behind two C entry points that take host pointers where the product takes device pointers.
#define MEDT_LANE_EMU 1
#include "../../medical-transformer_amd/csrc/block_small.hip"

namespace medt {
alignas(16) float smem[160 * 1024 / 4];          // the workgroup's LDS (extern __shared__ in the kernels)
void set_error(const char*, ...) {}
int launch_status(const char*) { return MEDT_OK; }
bool abl_skip(const char*) { return false; }
}  // namespace medt

extern "C" int emu_wopos_block_fwd(const medt_block_desc* d, const medt_block_params* p, const float* x, float* y,
                                   const medt_block_saved* sv, double* parts) {
    if (!medt::wopos_block_ok(*d)) return MEDT_EUNSUPPORTED;
    return medt::wopos_block_fwd(*d, *p, x, y, *sv, parts, nullptr);
}
extern "C" size_t emu_wopos_block_part_doubles(const medt_block_desc* d) { return medt::wopos_block_part_doubles(*d); }

// the backward kernel alone: `stats` = the eight statistics blocks (mean | rstd | scale | shift, [groups][CH] each) in the
// kernel's BatchNorm order, outputs as plain host arrays
extern "C" int emu_wopos_block_bwd(const medt_block_desc* d, const medt_block_params* p, const float* y, const float* dy,
                                   const float* dx_add, const medt_block_saved* sv, float* const* stats, float* dz2, float* dz1,
                                   float* const* dqkv, float* const* coef_q, float* const* part, float* dx) {
    const int gs = d->bn_groups, chs[8] = {d->width, 2 * d->width, d->G, d->width, 2 * d->width, d->G, d->width, d->C};
    medt::BnStats st[8];
    for (int b = 0; b < 8; ++b) st[b] = medt::BnStats(stats[b], gs * chs[b]);
    float* const dq[2] = {dqkv[0], dqkv[1]};
    float* const cq[2] = {coef_q[0], coef_q[1]};
    float* const pt[8] = {part[0], part[1], part[2], part[3], part[4], part[5], part[6], part[7]};
    return medt::wopos_block_bwd_launch(*d, *p, y, dy, dx_add, *sv, st, dz2, dz1, dq, cq, pt, dx, nullptr);
}
