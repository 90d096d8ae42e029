// emu_block.cpp -- TEST INFRASTRUCTURE: the one-launch block kernels' own source (csrc/block_small.hip) compiled for the CPU
// lane emulator, behind two C entry points that take host pointers where the product takes device pointers.
#define MEDT_LANE_EMU 1
#include "../../medical-transformer_amd/csrc/block_small.hip"


extern "C" int emu_wopos_block_fwd(const medt_block_desc* d, const medt_block_params* p, const float* x, float* y,
                                   const medt_block_saved* sv, double* parts) {
    if (!medt::wopos_block_ok(*d)) return MEDT_EUNSUPPORTED;
    return medt::wopos_block_fwd(*d, *p, x, y, *sv, parts, nullptr);
}
extern "C" size_t emu_wopos_block_part_doubles(const medt_block_desc* d) { return medt::wopos_block_part_doubles(*d); }

// the backward kernel alone, outputs as plain host arrays (part: the eight BatchNorms' partial rows, blk_part_off order)
extern "C" int emu_wopos_block_bwd(const medt_block_desc* d, const medt_block_params* p, const float* y, const float* dy,
                                   const float* dx_add, const medt_block_saved* sv, float* dz2, float* dz1, float* const* dqkv,
                                   float* const* coef_q, float* part, float* dx) {
    float* const dq[2] = {dqkv[0], dqkv[1]};
    float* const cq[2] = {coef_q[0], coef_q[1]};
    return medt::wopos_block_bwd_launch(*d, *p, y, dy, dx_add, *sv, dz2, dz1, dq, cq, part, dx, nullptr);
}
extern "C" size_t emu_wopos_block_bwd_part_floats(const medt_block_desc* d) {
    return medt::blk_part_off(8, d->bn_groups, d->width, d->C, d->G);
}
extern "C" void emu_set_block_pk(int on) { medt::block_pk_mode() = on ? 1 : 0; }
extern "C" void emu_set_block_bwd(int on) { medt::block_bwd_mode() = on ? 1 : 0; }
extern "C" void emu_set_block8(int on) { medt::block8_mode() = on ? 1 : 0; }
