// emu_conv_small.cpp -- TEST INFRASTRUCTURE: the fused small conv + BatchNorm kernels' own source (csrc/conv_small.hip:
// conv1x1_bn_small_fwd, bn_act_bwd_small / _chan, bn_dgrad1x1_small) compiled for the CPU lane emulator.
#define MEDT_LANE_EMU 1
#include "../../medical-transformer_amd/csrc/conv_small.hip"

// partials: [groups][Cout][2] DOUBLES
extern "C" int emu_conv_small_fwd(const medt_conv_desc* d, const float* x, const float* w, const medt_bn_ptrs* bn, const float* res,
                                  float* z, float* y, double* partials) {
    if (!medt::conv_small_ok(*d)) return MEDT_EUNSUPPORTED;
    return medt::conv_small_fwd(*d, x, w, *bn, res, z, y, (float*)partials, nullptr);
}
// stats: the block's statistics block (mean | rstd | scale | shift, [groups][Cout] each); partials [groups][Cout][2] floats
extern "C" int emu_bn_dgrad1x1_small(const medt_conv_desc* d, const float* dy, const float* y, const float* z, float* stats,
                                     const float* gamma, const float* w, const float* dx_add, float* g, float* dz, float* partials,
                                     float* dx) {
    if (!medt::bn_dgrad1x1_small_ok(*d)) return MEDT_EUNSUPPORTED;
    return medt::bn_dgrad1x1_small(*d, dy, y, z, medt::BnStats(stats, d->bn_groups * d->Cout), gamma, w, dx_add, g, dz, partials, dx,
                                   nullptr);
}
// which: 0 = one wave per (group, channel) (bn_act_bwd_small), 1 = one workgroup per (group, channel) (bn_act_bwd_chan)
extern "C" int emu_bn_act_bwd(const medt_conv_desc* d, int which, const float* dy, const float* y, const float* z, float* stats,
                              const float* gamma, float* g, float* dz, float* partials) {
    const int HoWo = (d->H / d->stride) * (d->W / d->stride);
    const medt::BnStats st(stats, d->bn_groups * d->Cout);
    if (which == 0) {
        if (!medt::conv_small_ok(*d)) return MEDT_EUNSUPPORTED;
        return medt::bn_act_bwd_small(*d, dy, y, z, st, gamma, g, dz, partials, HoWo, nullptr);
    }
    if (!medt::bn_chan_threads(*d, HoWo)) return MEDT_EUNSUPPORTED;
    return medt::bn_act_bwd_chan(*d, dy, y, z, st, gamma, g, dz, partials, HoWo, nullptr);
}
