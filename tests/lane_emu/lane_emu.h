// lane_emu.h -- TEST INFRASTRUCTURE: runs a HIP kernel's SOURCE on the CPU, one coroutine per work-item.
//
// The one-workgroup block kernels (medical-transformer_amd/csrc/block_small.hip) are written against a handful of gfx950
// cross-lane primitives (DPP moves, v_permlane{16,32}_swap, v_readlane) and workgroup barriers.  This emulator compiles the
// very same .hip file with g++ (shim/hip/hip_runtime.h maps the HIP vocabulary onto it) and executes a launch as
// grid x block cooperative coroutines: a work-item runs until it reaches a barrier or a cross-lane operation, where it
// yields until the other work-items of its workgroup / wavefront have arrived.  Host pointers play device memory, one
// global array plays the LDS.  It proves index arithmetic, phase ordering (a missing barrier shows up as a wrong result
// only when the interleaving exposes it -- set_order() runs the work-items in ascending, descending or shuffled order between
// synchronisation points, and a race-free kernel computes the same bits under all of them) and the mathematics of a kernel
// without a GPU; it says nothing about timing, register pressure or the real ISA.
// Nothing in the product links or imports this (tests/test_lane_emu.py only).
#pragma once
#include <stdint.h>
#include <functional>

namespace lane_emu {

struct Idx3 { unsigned x, y, z; };
extern Idx3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

void launch(Idx3 grid, Idx3 block, size_t lds_bytes, const std::function<void()>& body);
void add_lds(float* base, size_t bytes);      // an array that plays the LDS (poisoned / canaried around every workgroup)
void block_barrier();                         // __syncthreads / s_barrier
void wave_barrier();                          // a point every lane of the wavefront reaches together
uint64_t exchange(uint64_t v, int src_lane);  // every lane of the wavefront posts v and reads lane src_lane's (0..63)
int lane_id();
void set_order(int mode, uint64_t seed);   // order of the work-items between synchronisation points: 0 ascending, 1 descending, 2 random

}  // namespace lane_emu
