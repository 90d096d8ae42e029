// lane_emu.cpp -- see lane_emu.h (test infrastructure)
#include "lane_emu.h"
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <string.h>
#include <utility>
#include <vector>

namespace lane_emu {

Idx3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

namespace {
constexpr size_t STACK = 256 * 1024;
// Context switch.  x86-64: a dozen instructions (callee-saved registers + stack pointer) instead of swapcontext(), which saves the
// signal mask with a system call on every switch -- the emulator switches ~10^8 times in a whole-network run.
#if defined(__x86_64__)
#define LANE_EMU_FAST_SWITCH 1
extern "C" void lane_emu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl lane_emu_switch
    .type lane_emu_switch,@function
lane_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size lane_emu_switch, .-lane_emu_switch
)");
struct Lane { void* sp; bool done; int xpar; int wait; };     // wait: 0 running, 1 at a wavefront sync, 2 at the workgroup barrier
void* main_sp = nullptr;
#else
struct Lane { ucontext_t ctx; bool done; int xpar; int wait; };
#endif
struct Wave { int count, gen, alive; uint64_t slot[2][64]; };
std::vector<Lane> lanes;
std::vector<Wave> waves;
ucontext_t main_ctx;
int cur = -1, T = 0, alive = 0, bcount = 0, bgen = 0;
long progress = 0;
const std::function<void()>* body_fn = nullptr;

#ifdef LANE_EMU_FAST_SWITCH
void yield() { lane_emu_switch(&lanes[cur].sp, main_sp); }
#else
void yield() { swapcontext(&lanes[cur].ctx, &main_ctx); }
#endif

// Which work-item runs u-th in a pass over the workgroup.  Work-items only interact at the synchronisation points they yield at,
// so a kernel without data races computes the same bits under every order; a missing barrier shows up as a difference between
// orders (the consumer of a tile runs before / after its producer): tests run every kernel under all three.
int order_mode = 0;                 // 0: ascending, 1: descending, 2: pseudo-random, reshuffled every pass
uint64_t rng_state = 0x9E3779B97F4A7C15ull;
std::vector<int> perm;
int pick(int u) {
    if (order_mode == 0) return u;
    if (order_mode == 1) return T - 1 - u;
    if (u == 0) {                    // new pass: Fisher-Yates with a xorshift generator
        perm.resize(T);
        for (int i = 0; i < T; ++i) perm[i] = i;
        for (int i = T - 1; i > 0; --i) {
            rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
            std::swap(perm[i], perm[(int)(rng_state % (uint64_t)(i + 1))]);
        }
    }
    return perm[u];
}

void release_block_if_complete() {
    if (bcount > 0 && bcount == alive) { bcount = 0; ++bgen; ++progress; }
}
void release_wave_if_complete(Wave& w) {
    if (w.count > 0 && w.count == w.alive) { w.count = 0; ++w.gen; ++progress; }
}

void trampoline() {
    (*body_fn)();
    Lane& me = lanes[cur];
    me.done = true;
    --alive;
    Wave& w = waves[cur >> 6];
    --w.alive;
    ++progress;
    release_block_if_complete();               // work-items that have exited do not take part in later barriers
    release_wave_if_complete(w);
#ifdef LANE_EMU_FAST_SWITCH
    for (;;) lane_emu_switch(&me.sp, main_sp);  // (a finished work-item is never resumed)
#endif
    // ucontext: returns to main_ctx through uc_link
}
}  // namespace

int lane_id() { return cur & 63; }

void set_order(int mode, uint64_t seed) {
    order_mode = mode;
    rng_state = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    if (!rng_state) rng_state = 1;
}

void block_barrier() {
    const int gen = bgen;
    ++bcount;
    ++progress;
    lanes[cur].wait = 2;
    release_block_if_complete();
    while (bgen == gen) yield();
    lanes[cur].wait = 0;
}

void wave_barrier() {
    Wave& w = waves[cur >> 6];
    const int gen = w.gen;
    ++w.count;
    ++progress;
    lanes[cur].wait = 1;
    release_wave_if_complete(w);
    while (w.gen == gen) yield();
    lanes[cur].wait = 0;
}

uint64_t exchange(uint64_t v, int src_lane) {
    Wave& w = waves[cur >> 6];
    // two slot sets, chosen by the wavefront's synchronisation count (the same for every lane that arrives at this operation,
    // whatever it skipped before): a lane can be at most one operation ahead of the others, who may still be reading the other set
    const int p = w.gen & 1;
    w.slot[p][cur & 63] = v;
    wave_barrier();
    return w.slot[p][src_lane & 63];
}

// LDS discipline (every array registered with add_lds -- the kernels name their dynamic LDS differently): before every
// workgroup the bytes the launch asked for are filled with signalling garbage (a kernel that reads
// LDS it never wrote computes NaNs, as it would compute garbage on the GPU) and the rest of the array with a canary that must
// survive the workgroup (a kernel that writes past its allocation faults on the GPU).
struct LdsRegion { float* base; size_t bytes; };
static std::vector<LdsRegion>& lds_regions() { static std::vector<LdsRegion> r; return r; }
void add_lds(float* base, size_t bytes) { lds_regions().push_back(LdsRegion{base, bytes}); }
static const uint32_t LDS_GARBAGE = 0x7FA00000u /* NaN */, LDS_CANARY = 0xC0FFEE11u;
static void lds_prepare(size_t used) {
    for (const LdsRegion& r : lds_regions()) {
        uint32_t* w = reinterpret_cast<uint32_t*>(r.base);
        const size_t n = r.bytes / 4, u = used / 4 < n ? used / 4 : n;
        for (size_t i = 0; i < u; ++i) w[i] = LDS_GARBAGE;
        for (size_t i = u; i < n; ++i) w[i] = LDS_CANARY;
    }
}
static void lds_check(size_t used, unsigned bx, unsigned by) {
    for (const LdsRegion& r : lds_regions()) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(r.base);
        for (size_t i = (used + 3) / 4; i < r.bytes / 4; ++i)
            if (w[i] != LDS_CANARY) {
                fprintf(stderr, "lane_emu: workgroup (%u,%u) wrote LDS byte %zu, past the %zu bytes the launch requested\n", bx, by, i * 4, used);
                abort();
            }
    }
}

// Divergence: a cross-lane operation inside a branch that only some lanes of a wavefront take.  On the GPU the others are masked
// off; here they have run ahead and sit at the workgroup barrier behind the branch (or have exited), so they can never arrive at
// this operation.  When nothing else can move, a wavefront whose remaining lanes are all at the workgroup barrier lets its waiting
// lanes go (they read stale values for the absent lanes -- undefined on the GPU too).  Lanes of one wavefront waiting at two
// DIFFERENT cross-lane operations cannot be told apart from a real deadlock: that aborts.
static bool release_divergent_waves() {
    bool any = false;
    for (size_t wi = 0; wi < waves.size(); ++wi) {
        Wave& w = waves[wi];
        if (w.count == 0) continue;
        int at_wave = 0, at_block = 0;
        for (int t = (int)wi * 64; t < T && t < (int)wi * 64 + 64; ++t) {
            if (lanes[t].done) continue;
            at_wave += lanes[t].wait == 1;
            at_block += lanes[t].wait == 2;
        }
        if (at_wave == w.count && at_wave + at_block == w.alive && at_block > 0) {
            w.count = 0; ++w.gen; ++progress;
            any = true;
        }
    }
    return any;
}

void launch(Idx3 grid, Idx3 block, size_t lds_bytes, const std::function<void()>& body) {
    if (block.y != 1 || block.z != 1) { fprintf(stderr, "lane_emu: 1-D workgroups only\n"); abort(); }
    T = (int)block.x;
    g_blockDim = block;
    g_gridDim = grid;
    body_fn = &body;
    char* stacks = (char*)mmap(nullptr, STACK * T, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == MAP_FAILED) { perror("lane_emu: mmap"); abort(); }
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = Idx3{bx, by, bz};
                lds_prepare(lds_bytes);
                lanes.assign(T, Lane());
                waves.assign((T + 63) / 64, Wave());
                alive = T; bcount = 0; bgen = 0;
                for (int t = 0; t < T; ++t) {
                    Lane& l = lanes[t];
                    l.done = false; l.xpar = 0; l.wait = 0;
#ifdef LANE_EMU_FAST_SWITCH
                    // initial frame: six callee-saved registers + the entry point as return address; the stack pointer is
                    // 8 mod 16 when trampoline starts, as after a call
                    void** top = reinterpret_cast<void**>(stacks + STACK * (t + 1));
                    void** base = top - 8;
                    memset(base, 0, 8 * sizeof(void*));
                    base[6] = reinterpret_cast<void*>(&trampoline);
                    l.sp = base;
#else
                    getcontext(&l.ctx);
                    l.ctx.uc_stack.ss_sp = stacks + STACK * t;
                    l.ctx.uc_stack.ss_size = STACK;
                    l.ctx.uc_link = &main_ctx;
                    makecontext(&l.ctx, trampoline, 0);
#endif
                    ++waves[t >> 6].alive;
                }
                while (alive > 0) {
                    const long before = progress;
                    for (int u = 0; u < T; ++u) {
                        const int t = pick(u);
                        if (lanes[t].done) continue;
                        cur = t;
                        g_threadIdx = Idx3{(unsigned)t, 0, 0};
#ifdef LANE_EMU_FAST_SWITCH
                        lane_emu_switch(&main_sp, lanes[t].sp);
#else
                        swapcontext(&main_ctx, &lanes[t].ctx);
#endif
                    }
                    if (progress == before && alive > 0 && !release_divergent_waves()) {
                        fprintf(stderr, "lane_emu: deadlock in workgroup (%u,%u,%u): %d work-items alive, %d at the barrier\n",
                                bx, by, bz, alive, bcount);
                        abort();
                    }
                }
                lds_check(lds_bytes, bx, by);
            }
    munmap(stacks, STACK * T);
    cur = -1;
}

}  // namespace lane_emu
