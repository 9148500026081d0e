// TEST INFRASTRUCTURE ONLY -- fiber scheduler for hip_emu.h (x86-64 System V).
#include "hip_emu.h"
#include <cassert>
#include <cstdio>
#include <sys/mman.h>

emu_dim3 threadIdx, blockIdx, blockDim, gridDim;
namespace emu { uint64_t g_xchg[8192]; }

extern "C" void emu_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
    popq %rcx
    jmp *%rcx          # not `ret`: the return-address predictor holds another lane's history, an indirect jump predicts the common resume point
.size emu_switch,.-emu_switch
)");

namespace emu {
static std::vector<Fiber> g_fibers;
static void* g_sched_sp = nullptr;
static int g_cur = -1;
static const std::function<void()>* g_body = nullptr;
static const size_t kStack = 1u << 20;
static std::vector<char*> g_stacks;      // one mapping per lane, kept for the life of the process

// A lane's stack reads as zeros at the start of every block (as a fresh allocation would): the pages of the last use are dropped,
// which costs a page fault per page a lane touches instead of clearing 64 MiB per block.
static char* lane_stack(unsigned t) {
    if (t < g_stacks.size()) { madvise(g_stacks[t], kStack, MADV_DONTNEED); return g_stacks[t]; }
    void* p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("emu: mmap"); abort(); }
    g_stacks.push_back((char*)p);
    return (char*)p;
}

static unsigned g_block = 0, g_live = 0;

// Lanes run in cyclic order 0, 1, ..., block-1, 0, ... (finished ones skipped), each until its next yield point; a lane hands over to its
// successor directly (one stack switch per yield, none through the scheduler).  MXCSR / the x87 control word are not switched: no lane changes them.
static inline int next_live(int cur) {
    int t = cur;
    do { t = (t + 1 == (int)g_block) ? 0 : t + 1; } while (g_fibers[t].done && t != cur);
    return t;
}
static inline void hand_over(int cur) {
    const int nx = next_live(cur);
    if (nx == cur) return;
    g_cur = nx; threadIdx.x = (unsigned)nx;
    emu_switch(&g_fibers[cur].sp, g_fibers[nx].sp);
}

static void fiber_main() {
    (*g_body)();
    const int cur = g_cur;
    g_fibers[cur].done = true;
    if (--g_live == 0) emu_switch(&g_fibers[cur].sp, g_sched_sp);
    else hand_over(cur);
    abort();
}
extern "C" void emu_trampoline();
asm(R"(
.text
.globl emu_trampoline
.type emu_trampoline,@function
emu_trampoline:
    call *%r12
    ud2
.size emu_trampoline,.-emu_trampoline
)");

static uint64_t g_nb = 0, g_nl = 0;
struct StatsAtExit { ~StatsAtExit() { if (getenv("EMU_STATS")) fprintf(stderr, "emu: %llu barrier yields, %llu blocks\n", (unsigned long long)g_nb, (unsigned long long)g_nl); } } g_stats_at_exit;
void barrier() {
    ++g_nb;
    hand_over(g_cur);                               // resumed when every other live lane has reached its next yield point
}

void launch(unsigned grid, unsigned block, const std::function<void()>& body) {
    g_body = &body;
    gridDim.x = grid; blockDim.x = block;
    for (unsigned b = 0; b < grid; ++b) {
        blockIdx.x = b; ++g_nl;
        g_fibers.assign(block, Fiber());
        for (unsigned t = 0; t < block; ++t) {
            Fiber& f = g_fibers[t];
            f.stack = lane_stack(t);
            // the mappings are 1 MiB apart, so equal stack depths of the 64 lanes would share one cache set; 65 lines of stagger per lane spread them
            uintptr_t top = (((uintptr_t)f.stack + kStack) & ~(uintptr_t)63) - (uintptr_t)(t % 64) * 4160u;
            uint64_t* sp = (uint64_t*)top;
            *--sp = 0; *--sp = 0;                   // padding: rsp must be 16-byte aligned at the trampoline's call
            *--sp = (uint64_t)&emu_trampoline;      // ret target
            *--sp = 0;                              // rbp
            *--sp = 0;                              // rbx
            *--sp = (uint64_t)&fiber_main;          // r12
            *--sp = 0; *--sp = 0; *--sp = 0;        // r13 r14 r15
            f.sp = sp;
        }
        g_block = g_live = block;
        if (block) { g_cur = 0; threadIdx.x = 0; emu_switch(&g_sched_sp, g_fibers[0].sp); }   // returns when the last lane has finished
    }
    g_body = nullptr;
}
}  // namespace emu
