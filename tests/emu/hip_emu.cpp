// TEST INFRASTRUCTURE ONLY -- fiber scheduler for hip_emu.h (x86-64 System V).
#include "hip_emu.h"
#include <cassert>
#include <cstdio>

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
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

namespace emu {
static std::vector<Fiber> g_fibers;
static void* g_sched_sp = nullptr;
static int g_cur = -1;
static const std::function<void()>* g_body = nullptr;

static void fiber_main() {
    (*g_body)();
    g_fibers[g_cur].done = true;
    emu_switch(&g_fibers[g_cur].sp, g_sched_sp);
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

void barrier() {
    emu_switch(&g_fibers[g_cur].sp, g_sched_sp);   // back to the scheduler; resumed in the next round
}

void launch(unsigned grid, unsigned block, const std::function<void()>& body) {
    const size_t kStack = 1u << 20;
    g_body = &body;
    gridDim.x = grid; blockDim.x = block;
    for (unsigned b = 0; b < grid; ++b) {
        blockIdx.x = b;
        g_fibers.assign(block, Fiber());
        for (unsigned t = 0; t < block; ++t) {
            Fiber& f = g_fibers[t];
            f.stack.resize(kStack);
            uintptr_t top = ((uintptr_t)f.stack.data() + kStack) & ~(uintptr_t)63;
            uint64_t* sp = (uint64_t*)top;
            *--sp = 0; *--sp = 0;                   // padding: rsp must be 16-byte aligned at the trampoline's call
            *--sp = (uint64_t)&emu_trampoline;      // ret target
            *--sp = 0;                              // rbp
            *--sp = 0;                              // rbx
            *--sp = (uint64_t)&fiber_main;          // r12
            *--sp = 0; *--sp = 0; *--sp = 0;        // r13 r14 r15
            uint32_t csr[2]; asm volatile("stmxcsr %0" : "=m"(csr[0])); uint16_t cw; asm volatile("fnstcw %0" : "=m"(cw)); csr[1] = cw;
            --sp; std::memcpy(sp, csr, 8);
            f.sp = sp;
        }
        unsigned live = block;
        while (live) {
            live = 0;
            for (unsigned t = 0; t < block; ++t) {
                if (g_fibers[t].done) continue;
                g_cur = (int)t; threadIdx.x = t;
                emu_switch(&g_sched_sp, g_fibers[t].sp);
                if (!g_fibers[t].done) ++live;
            }
        }
    }
    g_body = nullptr;
}
}  // namespace emu
