// TEST INFRASTRUCTURE ONLY.
// Minimal single-host-thread SPMD emulator so that the *unmodified* device source
// (deepmimic_amd/csrc/dm_device.h) can be compiled with g++ and run lane-for-lane on the CPU:
// each of the 64 lanes of a wavefront is a cooperative fiber; __syncthreads() (and the wave
// cross-lane helpers built on it) yield to the next lane.  Used by `-m "not gpu"` tests to
// check the kernel logic against the oracle without a GPU.  Never part of the product build.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <functional>

#define DM_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct emu_dim3 { unsigned x = 1, y = 1, z = 1; };
extern emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {
struct Fiber { void* sp = nullptr; char* stack = nullptr; bool done = false; };
void barrier();                       // yield point: returns when every live lane arrived
void launch(unsigned grid, unsigned block, const std::function<void()>& body);
extern uint64_t g_xchg[8192];         // scratch for cross-lane helpers
}

static inline void __syncthreads() { emu::barrier(); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
