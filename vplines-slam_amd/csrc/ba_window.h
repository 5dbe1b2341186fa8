// k_window: the whole solve of one window -- k_prep, k_lin, [k_solve, k_cost, k_lin] x iterations, k_gauge, the
// marginalisation -- as ONE kernel, one workgroup per window, the phases being the very bodies of the stand-alone kernels
// called back to back with a workgroup barrier in between (a window's phases talk to each other through its own slices
// of the batch buffers only, so workgroup-scope visibility is all that is needed).
//
// Why it exists: windows are independent and their work per trust-region iteration differs (a window whose last step
// was rejected neither re-linearises nor re-factors), so a launch per phase over the batch waits for its unluckiest CU.
// Inside one kernel a CU finishes its window and takes the next one.
// MEASURED (512 windows, MI355X): 137.7 k solves/s against 150.5 k for the kernel-per-phase sequence -- NOT the default.
// The rocprof trace explains it: in iterations 1-3 every window is heavy anyway (two full rounds per launch), in
// iterations 4-5 fewer than 256 are and the launch already takes a single round; what imbalance is left (~9 %) is eaten by
// the call overhead (callee-saved registers through scratch, 480-720 B per phase) and by 8 instead of 10 waves in the
// preparation phase.  Kept for small batches / single-window latency (one launch instead of ~19); VPL_BA_FUSED=1.
#pragma once
#include "ba_lin.h"
#include "ba_solve.h"
#include "ba_marg.h"

namespace vpl {

constexpr int WINDOW_THREADS = 512;
static_assert(LIN_THREADS == WINDOW_THREADS && SOLVE_THREADS == WINDOW_THREADS && COST_THREADS == WINDOW_THREADS &&
              MARG_THREADS == WINDOW_THREADS, "the phase bodies are written for one workgroup size");

// Each phase is a real (not inlined) function: inlined into one body the register allocator keeps values of one phase alive
// through the others and spills several hundred registers; as calls every phase gets the allocation of its stand-alone
// kernel and k_window itself holds next to nothing across them.  Function arguments travel in vector registers, which
// would turn the batch descriptor's fields into per-lane loads; the descriptor therefore lives in constant memory (one slot
// per context) and the callees re-establish that slot and window are wave-uniform: scalar loads, as in the kernels.
constexpr int WINDOW_SLOTS = 16;
__constant__ DevBatch c_window_batch[WINDOW_SLOTS];

#define VPL_PHASE(name, call)                                                                  \
  __device__ __attribute__((noinline)) void name(int slot, int w) {                            \
    extern __shared__ double sm[];                                                             \
    const DevBatch& B = c_window_batch[__builtin_amdgcn_readfirstlane(slot)];                  \
    w = __builtin_amdgcn_readfirstlane(w);                                                     \
    call;                                                                                      \
  }
VPL_PHASE(lin0_call, lin_body<0>(B, w, sm))
VPL_PHASE(lin1_call, lin_body<1>(B, w, sm))
VPL_PHASE(lin2_call, lin_body<2>(B, w, sm))
VPL_PHASE(solve_call, solve_body(B, w, sm))
VPL_PHASE(cost_call, cost_body(B, w))
VPL_PHASE(gauge_call, gauge_body(B, w))
VPL_PHASE(marg_call, marg_body(B, w, sm))
#undef VPL_PHASE
__device__ __attribute__((noinline)) void prep_call(int slot, int w, int nstage) {
  extern __shared__ double sm[];
  const DevBatch& B = c_window_batch[__builtin_amdgcn_readfirstlane(slot)];
  prep_body(B, __builtin_amdgcn_readfirstlane(w), sm, __builtin_amdgcn_readfirstlane(nstage));
}

// marg_mode: 0 none, 1 MARGIN_OLD (k_lin<1> + k_marg), 2 MARGIN_SECOND_NEW (k_lin<2> + k_marg)
__global__ __launch_bounds__(WINDOW_THREADS) void k_window(int slot, int nstage, int iterations, int marg_mode) {
  const int w = blockIdx.x + c_window_batch[slot].w0;
  prep_call(slot, w, nstage);
  __syncthreads();
  for (int it = -1; it < iterations; ++it) {
    if (it >= 0) {
      solve_call(slot, w);
      __syncthreads();
      cost_call(slot, w);
      __syncthreads();
    }
    if (it + 1 < iterations) {
      lin0_call(slot, w);
      __syncthreads();
    }
  }
  gauge_call(slot, w);
  __syncthreads();
  if (marg_mode == 1) {
    lin1_call(slot, w);
    __syncthreads();
    marg_call(slot, w);
  } else if (marg_mode == 2) {
    lin2_call(slot, w);
    __syncthreads();
    marg_call(slot, w);
  }
}

// static LDS of the phases that k_window inherits (k_cost: 374 doubles + 1 int, k_gauge: 99 doubles, k_marg: 6 ints), rounded up
constexpr size_t WINDOW_STATIC_LDS = 4096;

}  // namespace vpl
