// Chains of small kernels executed by ONE workgroup in ONE launch (small models: M <= 64 inducing points).
//
// A training iteration of the Bayesian-optimisation surrogates (SO_BO.py:248-258: tens of inducing points) is ~90 launches
// of 3-13 us kernels, two thirds of them the per-layer small-matrix chains (Kuu -> Cholesky -> W, u -> KL ...; gradient
// finish: ~20 dependent launches per layer).  Each dependent launch costs a kernel boundary (~3 us on this runtime even
// for an empty kernel) plus ~7 us of host time through the library's launchers, so the chains, not their arithmetic, set
// the iteration time (NOTES.md: "small models are launch-bound").
//
// Here the SAME host code that enqueues a chain (prep_layer / finish_layer in dgp_ctx.h / dgp_abi.hip) is run once in
// RECORDING mode: every launcher of a chain kernel appends {kernel id, grid, arguments} to a list instead of launching.
// The list is uploaded once per (model, layer, chain) and replayed by chain_exec_kernel: one workgroup of 256 threads walks
// the list and runs each kernel's body over its "virtual blocks" (the body is literally the stand-alone kernel's: its
// blockIdx / gridDim / threadIdx / blockDim are parameters that shadow the built-ins), with a workgroup barrier between
// steps - global memory written by one step is read by the next through the CU's own L1, which is coherent for the
// wavefronts of one workgroup.  Results are bit-identical to the launch-by-launch chain (same bodies, same order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <type_traits>
#include <vector>

#include "gemm_f64.h"

namespace dgp {

struct VIdx { unsigned x, y, z; };

// how a body's barrier is realised: a stand-alone kernel / a chain step run by all 256 threads: __syncthreads(); a chain
// step whose (64-thread) virtual blocks are dealt to the four waves: the wave alone (LDS accesses of one wave are in order)
enum ChainMode : int { CM_KERNEL = 0, CM_CHAIN = 1, CM_CHAIN_WAVE = 2, CM_WAVE_EXT = 3 };   // 3: one wave inside another kernel, which lends the LDS (g_body_ext)
template <int MODE>
__device__ __forceinline__ void body_sync() {
  if constexpr (MODE == CM_CHAIN_WAVE || MODE == CM_WAVE_EXT) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

enum ChainKid : int {
  CK_RBF_KUU = 0, CK_LEAF32, CK_LEAF64, CK_COPY_BLOCK, CK_ZERO_BLOCK, CK_PACK_Q, CK_LQ_TO_WCAT, CK_COPY, CK_MAKE_Z1,
  CK_KUF_BWD_FINISH, CK_LOWER_DOT, CK_WCAT_T, CK_SUB_IDENTITY, CK_SYMMETRIZE, CK_SUB_SCALARS, CK_LAYER_KL, CK_WBAR_TOTAL,
  CK_LQBAR, CK_LUBAR, CK_PHI, CK_KUU_BWD, CK_UNPACK_Q, CK_ZERO1, CK_COPY2D,
  CK_GEMM_NN, CK_GEMM_NT, CK_GEMM_TN,
  CK_COUNT
};

union ChainSlot { const void* p; long l; double d; };
constexpr int kChainSlots = 16;
struct ChainOp {
  int kid;
  unsigned gx, gy, gz;
  union {
    ChainSlot s[kChainSlots];
    char raw[384];            // (GemmArgs by value for the small products)
  } a;
};
static_assert(sizeof(GemmArgs) <= 384, "GemmArgs must fit a chain op");

struct ChainRec {
  std::vector<ChainOp> ops;
  bool failed = false;      // a launch that cannot be recorded was attempted (nothing was issued): the chain is not replayable
};
// the recorder of the calling thread (nullptr: launch normally)
ChainRec* chain_recorder();
void chain_set_recorder(ChainRec* r);

template <typename T>
inline ChainSlot chain_slot(T v) {
  ChainSlot s;
  s.l = 0;
  if constexpr (std::is_pointer_v<T>) s.p = reinterpret_cast<const void*>(v);
  else if constexpr (std::is_floating_point_v<T>) s.d = (double)v;
  else s.l = (long)v;
  return s;
}
template <typename T>
__device__ __forceinline__ T chain_get(const ChainSlot& s) {
  if constexpr (std::is_pointer_v<T>) return reinterpret_cast<T>(const_cast<void*>(s.p));
  else if constexpr (std::is_floating_point_v<T>) return (T)s.d;
  else return (T)s.l;
}

// kernel tags: they only give the stand-alone kernels readable names in profiles (chain_global<dgp::pack_q_kernel, ...>)
struct rbf_kuu_kernel { static constexpr int id = CK_RBF_KUU; };
struct leaf_potrf_inv_32_kernel { static constexpr int id = CK_LEAF32; };
struct leaf_potrf_inv_64_kernel { static constexpr int id = CK_LEAF64; };
struct copy_block_kernel { static constexpr int id = CK_COPY_BLOCK; };
struct zero_block_kernel { static constexpr int id = CK_ZERO_BLOCK; };
struct pack_q_kernel { static constexpr int id = CK_PACK_Q; };
struct lq_to_wcat_kernel { static constexpr int id = CK_LQ_TO_WCAT; };
struct copy_kernel { static constexpr int id = CK_COPY; };
struct make_z1_kernel { static constexpr int id = CK_MAKE_Z1; };
struct rbf_kuf_bwd_finish_kernel { static constexpr int id = CK_KUF_BWD_FINISH; };
struct lower_dot_kernel { static constexpr int id = CK_LOWER_DOT; };
struct wcat_transpose_kernel { static constexpr int id = CK_WCAT_T; };
struct sub_identity_kernel { static constexpr int id = CK_SUB_IDENTITY; };
struct symmetrize_lower_kernel { static constexpr int id = CK_SYMMETRIZE; };
struct sub_scalars_kernel { static constexpr int id = CK_SUB_SCALARS; };
struct layer_kl_kernel { static constexpr int id = CK_LAYER_KL; };
struct wbar_total_kernel { static constexpr int id = CK_WBAR_TOTAL; };
struct lqbar_finish_kernel { static constexpr int id = CK_LQBAR; };
struct lubar_finish_kernel { static constexpr int id = CK_LUBAR; };
struct phi_tril_halfdiag_kernel { static constexpr int id = CK_PHI; };
struct rbf_kuu_bwd_kernel { static constexpr int id = CK_KUU_BWD; };
struct unpack_q_grads_kernel { static constexpr int id = CK_UNPACK_Q; };
struct zero_doubles_kernel { static constexpr int id = CK_ZERO1; };
struct copy_2d_kernel { static constexpr int id = CK_COPY2D; };

// ChainBody<KID>: kBlock (threads of the stand-alone kernel) and  template <int MODE> static void run(blockIdx, gridDim,
// threadIdx, blockDim, args...)  - defined in chain_bodies.h
template <int KID> struct ChainBody;

template <typename TAG, typename... A>
__global__ void __launch_bounds__(ChainBody<TAG::id>::kBlock) chain_global(A... a) {
  ChainBody<TAG::id>::template run<CM_KERNEL>(VIdx{blockIdx.x, blockIdx.y, blockIdx.z}, VIdx{gridDim.x, gridDim.y, gridDim.z},
                                          VIdx{threadIdx.x, 0u, 0u}, VIdx{blockDim.x, 1u, 1u}, a...);
}

// launch the kernel KID now, or append it to the calling thread's recording
template <typename TAG, typename... A>
inline hipError_t chain_launch(hipStream_t st, dim3 grid, A... a) {
  constexpr int KID = TAG::id;
  static_assert(sizeof...(A) <= kChainSlots, "too many arguments for a chain op");
  if (ChainRec* r = chain_recorder()) {
    ChainOp op;
    memset(&op, 0, sizeof op);
    op.kid = KID; op.gx = grid.x; op.gy = grid.y; op.gz = grid.z;
    int i = 0;
    ((op.a.s[i++] = chain_slot(a)), ...);
    r->ops.push_back(op);
    return hipSuccess;
  }
  hipLaunchKernelGGL((chain_global<TAG, A...>), grid, dim3(ChainBody<KID>::kBlock), 0, st, a...);
  return hipGetLastError();
}

// replay: n ops at dev_ops (device memory), one workgroup
hipError_t chain_exec(hipStream_t st, const ChainOp* dev_ops, int n);

}  // namespace dgp
