// One-workgroup products for the small-matrix chains of small models (Mp <= 64: the Bayesian-optimisation surrogates,
// SO_BO.py:248-258): C = alpha op(A) op(B) (+ C), M, N, K <= 64, batched.  On the 128 x 64 engine such a product is four
// k-tiles of one workgroup, each a round trip to L2 behind a barrier: 12-15 us, and a training iteration of config 1 has
// ~25 of them on its critical path.  Here both operands are staged once (two 33 KB LDS images, k-major), then every
// thread accumulates a 4 x 4 block of C with VALU FMAs: one load latency + 64 k-steps.
// Triangular hints are ignored, exactly as the engine does at this size (its skips work on whole 128 x 64 tiles and
// 64-column k-ranges: op(B)'s zeros are stored, the full C is written).
#include <cstdlib>

#include "gemm_f64.h"
#include "chain_bodies.h"

namespace dgp {

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs g) {
  gemm_small_body<TA, TB, CM_KERNEL>(VIdx{blockIdx.x, 0u, 0u}, VIdx{threadIdx.x, 0u, 0u}, g);
}

bool gemm_small_ok(GemmOp op, const GemmArgs& a) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("DGP_SMALL_GEMM"); enabled = e ? atoi(e) : 1; }
  if (!enabled) return false;
  if (a.M < 1 || a.N < 1 || a.K < 1 || a.M > SG || a.N > SG || a.K > SG) return false;
  if (a.splits > 1 || a.batch < 1 || a.batch > 65535) return false;
  if (a.A == nullptr || a.B == nullptr || a.C == nullptr) return false;
  if (a.ascale_mode != 0 || a.ascale != nullptr || a.rank != 0 || a.eadd != nullptr || a.emul != nullptr || a.C2 != nullptr || a.c2_only) return false;
  if (a.epi != 0 || a.a_blocked || a.c_blocked) return false;
  (void)op;
  return true;
}

hipError_t gemm_small(hipStream_t st, GemmOp op, const GemmArgs& a) {
  if (ChainRec* r = chain_recorder()) {          // a step of a recorded chain (chain.h)
    ChainOp o;
    memset(&o, 0, sizeof o);
    o.kid = op == GEMM_NN ? CK_GEMM_NN : (op == GEMM_NT ? CK_GEMM_NT : CK_GEMM_TN);
    o.gx = (unsigned)a.batch; o.gy = o.gz = 1;
    memcpy(o.a.raw, &a, sizeof(GemmArgs));
    r->ops.push_back(o);
    return hipSuccess;
  }
  const dim3 grid((unsigned)a.batch), block(256);
  switch (op) {
    case GEMM_NN: hipLaunchKernelGGL((gemm_small_kernel<false, false>), grid, block, 0, st, a); break;
    case GEMM_NT: hipLaunchKernelGGL((gemm_small_kernel<false, true>), grid, block, 0, st, a); break;
    case GEMM_TN: hipLaunchKernelGGL((gemm_small_kernel<true, false>), grid, block, 0, st, a); break;
  }
  return hipGetLastError();
}

}  // namespace dgp
