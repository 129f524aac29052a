// One-workgroup products for the small-matrix chains of small models (Mp <= 64: the Bayesian-optimisation surrogates,
// SO_BO.py:248-258): C = alpha op(A) op(B) (+ C), M, N, K <= 64, batched.  On the 128 x 64 engine such a product is four
// k-tiles of one workgroup, each a round trip to L2 behind a barrier: 12-15 us, and a training iteration of config 1 has
// ~25 of them on its critical path.  Here both operands are staged once (two 33 KB LDS images, k-major), then every
// thread accumulates a 4 x 4 block of C with VALU FMAs: one load latency + 64 k-steps.
// Triangular hints are ignored, exactly as the engine does at this size (its skips work on whole 128 x 64 tiles and
// 64-column k-ranges: op(B)'s zeros are stored, the full C is written).
#include <cstdlib>

#include "gemm_f64.h"

namespace dgp {

constexpr int SG = 64;          // largest M, N, K
constexpr int SG_LD = SG + 2;   // row pitch of the LDS images (doubles): 16-byte aligned rows, odd multiple of 16 bytes

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double As[SG][SG_LD];   // As[k][m] = op(A)[m][k]
  __shared__ __attribute__((aligned(16))) double Bs[SG][SG_LD];   // Bs[k][n] = op(B)[k][n]
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  const double* __restrict__ A = g.A + b * g.sA;
  const double* __restrict__ B = g.B + b * g.sB;
  double* __restrict__ C = g.C + b * g.sC;
  const int M = (int)g.M, N = (int)g.N, K = (int)g.K;
  // stage: the fast index of each global read runs along the operand's contiguous dimension.  All 32 loads of a thread are
  // requested before the first LDS write (clamped addresses + a select instead of a branch around the load: with the
  // branch the 16 rounds were 16 dependent trips to L2, 12 us per product)
  const int c = tid & 63, r0 = tid >> 6;
  const int ra_n = TA ? K : M, ca_n = TA ? M : K;      // rows / columns of A as stored
  const int rb_n = TB ? N : K, cb_n = TB ? K : N;      // rows / columns of B as stored
  double va[16], vb[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int r = r0 + 4 * it;
    va[it] = A[(long)(r < ra_n ? r : ra_n - 1) * g.lda + (c < ca_n ? c : ca_n - 1)];
    vb[it] = B[(long)(r < rb_n ? r : rb_n - 1) * g.ldb + (c < cb_n ? c : cb_n - 1)];
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int r = r0 + 4 * it;
    const double xa = (r < ra_n && c < ca_n) ? va[it] : 0.0, xb = (r < rb_n && c < cb_n) ? vb[it] : 0.0;
    if (TA) As[r][c] = xa; else As[c][r] = xa;         // As[k][m]
    if (TB) Bs[c][r] = xb; else Bs[r][c] = xb;         // Bs[k][n]
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  if (4 * ty < M && 4 * tx < N) {
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
      const d2_t a0 = *reinterpret_cast<const d2_t*>(&As[k][4 * ty]), a1 = *reinterpret_cast<const d2_t*>(&As[k][4 * ty + 2]);
      const d2_t b0 = *reinterpret_cast<const d2_t*>(&Bs[k][4 * tx]), b1 = *reinterpret_cast<const d2_t*>(&Bs[k][4 * tx + 2]);
      const double av[4] = {a0[0], a0[1], a1[0], a1[1]}, bv[4] = {b0[0], b0[1], b1[0], b1[1]};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 4 * ty + i;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = 4 * tx + j;
        if (n >= N) continue;
        double* c = C + (long)m * g.ldc + n;
        const double v = g.alpha * acc[i][j];
        *c = g.beta ? *c + v : v;
      }
    }
  }
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
  const dim3 grid((unsigned)a.batch), block(256);
  switch (op) {
    case GEMM_NN: hipLaunchKernelGGL((gemm_small_kernel<false, false>), grid, block, 0, st, a); break;
    case GEMM_NT: hipLaunchKernelGGL((gemm_small_kernel<false, true>), grid, block, 0, st, a); break;
    case GEMM_TN: hipLaunchKernelGGL((gemm_small_kernel<true, false>), grid, block, 0, st, a); break;
  }
  return hipGetLastError();
}

}  // namespace dgp
