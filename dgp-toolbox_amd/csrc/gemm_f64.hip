// Host-side dispatch of the fp64 MFMA GEMM engine (see gemm_f64.h).
#include "gemm_f64.h"

namespace dgp {

template <bool TA, bool TB, int BM, int BN, int BK, int WR, int WC, int VA, int VB>
static hipError_t launch(hipStream_t st, const GemmArgs& a) {
  const long tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  if (tiles <= 0 || a.batch <= 0) return hipSuccess;
  dim3 grid((unsigned)tiles, (unsigned)(a.batch * a.splits), 1);
  hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, BM, BN, BK, WR, WC, VA, VB>), grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

static inline bool vec2_ok(const double* p, long ld, long extent, long bstride, int batch) {
  return ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) && (ld % 2 == 0) && (extent % 2 == 0) &&
         (batch <= 1 || bstride % 2 == 0);
}

template <bool TA, bool TB>
static hipError_t dispatch(hipStream_t st, const GemmArgs& a) {
  // contiguous-direction extents: A is k-contiguous (N) or m-contiguous (T); B is n- (N) or k-contiguous (T)
  const bool va = vec2_ok(a.A, a.lda, TA ? a.M : a.K, a.sA, a.batch);
  const bool vb = vec2_ok(a.B, a.ldb, TB ? a.K : a.N, a.sB, a.batch);
  const bool skinny = a.N <= 16;
  if (!skinny) {
    if (va && vb) return launch<TA, TB, 128, 64, 16, 2, 2, 2, 2>(st, a);
    return launch<TA, TB, 128, 64, 16, 2, 2, 1, 1>(st, a);
  }
  if (va) return launch<TA, TB, 128, 16, 16, 4, 1, 2, 1>(st, a);
  return launch<TA, TB, 128, 16, 16, 4, 1, 1, 1>(st, a);
}

hipError_t gemm_f64(hipStream_t st, GemmOp op, const GemmArgs& args) {
  GemmArgs a = args;
  if (a.splits < 1) a.splits = 1;
  if (a.batch < 1) a.batch = 1;
  if (a.splits > 1) {
    // ksplit must be a multiple of BK (=16) so that split boundaries coincide with k-tiles
    long per = (a.K + a.splits - 1) / a.splits;
    per = ((per + 15) / 16) * 16;
    a.ksplit = per;
    a.splits = (int)((a.K + per - 1) / per);
    if (a.splits < 1) a.splits = 1;
  }
  switch (op) {
    case GEMM_NN: return dispatch<false, false>(st, a);
    case GEMM_NT: return dispatch<false, true>(st, a);
    case GEMM_TN: return dispatch<true, false>(st, a);
  }
  return hipErrorInvalidValue;
}

}  // namespace dgp
