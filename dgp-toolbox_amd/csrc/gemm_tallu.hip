// Host side of the tall-tile kernel of the backward dC product (gemm_tallu.h): eligibility test and launch.
#include "gemm_tallu.h"

#include <cstdlib>

namespace dgp {

bool gemm_tallu_ok(const GemmArgs& a) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("DGP_TALLU"); enabled = e ? atoi(e) : 1; }
  if (!enabled || !gemm_wide_ok(a)) return false;          // (a subset of what the wide-tile kernel takes)
  if (a.tri != TRI_B_UPPER || a.triblk != 256 || a.N != 256 || a.K % 256 != 0 || a.M % 128 != 0) return false;
  const long D = a.K / 256;
  static long min_d = -1;
  if (min_d < 0) { const char* e = getenv("DGP_TALLU_MIN_D"); min_d = e ? atol(e) : 1; }
  if (D < min_d || D > 8) return false;
  if (a.ascale_mode != 1 || a.ascale == nullptr || a.a_wrap != 0 || a.a_kblk != 256 || !a.a_blocked || a.lda != a.K) return false;
  if (a.as_ld < D) return false;
  if (a.epi != 0 || a.C == nullptr || a.c_blocked || a.C2 != nullptr || a.emul != nullptr) return false;
  if (a.beta != 0 || a.batch != 1 || a.splits > 1) return false;
  if (a.rank < 0 || a.rank > 8) return false;
  if (a.rank > 0 && (a.rowf == nullptr || a.colf == nullptr)) return false;
  if (a.eadd != nullptr && a.eadd_nsc != D) return false;
  if (a.ldb < 256 || a.ldb % 2 != 0 || a.ldc < 256 || a.ldc % 2 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15u) || (reinterpret_cast<uintptr_t>(a.B) & 15u) || (reinterpret_cast<uintptr_t>(a.C) & 15u) ||
      (reinterpret_cast<uintptr_t>(a.eadd) & 15u))
    return false;
  if (a.ldb * 8 * 2 >= (1L << 31)) return false;
  return true;
}

hipError_t gemm_tallu(hipStream_t st, const GemmArgs& a) {
  static int grid_env = -1;
  if (grid_env < 0) { const char* e = getenv("DGP_TALL_GRID"); grid_env = (e && atoi(e) > 0) ? atoi(e) : 0; }
  const int cus = grid_env ? grid_env : gemm_device_cus(a);
  TallUArgs g;
  g.A = a.A; g.lda = a.lda;
  g.B = a.B; g.ldb = a.ldb;
  g.C = a.C; g.ldc = a.ldc;
  g.s = a.ascale; g.as_ld = a.as_ld;
  g.eadd = a.eadd;
  g.rowf = a.rank ? a.rowf : nullptr; g.colf = a.rank ? a.colf : nullptr;
  g.rank = a.rank;
  g.alpha = a.alpha;
  g.M = a.M; g.D = (int)(a.K / 256);
  const long nb = (a.M + 255) / 256;
  const long gmax = gemm_persistent_grid(a, cus);
  const unsigned grid = (unsigned)(nb < gmax ? nb : gmax);
  if (a.rank == 8) hipLaunchKernelGGL(gemm_tallu_kernel<8>, dim3(grid), dim3(512), 0, st, g);
  else if (a.rank > 0) hipLaunchKernelGGL(gemm_tallu_kernel<-1>, dim3(grid), dim3(512), 0, st, g);
  else hipLaunchKernelGGL(gemm_tallu_kernel<0>, dim3(grid), dim3(512), 0, st, g);
  return hipGetLastError();
}

}  // namespace dgp
