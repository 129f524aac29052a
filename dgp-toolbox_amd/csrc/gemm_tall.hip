// Host side of the tall-tile kernel (gemm_tall.h): eligibility test and launch.
#include "gemm_tall.h"

#include <cstdlib>

namespace dgp {

bool gemm_tall_ok(const GemmArgs& a) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("DGP_TALL"); enabled = e ? atoi(e) : 1; }
  if (!enabled || !gemm_wide_ok(a)) return false;          // (a subset of what the wide-tile kernel takes: same row-norm planes)
  if (a.tri != TRI_B_LOWER || a.triblk != 256 || a.K != 256 || a.N % 256 != 0 || a.M % 128 != 0) return false;
  if (a.ldb != a.N || a.lda < 256 || a.alpha != 1.0 || a.beta != 0 || a.batch != 1 || a.splits > 1) return false;
  if (a.rank != 0 || a.eadd != nullptr || a.C2 != nullptr || a.emul != nullptr || a.ascale_mode != 0 || a.a_blocked) return false;
  if (a.epi == 2) { if (a.C == nullptr || a.ldc != a.N || a.ldc * 8 * 32 >= (1L << 31)) return false; }      // (blocked or row-major T)
  else if (a.epi != 1) return false;
  if (a.rowsq == nullptr) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15u) || (reinterpret_cast<uintptr_t>(a.B) & 15u) || (a.C && (reinterpret_cast<uintptr_t>(a.C) & 15u))) return false;
  if ((a.lda * 8) % 16 != 0 || a.lda * 8 * 8 * 4 >= (1L << 31) || a.ldb * 8 * 2 >= (1L << 31)) return false;
  return true;
}

hipError_t gemm_tall(hipStream_t st, const GemmArgs& a) {
  static int grid_env = -1;
  if (grid_env < 0) { const char* e = getenv("DGP_TALL_GRID"); grid_env = (e && atoi(e) > 0) ? atoi(e) : 0; }
  const int cus = grid_env ? grid_env : gemm_device_cus(a);
  TallArgs g;
  g.A = a.A; g.lda = a.lda;
  g.B = a.B; g.ldb = a.ldb;
  g.C = a.epi == 2 ? a.C : nullptr; g.ldc = a.ldc;
  g.rowsq = a.rowsq; g.rowsq_ld = a.rowsq_ld;
  g.M = a.M; g.D = (int)(a.N / 256);
  g.rowmajor = (a.epi == 2 && !a.c_blocked) ? 1 : 0;
  const long ntile = ((a.M + 255) / 256) * g.D;
  const long gmax = gemm_persistent_grid(a, cus);
  const unsigned grid = (unsigned)(ntile < gmax ? ntile : gmax);
  static int nt = -1;
  if (nt < 0) { const char* e = getenv("DGP_TALL_NT"); nt = e ? atoi(e) : 1; }   // (same-box A/B: 50.51 -> 50.36 ms of point-contraction time per iteration)
  if (g.C && nt) hipLaunchKernelGGL((gemm_tall_kernel<true, true>), dim3(grid), dim3(512), 0, st, g);
  else if (g.C) hipLaunchKernelGGL((gemm_tall_kernel<true, false>), dim3(grid), dim3(512), 0, st, g);
  else hipLaunchKernelGGL((gemm_tall_kernel<false, false>), dim3(grid), dim3(512), 0, st, g);
  return hipGetLastError();
}

}  // namespace dgp
