// Host side of the weighted Gram kernel (gemm_gram.h): eligibility test and launch.
#include "gemm_gram.h"

#include <cstdlib>

namespace dgp {

bool gemm_gram_ok(const GemmArgs& a) {
  static int enabled = -1;
  static long min_k = 0;
  if (enabled < 0) {
    const char* e = getenv("DGP_GRAM");
    enabled = e ? atoi(e) : 1;
    const char* k = getenv("DGP_GRAM_MIN_K");
    min_k = k ? atol(k) : 65536;       // below that a workgroup's range is a few k-tiles: the split-K engine is as good
  }
  if (!enabled) return false;
  if (a.A != a.B || a.M != 256 || a.N != 256 || a.lda != 256 || a.ldb != 256 || a.ldc != 256) return false;
  if (a.tri != TRI_OUT_LOWER || a.triblk != 256 || a.alpha != 1.0) return false;
  if (!(a.beta == 1 || a.splits > 1)) return false;                         // the kernel ADDS its partial triangles
  if (a.K < min_k || a.K % 16 != 0) return false;
  if (a.batch < 1 || a.batch > 16 || (a.batch > 1 && (a.sC != 65536 || a.sA != 0 || a.sB != 0))) return false;
  if (a.ascale_mode == 2) { if (a.ascale == nullptr || a.as_ld != a.batch) return false; }
  else if (a.ascale_mode != 0 || a.batch != 1) return false;
  if (a.epi != 0 || a.rank != 0 || a.eadd != nullptr || a.C2 != nullptr || a.a_blocked || a.c_blocked || a.tri_row0 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15u) || (reinterpret_cast<uintptr_t>(a.C) & 7u)) return false;
  if (a.ascale && (reinterpret_cast<uintptr_t>(a.ascale) & 15u)) return false;
  return true;
}

hipError_t gemm_gram(hipStream_t st, const GemmArgs& a) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    else cus = 256;
    const char* e = getenv("DGP_GRAM_GRID");
    if (e && atoi(e) > 0) cus = atoi(e);
  }
  GramArgs g;
  g.C = a.A;
  g.s = a.ascale_mode == 2 ? a.ascale : nullptr;
  g.G = a.C;
  g.P = a.K;
  g.D = a.batch;
  const long kt = a.K / 16;
  const unsigned grid = (unsigned)(kt < cus ? kt : cus);
  if (g.s) hipLaunchKernelGGL(gemm_gram_kernel<true>, dim3(grid), dim3(512), 0, st, g);
  else hipLaunchKernelGGL(gemm_gram_kernel<false>, dim3(grid), dim3(512), 0, st, g);
  return hipGetLastError();
}

}  // namespace dgp
