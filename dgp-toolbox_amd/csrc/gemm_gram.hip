// Host side of the weighted Gram kernel (gemm_gram.h): eligibility test and launch.
#include "gemm_gram.h"

#include <cstdlib>

namespace dgp {

// G[d] += sum of the partial triangles of d, in a fixed order (XCD by XCD, inside an XCD by workgroup index).  One
// workgroup of 256 threads per (block, d): thread = lane * 4 + e.  Inside an XCD the workgroups whose ranges meet d's local
// k-tiles [d KTx, (d + 1) KTx) are consecutive; only the first of them can have started in d - 1 (its partial of d is then
// its SECOND segment).
__global__ __launch_bounds__(256) void gram_reduce_kernel(GramArgs g, int grid) {
  const int id = blockIdx.x, d = blockIdx.y, t = threadIdx.x;
  const long KT = g.P / 16;
  const double* __restrict__ w = g.ws + (long)id * 256 + t;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (unsigned x = 0; x < 8u && x < (unsigned)grid; ++x) {
    const unsigned n_x = (unsigned)grid / 8u + (x < (unsigned)grid % 8u ? 1u : 0u);
    const GramRange r0 = gram_range(KT, g.D, x, (unsigned)grid);       // (idx 0: the XCD's k-tile share)
    const long Fx = r0.ktn * g.D, lo = (long)d * r0.ktn, hi = lo + r0.ktn;
    if (r0.ktn <= 0) continue;
    // first workgroup whose range ends after lo, first whose range starts at or after hi
    long i0 = (lo * n_x) / Fx;
    while (i0 > 0 && (Fx * i0) / n_x > lo) --i0;
    while ((Fx * (i0 + 1)) / n_x <= lo) ++i0;
    long i1 = (hi * n_x) / Fx;
    if (i1 > (long)n_x) i1 = n_x;
    while (i1 < (long)n_x && (Fx * i1) / n_x < hi) ++i1;
    while (i1 > i0 + 1 && (Fx * (i1 - 1)) / n_x >= hi) --i1;
    for (long i = i0; i < i1; ++i) {
      const long f_lo = (Fx * i) / n_x, f_hi = (Fx * (i + 1)) / n_x;
      if (f_hi <= f_lo) continue;                                       // (an empty range stores nothing)
      const int seg = d - (int)(f_lo / r0.ktn);                         // 0: the range starts in d; 1: it started in d - 1
      const double v = w[(((long)x + 8 * i) * 2 + seg) * GR_SLOT_DOUBLES];
      switch ((i - i0) & 3) { case 0: s0 += v; break; case 1: s1 += v; break; case 2: s2 += v; break; default: s3 += v; break; }
    }
  }
  const double sum = (s0 + s1) + (s2 + s3);
  int r = 0;
  while ((r + 1) * (r + 2) / 2 <= id) ++r;
  const int c = id - r * (r + 1) / 2;
  const int lane = t >> 2, e = t & 3, li = lane & 15, lk = lane >> 4;
  g.G[(long)d * 65536 + (long)(16 * r + 4 * (li >> 2) + lk) * 256 + 16 * c + 4 * (li & 3) + e] += sum;
}

static int gram_grid(int cu_count) {
  static int grid_env = -1;
  if (grid_env < 0) { const char* e = getenv("DGP_GRAM_GRID"); grid_env = (e && atoi(e) > 0) ? atoi(e) : 0; }
  if (grid_env) return grid_env;
  GemmArgs a;
  a.cu_count = cu_count;
  return gemm_device_cus(a);
}

long gemm_gram_ws_bytes(int cu_count) { return 2L * gram_grid(cu_count) * GR_SLOT_BYTES; }

bool gemm_gram_ok(const GemmArgs& a) {
  static int enabled = -1;
  static long min_k = 0;
  if (enabled < 0) {
    const char* e = getenv("DGP_GRAM");
    enabled = e ? atoi(e) : 1;
    const char* k = getenv("DGP_GRAM_MIN_K");
    min_k = k ? atol(k) : 8192;        // (measured: 3x faster than the split-K engine at 8192 points already)
  }
  if (!enabled) return false;
  if (a.M != 256 || a.N != 256 || a.lda != 256 || a.ldb != 256 || a.ldc != 256) return false;
  if (a.tri != TRI_OUT_LOWER || a.triblk != 256 || a.alpha != 1.0) return false;
  if (!(a.beta == 1 || a.splits > 1)) return false;                         // the triangles are ADDED to C
  if (a.K < min_k || a.K % 16 != 0) return false;
  if (a.batch < 1 || a.batch > 8 || (a.batch > 1 && (a.sC != 65536 || a.sA != 0 || a.sB != 0))) return false;
  if (a.ascale_mode == 2) { if (a.ascale == nullptr || a.as_ld != a.batch) return false; }
  else if (a.ascale_mode != 0 || a.batch != 1) return false;
  if (a.A != a.B && (a.ascale_mode != 0 || a.batch != 1)) return false;                  // two sources: unweighted, one output
  if (a.epi != 0 || a.rank != 0 || a.eadd != nullptr || a.C2 != nullptr || a.a_blocked || a.c_blocked || a.tri_row0 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15u) || (reinterpret_cast<uintptr_t>(a.B) & 15u) || (reinterpret_cast<uintptr_t>(a.C) & 7u)) return false;
  if (a.ascale && (reinterpret_cast<uintptr_t>(a.ascale) & 15u)) return false;
  if (a.gram_ws == nullptr || a.gram_ws_bytes < gemm_gram_ws_bytes(a.cu_count) || (reinterpret_cast<uintptr_t>(a.gram_ws) & 31u)) return false;
  if (gemm_persistent_grid(a, gram_grid(a.cu_count)) < 8 * a.batch) return false;        // (every XCD's share is cut into >= D ranges)
  if (a.K / 16 < 2L * gram_grid(a.cu_count)) return false;                                  // (every workgroup gets k-tiles)
  return true;
}

hipError_t gemm_gram(hipStream_t st, const GemmArgs& a) {
  GramArgs g;
  g.A = a.A == a.B ? nullptr : a.A;
  g.C = a.B;
  g.s = a.ascale_mode == 2 ? a.ascale : nullptr;
  g.G = a.C;
  g.ws = a.gram_ws;
  g.P = a.K;
  g.D = a.batch;
  const unsigned grid = (unsigned)gemm_persistent_grid(a, gram_grid(a.cu_count));          // (K >= min_k: every workgroup has k-tiles; the scratch is sized for the full grid)
  if (g.A) hipLaunchKernelGGL((gemm_gram_kernel<false, 2>), dim3(grid), dim3(512), 0, st, g);
  else if (g.s) hipLaunchKernelGGL((gemm_gram_kernel<true, 1>), dim3(grid), dim3(512), 0, st, g);
  else hipLaunchKernelGGL((gemm_gram_kernel<false, 1>), dim3(grid), dim3(512), 0, st, g);
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(136, g.D), dim3(256), 0, st, g, (int)grid);
  return hipGetLastError();
}

}  // namespace dgp
