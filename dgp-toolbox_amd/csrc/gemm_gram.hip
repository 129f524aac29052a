// Host side of the weighted Gram kernel (gemm_gram.h): eligibility test and launch.
#include "gemm_gram.h"

#include <cstdlib>

namespace dgp {

// G[d] += sum of the partial triangles of d, in a fixed order (XCD by XCD, inside an XCD by workgroup index).  One
// workgroup of 256 threads per (block, d): thread = lane * 4 + e.  Inside an XCD the workgroups whose ranges meet d's local
// k-tiles [d KTx, (d + 1) KTx) are consecutive; only the first of them can have started in d - 1 (its partial of d is then
// its SECOND segment).  Which workgroups those are is worked out on the host (GramReduceMap: a few hundred integer
// divisions per launch; done per thread in the kernel they made it 0.15 ms instead of 0.02).
struct GramReduceMap {
  short i0[8][8], i1[8][8];      // [d][xcd]: workgroup indices idx (b = xcd + 8 idx) with a partial of d: [i0, i1)
  unsigned char seg0[8][8];      // segment (0 / 1) of d in workgroup i0 (every later one: 0)
};

static GramReduceMap gram_reduce_map(long P, int D, int grid) {
  GramReduceMap m{};
  const long KT = P / 16;
  for (int d = 0; d < D; ++d)
    for (unsigned x = 0; x < 8u; ++x) {
      m.i0[d][x] = m.i1[d][x] = 0; m.seg0[d][x] = 0;
      if (x >= (unsigned)grid) continue;
      const long n_x = grid / 8 + (x < (unsigned)grid % 8u ? 1 : 0);
      const GramRange r0 = gram_range(KT, D, x, (unsigned)grid);       // (idx 0: the XCD's k-tile share)
      if (r0.ktn <= 0) continue;
      const long Fx = r0.ktn * D, lo = (long)d * r0.ktn, hi = lo + r0.ktn;
      auto cut = [&](long i) { return (Fx * i) / n_x; };
      long a = 0, b = n_x;
      while (a < n_x && cut(a + 1) <= lo) ++a;            // first workgroup whose range ends after lo
      while (b > a && cut(b - 1) >= hi) --b;             // one past the last whose range starts before hi
      while (a < b && cut(a + 1) <= cut(a)) ++a;          // (empty ranges store nothing)
      m.i0[d][x] = (short)a; m.i1[d][x] = (short)b;
      m.seg0[d][x] = (unsigned char)(a < b ? d - (int)(cut(a) / r0.ktn) : 0);
    }
  return m;
}

__global__ __launch_bounds__(256) void gram_reduce_kernel(GramArgs g, GramReduceMap m) {
  const int id = blockIdx.x, d = blockIdx.y, t = threadIdx.x;
  const double* __restrict__ w = g.ws + (long)id * 256 + t;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int x = 0; x < 8; ++x) {
    const int i0 = m.i0[d][x], i1 = m.i1[d][x];
    if (i1 <= i0) continue;
    s0 += w[(((long)x + 8 * i0) * 2 + m.seg0[d][x]) * GR_SLOT_DOUBLES];
    int i = i0 + 1;
    for (; i + 4 <= i1; i += 4) {                               // (four independent loads in flight; fixed association)
      const double v0 = w[((long)x + 8 * (i + 0)) * 2 * GR_SLOT_DOUBLES], v1 = w[((long)x + 8 * (i + 1)) * 2 * GR_SLOT_DOUBLES];
      const double v2 = w[((long)x + 8 * (i + 2)) * 2 * GR_SLOT_DOUBLES], v3 = w[((long)x + 8 * (i + 3)) * 2 * GR_SLOT_DOUBLES];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; i < i1; ++i) s0 += w[((long)x + 8 * i) * 2 * GR_SLOT_DOUBLES];
  }
  const double sum = (s0 + s1) + (s2 + s3);
  if (id == 136) { g.du[(long)t * g.D + d] += sum; return; }      // (the du block: thread = column of Ct)
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
  if ((a.gram_mb != nullptr) != (a.gram_du != nullptr)) return false;
  if (a.gram_mb && (a.ascale_mode != 2 || (reinterpret_cast<uintptr_t>(a.gram_mb) & 15u))) return false;   // du rides on the weighted form
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
  g.mb = a.gram_mb;
  g.du = a.gram_du;
  const unsigned grid = (unsigned)gemm_persistent_grid(a, gram_grid(a.cu_count));          // (K >= min_k: every workgroup has k-tiles; the scratch is sized for the full grid)
  if (g.A) hipLaunchKernelGGL((gemm_gram_kernel<false, 2>), dim3(grid), dim3(512), 0, st, g);
  else if (g.s && g.mb) hipLaunchKernelGGL((gemm_gram_kernel<true, 1, true>), dim3(grid), dim3(512), 0, st, g);
  else if (g.s) hipLaunchKernelGGL((gemm_gram_kernel<true, 1>), dim3(grid), dim3(512), 0, st, g);
  else hipLaunchKernelGGL((gemm_gram_kernel<false, 1>), dim3(grid), dim3(512), 0, st, g);
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(g.mb ? 137 : 136, g.D), dim3(256), 0, st, g, gram_reduce_map(g.P, g.D, (int)grid));
  return hipGetLastError();
}

}  // namespace dgp
