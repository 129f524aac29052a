// Host side of the row-panel kernel of the backward dC product (gemm_dcpanel.h): eligibility test and launch.
#include "gemm_dcpanel.h"

#include <cstdlib>

namespace dgp {


// (the shape test the forward pass uses to decide that T is written row-major for this kernel: dgp_ctx.h t_rowmajor)
bool dc_panel_shape_ok(long M, long Mp, long D) {
  // Measured (profiles/r4_dcpanel.txt): at 10^6 rows the kernel ties with gemm_tallu.h alone (10.0 ms at D = 8) and loses 0.2 ms
  // inside the iteration; at a rank's share of 8 GPUs (125 000 rows) it wins 1.25 against 1.41 ms - gemm_tallu.h's units (a
  // 256-row tile x half x all D blocks) are too coarse for 488 row tiles on 256 CUs.  So: a window of row counts.
  static int enabled = -1;
  static long min_rows = -1, max_rows = -1;
  if (enabled < 0) { const char* e = getenv("DGP_DCPANEL"); enabled = e ? atoi(e) : 1; }
  if (min_rows < 0) { const char* e = getenv("DGP_DCPANEL_MIN_ROWS"); min_rows = e ? atol(e) : 4096; }
  if (max_rows < 0) { const char* e = getenv("DGP_DCPANEL_MAX_ROWS"); max_rows = e ? atol(e) : 300000; }
  return enabled && Mp == 256 && D >= 1 && D <= 8 && M >= min_rows && M <= max_rows;
}

bool dc_panel_ok(const GemmArgs& a) {
  if (a.tri != TRI_B_UPPER || a.triblk != 256 || a.N != 256 || a.K % 256 != 0) return false;
  const long D = a.K / 256;
  if (!dc_panel_shape_ok(a.M, 256, D)) return false;
  if (a.ascale_mode != 1 || a.ascale == nullptr || a.a_wrap != 0 || a.a_kblk != 256 || a.a_blocked || a.lda != a.K) return false;
  if (a.as_ld < D) return false;
  if (a.epi != 0 || a.C == nullptr || a.c_blocked || a.C2 != nullptr || a.emul != nullptr) return false;
  if (a.beta != 0 || a.batch != 1 || a.splits > 1) return false;
  if (a.rank < 0 || a.rank > 8) return false;
  if (a.rank > 0 && (a.rowf == nullptr || a.colf == nullptr)) return false;
  if (a.eadd != nullptr && a.eadd_nsc != D) return false;
  if (a.ldb != 256 || a.ldc != 256) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15u) || (reinterpret_cast<uintptr_t>(a.B) & 15u) || (reinterpret_cast<uintptr_t>(a.C) & 15u) ||
      (reinterpret_cast<uintptr_t>(a.eadd) & 15u))
    return false;
  return true;
}

hipError_t dc_panel(hipStream_t st, const GemmArgs& a) {
  const int cus = gemm_persistent_grid(a, gemm_device_cus(a));
  DcPanelArgs g;
  g.T = a.A; g.ldt = a.lda;
  g.B = a.B;
  g.C = a.C;
  g.s = a.ascale; g.as_ld = a.as_ld;
  g.eadd = a.eadd;
  g.rowf = a.rank ? a.rowf : nullptr; g.colf = a.rank ? a.colf : nullptr;
  g.rank = a.rank;
  g.alpha = a.alpha;
  g.P = a.M; g.D = (int)(a.K / 256);
  const long npanel = (a.M + 127) / 128;
  const unsigned grid = (unsigned)(npanel < cus ? npanel : cus);
  if (a.rank == 8) hipLaunchKernelGGL(dc_panel_kernel<8>, dim3(grid), dim3(512), 0, st, g);
  else if (a.rank > 0) hipLaunchKernelGGL(dc_panel_kernel<-1>, dim3(grid), dim3(512), 0, st, g);
  else hipLaunchKernelGGL(dc_panel_kernel<0>, dim3(grid), dim3(512), 0, st, g);
  return hipGetLastError();
}

}  // namespace dgp
