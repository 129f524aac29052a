// Host side of the row-panel kernel of the backward pass through Kuf (gemm_gpanel.h): eligibility test and launch.
#include "gemm_gpanel.h"
#include "dgp_internal.h"

#include <cstdlib>

namespace dgp {

bool g_panel_ok(int Mp, int w1, long P) {
  static int enabled = -1;
  static long min_p = -1;
  if (enabled < 0) { const char* e = getenv("DGP_GPANEL"); enabled = e ? atoi(e) : 1; }
  if (min_p < 0) { const char* e = getenv("DGP_GPANEL_MIN_P"); min_p = e ? atol(e) : 4096; }
  return enabled && Mp == 256 && w1 >= 2 && w1 <= 9 && P >= min_p;
}

// R1 [P x w1] = g [Z | 1] and, with GX, GX [256 x w1] += g^T [X | 1] for g = (Cbar Linv) .* E, g never stored
hipError_t g_panel(hipStream_t st, const double* Cbar, const double* Linv, const double* E, long P, const double* Z1,
                   const double* X1, int w1, double* R1, double* GX, double* ws, int cu_count, int reserve_cus) {
  if (P <= 0) return hipSuccess;
  if (GX != nullptr && ws == nullptr) return hipErrorInvalidValue;
  int cus = cu_count;
  if (cus <= 0) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256; }
  if (reserve_cus > 0 && reserve_cus < cus / 4) cus -= reserve_cus;
  if (cus > 512) cus = 512;
  const long npanel = (P + 127) / 128;
  const long grid = npanel < cus ? npanel : cus;
  const long per = (npanel + grid - 1) / grid;
  const long used = (npanel + per - 1) / per;          // workgroups that get panels
  GPanelArgs g;
  g.A = Cbar; g.B = Linv; g.E = E; g.Z1 = Z1; g.X1 = X1; g.R1 = R1; g.slab = GX ? ws : nullptr; g.P = P; g.panels_per_wg = per;
#define GP_LAUNCH(W) hipLaunchKernelGGL(g_panel_kernel<W>, dim3((unsigned)used), dim3(512), 0, st, g)
  switch (w1) {
    case 2: GP_LAUNCH(2); break; case 3: GP_LAUNCH(3); break; case 4: GP_LAUNCH(4); break; case 5: GP_LAUNCH(5); break;
    case 6: GP_LAUNCH(6); break; case 7: GP_LAUNCH(7); break; case 8: GP_LAUNCH(8); break; case 9: GP_LAUNCH(9); break;
    default: return hipErrorInvalidValue;
  }
#undef GP_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (GX) return rg_reduce(st, ws, (int)used, 256 * w1, GX);
  return hipSuccess;
}

}  // namespace dgp
