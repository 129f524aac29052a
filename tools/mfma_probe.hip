// Probes the lane maps of v_mfma_f64_4x4x4_4b_f64: for every (lane of A, lane of B) one-hot pair, which output lanes are hit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned long long* hit) {   // hit[la*64+lb] = mask of output lanes with nonzero result
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      unsigned long long m = __ballot(d != 0.0);
      if (l == 0) hit[la * 64 + lb] = m;
    }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 4096 * 8);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<unsigned long long> h(4096);
  hipMemcpy(h.data(), d, 4096 * 8, hipMemcpyDeviceToHost);
  // for each output lane lo, list the (la, lb) pairs contributing
  for (int lo = 0; lo < 64; ++lo) {
    printf("out lane %2d <-", lo);
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >> lo & 1ull) printf(" (a%d,b%d)", la, lb);
    printf("\n");
  }
  return 0;
}
