// LDS read-rate microbenchmark for gfx950: bytes per clock per CU of ds_read_b64 / ds_read_b128 under the access
// patterns the GEMM engine uses (tools/, tuning only).   hipcc --offload-arch=gfx950 -O3 lds_bench.hip -o lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2_t __attribute__((ext_vector_type(2)));

// MODE 0: b64, lane -> consecutive doubles            (conflict-free, 512 B per instruction)
// MODE 1: b128, lane -> consecutive double2           (1024 B per instruction)
// MODE 2: b64, A-image pattern: row = lane&15 at pitch 18 doubles, k = lane>>4
// MODE 3: b128, lanes l, l+4, l+8, l+12 of a row of 16 share an address (broadcast groups), k = lane>>4 at pitch 80
// MODE 4: b64 pairs via ds_read2_b64-friendly pattern: B-image rotated columns (current engine)
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  __shared__ __attribute__((aligned(16))) double s[7168];
  for (int i = threadIdx.x; i < 7168; i += 256) s[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  int base;
  if (MODE == 0) base = lane;
  if (MODE == 1) base = lane * 2;
  if (MODE == 2) base = li * 18 + lk;
  if (MODE == 3) base = lk * 80 + (li & 3) * 4;
  if (MODE == 4) base = lk * 80 + li;
  if (MODE == 5) base = li * 18 + lk * 2;            // b128, A image k-contiguous at pitch 18, lane lk reads k = 2 lk, 2 lk + 1
  if (MODE == 6) base = (li & 3) * 18 + lk * 2;      // b128, transposed-B image: 4 distinct rows, 4-lane broadcast
  if (MODE == 7) base = li * 20 + lk * 2;            // as 5 at pitch 20
  if (MODE == 8) base = lk * 2 * 80 + (li & 3) * 4;  // b128 B image as the engine reads it: lane group lk reads row 2 lk
  if (MODE == 9) base = lk * 2 * 84 + (li & 3) * 4;  // same at pitch 84
  base += wave * 16;
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");   // the reads must be re-issued every iteration
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int off = base + u * 320 + (it & 1) * 2560;
      if (MODE == 1 || MODE == 3 || MODE >= 5) {
        d2_t v = *reinterpret_cast<const d2_t*>(s + off);
        d2_t w = *reinterpret_cast<const d2_t*>(s + off + (MODE == 1 ? 128 : ((MODE == 3 || MODE >= 8) ? 2 : 8)));
        acc0 += v[0]; acc1 += v[1]; acc2 += w[0]; acc3 += w[1];
      } else if (MODE == 4) {
        acc0 += s[off]; acc1 += s[base - li + ((li + 4) & 15) + u * 320 + (it & 1) * 2560];
        acc2 += s[base - li + ((li + 8) & 15) + u * 320 + (it & 1) * 2560];
        acc3 += s[base - li + ((li + 12) & 15) + u * 320 + (it & 1) * 2560];
      } else {
        acc0 += s[off]; acc1 += s[off + 64 * (MODE == 0) + 4 * (MODE == 2)];
        acc2 += s[off + 128 * (MODE == 0) + 8 * (MODE == 2)]; acc3 += s[off + 192 * (MODE == 0) + 12 * (MODE == 2)];
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}

template <int MODE>
void run(const char* name, double bytes_per_lane_iter) {
  double* out; hipMalloc(&out, 512 * 256 * 8);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // 2 workgroups per CU on 256 CUs; bytes per CU = 2 WG * 256 lanes * bytes
  const double bytes_cu = 2.0 * 256 * bytes_per_lane_iter * iters;
  const double clk = ms * 1e-3 * 2.4e9;
  printf("%-70s %8.3f ms  %6.1f B/clk/CU (at 2.4 GHz)\n", name, ms, bytes_cu / clk);
  hipFree(out);
}

int main() {
  run<0>("b64  consecutive doubles (4 reads per unrolled step)", 8 * 4 * 8.0);
  run<1>("b128 consecutive double2 (2 reads per step)", 8 * 2 * 16.0);
  run<2>("b64  A image: 16 rows at pitch 18 x 4 k (4 reads per step)", 8 * 4 * 8.0);
  run<3>("b128 B image, 4-lane broadcast groups (2 reads per step)", 8 * 2 * 16.0);
  run<4>("b64  B image, rotated columns (4 reads per step)", 8 * 4 * 8.0);
  run<5>("b128 A image pitch 18 (lane lk reads k pair) (2 reads per step)", 8 * 2 * 16.0);
  run<6>("b128 transposed-B image, 4 rows, broadcast (2 reads per step)", 8 * 2 * 16.0);
  run<7>("b128 A image pitch 20", 8 * 2 * 16.0);
  run<8>("b128 B image, rows 2 lk at pitch 80 (engine pattern)", 8 * 2 * 16.0);
  run<9>("b128 B image, rows 2 lk at pitch 84", 8 * 2 * 16.0);
  return 0;
}
