// Prototype: the fp64 MFMA main loop with LDS-DMA staging (global_load_lds_dwordx4) instead of register staging.
// NN product C[M x N] = A[M x K] * B[K x N], interior shapes only (M % 128 == 0, N % 64 == 0, K % 16 == 0), no
// epilogue features.  Purpose: measure what removing the VGPR round trip and the ds_write pass is worth before the
// engine (dgp-toolbox_amd/csrc/gemm_f64.h) is rebuilt around it.  tools/, tuning only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_glds_proto.hip -o gemm_glds_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 64, BK = 16;
constexpr int A_BYTES = BM * BK * 8;   // 16 KB: [128 rows][8 chunks of 16 B], chunk XOR-swizzled by (row >> 1) & 7
constexpr int B_BYTES = BK * BN * 8;   //  8 KB: [16 k rows][32 chunks of 16 B], chunk bit 3 flipped by (krow >> 1) & 1
constexpr int BUF = A_BYTES + B_BYTES;

#define LDS3(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB1(p) ((const __attribute__((address_space(1))) void*)(p))

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_glds(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, long M, long N, long K,
               long lda, long ldb, long ldc) {
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  const unsigned tiles_n = (unsigned)(N / BN), ntile = (unsigned)(M / BM) * tiles_n;
  for (unsigned lin = blockIdx.x; lin < ntile; lin += gridDim.x) {
    unsigned bid = lin;
    {
      const unsigned q = ntile / 8u, r = ntile % 8u, xcd = bid % 8u, idx = bid / 8u;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long m0 = (long)(bid / tiles_n) * BM, n0 = (long)(bid % tiles_n) * BN;
    const long ktiles = K / BK;
    // ---- per-lane global sources of this wave's LDS-DMA pieces (the LDS side is lane-linear: base + lane*16) ----
    // A: piece p of wave w covers rows (w*4 + p)*8 .. +7; lane -> row = lane/8, physical chunk = lane%8
    const char* ga[4];
    const char* gb[2];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = (wave * 4 + p) * 8 + (lane >> 3), pc = lane & 7;
      const int c = pc ^ ((row >> 1) & 7);                                   // logical chunk (2 doubles of k)
      ga[p] = reinterpret_cast<const char*>(A + (m0 + row) * lda + 2 * c);
    }
    // B: piece p of wave w covers k rows (w*2 + p)*2 .. +1; lane -> krow = lane/32, physical chunk = lane%32
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int krow = (wave * 2 + p) * 2 + (lane >> 5), pc = lane & 31;
      const int c = pc ^ (((krow >> 1) & 1) << 3);
      gb[p] = reinterpret_cast<const char*>(B + (long)krow * ldb + n0 + 2 * c);
    }
    const long a_step = (long)BK * 8, b_step = (long)BK * ldb * 8;
    auto issue = [&](int buf, long kt) {   // tile kt -> LDS buffer buf
      char* base = smem + buf * BUF;
#pragma unroll
      for (int p = 0; p < 4; ++p)
        __builtin_amdgcn_global_load_lds(GLB1(ga[p] + kt * a_step), LDS3(base + (wave * 4 + p) * 1024), 16, 0, 0);
#pragma unroll
      for (int p = 0; p < 2; ++p)
        __builtin_amdgcn_global_load_lds(GLB1(gb[p] + kt * b_step), LDS3(base + A_BYTES + (wave * 2 + p) * 1024), 16, 0, 0);
    };
    double acc[4][2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;
    double fa[2][2][4], fb[2][2][4];
    auto fragA = [&](int buf, int s8, double (&xa)[2][4]) {
      const char* as = smem + buf * BUF;
      const int c = s8 * 4 + lk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (i * 2 + wr) * 16 + li;
        const d2_t v = *reinterpret_cast<const d2_t*>(as + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        xa[0][i] = v[0]; xa[1][i] = v[1];
      }
    };
    auto fragB = [&](int buf, int s8, int q, double (&xb)[2][4]) {
      const char* bs = smem + buf * BUF + A_BYTES;
      const int krow = s8 * 8 + lk * 2 + q;
      const int sw = ((krow >> 1) & 1) << 3;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c0 = (j * 2 + wc) * 8 + 2 * (li & 3);
        const d2_t v0 = *reinterpret_cast<const d2_t*>(bs + krow * 512 + ((c0 ^ sw) << 4));
        const d2_t v1 = *reinterpret_cast<const d2_t*>(bs + krow * 512 + (((c0 + 1) ^ sw) << 4));
        xb[j][0] = v0[0]; xb[j][1] = v0[1]; xb[j][2] = v1[0]; xb[j][3] = v1[1];
      }
    };
    auto mma = [&](const double (&xa)[4], const double (&xb)[2][4]) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] = __builtin_amdgcn_mfma_f64_4x4x4f64(xa[i], xb[j][e], acc[i][j][e], 0, 0, 0);
    };
    // one LDS read, then a few MFMAs (32 MFMAs per half-step)
#define HINT(NR)                                                     \
  _Pragma("unroll") for (int n_ = 0; n_ < NR; ++n_) {                \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               \
    __builtin_amdgcn_sched_group_barrier(0x008, 32 / NR, 0);         \
  }
    // ---- prologue: two tiles in flight, then the first fragments ----
    issue(0, 0);
    if (ktiles > 1) issue(1, 1);
    __syncthreads();                       // drains vmcnt (the DMA is a pending LDS write) and publishes the tiles
    fragA(0, 0, fa[0]);
    fragB(0, 0, 0, fb[0]);
    for (long kt = 0; kt < ktiles; ++kt) {
      const int cur = (int)(kt & 1);
      // half-step (0,0): B of (0,1) flies
      fragB(cur, 0, 1, fb[1]);
      mma(fa[0][0], fb[0]);
      HINT(4)
      __builtin_amdgcn_sched_barrier(0);
      // half-step (0,1): fragments of step 1 fly
      fragA(cur, 1, fa[1]);
      fragB(cur, 1, 0, fb[0]);
      mma(fa[0][1], fb[1]);
      HINT(8)
      __builtin_amdgcn_sched_barrier(0);
      // half-step (1,0)
      fragB(cur, 1, 1, fb[1]);
      mma(fa[1][0], fb[0]);
      HINT(4)
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                     // tile kt+1 has landed (issued one k-tile ago); buffer cur is free
      if (kt + 2 < ktiles) issue(cur, kt + 2);
      __builtin_amdgcn_sched_barrier(0);
      // half-step (1,1): first fragments of the next k-tile fly (harmless after the last one)
      fragA(cur ^ 1, 0, fa[0]);
      fragB(cur ^ 1, 0, 0, fb[0]);
      mma(fa[1][1], fb[1]);
      HINT(8)
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                       // all waves are done with LDS before the next tile's prologue refills it
    // ---- epilogue: lane owns columns 4*(li&3)..+3 of rows 4*(li>>2)+lk of each 16x16 block ----
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const long row = m0 + (i * 2 + wr) * 16 + 4 * (li >> 2) + lk;
        const long col = n0 + (j * 2 + wc) * 16 + 4 * (li & 3);
        d2_t* p = reinterpret_cast<d2_t*>(C + row * ldc + col);
        p[0] = (d2_t){acc[i][j][0], acc[i][j][1]};
        p[1] = (d2_t){acc[i][j][2], acc[i][j][3]};
      }
  }
}

int main(int argc, char** argv) {
  const long P = ((argc > 1 ? atol(argv[1]) : 200000) / 128) * 128, N = 256, K = 2048;
  double *A, *B, *C;
  hipMalloc(&A, P * K * 8); hipMalloc(&B, K * N * 8); hipMalloc(&C, P * N * 8);
  std::vector<double> h(1 << 20);
  for (auto& x : h) x = (double)rand() / RAND_MAX - 0.5;
  for (size_t off = 0; off < (size_t)P * K; off += h.size()) hipMemcpy(A + off, h.data(), std::min(h.size(), (size_t)P * K - off) * 8, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), (size_t)K * N * 8, hipMemcpyHostToDevice);
  const long tiles = (P / BM) * (N / BN);
  dim3 grid((unsigned)(tiles > 4088 ? 4088 : tiles));
  hipLaunchKernelGGL(gemm_glds, grid, dim3(256), 0, 0, A, B, C, P, N, K, K, N, N);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  // check a few entries against a host dot product
  std::vector<double> c(1024);
  double err = 0.0;
  for (int t = 0; t < 6; ++t) {
    const long row = (t * 7919L * 131) % P, col0 = (t * 37) % (N - 4);
    hipMemcpy(c.data(), C + row * N + col0, 4 * 8, hipMemcpyDeviceToHost);
    for (int q = 0; q < 4; ++q) {
      double s = 0.0;
      for (long k = 0; k < K; ++k) s += h[(row * K + k) % h.size()] * h[(k * N + col0 + q) % h.size()];
      err = fmax(err, fabs(s - c[q]));
    }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(gemm_glds, grid, dim3(256), 0, 0, A, B, C, P, N, K, K, N, N);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("LDS-DMA staged NN product P=%ld K=%ld N=%ld: %.3f ms  %.1f TFLOP/s   max |err| on samples %.2e\n", P, K, N, ms,
         2.0 * P * N * K / ms / 1e9, err);
  return 0;
}
