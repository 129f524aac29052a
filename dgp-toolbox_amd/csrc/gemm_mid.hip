// Latency-oriented products for the Mp x Mp matrices of the per-layer chains (Kuu -> Cholesky -> inverse -> W, u, KL;
// the gradient finish; the natural-gradient step) at Mp = 128 .. 512:  C = alpha op(A) op(B) (+ C), K <= 256, batched.
//   reference: what tf.linalg.cholesky / triangular_solve / matmul and their adjoints do on [M, M] operands per layer
//   (layers.py:227-241, 277-308); ~25 such products sit on the critical path of an iteration.
// On the 128 x 64 engine a 256 x 256 x 256 product is 8 workgroups walking 16 k-tiles each, a global -> LDS -> MFMA round
// trip per k-tile: 23-25 us, whatever the chip could do (it is 0.03 GFLOP).  One rank's share of a strongly-scaled iteration
// (N / 8 points) is ~9 ms of which ~1 ms is the first layer's chain in front of and behind everything else, so these 25 us
// launches are what the 8-GPU time is made of (NOTES.md, round 3).  Here an output tile is 32 x 32 (64 workgroups for
// 256 x 256, 512 with D = 8 in the batch).  The k range is walked in chunks of 128: a chunk of both operands (A panel [k][32],
// B panel [k][32], k-major images, 70 KB: two workgroups per CU) goes global -> registers with every load of the chunk in flight
// at once (16-byte loads, 16 per thread and operand), -> LDS, and while the four waves run its 4 x 32 MFMAs each on their
// 16 x 16 quadrants the next chunk's loads are already under way.
// Triangular hints are ignored (the zeros are stored; at this size the skip does not pay).
#include <cstdlib>

#include "gemm_f64.h"

namespace dgp {

constexpr int MD_T = 32;            // output tile
constexpr int MD_KC = 128;          // k chunk
constexpr int MD_K = 256;           // largest k range
constexpr int MD_PA = 34;           // pitch of the A image [k][m] (doubles)
constexpr int MD_PB = 36;           // pitch of the B image [k][n]: 32-byte fragment reads stay aligned
constexpr int MD_NL = MD_KC * MD_T / 2 / 256;      // 16-byte loads per thread, operand and chunk: 8

// One operand's chunk, as [k][32] in the image whatever its storage.  KMAJ: stored k-major (a k row of the panel = 32
// contiguous doubles; thread t: columns 2 (t & 15), k = t / 16 + 16 j); else stored panel-row-major (a panel row = the k range,
// contiguous; thread t: panel row t / 8, k = 2 (t & 7) + 16 j) and written transposed.
template <bool KMAJ>
__device__ __forceinline__ void md_load(d2_t (&r)[MD_NL], const double* __restrict__ src, long ld, long base, long lim, int k0, int K, int tid) {
#pragma unroll
  for (int j = 0; j < MD_NL; ++j) {
    d2_t v = {0.0, 0.0};
    if constexpr (KMAJ) {
      const int k = k0 + (tid >> 4) + 16 * j, c = 2 * (tid & 15);
      if (k < K && base + c < lim) {       // (lim even or the pair's second element masked below)
        const double* p = src + (long)k * ld + base + c;
        if (base + c + 1 < lim) v = *reinterpret_cast<const d2_t*>(p); else v[0] = p[0];
      }
    } else {
      const int row = tid >> 3, k = k0 + 2 * (tid & 7) + 16 * j;
      if (base + row < lim && k < K) {
        const double* p = src + (base + row) * ld + k;
        if (k + 1 < K) v = *reinterpret_cast<const d2_t*>(p); else v[0] = p[0];
      }
    }
    r[j] = v;
  }
}
template <bool KMAJ, int PITCH>
__device__ __forceinline__ void md_store(const d2_t (&r)[MD_NL], double* img, int tid) {
#pragma unroll
  for (int j = 0; j < MD_NL; ++j) {
    if constexpr (KMAJ) {
      const int k = (tid >> 4) + 16 * j, c = 2 * (tid & 15);
      *reinterpret_cast<d2_t*>(img + k * PITCH + c) = r[j];
    } else {
      const int row = tid >> 3, k = 2 * (tid & 7) + 16 * j;
      img[k * PITCH + row] = r[j][0];
      img[(k + 1) * PITCH + row] = r[j][1];
    }
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_mid_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double As[MD_KC * MD_PA];
  __shared__ __attribute__((aligned(16))) double Bs[MD_KC * MD_PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4, wr = wave >> 1, wc = wave & 1;
  const long m0 = (long)blockIdx.y * MD_T, n0 = (long)blockIdx.x * MD_T;
  const double* __restrict__ A = g.A + (long)blockIdx.z * g.sA;
  const double* __restrict__ B = g.B + (long)blockIdx.z * g.sB;
  double* __restrict__ C = g.C + (long)blockIdx.z * g.sC;
  const int K = (int)g.K;

  d2_t ra[MD_NL], rb[MD_NL];
  md_load<TA>(ra, A, g.lda, m0, g.M, 0, K, tid);         // op(A)[m][k]: k-major storage when A is transposed
  md_load<!TB>(rb, B, g.ldb, n0, g.N, 0, K, tid);        // op(B)[k][n]: k-major storage when B is not
  // the wave's 16 x 16 quadrant: lane (li, lk) supplies A[16 wr + li][4 s + lk] and B[4 s + lk][16 wc + 4 (li & 3) + e] to MFMA e and
  // receives C[16 wr + 4 (li >> 2) + lk][16 wc + 4 (li & 3) + e]  (lane maps: gemm_f64.h)
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const double* ap = As + lk * MD_PA + 16 * wr + li;
  const double* bp = Bs + lk * MD_PB + 16 * wc + 4 * (li & 3);
  for (int k0 = 0; k0 < K; k0 += MD_KC) {
    if (k0) __syncthreads();                              // (every wave has left the previous chunk's images)
    md_store<TA, MD_PA>(ra, As, tid);
    md_store<!TB, MD_PB>(rb, Bs, tid);
    __syncthreads();
    if (k0 + MD_KC < K) {
      md_load<TA>(ra, A, g.lda, m0, g.M, k0 + MD_KC, K, tid);
      md_load<!TB>(rb, B, g.ldb, n0, g.N, k0 + MD_KC, K, tid);
    }
    const int kc = K - k0 < MD_KC ? K - k0 : MD_KC, ns = (kc + 3) >> 2;     // (the images are zero past K)
#pragma unroll 8
    for (int s = 0; s < ns; ++s) {
      const double a = ap[s * 4 * MD_PA];
      const d4_t b = *reinterpret_cast<const d4_t*>(bp + s * 4 * MD_PB);
      acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[1], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[2], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[3], acc[3], 0, 0, 0);
    }
  }
  const long row = m0 + 16 * wr + 4 * (li >> 2) + lk, col = n0 + 16 * wc + 4 * (li & 3);
  if (row < g.M) {
    double* c = C + row * g.ldc + col;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col + e < g.N) c[e] = g.beta ? c[e] + g.alpha * acc[e] : g.alpha * acc[e];
  }
}

bool gemm_mid_ok(GemmOp op, const GemmArgs& a) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("DGP_MID_GEMM"); enabled = e ? atoi(e) : 1; }
  if (!enabled) return false;
  if (a.M < 1 || a.N < 1 || a.K < 1 || a.K > MD_K) return false;
  if (a.M > 1024 || a.N > 1024) return false;                       // (the chains' matrices; the point contractions have their own kernels)
  if (a.splits > 1 || a.batch < 1 || a.batch > 64) return false;
  if (a.A == nullptr || a.B == nullptr || a.C == nullptr) return false;
  if (a.ascale_mode != 0 || a.ascale != nullptr || a.rank != 0 || a.eadd != nullptr || a.emul != nullptr || a.C2 != nullptr || a.c2_only) return false;
  if (a.epi != 0 || a.a_blocked || a.c_blocked || a.tri == TRI_OUT_LOWER) return false;    // (an output-triangle product leaves the other half of C alone)
  // 16-byte loads: even leading dimensions and batch strides, 16-byte aligned operands
  if ((a.lda & 1) || (a.ldb & 1) || (a.sA & 1) || (a.sB & 1)) return false;
  if ((reinterpret_cast<uintptr_t>(a.A) & 15u) || (reinterpret_cast<uintptr_t>(a.B) & 15u)) return false;
  // the engine's 128 x 64 tiles fill the chip from ~256 of them on: beyond that it is the throughput machine it was built as
  const long tiles = ((a.M + 127) / 128) * ((a.N + 63) / 64) * a.batch;
  if (tiles > 128) return false;
  (void)op;
  return true;
}

hipError_t gemm_mid(hipStream_t st, GemmOp op, const GemmArgs& a) {
  const dim3 grid((unsigned)((a.N + MD_T - 1) / MD_T), (unsigned)((a.M + MD_T - 1) / MD_T), (unsigned)a.batch), block(256);
  switch (op) {
    case GEMM_NN: hipLaunchKernelGGL((gemm_mid_kernel<false, false>), grid, block, 0, st, a); break;
    case GEMM_NT: hipLaunchKernelGGL((gemm_mid_kernel<false, true>), grid, block, 0, st, a); break;
    case GEMM_TN: hipLaunchKernelGGL((gemm_mid_kernel<true, false>), grid, block, 0, st, a); break;
  }
  return hipGetLastError();
}

}  // namespace dgp
