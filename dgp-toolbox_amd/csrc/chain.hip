// Replay of recorded small-kernel chains by one workgroup (chain.h).
#include "chain_bodies.h"

namespace dgp {

static thread_local ChainRec* t_chain_rec = nullptr;
ChainRec* chain_recorder() { return t_chain_rec; }
void chain_set_recorder(ChainRec* r) { t_chain_rec = r; }

namespace {

template <int KID, int MODE, typename... T, size_t... I>
__device__ __forceinline__ void chain_call(const VIdx b, const VIdx g, const VIdx t, const VIdx bd, const ChainSlot* s, ArgList<T...>,
                                           std::index_sequence<I...>) {
  ChainBody<KID>::template run<MODE>(b, g, t, bd, chain_get<T>(s[I])...);
}

template <typename L> struct ArgCount;
template <typename... T> struct ArgCount<ArgList<T...>> { static constexpr size_t value = sizeof...(T); };

// one step: the kernel KID over its virtual grid
template <int KID>
__device__ __forceinline__ void chain_step(const ChainOp& op, int tid) {
  using B = ChainBody<KID>;
  using Args = typename B::Args;
  constexpr auto seq = std::make_index_sequence<ArgCount<Args>::value>{};
  const VIdx g{op.gx, op.gy, op.gz};
  if constexpr (B::kBlock == 256) {
    for (unsigned z = 0; z < op.gz; ++z)
      for (unsigned y = 0; y < op.gy; ++y)
        for (unsigned x = 0; x < op.gx; ++x) {
          chain_call<KID, CM_CHAIN>(VIdx{x, y, z}, g, VIdx{(unsigned)tid, 0u, 0u}, VIdx{256u, 1u, 1u}, op.a.s, Args{}, seq);
          __syncthreads();                       // (the body's LDS is reused by the next virtual block)
        }
  } else if constexpr (B::kBlock == 64) {
    const unsigned wave = (unsigned)tid >> 6, lane = (unsigned)tid & 63u;
    const unsigned total = op.gx * op.gy * op.gz;
    if (B::kWaveParallel) {                      // virtual blocks dealt to the four waves (no LDS, no block barrier inside)
      for (unsigned lin = wave; lin < total; lin += 4) {
        const unsigned x = lin % op.gx, y = (lin / op.gx) % op.gy, z = lin / (op.gx * op.gy);
        chain_call<KID, CM_CHAIN_WAVE>(VIdx{x, y, z}, g, VIdx{lane, 0u, 0u}, VIdx{64u, 1u, 1u}, op.a.s, Args{}, seq);
      }
    } else if (wave == 0) {                      // wave 0 walks the batch
      for (unsigned lin = 0; lin < total; ++lin) {
        const unsigned x = lin % op.gx, y = (lin / op.gx) % op.gy, z = lin / (op.gx * op.gy);
        chain_call<KID, CM_CHAIN_WAVE>(VIdx{x, y, z}, g, VIdx{lane, 0u, 0u}, VIdx{64u, 1u, 1u}, op.a.s, Args{}, seq);
      }
    }
  } else {                                       // single-thread kernels
    if (tid == 0) chain_call<KID, CM_CHAIN_WAVE>(VIdx{0u, 0u, 0u}, g, VIdx{0u, 0u, 0u}, VIdx{1u, 1u, 1u}, op.a.s, Args{}, seq);
  }
}

template <bool TA, bool TB>
__device__ __forceinline__ void chain_gemm(const ChainOp& op, int tid) {
  GemmArgs g;
  memcpy(&g, op.a.raw, sizeof(GemmArgs));
  for (unsigned b = 0; b < op.gx; ++b) {
    gemm_small_body<TA, TB, CM_CHAIN>(VIdx{b, 0u, 0u}, VIdx{(unsigned)tid, 0u, 0u}, g);
    __syncthreads();
  }
}

template <int... K>
__device__ __forceinline__ void chain_dispatch(const ChainOp& op, int tid, std::integer_sequence<int, K...>) {
  ((op.kid == K ? (chain_step<K>(op, tid), 0) : 0), ...);
}

}  // namespace

__global__ __launch_bounds__(256) void chain_exec_kernel(const ChainOp* __restrict__ ops, int n) {
  __shared__ ChainOp cur;
  const int tid = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    // the op descriptor into LDS (256 threads x 2 bytes... : as dwords)
    {
      const unsigned* src = reinterpret_cast<const unsigned*>(ops + i);
      unsigned* dst = reinterpret_cast<unsigned*>(&cur);
      for (int w = tid; w < (int)(sizeof(ChainOp) / 4); w += 256) dst[w] = src[w];
    }
    __syncthreads();
    const int kid = cur.kid;
    if (kid == CK_GEMM_NN) chain_gemm<false, false>(cur, tid);
    else if (kid == CK_GEMM_NT) chain_gemm<false, true>(cur, tid);
    else if (kid == CK_GEMM_TN) chain_gemm<true, false>(cur, tid);
    else chain_dispatch(cur, tid, std::make_integer_sequence<int, CK_GEMM_NN>{});
    // the step's global writes are visible to the whole workgroup behind the barrier (one CU, one L1); the scalar cache is
    // not written through by vector stores: drop it, a later step may read what this one wrote through scalar loads
    __syncthreads();
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

hipError_t chain_exec(hipStream_t st, const ChainOp* dev_ops, int n) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(chain_exec_kernel, dim3(1), dim3(256), 0, st, dev_ops, n);
  return hipGetLastError();
}

}  // namespace dgp
