// gemm_pair.h -- internal hooks of the paired (data-gradient + weight-gradient) GEMM launch, see gemm_pair.hip.
#pragma once
#include "gemm_core.h"

namespace avsr_pair {
// Called by avsr_gemm_bf16_nt / avsr_gemm_bf16_tn with their fully prepared problem right before the launch.
// Returns true when the problem was taken into the open pair of this thread (the caller then returns without launching).
bool stash_nt(const avsr_gemm_impl::Params& p, int tile, int split_k, hipStream_t stream);
bool stash_tn(const avsr_gemm_impl::Params& p, int split_k, hipStream_t stream);
}  // namespace avsr_pair
