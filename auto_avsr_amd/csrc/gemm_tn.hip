// gemm_tn.hip -- instantiations of gemm_core.h for operand layout LA=1, LB=1 (see gemm_core.h).
#include "gemm_core.h"
namespace avsr_gemm_impl {
int run_tn(const Params& p, int a_dtype, int b_dtype, int precise, int force_tile, int split_k,
           hipStream_t stream) {
    return dispatch<1, 1>(p, a_dtype, b_dtype, precise, force_tile, split_k, stream);
}
}  // namespace avsr_gemm_impl
