// gemm_nn.hip -- instantiations of gemm_core.h for operand layout LA=0, LB=1 (see gemm_core.h).
#include "gemm_core.h"
namespace avsr_gemm_impl {
int run_nn(const Params& p, int a_dtype, int b_dtype, int precise, int force_tile, int split_k,
           hipStream_t stream) {
    return dispatch<0, 1>(p, a_dtype, b_dtype, precise, force_tile, split_k, stream);
}
}  // namespace avsr_gemm_impl
