// gemm_nt.hip -- instantiations of gemm_core.h for operand layout LA=0, LB=0 (see gemm_core.h).
#include "gemm_core.h"
namespace avsr_gemm_impl {
int run_nt(const Params& p, int a_dtype, int b_dtype, int precise, int force_tile, int split_k,
           hipStream_t stream) {
    return dispatch<0, 0>(p, a_dtype, b_dtype, precise, force_tile, split_k, stream);
}
}  // namespace avsr_gemm_impl
