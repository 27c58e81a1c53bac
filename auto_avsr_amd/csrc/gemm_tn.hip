// gemm_tn.hip -- instantiations of gemm_core.h for operand layout LA=1, LB=1 (see gemm_core.h).
#include "gemm_core.h"
namespace avsr_gemm_impl {
int run_tn(const Params& p, int a_dtype, int b_dtype, int precise, int force_tile, int split_k,
           hipStream_t stream) {
    return dispatch<1, 1>(p, a_dtype, b_dtype, precise, force_tile, split_k, stream);
}
// several TN problems with the same operand dtypes in one launch (see gemm_core.h MultiParams)
int run_tn_multi(const Params* ps, int n, int a_dtype, int b_dtype, int precise, hipStream_t stream) {
    if (precise) {
        if (a_dtype != 0 || b_dtype != 0) return -1;
        return launch_multi<float, float, 2, 1, 1>(ps, n, stream);
    }
    if (a_dtype == 1 && b_dtype == 1) return launch_multi<bf16_t, bf16_t, 1, 1, 1>(ps, n, stream);
    if (a_dtype == 0 && b_dtype == 0) return launch_multi<float, float, 1, 1, 1>(ps, n, stream);
    return -1;
}
}  // namespace avsr_gemm_impl
