// gemm_pair.hip -- the two GEMMs of a Linear layer's backward pass in ONE launch.
//
// Given the output gradient dY of y = x W^T, the data gradient dX = dY W (an NT GEMM against the transposed bf16 weight
// copy, gemm_fast.hip) and the weight gradient dW = dY^T x (a TN GEMM, gemm_tn_fast.hip) are independent.  At the
// reference's batch geometry (M = B*T <= 1600 rows) either one is ~300 tiles of 64x64 -- barely more than one block per
// CU on a 256-CU part, so each launch is a latency-bound tail of a few k-steps (measured: 22 us each, 44.5 us back to
// back, 29.8 us when the two run concurrently on two streams).  Here the blocks of both problems share one grid:
// blocks [0, n_nt) run the NT tile kernel, the rest the TN tile kernel, so every CU holds 2-3 blocks of mixed work
// and no stream fork / join or extra launch boundary is paid.
//
// Interface: avsr_gemm_pair_begin() opens a pair on the calling thread; the next avsr_gemm_bf16_nt and
// avsr_gemm_bf16_tn calls on that thread are recorded instead of launched (only the tile shapes the pair kernel is
// built for; anything else launches immediately as usual); avsr_gemm_pair_end() launches what was recorded.  The two
// problems must be independent (neither reads the other's output).
#include "gemm_fast_kernel.h"
#include "gemm_tn_kernel.h"
#include "gemm_pair.h"
#include "avsr_hip.h"

namespace {

using avsr_gemm_impl::Params;
using avsr_fast::FastKernel;
using Tn = avsr_tn::TnKernel<3, 0>;
using Tn4 = avsr_tn::TnKernel<3, 0, 1, 1, 4>;  // all four k-steps' transpose reads of a tile in flight together (knob 20 = 1)
using Tn128 = avsr_tn::TnKernel<2, 0, 2, 2>;  // 128 x 128 weight-gradient tiles (gemm_tn_kernel.h): the large dW of FFN / fused Q / K / V

struct PairState {
    bool active = false, have_nt = false, have_tn = false;
    Params nt, tn;
    int nt_tile = 0;
    int gxa = 0, gya = 0;            // NT grid
    int gxb = 0, gyb = 0, gzb = 0;   // TN grid (z = k-split)
    int tn_big = 0;                  // 1: 128 x 128 TN tiles
    hipStream_t stream = nullptr;
};
thread_local PairState g_pair;

template <class KA, class KB>
__global__ __launch_bounds__(256) void gemm_pair_kernel(Params pa, Params pb, int na, int gxa, int gya, int gxb, int gyb) {
    AVSR_DYN_SMEM(smem);
    int b = blockIdx.x;
    if (b < na) {
        const int by = b / gxa;
        KA::run_at(pa, smem, b - by * gxa, by, 0, gxa, gya, 1);
    } else {
        b -= na;
        const int r = b / gxb;
        KB::run_at(pb, smem, b - r * gxb, r % gyb, r / gyb);
    }
}

template <class KA, class KB>
void launch_pair(const PairState& s) {
    const int na = s.have_nt ? s.gxa * s.gya : 0;
    const int nb = s.have_tn ? s.gxb * s.gyb * s.gzb : 0;
    const size_t lds = KA::LDS_BYTES > KB::LDS_BYTES ? KA::LDS_BYTES : KB::LDS_BYTES;
    AVSR_LAUNCH((gemm_pair_kernel<KA, KB>), dim3(na + nb), dim3(256), lds, s.stream, s.nt, s.tn, na, s.gxa > 0 ? s.gxa : 1,
                s.gya > 0 ? s.gya : 1, s.gxb > 0 ? s.gxb : 1, s.gyb > 0 ? s.gyb : 1);
}

}  // namespace

namespace avsr_pair {

bool stash_nt(const Params& p, int tile, int split_k, hipStream_t stream) {
    PairState& s = g_pair;
    if (!s.active || s.have_nt || split_k > 1 || p.accumulate || (tile != 1 && tile != 7)) return false;
    if (s.have_tn && s.stream != stream) return false;
    s.nt = p;
    s.nt.xcd_order = (tile == 1 && avsr_tune_knobs[1] != 2) ? 1 : 0;  // NT blocks are blocks [0, na) of the pair grid: id % 8 still names the XCD
    s.nt.k_chunk = ((p.K + 63) / 64) * 64;
    s.nt_tile = tile;
    const int BM = tile == 1 ? 64 : 128;
    s.gxa = (p.N + 63) / 64;
    s.gya = (p.M + BM - 1) / BM;
    s.stream = stream;
    s.have_nt = true;
    return true;
}

bool stash_tn(const Params& p, int split_k, hipStream_t stream) {
    PairState& s = g_pair;
    if (!s.active || s.have_tn) return false;
    if (s.have_nt && s.stream != stream) return false;
    if (split_k < 1) split_k = 1;
    int kc = (p.K + split_k - 1) / split_k;
    kc = ((kc + 63) / 64) * 64;
    split_k = (p.K + kc - 1) / kc;
    s.tn = p;
    // 128 x 128 tiles for the weight gradient (half the operand bytes per output, half the transpose reads per MFMA) -- MEASURED
    // SLOWER on the MI355X (round 5, fixed batch A: 23.85 ms with 64 x 64 tiles everywhere, 24.1 - 24.2 with 128 x 128 for the FFN /
    // fused Q / K / V weight gradients, 24.8 with 128 x 128 everywhere; the FFN pair 50.8 -> 60.3 us): the TN tile is bound by the
    // latency of its transpose-read -> MFMA chain at one wave per SIMD, not by the operand stream, and four times fewer blocks
    // leave less to overlap it with.  Kept behind knob 19 = 2 (tests/test_kernels_basic.py) as the record of the experiment.
    s.tn_big = avsr_tune_knobs[19] == 2 && p.M >= 128 && p.N >= 128;
    if (s.tn_big) {
        split_k = 1;  // (the data-gradient tiles of the pair fill the chip; no atomics, no zeroed output needed -- but honour accumulate)
        kc = ((p.K + 63) / 64) * 64;
    }
    const int tb = s.tn_big ? 128 : 64;
    s.tn.k_chunk = kc;
    s.gxb = (p.N + tb - 1) / tb;
    s.gyb = (p.M + tb - 1) / tb;
    s.gzb = split_k;
    s.stream = stream;
    s.have_tn = true;
    return true;
}

}  // namespace avsr_pair

extern "C" int avsr_gemm_pair_begin(void) {
    AVSR_REQUIRE(!g_pair.active, "gemm_pair_begin: a pair is already open on this thread");
    g_pair = PairState{};
    g_pair.active = true;
    return 0;
}

extern "C" int avsr_gemm_pair_end(void) {
    PairState s = g_pair;
    g_pair = PairState{};
    AVSR_REQUIRE(s.active, "gemm_pair_end: no open pair on this thread");
    if (!s.have_nt && !s.have_tn) return 0;
    if (s.tn_big) {
        if (s.have_nt && s.nt_tile == 7) launch_pair<FastKernel<128, 64, 2, 0>, Tn128>(s);
        else launch_pair<FastKernel<64, 64, 3, 0>, Tn128>(s);
    } else if (avsr_tune_knobs[20] == 1) {
        if (s.have_nt && s.nt_tile == 7) launch_pair<FastKernel<128, 64, 2, 0>, Tn4>(s);
        else launch_pair<FastKernel<64, 64, 3, 0>, Tn4>(s);
    } else if (s.have_nt && s.nt_tile == 7) launch_pair<FastKernel<128, 64, 2, 0>, Tn>(s);
    else launch_pair<FastKernel<64, 64, 3, 0>, Tn>(s);
    AVSR_CHECK_LAUNCH("gemm_pair");
    return 0;
}
