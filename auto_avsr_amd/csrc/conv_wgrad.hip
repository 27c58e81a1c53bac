// conv_wgrad.hip -- weight gradient of the 3x3 / pad 1 convolutions of the ResNet trunks (resnet.py:10-35), bf16,
// channels-last:     dwp[co][kh][kw][ci] += sum_{n,oh,ow} dy[n,oh,ow,co] * x[n, oh*s+kh-1, ow*s+kw-1, ci]
//
// Why a dedicated kernel: as an implicit GEMM (gemm_tn_fast.hip) every one of the nine taps is its own B operand, so
// the same dy rows and (shifted) x rows are staged nine times and every staged row costs an im2col address decode.
// Here a block owns a 64 (co) x 64 (ci) x ALL NINE TAPS slice of dwp and walks the pixels in tiles of G images x R
// output rows:
//   * one tile = the dy rows of those pixels (k-major, 128 B per pixel = the block's 64 co) and ONE zero-padded x
//     patch (G x ((R-1)s+3) x ((OW-1)s+3) pixels, 128 B each = the block's 64 ci), staged by LDS-DMA into a 2-stage
//     ring; padding pixels and rows past the image come from a page of zeros;
//   * the nine taps are nine SHIFTED VIEWS of that patch: LDS row of pixel k for tap (kh,kw) = view[k] + kh*XW + kw,
//     view[] built once per block (the tile geometry is the same for every tile), so no address decode in the k loop;
//   * both operands have the contraction index (pixels) as the slow LDS dimension: MFMA fragments are fetched with the
//     CDNA4 transpose read ds_read_b64_tr_b16; a 64-byte XOR swizzle keeps the 4-row x 64-byte footprint of a
//     transpose read on distinct banks (rows r and r+2 are 256 bytes apart);
//   * 4 waves = 2 (co halves) x 2 (ci halves), each 9 accumulators of v_mfma_f32_32x32x16_bf16 (144 registers);
//   * the fragment requests are software-pipelined ACROSS the 16-pixel k-steps (round 3: started cold, the chain table ->
//     address -> transpose read -> MFMA left the matrix pipe idle 3/4 of the time) and every LDS read of the loop is in the
//     caller-ordered asm form: a plain LDS load beside an LDS-DMA ring makes the compiler drain the DMA queue first;
//   * split over the pixel tiles across blockIdx.z; every block stores its partial gradient (plain 128-byte-line
//     stores) and a second kernel sums the partials in a fixed order -- deterministic, and cheaper than device-scope
//     float atomics from 512 blocks onto one small tensor (atomics remain available without a workspace).
// Staged bytes per MFMA drop ~9x and the per-load integer work disappears from the inner loop.
#include <math.h>
#include "prims.h"
#include "avsr_hip.h"

namespace {

struct WgParams {
    const bf16_t* dy;
    const bf16_t* x;
    float* dw;         // atomics: dwp ; partial mode: workspace [split][Cout][9][Cin]
    int partial;       // 0: atomicAdd into dw, 1: plain stores of this block's slice into dw + z * |dwp|, 2: none
    int xcd_order;     // 1: XCD-aware work order
    int abl;           // benchmarks only (wrong results): 4 no staging after tile 0, 16 no tiles (per-k-step switches cost time themselves)
    const bf16_t* zero;
    int N, H, W, OH, OW, Cin, Cout, S;
    int G, R;          // images / output rows per tile
    int KP;            // dy rows per tile (G*R*OW rounded up to 16)
    int XW, XR;        // patch width / rows per image: (OW-1)*S+3, (R-1)*S+3
    int XROWS;         // patch rows per tile (G*XR*XW rounded up to 8)
    int nbands;        // ceil(OH / R)
    int ntiles;        // ceil(N / G) * nbands
    int tiles_per_block;
};

// staging descriptor of one LDS row: byte offset from the tile's base pointer; image | row << 8 | swizzle << 24
// (row = kNever for rows that are always zero)
struct RowDesc { int off, meta; };
constexpr int kNever = 0xfff;
// patch-view table entry of pixel k: byte address (relative to the patch) of its row for kw = 0, 1, 2 with the row's
// swizzle bit already placed in bit 6, so that the address of byte column c is simply v[kw] ^ c
struct ViewEnt { int v[4]; };

// LDS image.  Every row is 128 bytes (64 channels of one pixel) in eight 16-byte chunks; chunk c of a row is stored at
// chunk c ^ 4*s where s is one bit of the row's identity, chosen so that the four rows a transpose read touches
// (4 consecutive pixels) alternate s in pairs: rows 256 bytes apart would otherwise hit the same banks.
//   dy rows: s = bit 1 of the row index k;   x patch rows: s = bit 1 of the patch COLUMN (invariant under the kh
//   shift of a tap, so a pixel needs only three pre-swizzled addresses, one per kw).
//
// KG = number of k groups, MW = 32-row co fragments per wave.
//   <1, 1>: 4 waves = 2 co halves x 2 ci halves, two blocks per CU, 2-stage ring.
//   <2, 2>: 4 "fat" waves = 2 k groups x 2 ci halves, each wave BOTH co halves (18 accumulators, 288 registers: one wave
//           per SIMD, one block per CU): a patch fragment feeds two MFMAs, so the LDS bytes per MFMA halve (the thin
//           variant is LDS-read-bound: 22 transpose reads per 9 MFMAs); the two k groups share every staged tile and take
//           alternate 16-pixel k-steps of it; 3-stage ring with a counted vmcnt wait (every wave issues exactly NPT DMA
//           instructions per tile: rows past the tile go from the zero page to a scratch kilobyte); the groups' accumulators
//           are added through LDS before the store -- half the partial-gradient bytes per launch.
//   <2, 1>: 8 thin waves = 2 k groups x the four (co half, ci half) waves (measured slower than <1, 1>; kept for A/B).
template <int KG, int MW>
__global__ __launch_bounds__(64 * (MW == 2 ? 2 : 4) * KG, (KG == 1 && MW == 1) ? 2 : 1) void conv3x3_wgrad_kernel(WgParams p) {
    AVSR_DYN_SMEM(smem);
    constexpr int NW = (MW == 2 ? 2 : 4) * KG, NT = 64 * NW, NSTAGE = KG == 1 ? 2 : 3, WPG = NW / KG;  // waves per k group
    const int lane = threadIdx.x & 63, wave = wave_id(), tid = threadIdx.x;
    const int kg = wave / WPG, wm = MW == 2 ? 0 : (wave >> 1) & 1, wn = wave & 1;
    // Work item w = (pixel-tile range z, ci block, co block).  All (ci, co) blocks of one z stage the same dy / x rows
    // (a different 128-byte slice each), so they should meet in one L2: block b runs on XCD b % 8 (observed), hence
    // XCD x is handed the contiguous run of work items [x * total/8, (x+1) * total/8) with z slowest.
    int w = blockIdx.x;
    if (p.xcd_order) {
        const int total = gridDim.x, xcd = w & 7, slot = w >> 3, q = total >> 3, r = total & 7;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int pairs = (p.Cin >> 6) * (p.Cout >> 6);
    const int zid = w / pairs, pr = w - zid * pairs;
    const int ci0 = (pr % (p.Cin >> 6)) * 64, co0 = (pr / (p.Cin >> 6)) * 64;
    const int KP = p.KP, XROWS = p.XROWS, nbands = p.nbands, G = p.G, R = p.R, S = p.S, N = p.N, H = p.H, OH = p.OH;
    const int stage_bytes = (KP + XROWS) * 128;
    ViewEnt* view = reinterpret_cast<ViewEnt*>(smem + NSTAGE * stage_bytes);  // [KP]
    RowDesc* desc = reinterpret_cast<RowDesc*>(view + KP);                    // [KP + XROWS]
    char* const scratch = reinterpret_cast<char*>(desc + KP + XROWS) + wave * 1024;  // KG = 2: target of the padding DMAs
    const int kvalid = G * R * p.OW;

    // ---- once per block: the tile geometry (identical for every tile)
    for (int k = tid; k < KP; k += NT) {
        int g = 0, y = 0, xx = 0;
        const bool ok = k < kvalid;
        if (ok) {
            g = k / (R * p.OW);
            const int rem = k - g * R * p.OW;
            y = rem / p.OW;
            xx = rem - y * p.OW;
        }
        const int row = ok ? (g * p.XR + y * S) * p.XW + xx * S : 0, col = ok ? xx * S : 0;
        ViewEnt e;
#pragma unroll
        for (int kw = 0; kw < 3; kw++) e.v[kw] = (row + kw) * 128 | (((col + kw) & 2) << 5);
        e.v[3] = 0;
        view[k] = e;
        // dy row k: byte offset relative to pixel (n0, r0, 0), image g, row y inside the band
        desc[k] = ok ? RowDesc{((g * OH + y) * p.OW + xx) * p.Cout * 2, g | (y << 8) | ((k & 2) << 23)}
                     : RowDesc{0, kNever << 8};
    }
    for (int j = tid; j < XROWS; j += NT) {
        const int g = j / (p.XR * p.XW);
        const int rem = j - g * p.XR * p.XW;
        const int yy = rem / p.XW, xx = rem - yy * p.XW;
        const bool ok = g < G && xx >= 1 && xx <= p.W;  // column padding is static, row padding depends on the band
        desc[KP + j] = ok ? RowDesc{((g * H + yy) * p.W + xx) * p.Cin * 2, g | (yy << 8) | ((xx & 2) << 23)}
                          : RowDesc{0, kNever << 8};
    }
    __syncthreads();

    f32x16 acc[MW][9];
#pragma unroll
    for (int m = 0; m < MW; m++)
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][t][r] = 0.f;

    const int t_begin = zid * p.tiles_per_block;
    const int t_end = (p.abl & 16) ? t_begin : min(p.ntiles, t_begin + p.tiles_per_block);

    // stage tile t: every wave instruction moves 8 LDS rows (64 lanes x 16 B); rows are padded to whole instructions.
    // Fixed trip counts (KP <= 256, XROWS <= 288) with wave-uniform guards: the descriptors of all of a lane's rows are
    // fetched first, then turned into addresses -- one LDS round trip per tile instead of one per row.
    constexpr int RPP = 8 * NW, DY_PASSES = 256 / RPP, X_PASSES = (288 + RPP - 1) / RPP, NPT = DY_PASSES + X_PASSES;  // RPP rows per pass
    const char* const zero = reinterpret_cast<const char*>(p.zero);
    const int rsub = lane >> 3;
    const int chunk0 = (lane & 7) * 16, chunk1 = ((lane & 7) ^ 4) * 16;  // source chunk of this lane for s = 0 / 1
    const char* const dy_blk = reinterpret_cast<const char*>(p.dy + co0);
    const char* const x_blk = reinterpret_cast<const char*>(p.x + ci0);
    auto issue = [&](int t, char* stage) {
        const int grp = t / nbands;
        const int n0 = grp * G, r0 = (t - grp * nbands) * R;
        const int gmax = N - n0;      // images g >= gmax do not exist
        const int ymax = OH - r0;     // band rows y >= ymax do not exist
        const int ytop = r0 * S - 1;  // image row of patch row 0
        const char* const dy_tile = dy_blk + (((long)n0 * OH + r0) * p.OW) * p.Cout * 2;
        const char* const x_tile = x_blk + (((long)n0 * H + ytop) * p.W - 1) * p.Cin * 2;
        // (caller-ordered reads, prims.h: a plain LDS load here makes the compiler drain every LDS-DMA still in flight)
        i32x2 dd[DY_PASSES], dx[X_PASSES];
#pragma unroll
        for (int j = 0; j < DY_PASSES; j++)
            if (wave * 8 + j * RPP < KP) dd[j] = lds_read8_async(desc + wave * 8 + j * RPP + rsub);
#pragma unroll
        for (int j = 0; j < X_PASSES; j++)
            if (wave * 8 + j * RPP < XROWS) dx[j] = lds_read8_async(desc + KP + wave * 8 + j * RPP + rsub);
        lds_wait<0>();
#pragma unroll
        for (int j = 0; j < DY_PASSES; j++) lds_tie(dd[j]);
#pragma unroll
        for (int j = 0; j < X_PASSES; j++) lds_tie(dx[j]);
#pragma unroll
        for (int j = 0; j < DY_PASSES; j++) {
            const int row0 = wave * 8 + j * RPP;
            if (row0 < KP) {
                const int m = dd[j][1];
                const bool ok = (m & 255) < gmax && ((m >> 8) & 0xfff) < ymax;
                glds16(ok ? dy_tile + (dd[j][0] + ((m >> 24) ? chunk1 : chunk0)) : zero, stage + row0 * 128);
            } else if (KG > 1) {
                glds16(zero, scratch);
            }
        }
        char* xs = stage + KP * 128;
#pragma unroll
        for (int j = 0; j < X_PASSES; j++) {
            const int row0 = wave * 8 + j * RPP;
            if (row0 < XROWS) {
                const int m = dx[j][1];
                const bool ok = (m & 255) < gmax && (unsigned)(ytop + ((m >> 8) & 0xfff)) < (unsigned)H;
                glds16(ok ? x_tile + (dx[j][0] + ((m >> 24) ? chunk1 : chunk0)) : zero, xs + row0 * 128);
            } else if (KG > 1) {
                glds16(zero, scratch);
            }
        }
    };
    auto issue_padding = [&]() {  // KG = 2: keeps the number of DMA instructions per loop trip constant
#pragma unroll
        for (int j = 0; j < NPT; j++) glds16(zero, scratch);
    };

    if (t_begin < t_end) issue(t_begin, smem);
    if (KG > 1) {
        if (t_begin + 1 < t_end) issue(t_begin + 1, smem + stage_bytes);
        else issue_padding();
    }
    // per-lane constants of the transpose reads (prims.h lds_tr16): lane (g4, i) addresses row 8*(g4>>1) + (i>>2) (+4)
    // of a 16-row k-step, 4 consecutive columns starting at 16*(g4&1) + 4*(i&3) of the wave's 32-column slice
    const int g4 = lane >> 4, li = lane & 15;
    const int krow = 8 * (g4 >> 1) + (li >> 2);  // bit 1 of krow == bit 1 of krow + 4 == bit 1 of the dy row
    int acol[MW];  // swizzled byte column in the dy tile, per co fragment
#pragma unroll
    for (int m = 0; m < MW; m++) acol[m] = (((wm + m) * 32 + 16 * (g4 & 1) + 4 * (li & 3)) * 2) ^ ((krow & 2) << 5);
    const int bcol = (wn * 32 + 16 * (g4 & 1) + 4 * (li & 3)) * 2;                         // byte column, x patch
    const int kh_bytes = p.XW * 128;                                                       // one patch row of pixels

    struct FragSet {
        bf16x4 alo[MW], ahi[MW];  // dy fragment(s) of the step
        bf16x4 blo[5], bhi[5];  // patch fragments of taps 0..4
        i32x4 cvlo, cvhi;       // view entries of the step itself (its taps 5..8 are requested while it runs)
        i32x4 vlo, vhi;         // view entries of the FOLLOWING step
    };
    const int nks = KP / 16;
    auto view_of = [&](int ks_, int hi) { return view + (ks_ < nks ? ks_ : nks - 1) * 16 + krow + 4 * hi; };
    auto tap_lo = [&](const i32x4& v, const char* xs_, int tap) {
        return reinterpret_cast<const bf16_t*>(xs_ + (v[tap % 3] ^ bcol) + (tap / 3) * kh_bytes);
    };
    // dy fragment(s) + taps 0..4 of step ks_ whose view entries are (vl, vh): 2 MW + 10 reads
    // (dys_ = the tile's dy rows at this lane's krow; xs_ = its patch)
    auto request_a = [&](FragSet& f, const char* dys_, int ks_) {
#pragma unroll
        for (int m = 0; m < MW; m++) {
            f.alo[m] = lds_tr16_async(reinterpret_cast<const bf16_t*>(dys_ + acol[m] + ks_ * 2048));
            f.ahi[m] = lds_tr16_async(reinterpret_cast<const bf16_t*>(dys_ + acol[m] + ks_ * 2048 + 512));
        }
    };
    auto request_set = [&](FragSet& f, const i32x4& vl, const i32x4& vh, const char* dys_, const char* xs_, int ks_) {
        request_a(f, dys_, ks_);
#pragma unroll
        for (int tap = 0; tap < 5; tap++) {
            f.blo[tap] = lds_tr16_async(tap_lo(vl, xs_, tap));
            f.bhi[tap] = lds_tr16_async(tap_lo(vh, xs_, tap));
        }
    };
    // One k-step on set `cur`, which it replaces by the set of step ks_ + KG.  Read order of a step:
    //   [view entries of step + 2] | tap 0..3: MFMA, request tap + 5 | tap 4: MFMA, request next dy + next tap 0 |
    //   tap 5..8: MFMA, request next tap 1..4     -- 10 (taps 5..8: 8 + 2 MW) reads are issued between a fragment and its use.
    // A tile's last step requests a set nobody uses (clamped view entries, addresses inside the ring): one loop body, no
    // tail variants -- the caller drains the queue after the last step.
    auto run_step = [&](FragSet& cur, const char* dys_, const char* xs_, int ks_, f32x16 (&acc_)[MW][9]) {
        FragSet nxt;
        nxt.vlo = lds_read16_async(view_of(ks_ + 2 * KG, 0));
        nxt.vhi = lds_read16_async(view_of(ks_ + 2 * KG, 1));
        bf16x4 blo[4], bhi[4];
        bf16x8 a[MW];
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            if (tap < 5) lds_wait<10>();
            else lds_wait<8 + 2 * MW>();
            bf16x8 b;
            if (tap == 0) {
#pragma unroll
                for (int m = 0; m < MW; m++) {
                    lds_tie(cur.alo[m]);
                    lds_tie(cur.ahi[m]);
                    a[m] = bf16x8{cur.alo[m][0], cur.alo[m][1], cur.alo[m][2], cur.alo[m][3],
                                  cur.ahi[m][0], cur.ahi[m][1], cur.ahi[m][2], cur.ahi[m][3]};
                }
            }
            if (tap < 5) {
                lds_tie(cur.blo[tap]);
                lds_tie(cur.bhi[tap]);
                b = bf16x8{cur.blo[tap][0], cur.blo[tap][1], cur.blo[tap][2], cur.blo[tap][3],
                           cur.bhi[tap][0], cur.bhi[tap][1], cur.bhi[tap][2], cur.bhi[tap][3]};
            } else {
                lds_tie(blo[tap - 5]);
                lds_tie(bhi[tap - 5]);
                b = bf16x8{blo[tap - 5][0], blo[tap - 5][1], blo[tap - 5][2], blo[tap - 5][3],
                           bhi[tap - 5][0], bhi[tap - 5][1], bhi[tap - 5][2], bhi[tap - 5][3]};
            }
#pragma unroll
            for (int m = 0; m < MW; m++) {
                // 18 accumulators = 288 registers: four of them live in architectural VGPRs (prims.h mfma32_vgpr)
                if (MW == 2 && m == 1 && tap >= 5) acc_[m][tap] = mfma32_vgpr(a[m], b, acc_[m][tap]);
                else acc_[m][tap] = mfma32(a[m], b, acc_[m][tap]);
            }
            if (tap < 4) {
                blo[tap] = lds_tr16_async(tap_lo(cur.cvlo, xs_, tap + 5));
                bhi[tap] = lds_tr16_async(tap_lo(cur.cvhi, xs_, tap + 5));
            } else {
                if (tap == 4) {
                    lds_tie(cur.vlo);  // requested two steps ago
                    lds_tie(cur.vhi);
                    nxt.cvlo = cur.vlo;
                    nxt.cvhi = cur.vhi;
                    request_a(nxt, dys_, ks_ + KG);
                }
                nxt.blo[tap - 4] = lds_tr16_async(tap_lo(nxt.cvlo, xs_, tap - 4));
                nxt.bhi[tap - 4] = lds_tr16_async(tap_lo(nxt.cvhi, xs_, tap - 4));
            }
            sched_fence();  // keep each MFMA right behind its own wait
        }
        cur = nxt;
    };
    // view entries of this wave's first two steps: the same for every tile
    i32x4 v_first[4] = {lds_read16_async(view_of(kg, 0)), lds_read16_async(view_of(kg, 1)),
                        lds_read16_async(view_of(kg + KG, 0)), lds_read16_async(view_of(kg + KG, 1))};
    lds_wait<0>();
#pragma unroll
    for (int i = 0; i < 4; i++) lds_tie(v_first[i]);

    int buf = 0;
    for (int t = t_begin; t < t_end; t++) {
        if (KG == 1) {
            wait_vmcnt<0>();
            __syncthreads();  // tile t has landed for every wave; everyone is done with the other buffer
            buf = (t - t_begin) & 1;
            if (t + 1 < t_end && !(p.abl & 4)) issue(t + 1, smem + (buf ^ 1) * stage_bytes);
        } else {
            wait_vmcnt<NPT>();    // everything but the newest group (tile t + 1 or its padding) has landed: tile t is in
            block_barrier_raw();  // ... for every wave, and everyone is done with tile t - 1 = the buffer of tile t + 2
                                  // (__syncthreads() would drain the DMA queue; this wave's LDS reads ended with the last tap)
            const int nb = buf >= 1 ? buf - 1 : 2;  // (buf + 2) % 3
            if (t + 2 < t_end && !(p.abl & 4)) issue(t + 2, smem + nb * stage_bytes);
            else issue_padding();
        }
        const char* dys = smem + buf * stage_bytes + krow * 128;
        const char* xs = smem + buf * stage_bytes + KP * 128;
        if (KG > 1) buf = buf == 2 ? 0 : buf + 1;
        // ---- the tile's k-steps (16 pixels each; this wave's are ks = kg, kg + KG, ...), software-pipelined ACROSS steps.
        // The chain view table -> patch addresses -> transpose reads -> MFMA is two LDS round trips long; started cold in
        // every step it left the matrix pipe idle 3/4 of the time.  Step k therefore runs on a fragment set that step k - 1
        // requested: its dy fragment, its first five patch taps and the view entries of step k + 1; it requests its own last
        // four taps, and -- from tap 4 on, interleaved with its MFMAs -- the set of step k + 1.  All reads are in the
        // caller-ordered asm form (prims.h) and return in order, so every wait is "lgkmcnt(number of reads issued after the
        // fragment needed now)": 10 in the steady state, 12 reads in flight at most (the counter has 4 bits).
        const int nsteps = (KP / 16 - kg + KG - 1) / KG;
        if (nsteps > 0) {
            FragSet f;
            lds_tie(v_first[0]);
            lds_tie(v_first[1]);
            request_set(f, v_first[0], v_first[1], dys, xs, kg);  // (the first steps' view entries are tile-invariant)
            f.cvlo = v_first[0];
            f.cvhi = v_first[1];
            f.vlo = v_first[2];
            f.vhi = v_first[3];
            for (int i = 0, ks = kg; i < nsteps; i++, ks += KG) run_step(f, dys, xs, ks, acc);
            lds_wait<0>();  // the set requested by the last step
            lds_tie(f.alo[0]);
        }
    }

    if (KG > 1) {
        // the second k group hands its accumulators over through LDS (lane-contiguous dwords: no bank conflicts); the staging
        // ring, the tables and the padding DMAs are finished with first
        wait_vmcnt<0>();
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem) + (size_t)(wave % WPG) * MW * 144 * 64 + lane;
        if (kg == 1) {
#pragma unroll
            for (int m = 0; m < MW; m++)
#pragma unroll
                for (int tap = 0; tap < 9; tap++)
#pragma unroll
                    for (int r = 0; r < 16; r++) red[((m * 9 + tap) * 16 + r) * 64] = acc[m][tap][r];
        }
        __syncthreads();
        if (kg == 1) return;
    }
    const float* const red = reinterpret_cast<const float*>(smem) + (size_t)(wave % WPG) * MW * 144 * 64 + lane;
    if (p.partial == 2) return;
    float* out = p.dw + (p.partial ? (size_t)zid * p.Cout * 9 * p.Cin : 0);
#pragma unroll
    for (int m = 0; m < MW; m++)
#pragma unroll
        for (int tap = 0; tap < 9; tap++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + (wm + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ci = ci0 + wn * 32 + (lane & 31);
                float* dst = out + ((size_t)co * 9 + tap) * p.Cin + ci;
                const float v = KG > 1 ? acc[m][tap][r] + red[((m * 9 + tap) * 16 + r) * 64] : acc[m][tap][r];
                if (p.partial) *dst = v;
                else atomicAdd(dst, v);
            }
}

// dwp[i] = sum_z ws[z][i]: the blocks' partial gradients are combined in a fixed order (deterministic, no atomics).
// 1024 threads = 64 float4 columns x 16 z-lanes; a z-lane sums every 16th partial with independent loads in flight,
// then the 16 lane sums are combined through LDS in a fixed order.
// torch_layout: dw is [Cout][Cin][3][3] (what autograd hands to the optimizer) instead of [Cout][3][3][Cin]
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int split,
                                                            int Cin, int torch_layout) {
    __shared__ f32x4 part[16][64];
    const int tx = threadIdx.x & 63, tz = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + tx;
    f32x4 s{0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        const f32x4* src = reinterpret_cast<const f32x4*>(ws) + i;
        int z = tz;
        for (; z + 48 < split; z += 64) {
            const f32x4 v0 = src[(long)z * n4], v1 = src[(long)(z + 16) * n4], v2 = src[(long)(z + 32) * n4],
                        v3 = src[(long)(z + 48) * n4];
#pragma unroll
            for (int e = 0; e < 4; e++) s[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
        }
        for (; z < split; z += 16) {
            const f32x4 v = src[(long)z * n4];
#pragma unroll
            for (int e = 0; e < 4; e++) s[e] += v[e];
        }
    }
    part[tz][tx] = s;
    __syncthreads();
    if (tz == 0 && i < n4) {
#pragma unroll
        for (int q = 1; q < 16; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) s[e] += part[q][tx][e];
        if (!torch_layout) {
            reinterpret_cast<f32x4*>(dw)[i] = s;
        } else {
            const long e0 = i * 4;  // = (co * 9 + tap) * Cin + ci
            const int ci = (int)(e0 % Cin);
            const long ct = e0 / Cin;
            const int tap = (int)(ct % 9);
            const long co = ct / 9;
#pragma unroll
            for (int e = 0; e < 4; e++) dw[(co * Cin + ci + e) * 9 + tap] = s[e];
        }
    }
}

struct Plan { WgParams p; int split; size_t lds; int kg, variant; };

// tile geometry: G whole images (small images) or one band of R output rows; at most 256 pixels and 38 KiB per stage
// (two stages, two blocks per CU); maximise useful pixels per unit of max(MFMA time, staging time)
bool make_plan(int N, int H, int W, int Cin, int Cout, int stride, Plan& pl) {
    const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
    WgParams& p = pl.p;
    p = WgParams{};
    p.N = N; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.Cin = Cin; p.Cout = Cout; p.S = stride;
    p.XW = (OW - 1) * stride + 3;
    double best = -1.0;
    for (int mode = 0; mode < 2; mode++)
        for (int v = 1; v <= (mode == 0 ? OH : 64); v++) {
            const int G = mode == 0 ? 1 : v, R = mode == 0 ? v : OH;
            if (G > N || G > 255 || R > 1000) break;
            const int XR = (R - 1) * stride + 3;
            const int KP = (G * R * OW + 15) / 16 * 16, XROWS = (G * XR * p.XW + 7) / 8 * 8;
            if (KP > 256 || XROWS > 288 || (KP + XROWS) * 128 > 37 * 1024) break;
            const int nb = (OH + R - 1) / R;
            const double useful = (double)G * OH * OW / nb;                                  // real pixels per tile
            const double cost = fmax(KP / 16 * 9 * 32.0, (KP + XROWS) * 128 / 20.0) + 300.0;  // clocks per tile
            if (useful / cost > best) {
                best = useful / cost;
                p.G = G; p.R = R; p.XR = XR; p.KP = KP; p.XROWS = XROWS; p.nbands = nb;
            }
        }
    if (best <= 0.0) return false;
    p.ntiles = (N + p.G - 1) / p.G * p.nbands;
    const int pairs = (Cin / 64) * (Cout / 64);
    // knob 16: kernel variant (see the kernel): 0 / 1 = <1, 1>, 2 = <2, 2> "fat waves", 3 = <2, 1>; knob 15: block-count target
    pl.variant = avsr_tune_knobs[16] == 2 ? 2 : (avsr_tune_knobs[16] == 3 ? 3 : 1);
    pl.kg = pl.variant == 1 ? 1 : 2;
    const int target = avsr_tune_knobs[15] > 0 ? avsr_tune_knobs[15] : (pl.kg == 1 ? 512 : 256);
    int split = (target + pairs - 1) / pairs;  // default: fill every CU
    if (split > p.ntiles) split = p.ntiles;
    p.tiles_per_block = (p.ntiles + split - 1) / split;
    pl.split = (p.ntiles + p.tiles_per_block - 1) / p.tiles_per_block;
    pl.lds = (size_t)(pl.kg == 1 ? 2 : 3) * (p.KP + p.XROWS) * 128 + (size_t)p.KP * 16 + (size_t)(p.KP + p.XROWS) * 8;
    if (pl.kg == 2) {
        pl.lds += 8 * 1024;                                                     // one scratch kilobyte per wave (at most 8)
        if (pl.lds < (size_t)4 * 144 * 64 * 4) pl.lds = (size_t)4 * 144 * 64 * 4;  // the k groups' hand-over
    }
    return true;
}

}  // namespace

// Bytes of workspace with which avsr_conv3x3_wgrad_bf16 runs in its deterministic (partial sums + ordered reduce) mode.
extern "C" int64_t avsr_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int stride) {
    Plan pl;
    if (N <= 0 || Cin % 64 || Cout % 64 || (stride != 1 && stride != 2) || !make_plan(N, H, W, Cin, Cout, stride, pl)) return 0;
    return (int64_t)pl.split * Cout * 9 * Cin * 4;
}

// dwp[Cout][3][3][Cin] (f32) = dy[N,OH,OW,Cout]^T (x) shifted x[N,H,W,Cin]; 3x3, pad 1, stride 1 or 2; Cin % 64 == 0,
// Cout % 64 == 0; zero_page: >= 16 zero bytes of device memory.  With a workspace of at least
// avsr_conv3x3_wgrad_workspace_bytes() dwp is OVERWRITTEN with the ordered sum of the blocks' partial gradients;
// without one (workspace = NULL) the blocks atomicAdd into dwp, which the caller must have zeroed.
extern "C" int avsr_conv3x3_wgrad_bf16(const void* dy, const void* x, float* dwp, const void* zero_page, void* workspace,
                                       int64_t workspace_bytes, int N, int H, int W, int Cin, int Cout, int stride,
                                       int torch_layout, hipStream_t stream) {
    AVSR_REQUIRE(!torch_layout || workspace != nullptr, "conv3x3_wgrad_bf16: the torch layout needs the workspace mode");
    AVSR_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "conv3x3_wgrad_bf16: channel counts must be multiples of 64");
    AVSR_REQUIRE(stride == 1 || stride == 2, "conv3x3_wgrad_bf16: stride must be 1 or 2");
    AVSR_REQUIRE(zero_page != nullptr, "conv3x3_wgrad_bf16: zero page required");
    AVSR_REQUIRE((long)N * H * W < (1l << 31), "conv3x3_wgrad_bf16: pixel count exceeds int32");
    if (N <= 0) return 0;
    Plan pl;
    AVSR_REQUIRE(make_plan(N, H, W, Cin, Cout, stride, pl), "conv3x3_wgrad_bf16: image row too wide for one LDS tile");
    WgParams& p = pl.p;
    p.dy = reinterpret_cast<const bf16_t*>(dy); p.x = reinterpret_cast<const bf16_t*>(x);
    p.zero = reinterpret_cast<const bf16_t*>(zero_page);
    const int64_t need = (int64_t)pl.split * Cout * 9 * Cin * 4;
    const bool partial = workspace != nullptr && (avsr_tune_knobs[3] != 1 || torch_layout);
    AVSR_REQUIRE(!partial || workspace_bytes >= need, "conv3x3_wgrad_bf16: workspace too small");
    p.dw = partial ? reinterpret_cast<float*>(workspace) : dwp;
    p.partial = avsr_tune_knobs[3] == 2 ? 2 : (partial ? 1 : 0);
    p.xcd_order = avsr_tune_knobs[4] != 1;
    p.abl = avsr_tune_knobs[5];
    dim3 grid((Cin / 64) * (Cout / 64) * pl.split);
    if (pl.variant == 1) AVSR_LAUNCH((conv3x3_wgrad_kernel<1, 1>), grid, dim3(256), pl.lds, stream, p);
    else if (pl.variant == 2) AVSR_LAUNCH((conv3x3_wgrad_kernel<2, 2>), grid, dim3(256), pl.lds, stream, p);
    else AVSR_LAUNCH((conv3x3_wgrad_kernel<2, 1>), grid, dim3(512), pl.lds, stream, p);
    if (partial) {
        const long n4 = (long)Cout * 9 * Cin / 4;
        AVSR_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(1024), 0, stream,
                    reinterpret_cast<const float*>(workspace), dwp, n4, pl.split, Cin, torch_layout);
    }
    AVSR_CHECK_LAUNCH("conv3x3_wgrad_bf16");
    return 0;
}
