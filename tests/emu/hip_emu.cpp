// hip_emu.cpp -- TEST INFRASTRUCTURE ONLY (see hip_emu.h).
// Fiber-based SIMT emulator: one OS worker per in-flight workgroup, one fiber
// per GPU thread, cooperative scheduling with rendezvous at __syncthreads()
// and at wavefront collectives.
#include "hip_emu.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {

extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_ctx_switch,.-emu_ctx_switch
)");

static constexpr size_t kStack = 256 * 1024;
static constexpr size_t kStageStride = 160;  // bytes per lane per wave op (>= 2*8 bf16 + 16 f32)

enum State { READY, WAIT_BAR, WAIT_WAVE, DONE };

struct Wave {
    int live = 0;
    int arrived = 0;
    long gen_done = 0;  // number of completed wave ops
    std::vector<unsigned char> stage[2];
};

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int tid = 0;
    dim3 tidx;
    State st = READY;
    long wait_gen = 0;
};

struct Worker {
    std::vector<char*> stacks;
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    void* main_sp = nullptr;
    int cur = -1;
    int live = 0;
    int bar_arrived = 0;
    long bar_gen = 0;
    const std::function<void()>* body = nullptr;
    std::vector<char> smem;
    ~Worker() {
        for (char* s : stacks) free(s);
    }
};

static thread_local Worker* W = nullptr;

static void fiber_exit_bookkeeping() {
    Fiber& f = W->fibers[W->cur];
    f.st = DONE;
    W->live--;
    Wave& wv = W->waves[f.tid / 64];
    wv.live--;
    if (wv.live > 0 && wv.arrived == wv.live) {
        wv.arrived = 0;
        wv.gen_done++;
    }
    if (W->live > 0 && W->bar_arrived == W->live) {
        W->bar_arrived = 0;
        W->bar_gen++;
    }
}

static void fiber_entry() {
    (*W->body)();
    fiber_exit_bookkeeping();
    Fiber& f = W->fibers[W->cur];
    emu_ctx_switch(&f.sp, W->main_sp);
    std::fprintf(stderr, "emu: resumed a finished fiber\n");
    std::abort();
}

static void yield_to_main() {
    Fiber& f = W->fibers[W->cur];
    int me = W->cur;
    emu_ctx_switch(&f.sp, W->main_sp);
    // resumed
    W->cur = me;
    threadIdx = W->fibers[me].tidx;
}

void sync_threads() {
    Fiber& f = W->fibers[W->cur];
    W->bar_arrived++;
    if (W->bar_arrived == W->live) {
        W->bar_arrived = 0;
        W->bar_gen++;
        return;
    }
    f.st = WAIT_BAR;
    f.wait_gen = W->bar_gen + 1;
    yield_to_main();
}

int lane_id() { return W->fibers[W->cur].tid & 63; }

char* dyn_smem() { return W->smem.data(); }

const unsigned char* wave_gather(const void* mine, size_t bytes, size_t* stride) {
    if (bytes > kStageStride) {
        std::fprintf(stderr, "emu: wave op payload too large (%zu)\n", bytes);
        std::abort();
    }
    Fiber& f = W->fibers[W->cur];
    Wave& wv = W->waves[f.tid / 64];
    long my_gen = wv.gen_done;  // the op being formed
    // a lane that already deposited for gen `my_gen` cannot be here again before it completes
    std::vector<unsigned char>& st = wv.stage[my_gen & 1];
    std::memcpy(st.data() + (size_t)(f.tid & 63) * kStageStride, mine, bytes);
    wv.arrived++;
    *stride = kStageStride;
    if (wv.arrived == wv.live) {
        wv.arrived = 0;
        wv.gen_done++;
        return st.data();
    }
    f.st = WAIT_WAVE;
    f.wait_gen = my_gen + 1;
    yield_to_main();
    return W->waves[W->fibers[W->cur].tid / 64].stage[my_gen & 1].data();
}

static void run_block(Worker& w, dim3 grid, dim3 block, dim3 bidx, size_t dyn_smem_bytes,
                      const std::function<void()>& body) {
    W = &w;
    const int n = (int)(block.x * block.y * block.z);
    while ((int)w.stacks.size() < n) w.stacks.push_back((char*)aligned_alloc(64, kStack));
    w.fibers.assign(n, Fiber());
    const int nw = (n + 63) / 64;
    w.waves.assign(nw, Wave());
    for (int i = 0; i < nw; i++) {
        w.waves[i].stage[0].assign(64 * kStageStride, 0);
        w.waves[i].stage[1].assign(64 * kStageStride, 0);
    }
    w.live = n;
    w.bar_arrived = 0;
    w.bar_gen = 0;
    w.body = &body;
    if (w.smem.size() < dyn_smem_bytes + 64) w.smem.assign(dyn_smem_bytes + 64, 0);
    // LDS is not zeroed by the hardware: poison the dynamic segment (0xFF bytes = NaN in f32 and bf16) so that a kernel
    // relying on uninitialised shared memory fails on the emulator too
    std::memset(w.smem.data(), 0xFF, w.smem.size());
    blockIdx = bidx;
    blockDim = block;
    gridDim = grid;
    for (int t = 0; t < n; t++) {
        Fiber& f = w.fibers[t];
        f.tid = t;
        f.tidx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        f.stack = w.stacks[t];
        f.st = READY;
        w.waves[t / 64].live++;
        uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;               // fake return address for fiber_entry's frame
        *--sp = (void*)&fiber_entry;   // `ret` target
        for (int r = 0; r < 6; r++) *--sp = nullptr;  // rbp rbx r12..r15
        f.sp = (void*)sp;
    }
    int remaining = n;
    while (remaining > 0) {
        bool progressed = false;
        for (int t = 0; t < n; t++) {
            Fiber& f = w.fibers[t];
            if (f.st == DONE) continue;
            if (f.st == WAIT_BAR && w.bar_gen < f.wait_gen) continue;
            if (f.st == WAIT_WAVE && w.waves[t / 64].gen_done < f.wait_gen) continue;
            f.st = READY;
            w.cur = t;
            threadIdx = f.tidx;
            emu_ctx_switch(&w.main_sp, f.sp);
            progressed = true;
            if (w.fibers[t].st == DONE) remaining--;
        }
        if (!progressed) {
            std::fprintf(stderr,
                         "emu: deadlock in block (%u,%u,%u): divergent __syncthreads or wave op "
                         "(live=%d bar_arrived=%d)\n",
                         bidx.x, bidx.y, bidx.z, w.live, w.bar_arrived);
            std::abort();
        }
    }
}

void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body) {
    const long nblocks = (long)grid.x * grid.y * grid.z;
    if (nblocks <= 0) return;
    static int max_workers = [] {
        const char* e = std::getenv("AVSR_EMU_THREADS");
        int n = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
        return n < 1 ? 1 : n;
    }();
    int nworkers = (int)std::min<long>(nblocks, max_workers);
    std::atomic<long> next{0};
    auto work = [&]() {
        Worker w;
        for (;;) {
            long b = next.fetch_add(1);
            if (b >= nblocks) break;
            dim3 bidx((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y),
                      (unsigned)(b / ((long)grid.x * grid.y)));
            run_block(w, grid, block, bidx, dyn_smem_bytes, body);
        }
        W = nullptr;
    };
    if (nworkers == 1) {
        work();
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nworkers; i++) ts.emplace_back(work);
        for (auto& t : ts) t.join();
    }
}

}  // namespace emu
