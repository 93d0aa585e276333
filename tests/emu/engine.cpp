// Fiber scheduler of the CPU emulation (see hip/hip_runtime.h).  TEST INFRASTRUCTURE ONLY.
#include "hip/hip_runtime.h"

namespace shim {

static Engine g_engine;
Engine& eng() { return g_engine; }

// Scheduling order of the fibers between synchronisation points: 0 = ascending thread id, 1 = descending, 2 = waves descending
// with lanes ascending.  A kernel whose result depends on it is missing a barrier (or relies on an unordered float atomic).
static int g_order = 0;
static inline int pick(int i, int n) {
    if (g_order == 2 && n % WAVE == 0) return (n / WAVE - 1 - i / WAVE) * WAVE + i % WAVE;
    if (g_order >= 1) return n - 1 - i;
    return i;
}

constexpr size_t STACK = 256 * 1024;

static void trampoline() {
    Engine& e = eng();
    e.body();
    e.fib[e.cur].st = DONE;
    swapcontext(&e.fib[e.cur].ctx, &e.sched);
}

void yield(State s) {
    Engine& e = eng();
    Fiber& f = e.fib[e.cur];
    f.st = s;
    swapcontext(&f.ctx, &e.sched);
}

static void release_waves(Engine& e) {
    for (int base = 0; base < e.nthreads; base += WAVE) {
        const int n = min(WAVE, e.nthreads - base);
        bool any = false, all = true;
        for (int l = 0; l < n; ++l) {
            const State s = e.fib[base + l].st;
            if (s == AT_WAVE) any = true;
            else if (s != DONE) all = false;
        }
        if (!any || !all) continue;
        for (int l = 0; l < n; ++l) {
            Fiber& f = e.fib[base + l];
            e.snap[(size_t)(base + l) * 2] = f.pub[0];
            e.snap[(size_t)(base + l) * 2 + 1] = f.pub[1];
            e.snap_pred[base + l] = f.st == AT_WAVE ? f.pred : -1;      // -1: the lane has exited
            if (f.st == AT_WAVE) f.st = RUNNABLE;
        }
    }
}

static bool release_block(Engine& e) {
    int waiting = 0, count = 0;
    for (int t = 0; t < e.nthreads; ++t) {
        const State s = e.fib[t].st;
        if (s == AT_BLOCK) { ++waiting; count += e.fib[t].pred > 0; }
        else if (s != DONE) return false;
    }
    if (!waiting) return false;
    e.block_count = count;
    for (int t = 0; t < e.nthreads; ++t)
        if (e.fib[t].st == AT_BLOCK) { e.fib[t].st = RUNNABLE; e.fib[t].pred = 0; }
    return true;
}

static void run_block(Engine& e) {
    for (int t = 0; t < e.nthreads; ++t) {
        Fiber& f = e.fib[t];
        f.st = RUNNABLE;
        f.lin = t;
        f.pred = 0;
        f.tid = dim3(t % e.block.x, (t / e.block.x) % e.block.y, t / (e.block.x * e.block.y));
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = e.stacks.data() + (size_t)t * STACK;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = &e.sched;
        makecontext(&f.ctx, trampoline, 0);
    }
    for (;;) {
        bool progressed = false, live = false;
        for (int i = 0; i < e.nthreads; ++i) {
            const int t = pick(i, e.nthreads);
            if (e.fib[t].st != RUNNABLE) continue;
            e.cur = t;
            swapcontext(&e.sched, &e.fib[t].ctx);
            progressed = true;
        }
        for (int t = 0; t < e.nthreads; ++t) live |= e.fib[t].st != DONE;
        if (!live) return;
        release_waves(e);
        bool runnable = false;
        for (int t = 0; t < e.nthreads; ++t) runnable |= e.fib[t].st == RUNNABLE;
        if (!runnable && !release_block(e) && !progressed) {
            std::fprintf(stderr, "emu: deadlock in block (%u,%u,%u): threads wait at different synchronisation points\n", e.bid.x, e.bid.y, e.bid.z);
            std::abort();
        }
    }
}

void run_grid(dim3 grid, dim3 block, size_t lds, std::function<void()> body) {
    Engine& e = eng();
    e.grid = grid; e.block = block;
    e.nthreads = (int)(block.x * block.y * block.z);
    if ((int)e.fib.size() < e.nthreads) {
        e.fib.resize(e.nthreads);
        e.stacks.resize((size_t)e.nthreads * STACK);
        e.snap.resize((size_t)e.nthreads * 2);
        e.snap_pred.resize(e.nthreads);
    }
    constexpr size_t GUARD = 256;                         // canary behind the dynamic LDS: a kernel that writes past its allocation trips it
    e.dyn.assign(lds + GUARD, 0);
    std::memset(e.dyn.data() + lds, 0xA5, GUARD);
    e.body = std::move(body);
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                e.bid = dim3(x, y, z);
                run_block(e);
            }
    for (size_t i = 0; i < GUARD; ++i)
        if ((unsigned char)e.dyn[lds + i] != 0xA5) {
            std::fprintf(stderr, "emu: a kernel wrote %zu bytes past its %zu bytes of dynamic LDS\n", i + 1, lds);
            std::abort();
        }
    e.cur = -1;
}

}  // namespace shim

extern "C" void rcmvs_emu_set_order(int order) { shim::g_order = order; }
