// hip_emu.cpp — TEST-ONLY fiber scheduler of the HIP emulator (see hip_emu.h).
#include "hip_emu.h"
#include <chrono>

namespace hipemu {

thread_local Block* g_blk = nullptr;
static const size_t STACK = 256 * 1024;

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// Fiber switch. swapcontext() saves and restores the signal mask with two system calls per switch, and a collective of a 1024-thread
// workgroup is thousands of switches: the suite spent more time in the kernel than in the emulated kernels. On x86-64 the switch is
// done by hand instead (callee-saved registers + stack pointer; everything runs on one OS thread, so nothing else needs saving).
#if defined(__x86_64__)
#define HIPEMU_FAST_SWITCH 1
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n"
    ".globl hipemu_switch\n"
    ".type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size hipemu_switch,.-hipemu_switch\n");
static inline void to_sched(Block* b, Lane& l) { hipemu_switch(&l.sp, b->sched_sp); }
static inline void to_lane(Block* b, Lane& l) { hipemu_switch(&b->sched_sp, l.sp); }
#else
static inline void to_sched(Block* b, Lane& l) { swapcontext(&l.ctx, &b->sched); }
static inline void to_lane(Block* b, Lane& l) { swapcontext(&b->sched, &l.ctx); }
#endif

void yield() {
    Block* b = g_blk;
    to_sched(b, b->lanes[b->cur]);
}

static void rendezvous(Rendezvous& rv, int& alive) {
    int gen = rv.gen;
    if (++rv.count >= alive) { rv.count = 0; rv.gen++; }
    else { while (rv.gen == gen) yield(); }
}

void wave_barrier() { Block* b = g_blk; unsigned w = b->lanes[b->cur].tid >> 6; rendezvous(b->wave_rv[w], b->wave_alive[w]); }
void row_barrier() { Block* b = g_blk; unsigned r = b->lanes[b->cur].tid >> 4; rendezvous(b->row_rv[r], b->row_alive[r]); }
void block_barrier() { Block* b = g_blk; rendezvous(b->block_rv, b->block_alive); }

static void trampoline() {
    Block* b = g_blk;
    Lane& l = b->lanes[b->cur];
    b->body();
    l.done = true;
    unsigned w = l.tid >> 6;
    b->wave_alive[w]--; b->block_alive--;
    { const unsigned r = l.tid >> 4; b->row_alive[r]--; if (b->row_alive[r] > 0 && b->row_rv[r].count >= b->row_alive[r]) { b->row_rv[r].count = 0; b->row_rv[r].gen++; } }
    // a lane leaving may complete a rendezvous the others are waiting on
    if (b->wave_alive[w] > 0 && b->wave_rv[w].count >= b->wave_alive[w]) { b->wave_rv[w].count = 0; b->wave_rv[w].gen++; }
    if (b->block_alive > 0 && b->block_rv.count >= b->block_alive) { b->block_rv.count = 0; b->block_rv.gen++; }
    to_sched(b, l);
    abort();                                    // a finished fiber is never resumed
}

static void run_block(Block& b, std::vector<char*>& stacks) {
    g_blk = &b;
    unsigned n = b.nthreads;
    unsigned nw = (n + 63) / 64;
    b.slots.assign(n, 0); b.slots2.assign(n, 0);
    b.wave_rv.assign(nw, Rendezvous()); b.block_rv = Rendezvous();
    b.wave_alive.assign(nw, 0);
    b.row_rv.assign((n + 15) / 16, Rendezvous()); b.row_alive.assign((n + 15) / 16, 0);
    for (unsigned t = 0; t < n; ++t) b.row_alive[t >> 4]++;
    for (unsigned t = 0; t < n; ++t) b.wave_alive[t >> 6]++;
    b.block_alive = (int)n;
    b.lanes.resize(n);
    for (unsigned t = 0; t < n; ++t) {
        Lane& l = b.lanes[t];
        l.blk = &b; l.tid = t; l.done = false;
#ifdef HIPEMU_FAST_SWITCH
        {   // initial frame: six zeroed callee-saved registers, then the entry point as the return address (16-byte aligned slot, so that
            // the entry sees the stack as after a call)
            uintptr_t top = ((uintptr_t)stacks[t] + STACK) & ~(uintptr_t)15;
            void** A = (void**)(top - 32);
            A[0] = (void*)trampoline; A[1] = nullptr;
            for (int i = 1; i <= 6; ++i) A[-i] = nullptr;
            l.sp = (void*)(A - 6);
        }
#else
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = stacks[t]; l.ctx.uc_stack.ss_size = STACK; l.ctx.uc_link = &b.sched;
        makecontext(&l.ctx, (void (*)())trampoline, 0);
#endif
    }
    unsigned live = n;
    uint64_t spins = 0;
    while (live) {
        live = 0;
        for (unsigned t = 0; t < n; ++t) {
            if (b.lanes[t].done) continue;
            b.cur = t;
            to_lane(&b, b.lanes[t]);
            if (!b.lanes[t].done) ++live;
        }
        if (++spins > (1ull << 34)) { fprintf(stderr, "hip_emu: deadlock suspected (non-uniform collective?)\n"); abort(); }
    }
    g_blk = nullptr;
}

void run_grid(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    unsigned nthreads = block.x * block.y * block.z;
    if (!nblocks || !nthreads) return;
    unsigned nos = std::thread::hardware_concurrency(); if (nos == 0) nos = 4; if (nos > 8) nos = 8;
    if (nblocks < nos) nos = (unsigned)nblocks;
    std::atomic<size_t> next(0);
    auto worker = [&]() {
        std::vector<char*> stacks(nthreads);
        for (unsigned t = 0; t < nthreads; ++t) stacks[t] = (char*)malloc(STACK);
        std::vector<char> dyn(shmem + 64);
        Block b;
        b.grid = grid; b.block = block; b.nthreads = nthreads; b.body = body;
        while (true) {
            size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            b.bidx = dim3((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((size_t)grid.x * grid.y)));
            b.dynshared = (void*)(((uintptr_t)dyn.data() + 15) & ~(uintptr_t)15);
            run_block(b, stacks);
        }
        for (char* s : stacks) free(s);
    };
    if (nos <= 1) worker();
    else { std::vector<std::thread> th; for (unsigned t = 0; t < nos; ++t) th.emplace_back(worker); for (auto& t : th) t.join(); }
}

}  // namespace hipemu
