// Fiber scheduler for the SIMT emulator (see hip_emu.h).  TEST INFRASTRUCTURE ONLY.
#include "hip_emu.h"
#include <vector>

namespace emu {
Ctx g;

extern "C" void emu_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
)");

static void to_sched() { emu_switch(&g.cur->sp, g.sched_sp); }

static void release_if_complete(Group& gr) {
    if (gr.count > 0 && gr.arrived == gr.count) { gr.arrived = 0; gr.gen++; }
}

static void fiber_entry() {
    g.fn();
    Fiber* f = g.cur;
    f->state = DONE;
    g.block.count--;
    g.waves[f->wave].count--;
    release_if_complete(g.block);
    release_if_complete(g.waves[f->wave]);
    to_sched();
    abort();
}

static void barrier(Group& gr, int st) {
    Fiber* f = g.cur;
    f->wait_gen = gr.gen;
    if (++gr.arrived == gr.count) { gr.arrived = 0; gr.gen++; return; }
    f->state = st;
    to_sched();
    f->state = RUNNABLE;
}
void block_barrier() { barrier(g.block, WAIT_BLOCK); }
void wave_barrier() { barrier(g.waves[g.cur->wave], WAIT_WAVE); }

static const size_t STACK = 192 * 1024;
static std::vector<char*> stacks;
static std::vector<Fiber> fibers;

void launch(std::function<void()> fn, dim3 grid, dim3 block) {
    int n = block.x * block.y * block.z;
    if (n > 1024 || n <= 0) { fprintf(stderr, "emu: bad block size %d\n", n); abort(); }
    while ((int)stacks.size() < n) stacks.push_back((char*)aligned_alloc(64, STACK));
    fibers.resize(n);
    g.fn = fn;
    g.gDim = grid; g.bDim = block;
    int nw = (n + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
        g.bIdx = dim3(bx, by, bz);
        g.block = Group{(unsigned)n, 0, 0};
        for (int w = 0; w < nw; w++) g.waves[w] = Group{(unsigned)((w + 1) * 64 <= n ? 64 : n - w * 64), 0, 0};
        memset(g.xpar, 0, sizeof(g.xpar));
        for (int i = 0; i < n; i++) {
            Fiber& f = fibers[i];
            f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
            f.lane = i & 63; f.wave = i >> 6; f.state = RUNNABLE; f.wait_gen = 0;
            f.stack = stacks[i];
            uintptr_t top = ((uintptr_t)(f.stack + STACK)) & ~(uintptr_t)15;
            void** sp = (void**)(top - 64);
            for (int k = 0; k < 6; k++) sp[k] = nullptr;
            sp[6] = (void*)&fiber_entry;
            sp[7] = nullptr;
            f.sp = sp;
        }
        g.fibers = fibers.data(); g.nfibers = n;
        int ndone = 0;
        while (ndone < n) {
            bool progress = false;
            for (int w = 0; w < nw; w++) {
                bool again = true;
                while (again) {
                    again = false;
                    int lo = w * 64, hi = lo + 64 < n ? lo + 64 : n;
                    for (int i = lo; i < hi; i++) {
                        Fiber& f = fibers[i];
                        if (f.state == DONE) continue;
                        if (f.state == WAIT_BLOCK && g.block.gen == f.wait_gen) continue;
                        if (f.state == WAIT_WAVE && g.waves[w].gen == f.wait_gen) continue;
                        g.cur = &f;
                        emu_switch(&g.sched_sp, f.sp);
                        if (f.state == DONE) ndone++;
                        again = true; progress = true;
                    }
                }
            }
            if (!progress && ndone < n) {
                fprintf(stderr, "emu: deadlock in block (%u,%u,%u): %d/%d done (divergent barrier?)\n", bx, by, bz, ndone, n);
                abort();
            }
        }
    }
}
}  // namespace emu
