// TEST INFRASTRUCTURE ONLY — runtime of the lock-step host emulator (see hip_emu.h).
#include "hip_emu.h"

#include <mutex>

namespace emu {

thread_local Block* g_block = nullptr;
thread_local Fiber* g_cur = nullptr;
thread_local dim3 g_blockIdx, g_blockDim, g_gridDim;

static void fiber_main() {
    g_block->body();
    g_cur->done = true;
    for (;;) yield();
}

static void run_block(const std::function<void()>& body, dim3 bidx, dim3 grid, dim3 block, size_t dyn_smem_bytes,
                      std::vector<char*>& stacks) {
    Block blk;
    const int n = (int)(block.x * block.y * block.z);
    blk.nthreads = n;
    blk.body = body;
    blk.fibers.resize(n);
    blk.waves.resize((n + WAVE - 1) / WAVE);
    std::vector<char> smem(dyn_smem_bytes + 64);
    blk.dyn_smem = (char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
    g_block = &blk;
    g_blockIdx = bidx;
    g_blockDim = block;
    g_gridDim = grid;
    while ((int)stacks.size() < n) stacks.push_back((char*)aligned_alloc(64, STACK_BYTES));
    for (int t = 0; t < n; ++t) {
        Fiber& f = blk.fibers[t];
        f.stack = stacks[t];
        f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        f.lane = t % WAVE;
        f.wave = t / WAVE;
        uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;                 // fake return address of fiber_main (never used)
        *--sp = (void*)&fiber_main;      // popped by `ret` in dvs_emu_switch
        for (int i = 0; i < 6; ++i) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
        f.sp = sp;
    }
    int remaining = n;
    while (remaining > 0) {
        remaining = 0;
        for (int t = 0; t < n; ++t) {
            Fiber& f = blk.fibers[t];
            if (f.done) continue;
            g_cur = &f;
            dvs_emu_switch(&blk.sched_sp, f.sp);
            if (!f.done) ++remaining;
        }
    }
    g_block = nullptr;
    g_cur = nullptr;
}

void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t dyn_smem_bytes) {
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    int nthreads = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("DVS_EMU_THREADS")) nthreads = atoi(e);
    nthreads = std::max(1, std::min(nthreads, nblocks));
    std::atomic<int> next{0};
    auto worker = [&]() {
        std::vector<char*> stacks;
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= nblocks) break;
            dim3 bidx(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y));
            run_block(body, bidx, grid, block, dyn_smem_bytes, stacks);
        }
        for (char* s : stacks) free(s);
    };
    if (nthreads == 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < nthreads; ++i) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
}

}  // namespace emu

// x86-64 SysV context switch: save callee-saved registers on the current stack, swap stacks, restore.
asm(R"(
.text
.globl dvs_emu_switch
.type dvs_emu_switch,@function
dvs_emu_switch:
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
.size dvs_emu_switch,.-dvs_emu_switch
)");
