// TEST INFRASTRUCTURE ONLY -- runtime of the fiber emulator declared in hip_emu.h.
#include "hip_emu.h"

#include <vector>

EmuFiber* emu_cur = nullptr;
dim3 emu_blockIdx, emu_blockDim, emu_gridDim;
char* emu_dyn_smem = nullptr;
float emu_wave_xchg[16][64][8];

static void* emu_sched_sp = nullptr;
static const std::function<void()>* emu_body = nullptr;
static const size_t EMU_STACK = 256 * 1024;

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

static void emu_yield() { emu_switch(&emu_cur->sp, emu_sched_sp); }

static void emu_trampoline() {
    (*emu_body)();
    emu_cur->state = 3;
    emu_yield();
    abort();
}

void emu_block_barrier() { emu_cur->state = 1; emu_yield(); }
void emu_wave_sync() { emu_cur->state = 2; emu_yield(); }

void emu_launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    const int nthr = block.x * block.y * block.z;
    if (nthr > 1024 || nthr <= 0) { fprintf(stderr, "emu: bad block size %d\n", nthr); abort(); }
    std::vector<EmuFiber> fibers(nthr);
    std::vector<char> smem_buf(smem + 64);
    static std::vector<char*> stacks;
    while ((int)stacks.size() < nthr) stacks.push_back((char*)aligned_alloc(64, EMU_STACK));
    emu_gridDim = grid; emu_blockDim = block;
    emu_body = &body;
    emu_dyn_smem = (char*)(((uintptr_t)smem_buf.data() + 63) & ~(uintptr_t)63);
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        emu_blockIdx = dim3(bx, by, bz);
        for (int t = 0; t < nthr; ++t) {
            EmuFiber& f = fibers[t];
            f.lin = t;
            f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.stack = stacks[t];
            f.state = 0;
            // initial frame: 6 callee-saved slots, then the return address (16-byte aligned slot)
            uintptr_t top = ((uintptr_t)f.stack + EMU_STACK) & ~(uintptr_t)15;
            void** sp = (void**)(top - 16);          // slot of the return address: ==0 mod 16
            sp[0] = (void*)&emu_trampoline;
            sp -= 6;
            for (int i = 0; i < 6; ++i) sp[i] = nullptr;
            f.sp = (void*)sp;
        }
        int done = 0;
        while (done < nthr) {
            bool progressed = false;
            for (int t = 0; t < nthr; ++t) {
                EmuFiber& f = fibers[t];
                if (f.state != 0) continue;
                emu_cur = &f;
                emu_switch(&emu_sched_sp, f.sp);
                progressed = true;
                if (f.state == 3) ++done;
            }
            // release wave syncs: every not-done lane of the wave must be at state 2
            const int nw = (nthr + 63) / 64;
            for (int w = 0; w < nw; ++w) {
                int lo = w * 64, hi = lo + 64 < nthr ? lo + 64 : nthr;
                bool all = true, any = false;
                for (int t = lo; t < hi; ++t) {
                    if (fibers[t].state == 2) any = true;
                    else if (fibers[t].state != 3) all = false;
                }
                if (any && all) { for (int t = lo; t < hi; ++t) if (fibers[t].state == 2) fibers[t].state = 0; progressed = true; }
            }
            // release the block barrier: every not-done thread must be at state 1
            {
                bool all = true, any = false;
                for (int t = 0; t < nthr; ++t) {
                    if (fibers[t].state == 1) any = true;
                    else if (fibers[t].state != 3) all = false;
                }
                if (any && all) { for (int t = 0; t < nthr; ++t) if (fibers[t].state == 1) fibers[t].state = 0; progressed = true; }
            }
            if (!progressed) {
                fprintf(stderr, "emu: DEADLOCK in block (%u,%u,%u): divergent barrier / wave collective\n", bx, by, bz);
                for (int t = 0; t < nthr; ++t) if (fibers[t].state != 3) { fprintf(stderr, "  thread %d state %d\n", t, fibers[t].state); break; }
                abort();
            }
        }
    }
    emu_cur = nullptr;
}
