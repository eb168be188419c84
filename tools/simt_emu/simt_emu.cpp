// TEST INFRASTRUCTURE ONLY — see simt_emu.h.
#include "simt_emu.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

// x86-64 SysV stack switch: save callee-saved registers + rsp, load the other stack.
extern "C" void emu_switch(void** save_sp, void* next_sp);
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
.size emu_switch,.-emu_switch
)");

namespace emu {
Ctx* cur = nullptr;

namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    Ctx ctx;
};
struct Wave {
    int arrived = 0, gen = 0, nlanes = 64;
    alignas(16) unsigned char slot[64][32];
};
struct BlockState {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int bar_arrived = 0, bar_gen = 0, nthreads = 0, live = 0;
    std::vector<char> smem;
    const std::function<void()>* body = nullptr;
    void* sched_sp = nullptr;
    int cur_idx = -1;
    long idle_switches = 0;
} B;

void yield() {
    Fiber& f = B.fibers[B.cur_idx];
    emu_switch(&f.sp, B.sched_sp);
}

void fiber_entry() {
    (*B.body)();
    Fiber& f = B.fibers[B.cur_idx];
    f.done = true;
    B.live--;
    emu_switch(&f.sp, B.sched_sp);
    abort();
}

void progress() { B.idle_switches = 0; }

void wave_sync() {
    Wave& w = B.waves[cur->lin_tid >> 6];
    int g = w.gen;
    if (++w.arrived == w.nlanes) {
        w.arrived = 0;
        w.gen++;
        progress();
    } else {
        while (w.gen == g) yield();
    }
}
}  // namespace

void syncthreads() {
    int g = B.bar_gen;
    if (++B.bar_arrived == B.nthreads) {
        B.bar_arrived = 0;
        B.bar_gen++;
        progress();
    } else {
        while (B.bar_gen == g) yield();
    }
}

char* smem() { return B.smem.data(); }

uint32_t shfl_u32(uint32_t v, int src_lane) {
    Wave& w = B.waves[cur->lin_tid >> 6];
    int lane = cur->lin_tid & 63;
    memcpy(w.slot[lane], &v, 4);
    wave_sync();
    uint32_t r;
    int s = src_lane & 63;
    if (s >= w.nlanes) s = lane;
    memcpy(&r, w.slot[s], 4);
    wave_sync();
    return r;
}

static inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// v_mfma_f32_32x32x16_bf16: A[i][k] in lane i+32*(k/8) elem k%8; B[k][j] in lane j+32*(k/8) elem k%8;
// D[i][j] in lane j+32*((i/4)%2), reg (i%4)+4*(i/8)   (cdna_hip_programming.md §3 "Fragment layout").
void mfma_32x32x16_bf16(const uint16_t* a8, const uint16_t* b8, float* c16) {
    Wave& w = B.waves[cur->lin_tid >> 6];
    int lane = cur->lin_tid & 63;
    if (w.nlanes != 64) { fprintf(stderr, "emu: mfma in a partial wave\n"); abort(); }
    memcpy(w.slot[lane], a8, 16);
    memcpy(w.slot[lane] + 16, b8, 16);
    wave_sync();
    int j = lane & 31, half = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        float s = c16[r];
        for (int k = 0; k < 16; ++k) {
            const uint16_t* pa = (const uint16_t*)(w.slot[i + 32 * (k >> 3)]);
            const uint16_t* pb = (const uint16_t*)(w.slot[j + 32 * (k >> 3)] + 16);
            s += bf2f(pa[k & 7]) * bf2f(pb[k & 7]);
        }
        c16[r] = s;
    }
    wave_sync();
}

// v_mfma_f32_16x16x32_bf16: A[i][k] in lane i+16*(k/8) elem k%8; B[k][j] in lane j+16*(k/8) elem k%8;
// D[i][j] in lane j+16*(i/4), reg i%4   (cdna_hip_programming.md §3 "Fragment layout").
void mfma_16x16x32_bf16(const uint16_t* a8, const uint16_t* b8, float* c4) {
    Wave& w = B.waves[cur->lin_tid >> 6];
    int lane = cur->lin_tid & 63;
    if (w.nlanes != 64) { fprintf(stderr, "emu: mfma in a partial wave\n"); abort(); }
    memcpy(w.slot[lane], a8, 16);
    memcpy(w.slot[lane] + 16, b8, 16);
    wave_sync();
    int j = lane & 15, q = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * q + r;
        float s = c4[r];
        for (int k = 0; k < 32; ++k) {
            const uint16_t* pa = (const uint16_t*)(w.slot[i + 16 * (k >> 3)]);
            const uint16_t* pb = (const uint16_t*)(w.slot[j + 16 * (k >> 3)] + 16);
            s += bf2f(pa[k & 7]) * bf2f(pb[k & 7]);
        }
        c4[r] = s;
    }
    wave_sync();
}

// v_mfma_f32_16x16x16_bf16: A[i][k] in lane i+16*(k/4) elem k%4; B[k][j] in lane j+16*(k/4) elem k%4;
// D[i][j] in lane j+16*(i/4), reg i%4   (checked on the hardware by tools/tr_probe.hip).
void mfma_16x16x16_bf16(const uint16_t* a4, const uint16_t* b4, float* c4) {
    Wave& w = B.waves[cur->lin_tid >> 6];
    int lane = cur->lin_tid & 63;
    if (w.nlanes != 64) { fprintf(stderr, "emu: mfma in a partial wave\n"); abort(); }
    memcpy(w.slot[lane], a4, 8);
    memcpy(w.slot[lane] + 16, b4, 8);
    wave_sync();
    int j = lane & 15, q = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * q + r;
        float s = c4[r];
        for (int k = 0; k < 16; ++k) {
            const uint16_t* pa = (const uint16_t*)(w.slot[i + 16 * (k >> 2)]);
            const uint16_t* pb = (const uint16_t*)(w.slot[j + 16 * (k >> 2)] + 16);
            s += bf2f(pa[k & 3]) * bf2f(pb[k & 3]);
        }
        c4[r] = s;
    }
    wave_sync();
}

// ds_read_b64_tr_b16: per 16 lanes a [4][16] block of 16-bit elements, lane 4a + b supplying row a, columns 4b..4b+3; lane i
// receives column i, rows 0..3   (checked on the hardware by tools/tr_probe.hip).
void ds_read_tr16_b64(const void* lds_lane, uint16_t* out4) {
    Wave& w = B.waves[cur->lin_tid >> 6];
    int lane = cur->lin_tid & 63;
    if (w.nlanes != 64) { fprintf(stderr, "emu: transposed LDS read in a partial wave\n"); abort(); }
    if (((uintptr_t)lds_lane) & 7) { fprintf(stderr, "emu: ds_read_b64_tr_b16 address not 8-byte aligned\n"); abort(); }
    memcpy(w.slot[lane], lds_lane, 8);
    wave_sync();
    const int base = lane & ~15, i = lane & 15;
    for (int j = 0; j < 4; ++j) out4[j] = ((const uint16_t*)w.slot[base + 4 * j + (i >> 2)])[i & 3];
    wave_sync();
}

// global_load_lds_dwordx4: LDS destination = wave-uniform base + lane*16 (cdna_hip_programming.md §5 caveat)
void glds16(const void* gsrc_lane, void* lds_wave_base) {
    int lane = cur->lin_tid & 63;
    memcpy((char*)lds_wave_base + lane * 16, gsrc_lane, 16);
}

void launch(Dim3 grid, Dim3 block, size_t shmem, const std::function<void()>& body) {
    int nthreads = block.x * block.y * block.z;
    if (nthreads <= 0 || nthreads > 1024) { fprintf(stderr, "emu: bad block size %d\n", nthreads); abort(); }
    static std::vector<char*> stacks;
    while ((int)stacks.size() < nthreads) stacks.push_back((char*)aligned_alloc(64, kStack));
    Ctx* saved = cur;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        B.fibers.assign(nthreads, Fiber());
        B.waves.assign((nthreads + 63) / 64, Wave());
        for (size_t wv = 0; wv < B.waves.size(); ++wv)
            B.waves[wv].nlanes = std::min(64, nthreads - (int)wv * 64);
        B.bar_arrived = 0; B.bar_gen = 0; B.nthreads = nthreads; B.live = nthreads;
        B.smem.assign(shmem + 64, 0);
        B.body = &body;
        B.idle_switches = 0;
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = B.fibers[t];
            f.stack = stacks[t];
            f.ctx.tid = Dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.ctx.bid = Dim3(bx, by, bz);
            f.ctx.bdim = block;
            f.ctx.gdim = grid;
            f.ctx.lin_tid = t;
            uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
            void** sp = (void**)top;
            *--sp = nullptr;                 // fake return address of the entry function
            *--sp = (void*)&fiber_entry;     // popped by `ret` in emu_switch
            for (int i = 0; i < 6; ++i) *--sp = nullptr;
            f.sp = sp;
        }
        int idx = 0;
        while (B.live > 0) {
            Fiber& f = B.fibers[idx];
            if (!f.done) {
                B.cur_idx = idx;
                cur = &f.ctx;
                B.idle_switches++;
                emu_switch(&B.sched_sp, f.sp);
                if (B.idle_switches > 64L * 1024 * 1024) {
                    fprintf(stderr, "emu: deadlock (a thread exited or diverged before a collective?)\n");
                    abort();
                }
            }
            idx = (idx + 1) % nthreads;
        }
    }
    cur = saved;
}
}  // namespace emu
