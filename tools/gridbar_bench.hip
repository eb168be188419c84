// Feasibility microbenchmark (not part of the product): cost of hand-rolled grid barriers on MI355X with one
// 1024-thread workgroup per CU, plus the "every block writes a slice, barrier, every block reads everything" exchange
// that a persistent decode-chain kernel would do between projections.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define SPIN_LIMIT (1u << 20)

struct Bar {
    unsigned* ctr;       // [0] global counter, [64 + 64*x] per-XCD counters, flags at [1024 + 64*b]
    int* err;
    int variant, sleep, fence;
};

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_wg(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int S>
__device__ __forceinline__ void nap() { __builtin_amdgcn_s_sleep(S); }

__device__ __forceinline__ void grid_barrier(const Bar& B, unsigned round, unsigned xcc, unsigned n_xcc_blocks) {
    const unsigned G = gridDim.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (B.fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        unsigned spins = 0;
        if (B.variant == 0) {                 // one counter, everyone polls it
            atomicAdd(B.ctr, 1u);
            while (ld_agent(B.ctr) < G * (round + 1)) {
                if (B.sleep == 1) nap<1>(); else if (B.sleep == 8) nap<8>(); else nap<32>();
                if (++spins > SPIN_LIMIT) { atomicExch(B.err, 1); break; }
            }
        } else if (B.variant == 1) {          // one counter, last arriver raises every block's own flag
            const unsigned old = atomicAdd(B.ctr, 1u);
            if (old == G * (round + 1) - 1) {
                for (unsigned b = 0; b < G; ++b) __hip_atomic_store(B.ctr + 1024 + 64 * b, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            while (ld_agent(B.ctr + 1024 + 64 * blockIdx.x) < round + 1) {
                nap<1>();
                if (++spins > SPIN_LIMIT) { atomicExch(B.err, 1); break; }
            }
        } else {                              // per-XCD counter in the XCD's own L2, then one arrival per XCD
            unsigned* lc = B.ctr + 64 + 64 * xcc;
            unsigned* lf = B.ctr + 64 + 64 * (8 + xcc);
            const unsigned old = __hip_atomic_fetch_add(lc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == n_xcc_blocks * (round + 1) - 1) {
                atomicAdd(B.ctr, 1u);
                while (ld_agent(B.ctr) < 8 * (round + 1)) {
                    nap<1>();
                    if (++spins > SPIN_LIMIT) { atomicExch(B.err, 1); break; }
                }
                __hip_atomic_store(lf, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                while (ld_wg(lf) < round + 1) {
                    nap<1>();
                    if (++spins > SPIN_LIMIT) { atomicExch(B.err, 1); break; }
                }
            }
        }
        if (B.fence == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (B.fence == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(1024) void bar_kernel(Bar B, unsigned* buf0, unsigned* buf1, int rounds, int mode, int words_per_block,
                                                    unsigned* sink, unsigned* xcc_out) {
    const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (tid == 0) xcc_out[b] = xcc;
    unsigned acc = 0;
    for (int r = 0; r < rounds; ++r) {
        unsigned* buf = (r & 1) ? buf1 : buf0;
        if (mode >= 1)
            for (int i = tid; i < words_per_block; i += 1024) buf[(size_t)b * words_per_block + i] = (unsigned)(r * 131 + b);
        grid_barrier(B, (unsigned)r, xcc, (unsigned)(G / 8));
        if (*(volatile int*)B.err) return;
        if (mode >= 1) {
            const int total = G * words_per_block;
            for (int i = tid * 4; i < total; i += 4096) {
                const uint4 v = *(const uint4*)(buf + i);
                const unsigned want = (unsigned)(r * 131 + i / words_per_block);
                if (v.x != want || v.w != want) atomicExch(B.err, 2);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int G = prop.multiProcessorCount;
    if (argc > 1) G = atoi(argv[1]);
    printf("CUs %d, grid %d\n", prop.multiProcessorCount, G);
    unsigned *ctr, *buf0, *buf1, *sink, *xcc; int* err;
    const int wpb_max = 1024, ctr_words = 1024 + 64 * 1024;
    CK(hipMalloc(&ctr, ctr_words * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&xcc, G * 4));
    CK(hipMalloc(&buf0, (size_t)G * wpb_max * 4)); CK(hipMalloc(&buf1, (size_t)G * wpb_max * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Cfg { int variant, sleep, mode, words, fence; };
    const Cfg cfgs[] = {{0, 1, 0, 0, 0}, {0, 1, 0, 0, 1}, {0, 1, 0, 0, 2}, {1, 1, 0, 0, 0}, {1, 1, 0, 0, 1}, {0, 1, 1, 256, 1}, {1, 1, 1, 256, 1}, {1, 1, 1, 64, 1}, {1, 1, 1, 1024, 1}, {1, 1, 1, 256, 0}};
    for (const Cfg& c : cfgs) {
        float best = 1e9f; int herr = 0;
        for (int rep = 0; rep < 3; ++rep) {
            const int rounds = 500;
            CK(hipMemsetAsync(ctr, 0, ctr_words * 4, st)); CK(hipMemsetAsync(err, 0, 4, st));
            Bar B{ctr, err, c.variant, c.sleep, c.fence};
            CK(hipEventRecord(e0, st));
            bar_kernel<<<G, 1024, 0, st>>>(B, buf0, buf1, rounds, c.mode, c.words ? c.words : 1, sink, xcc);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            if (ms * 1e3f / rounds < best) best = ms * 1e3f / rounds;
            if (herr) break;
        }
        printf("variant %d fence %d sleep %2d mode %d, %4d B written per block (%7d B read per block): %.3f us per round, err %d\n", c.variant, c.fence, c.sleep,
               c.mode, c.words * 4, c.words * 4 * G, best, herr);
    }
    std::vector<unsigned> hx(G);
    CK(hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost));
    int per[16] = {0}; bool rr = true;
    for (int b = 0; b < G; ++b) { per[hx[b] & 15]++; if (hx[b] != (unsigned)(b % 8)) rr = false; }
    printf("blocks per XCC:"); for (int x = 0; x < 8; ++x) printf(" %d", per[x]); printf("  round-robin mapping: %s\n", rr ? "yes" : "no");
    return 0;
}
