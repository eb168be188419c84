// Feasibility microbenchmark (not part of the product): a chain of dependent small kernels, (a) on one in-order stream
// (hardware barrier between launches) versus (b) alternating between two streams with a software dependency: every
// workgroup of kernel i releases (fence + atomic add) when done, kernel i+1 starts early, does its independent prologue
// (here: streams its "weights") and only then polls the counter before touching kernel i's output.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NB = 256, NT = 512, WBYTES_PER_BLOCK = 64 * 1024;

// one "projection": every block streams its private 64 KB of weights, then reads the whole 64 KB activation written by
// the previous kernel, reduces, and writes its 256 B slice of the next activation.
__global__ __launch_bounds__(NT) void step_kernel(const uint4* __restrict__ w, const float* __restrict__ xin, float* __restrict__ xout,
                                                  unsigned* flags, int idx, int soft, int* err) {
    const int tid = threadIdx.x, b = blockIdx.x;
    // independent prologue: weights
    const uint4* wb = w + (size_t)b * (WBYTES_PER_BLOCK / 16);
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WBYTES_PER_BLOCK / 16 / NT; ++i) {
        const uint4 v = wb[tid + i * NT];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (soft && idx > 0) {
        if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(flags + (idx - 1) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NB) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) { atomicExch(err, 1); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // dependent part: read all of xin (16384 floats = 64 KB)
    float s = 0.f;
    for (int i = tid; i < 16384; i += NT) s += xin[i];
    s += (float)((acc.x ^ acc.y ^ acc.z ^ acc.w) & 1);
    __shared__ float red[NT / 64];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid < 64) {
        float t = 0.f;
        for (int i = 0; i < NT / 64; ++i) t += red[i];
        xout[b * 64 + tid] = t * 1e-6f + (float)tid;
    }
    if (soft) {
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            atomicAdd(flags + idx * 64, 1u);
        }
    }
}

int main() {
    CK(hipSetDevice(0));
    const int CHAIN = 144;      // one decoder step's worth of launches
    uint4* w; float *x0, *x1; unsigned* flags; int* err;
    const size_t wbytes = (size_t)NB * WBYTES_PER_BLOCK;
    const int NCOPY = 24;
    CK(hipMalloc(&w, wbytes * NCOPY)); CK(hipMemset(w, 1, wbytes * NCOPY));
    CK(hipMalloc(&x0, 65536)); CK(hipMalloc(&x1, 65536)); CK(hipMemset(x0, 0, 65536)); CK(hipMemset(x1, 0, 65536));
    CK(hipMalloc(&flags, CHAIN * 256)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(flags, 0, CHAIN * 256, s[0]));
            CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
            CK(hipEventRecord(e0, s[0]));
            for (int i = 0; i < CHAIN; ++i) {
                hipStream_t st = mode ? s[i & 1] : s[0];
                const uint4* wi = (const uint4*)((const char*)w + (size_t)(i % NCOPY) * wbytes);
                step_kernel<<<NB, NT, 0, st>>>(wi, (i & 1) ? x1 : x0, (i & 1) ? x0 : x1, flags, i, mode, err);
            }
            CK(hipStreamSynchronize(s[1]));
            CK(hipEventRecord(e1, s[0]));
            CK(hipStreamSynchronize(s[0]));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("%s: %.2f us per kernel (chain of %d), err %d\n", mode ? "two streams + software dependency" : "one stream (hardware barrier)      ",
               best * 1e3 / CHAIN, CHAIN, herr);
    }
    return 0;
}
