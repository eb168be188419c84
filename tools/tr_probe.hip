// Probe (gfx950): semantics of ds_read_b64_tr_b16 and the operand layout of v_mfma_f32_16x16x16_bf16, checked against the
// formulas tools/simt_emu implements.  Build: hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o gpurun_out/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void tr_kernel(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 4];
    const int lane = threadIdx.x;
    // lane l stores 4 elements (l, 0..3) at its own 8-byte slot
    for (int j = 0; j < 4; ++j) lds[lane * 4 + j] = (uint16_t)(lane * 4 + j);
    __syncthreads();
    s16x4 r;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t*)(lds + lane * 4);
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)r[j];
}

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

// D = A (16 x 16) * B (16 x 16): A[i][k] in lane i + 16*(k/4) elem k%4; B[k][j] in lane j + 16*(k/4) elem k%4; D[i][j] in lane j + 16*(i/4) reg i%4
__global__ void mfma_kernel(const uint16_t* A, const uint16_t* Bm, float* D) {
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    s16x4 a, b;
    for (int e = 0; e < 4; ++e) { a[e] = (short)A[i * 16 + 4 * g + e]; b[e] = (short)Bm[(4 * g + e) * 16 + i]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}

int main() {
    uint16_t* d_out;
    hipMalloc(&d_out, 256 * 2);
    tr_kernel<<<1, 64>>>(d_out);
    std::vector<uint16_t> h(256);
    hipMemcpy(h.data(), d_out, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int base = (l / 16) * 16, i = l % 16;
        for (int j = 0; j < 4; ++j) {
            const int want = (base + j * 4 + i / 4) * 4 + (i % 4);      // element (i % 4) of lane base + 4j + i/4
            if (h[l * 4 + j] != want) ++bad;
        }
    }
    printf("tr_b16: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    for (int l = 0; l < 20; ++l) printf("  lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);

    std::vector<uint16_t> A(256), Bm(256);
    std::vector<float> Af(256), Bf(256), Dh(256);
    for (int x = 0; x < 256; ++x) {
        Af[x] = (float)((x * 7 + 3) % 11) - 5.f; Bf[x] = (float)((x * 5 + 1) % 13) - 6.f;
        A[x] = f2bf(Af[x]); Bm[x] = f2bf(Bf[x]);
    }
    uint16_t *dA, *dB; float* dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bm.data(), 512, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(Dh.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad2 = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += Af[i * 16 + k] * Bf[k * 16 + j];
        if (s != Dh[i * 16 + j]) ++bad2;
    }
    printf("mfma_16x16x16_bf16: %s (%d mismatches)\n", bad2 ? "FAIL" : "PASS", bad2);
    return (bad || bad2) ? 1 : 0;
}
