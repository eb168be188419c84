// Which compute units does a hipExtStreamCreateWithCUMask stream run on?  (Not part of the product.)  For a few masks: launch
// 8192 single-wave workgroups that spin ~20 us each, record HW_REG_XCC_ID and the CU / SE / SH fields of HW_REG_HW_ID, print how
// many distinct (XCC, SE, CU) the stream touched and the workgroups per XCC.   hipcc --offload-arch=gfx950 -o cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void where_kernel(unsigned* out, int spin) {
    unsigned xcc = 0, hw = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 32)" : "=s"(hw));
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

int main() {
    CK(hipSetDevice(0));
    const int G = 8192;
    unsigned* d; CK(hipMalloc(&d, G * 8));
    std::vector<unsigned> h(2 * G);
    struct M { const char* name; std::vector<uint32_t> w; };
    std::vector<M> masks;
    auto mk = [&](const char* n, auto pred) { M m{n, std::vector<uint32_t>(8, 0)}; for (int b = 0; b < 256; ++b) if (pred(b)) m.w[b / 32] |= 1u << (b % 32); masks.push_back(m); };
    mk("all 256 bits", [](int) { return true; });
    mk("bits 0..31", [](int b) { return b < 32; });
    mk("bits 0..63", [](int b) { return b < 64; });
    mk("every 8th bit (32)", [](int b) { return b % 8 == 0; });
    mk("every 4th bit (64)", [](int b) { return b % 4 == 0; });
    mk("bits 0..7", [](int b) { return b < 8; });
    mk("bits 128..159", [](int b) { return b >= 128 && b < 160; });
    for (const M& m : masks) {
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)m.w.size(), m.w.data());
        if (e != hipSuccess) { printf("%-22s: stream creation failed (%s)\n", m.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipMemsetAsync(d, 0xff, G * 8, st));
        CK(hipEventRecord(e0, st));
        where_kernel<<<G, 64, 0, st>>>(d, 40000);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost));
        int per[16] = {0};
        std::set<unsigned> cus;
        for (int b = 0; b < G; ++b) {
            const unsigned xcc = h[2 * b] & 15, hw = h[2 * b + 1];
            per[xcc]++;
            const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;      // HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
            cus.insert(xcc << 16 | se << 8 | sh << 4 | cu);
        }
        printf("%-22s: %6.2f ms, %3zu distinct (XCC, SE, SH, CU); workgroups per XCC:", m.name, ms, cus.size());
        for (int x = 0; x < 8; ++x) printf(" %4d", per[x]);
        printf("\n");
        CK(hipStreamDestroy(st));
    }
    return 0;
}
