// Probe of the lane-exchange helpers in mg_device.h on the device: prints, per STEP, whether lane_xor<STEP>(lane) == lane ^ STEP
// and whether lane_halve<STEP> returns own-kept + partner's same part.   hipcc --offload-arch=gfx950 -I markushgrapher_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include "mg_device.h"
using namespace mg;
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    const float v = (float)lane;
    out[lane] = lane_xor<8>(v, lane);
    out[64 + lane] = lane_xor<16>(v, lane);
    out[128 + lane] = lane_xor<32>(v, lane);
    const float lo = 1000.f + lane, hi = 2000.f + lane;
    out[192 + lane] = lane_halve<8>(lo, hi, 1.f, 10000.f, lane);
    out[256 + lane] = lane_halve<16>(lo, hi, 1.f, 10000.f, lane);
    out[320 + lane] = lane_halve<32>(lo, hi, 1.f, 10000.f, lane);
}
int main() {
    float* d; hipMalloc(&d, 384 * 4);
    probe<<<1, 64>>>(d);
    float h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const int steps[3] = {8, 16, 32};
    for (int s = 0; s < 3; ++s) {
        int bad = 0, badh = 0;
        for (int l = 0; l < 64; ++l) {
            if (h[s * 64 + l] != (float)(l ^ steps[s])) ++bad;
            const int p = l ^ steps[s]; const bool up = l & steps[s];
            const float want = (up ? 2000.f + l : 1000.f + l) + 10000.f * (up ? 2000.f + p : 1000.f + p);
            if (h[192 + s * 64 + l] != want) ++badh;
        }
        printf("step %d: xor mismatches %d, halve mismatches %d\n", steps[s], bad, badh);
        if (bad) { for (int l = 0; l < 64; ++l) printf("%g ", h[s * 64 + l]); printf("\n"); }
        if (badh) { for (int l = 0; l < 64; ++l) printf("%g ", h[192 + s * 64 + l]); printf("\n"); }
    }
    return 0;
}
