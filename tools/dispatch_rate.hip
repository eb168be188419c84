// Probe: how many dependent kernel launches per second does the chip sustain on 1..6 streams (hardware queues) at once?
// Each stream replays a captured chain of `len` kernels (grid x 512 threads, each spinning `spin` cycles).
//   hipcc --offload-arch=gfx950 -O2 tools/dispatch_rate.hip -o /tmp/dispatch_rate && GPU_MAX_HW_QUEUES=8 /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void spin_kernel(int* out, long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0 && cycles < 0) out[0] = 1;
}

int main() {
    const int len = 2000;
    for (int grid : {1, 256, 512}) {
        for (long long spin : {0LL, 400LL, 1000LL}) {      // clock64 ticks at 100 MHz: 400 = 4 us, 1000 = 10 us
            for (int ns : {1, 2, 3, 4, 6}) {
                std::vector<hipStream_t> st(ns);
                std::vector<hipGraphExec_t> ex(ns);
                for (int i = 0; i < ns; ++i) {
                    hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
                    hipGraph_t g;
                    hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal);
                    for (int k = 0; k < len; ++k) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(512), 0, st[i], (int*)nullptr, spin);
                    hipStreamEndCapture(st[i], &g);
                    hipGraphInstantiate(&ex[i], g, nullptr, nullptr, 0);
                    hipGraphDestroy(g);
                }
                auto run = [&]() {
                    std::vector<std::thread> th;
                    for (int i = 0; i < ns; ++i) th.emplace_back([&, i]() { hipGraphLaunch(ex[i], st[i]); hipStreamSynchronize(st[i]); });
                    for (auto& t : th) t.join();
                };
                run();
                auto t0 = std::chrono::steady_clock::now();
                run();
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                printf("grid %3d spin %4lld ticks  streams %d: %.2f us per kernel per stream, %.0f k kernels/s total\n", grid, spin, ns, us / len,
                       1e3 * ns * len / us);
                for (int i = 0; i < ns; ++i) { hipGraphExecDestroy(ex[i]); hipStreamDestroy(st[i]); }
            }
        }
    }
    return 0;
}
