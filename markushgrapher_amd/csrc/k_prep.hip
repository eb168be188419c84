// Page preprocessing on the device ("next" row f-3): the reference resizes each stored 1024x1024 page
// to the model's 512x512 input with Pillow's LANCZOS filter on the host (ref: markushgrapher/core/datasets/
// mdu_dataset.py:118) and the image processor rescales by 1/255 and normalises with mean = std = 0.5
// (ref: core/common/begin.py:105-109).  This is the same arithmetic, bit for bit: Pillow's 8-bit resampler
// (libImaging/Resample.c) works in 22-bit fixed point — integer coefficient tables (built on the host exactly as
// precompute_coeffs / normalize_coeffs_8bpc do), horizontal pass to u8, vertical pass to u8 — followed by
// float32(double(u8) * (1/255)), (x - 0.5) / 0.5 in float32, CHW.  HBM-bound streaming kernels.
#include "mg_kernels.h"

#include <math.h>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace mg {

constexpr int PREP_BITS = 32 - 8 - 2;

namespace {
struct CoefTable { std::vector<int> data; int ksize; };   // [out][2 + ksize]: xmin, count, coefficients

double lanczos3(double x) {
    if (-3.0 <= x && x < 3.0) {
        auto sinc = [](double v) { if (v == 0.0) return 1.0; v = v * M_PI; return sin(v) / v; };
        return sinc(x) * sinc(x / 3.0);
    }
    return 0.0;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full source range
const CoefTable& coef_table(int in_size, int out_size) {
    static std::map<std::pair<int, int>, CoefTable> cache;    // host tables persist: async uploads may still read them
    static std::mutex mu;                                     // callers on several host threads (one execution context each)
    std::lock_guard<std::mutex> lock(mu);                     // map nodes keep their address across later inserts
    auto key = std::make_pair(in_size, out_size);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    CoefTable t;
    const double scale = (double)in_size / out_size;
    const double fscale = scale < 1.0 ? 1.0 : scale;
    const double support = 3.0 * fscale;
    t.ksize = (int)ceil(support) * 2 + 1;
    t.data.assign((size_t)out_size * (2 + t.ksize), 0);
    const double ss = 1.0 / fscale;
    std::vector<double> w(t.ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int lo = (int)(center - support + 0.5);
        if (lo < 0) lo = 0;
        int hi = (int)(center + support + 0.5);
        if (hi > in_size) hi = in_size;
        const int n = hi - lo;
        double ww = 0.0;
        for (int x = 0; x < n; ++x) { w[x] = lanczos3((x + lo - center + 0.5) * ss); ww += w[x]; }
        int* row = &t.data[(size_t)xx * (2 + t.ksize)];
        row[0] = lo; row[1] = n;
        for (int x = 0; x < n; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            row[2 + x] = v < 0 ? (int)(-0.5 + v * (1 << PREP_BITS)) : (int)(0.5 + v * (1 << PREP_BITS));
        }
    }
    return cache.emplace(key, std::move(t)).first->second;
}
}  // namespace

MG_DEV int clip8(int v) { v >>= PREP_BITS; return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass: src [B][Hs][Ws][3] u8 -> tmp [B][Hs][Wo][3] u8
__global__ __launch_bounds__(256) void prep_horizontal_kernel(const uint8_t* src, uint8_t* tmp, const int* tab, int ksize, int rows, int Ws, int Wo) {
    const size_t total = (size_t)rows * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t y = i / Wo;
        const int xx = (int)(i - y * Wo);
        const int* t = tab + (size_t)xx * (2 + ksize);
        const int lo = t[0], n = t[1];
        const uint8_t* p = src + (y * Ws + lo) * 3;
        int s0 = 1 << (PREP_BITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < n; ++x) {
            const int k = t[2 + x];
            s0 += p[x * 3] * k; s1 += p[x * 3 + 1] * k; s2 += p[x * 3 + 2] * k;
        }
        uint8_t* o = tmp + i * 3;
        o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
    }
}

// vertical pass + rescale + normalise: tmp [B][Hs][Wo][3] u8 -> out [B][3][Ho][Wo] f32
__global__ __launch_bounds__(256) void prep_vertical_norm_kernel(const uint8_t* tmp, float* out, const int* tab, int ksize, int B, int Hs, int Ho, int Wo) {
    const size_t total = (size_t)B * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wo);
        const size_t r = i / Wo;
        const int yy = (int)(r % Ho), b = (int)(r / Ho);
        const int* t = tab + (size_t)yy * (2 + ksize);
        const int lo = t[0], n = t[1];
        const uint8_t* p = tmp + (((size_t)b * Hs + lo) * Wo + xx) * 3;
        int s[3] = {1 << (PREP_BITS - 1), 1 << (PREP_BITS - 1), 1 << (PREP_BITS - 1)};
        for (int y = 0; y < n; ++y) {
            const int k = t[2 + y];
            const uint8_t* q = p + (size_t)y * Wo * 3;
            s[0] += q[0] * k; s[1] += q[1] * k; s[2] += q[2] * k;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = (float)((double)clip8(s[c]) * (1.0 / 255.0));
            out[(((size_t)b * 3 + c) * Ho + yy) * Wo + xx] = (x - 0.5f) / 0.5f;
        }
    }
}

size_t preprocess_scratch_bytes(int B, int Hs, int Ws, int out_size) {
    const size_t tabs = ((size_t)coef_table(Ws, out_size).data.size() + coef_table(Hs, out_size).data.size()) * sizeof(int);
    return ((tabs + 255) / 256) * 256 + (size_t)B * Hs * out_size * 3;
}

void preprocess_pages(const uint8_t* pages, int B, int Hs, int Ws, int out_size, float* pixel_values, void* scratch, mgStream_t stream) {
    const CoefTable& th = coef_table(Ws, out_size);
    const CoefTable& tv = coef_table(Hs, out_size);
    int* d_th = (int*)scratch;
    int* d_tv = d_th + th.data.size();
    const size_t tabs = (th.data.size() + tv.data.size()) * sizeof(int);
    uint8_t* tmp = (uint8_t*)scratch + ((tabs + 255) / 256) * 256;
    mg_memcpy_async(d_th, th.data.data(), th.data.size() * sizeof(int), stream);
    mg_memcpy_async(d_tv, tv.data.data(), tv.data.size() * sizeof(int), stream);
    const size_t n1 = (size_t)B * Hs * out_size, n2 = (size_t)B * out_size * out_size;
    const int g1 = (int)((n1 + 255) / 256 > 8192 ? 8192 : (n1 + 255) / 256);
    const int g2 = (int)((n2 + 255) / 256 > 8192 ? 8192 : (n2 + 255) / 256);
    MG_LAUNCH(prep_horizontal_kernel, dim3(g1), dim3(256), 0, stream, pages, tmp, (const int*)d_th, th.ksize, B * Hs, Ws, out_size);
    MG_LAUNCH(prep_vertical_norm_kernel, dim3(g2), dim3(256), 0, stream, (const uint8_t*)tmp, pixel_values, (const int*)d_tv, tv.ksize, B, Hs,
              out_size, out_size);
}

}  // namespace mg
