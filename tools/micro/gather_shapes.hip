// Measurement aid (not part of the library): L2-resident gather rate per CU by ACCESS SHAPE of one wave
// instruction — how many lanes share a 128-byte line and how wide each lane's load is — with and without
// bf16 MFMAs issued beside the loads.  Shapes:
//   0: 8 lanes x 16 B per line, 8 lines per instruction (the gather legs' shape, gather_ceiling.hip)
//   1: 32 lanes x 4 B per line, 2 lines per instruction (MFMA B-fragment shape: lane = column)
//   2: 16 lanes x 8 B per line, 4 lines per instruction
//   3: 2 lanes x 16 B per line, 32 lines per instruction, 4 instructions complete the lines
//   4: 4 lanes x 16 B = 64-byte half lines, 16 per instruction
//   hipcc -O3 --offload-arch=gfx950 tools/micro/gather_shapes.hip -o build/gather_shapes && build/gather_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned nxt(unsigned& h, unsigned rows) {
    h = h * 1664525u + 1013904223u;
    return (h >> 8) % rows;
}

template <int SHAPE, int MFMA>
__global__ __launch_bounds__(256) void k(const char* __restrict__ base, int rows, int iters, float* out) {
    const int lane = threadIdx.x & 63;
    const int slice = blockIdx.x & 7;
    const char* s = base + size_t(slice) * rows * 128;
    constexpr int LPL = SHAPE == 0 ? 8 : SHAPE == 1 ? 32 : SHAPE == 2 ? 16 : SHAPE == 3 ? 2 : 4;   // lanes per line
    unsigned h = (blockIdx.x * 256u + threadIdx.x / unsigned(LPL)) * 2654435761u + 12345u;
    const int q = lane % LPL;
    float acc = 0.f;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)(float(lane)); }
    for (int it = 0; it < iters; ++it) {
        if constexpr (SHAPE == 0 || SHAPE == 4) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(s + size_t(nxt(h, rows)) * 128 + (SHAPE == 4 ? (it & 1) * 64 : 0) + q * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        } else if constexpr (SHAPE == 1) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const float*>(s + size_t(nxt(h, rows)) * 128 + q * 4);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += v[j];
        } else if constexpr (SHAPE == 2) {
            float2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float2*>(s + size_t(nxt(h, rows)) * 128 + q * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y;
        } else {
            float4 v[4];
            const size_t r = size_t(nxt(h, rows)) * 128;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(s + r + j * 32 + q * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
        }
        if constexpr (MFMA > 0) {
#pragma unroll
            for (int m = 0; m < MFMA; ++m) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        }
    }
    float t = acc;
    for (int i = 0; i < 16; ++i) t += c[i];
    if (t == 12345.678f) out[0] = t;
}

template <int SHAPE, int MFMA>
void run(const char* buf, int rows, int cus, double clk, float* out, int wgs_per_cu) {
    const int grid = cus * wgs_per_cu, iters = 1024;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<SHAPE, MFMA>), dim3(grid), dim3(256), 0, 0, buf, rows, iters, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<SHAPE, MFMA>), dim3(grid), dim3(256), 0, 0, buf, rows, iters, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= 3;
    const double total = double(grid) * 256 * 64.0 * iters;      // every shape moves 64 B per lane and iteration
    const double tbs = total / (ms * 1e-3) / 1e12;
    const double mf = double(grid) * 4 * iters * MFMA * 32768.0 / (ms * 1e-3) / 1e12;
    printf("shape %d  mfma/iter %d  slice %5.2f MiB  waves/SIMD %d: %6.2f TB/s = %4.1f B/clk/CU   %7.1f TFLOP/s\n", SHAPE, MFMA,
           rows * 128.0 / (1 << 20), wgs_per_cu, tbs, tbs * 1e12 / cus / clk, mf);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    printf("%s: %d CUs, %.2f GHz\n", prop.name, cus, clk / 1e9);
    float* out;
    CHECK(hipMalloc(&out, 64));
    for (int rows : {8192, 32768}) {
        char* buf;
        const size_t bytes = size_t(8) * rows * 128;
        CHECK(hipMalloc(&buf, bytes));
        CHECK(hipMemset(buf, 0, bytes));
        for (int w : {2, 4, 8}) {
            run<0, 0>(buf, rows, cus, clk, out, w);
            run<1, 0>(buf, rows, cus, clk, out, w);
            run<2, 0>(buf, rows, cus, clk, out, w);
            run<3, 0>(buf, rows, cus, clk, out, w);
            run<4, 0>(buf, rows, cus, clk, out, w);
        }
        for (int w : {2, 4}) {
            run<0, 1>(buf, rows, cus, clk, out, w);
            run<0, 2>(buf, rows, cus, clk, out, w);
            run<0, 4>(buf, rows, cus, clk, out, w);
            run<1, 1>(buf, rows, cus, clk, out, w);
            run<1, 2>(buf, rows, cus, clk, out, w);
            run<1, 4>(buf, rows, cus, clk, out, w);
            run<1, 8>(buf, rows, cus, clk, out, w);
        }
        CHECK(hipFree(buf));
    }
    return 0;
}
