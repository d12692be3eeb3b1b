// Prototype for the round-2 kernel (DESIGN.md §6 "Round-2 plan"): how fast can a CU
//  (a) stage contiguous 128 KiB tiles of X (8192 rows x 4 floats) from HBM/L2 into LDS, and
//  (b) gather 16-byte entries from that tile at random (one output row per lane, neighbour
//      ids as u16 in SELL order: 64 consecutive ids per wave-instruction) and accumulate?
// It measures rates only; the arithmetic is a stand-in.   build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int TILE_ROWS = 8192;            // 16 B each -> 128 KiB
constexpr int THREADS = 512;

template <bool STAGE, bool GATHER>
__global__ __launch_bounds__(THREADS) void proto(const float4* __restrict__ X, int tiles_total,
                                                 int tiles_per_block, const unsigned short* __restrict__ ids,
                                                 int gathers_per_lane, float4* out) {
    extern __shared__ float4 tile[];       // TILE_ROWS entries
    const int tid = threadIdx.x;
    float4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int t = 0; t < tiles_per_block; ++t) {
        const int tile_id = (blockIdx.x * tiles_per_block + t) % tiles_total;
        if (STAGE) {
            const float4* src = X + size_t(tile_id) * TILE_ROWS;
#pragma unroll
            for (int i = 0; i < TILE_ROWS / THREADS; ++i) tile[i * THREADS + tid] = src[i * THREADS + tid];
        }
        __syncthreads();
        if (GATHER) {
            // SELL order: entry j of the 64 rows of a wave is 64 consecutive u16
            const unsigned short* my = ids + size_t(tid >> 6) * 64 * gathers_per_lane + (tid & 63);
            for (int j = 0; j < gathers_per_lane; j += 8) {
                unsigned short id[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) id[u] = my[(j + u) * 64];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 v = tile[id[u]];
                    acc[u & 3].x += v.x; acc[u & 3].y += v.y; acc[u & 3].z += v.z; acc[u & 3].w += v.w;
                }
            }
        }
        __syncthreads();
    }
    float4 r = acc[0];
    r.x += acc[1].x + acc[2].x + acc[3].x; r.y += acc[1].y + acc[2].y + acc[3].y;
    r.z += acc[1].z + acc[2].z + acc[3].z; r.w += acc[1].w + acc[2].w + acc[3].w;
    if (!STAGE && !GATHER) r = tile[tid];
    out[size_t(blockIdx.x) * THREADS + tid] = r;
}

template <bool S, bool G>
static float run(const float4* X, int tiles_total, int tpb, const unsigned short* ids, int gpl, float4* out, int blocks) {
    auto k = proto<S, G>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, TILE_ROWS * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(THREADS), TILE_ROWS * 16, 0, X, tiles_total, tpb, ids, gpl, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(THREADS), TILE_ROWS * 16, 0, X, tiles_total, tpb, ids, gpl, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 3;
}

int main() {
    const int tiles_total = 16384;                       // 2 GiB of tiles
    const int blocks = 256, tpb = 64, gpl = 96;          // 96 gathers per lane per tile
    float4* X; CK(hipMalloc(&X, size_t(tiles_total) * TILE_ROWS * 16));
    CK(hipMemset(X, 0, size_t(tiles_total) * TILE_ROWS * 16));
    std::vector<unsigned short> h(size_t(THREADS / 64) * 64 * gpl);
    srand(1);
    for (auto& v : h) v = (unsigned short)(rand() % TILE_ROWS);
    unsigned short* ids; CK(hipMalloc(&ids, h.size() * 2));
    CK(hipMemcpy(ids, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    float4* out; CK(hipMalloc(&out, size_t(blocks) * THREADS * 16));
    const double stage_bytes = double(blocks) * tpb * TILE_ROWS * 16;
    const double gather_bytes = double(blocks) * tpb * THREADS * gpl * 16;
    float ms;
    ms = run<true, false>(X, tiles_total, tpb, ids, gpl, out, blocks);
    printf("stage only : %.3f ms  %.2f TB/s into LDS (%.1f GB/s per CU)\n", ms, stage_bytes / ms / 1e9, stage_bytes / ms / 1e6 / 256);
    ms = run<false, true>(X, tiles_total, tpb, ids, gpl, out, blocks);
    printf("gather only: %.3f ms  %.2f TB/s of 16-B LDS gathers (%.1f B/clk/CU at 2.4 GHz)\n", ms, gather_bytes / ms / 1e9, gather_bytes / ms / 1e6 / 256 / 2.4);
    ms = run<true, true>(X, tiles_total, tpb, ids, gpl, out, blocks);
    printf("both       : %.3f ms  (%.2f TB/s gathers, %.2f TB/s staging)\n", ms, gather_bytes / ms / 1e9, stage_bytes / ms / 1e9);
    return 0;
}
