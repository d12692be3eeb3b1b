// What v_smfmac_f32_32x32x32_bf16 (gfx950: A 2:4-sparse, 8 stored bf16 + 2-bit positions per lane; B dense, 16 bf16 per lane)
// computes, found by probing: which K a stored A value of lane (m, h), slot j with position p multiplies, and which K element e
// of B's lane (n, h) holds.  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/smfmac_probe.hip -o smfmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <utility>
#ifndef ORDER_SWAP
#define ORDER_SWAP 0
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one launch per probe: A has 1.0 in slot j of the lanes with h == hA (position p in every 2-bit field of idx),
// B has 1.0 in element e of the lanes with h == hB; out[probe] = C[0][0] .. summed over the tile (non-zero: the Ks meet)
__global__ void probe(float* out) {
    const int l = threadIdx.x, h = l >> 5;
    int n = 0;
    for (int hA = 0; hA < 2; ++hA)
        for (int j = 0; j < 8; ++j)
            for (int p = 0; p < 4; ++p)
                for (int hB = 0; hB < 2; ++hB)
                    for (int e = 0; e < 16; ++e, ++n) {
                        bf16x8 a;
                        bf16x16 b;
                        for (int i = 0; i < 8; ++i) a[i] = (__bf16)((h == hA && i == j) ? 1.0f : 0.0f);
                        for (int i = 0; i < 16; ++i) b[i] = (__bf16)((h == hB && i == e) ? 1.0f : 0.0f);
                        f32x16 c;
                        for (int i = 0; i < 16; ++i) c[i] = 0.f;
                        const int idx = p * 0x55555555;
                        c = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b, c, idx, 0, 0);
                        float s = 0.f;
                        for (int i = 0; i < 16; ++i) s += c[i];
                        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
                        if (l == 0) out[n] = s;
                    }
}

// which bits of idx belong to slot j: A slot j of h == 0 lanes is 1.0, B all ones in element e only ... simpler: B = K + 1
// under the layout found above is checked by the full test below instead.

// full test under a hypothesis: A[m][K] 0/1 with at most two ones per group of four Ks, B random small integers
__global__ void full(const bf16x8* a, const int* idx, const bf16x16* b, float* c_out, int abid) {
    const int l = threadIdx.x;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    if (abid == 0) c = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[l], b[l], c, idx[l], 0, 0);
    else c = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[l], b[l], c, idx[l], 0, 1);
    for (int i = 0; i < 16; ++i) c_out[l * 16 + i] = c[i];
}

// throughput: 4 independent accumulators per wave, 4 waves per SIMD (1024 workgroups of 256), N instructions each
template <bool SPARSE>
__global__ void rate(float* out, int n_iter) {
    bf16x8 a;
    bf16x16 b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)1.0f;
    for (int i = 0; i < 16; ++i) b[i] = (__bf16)(float)(threadIdx.x & 3);
    f32x16 c[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) c[t][i] = 0.f;
    bf16x8 b8;
    for (int i = 0; i < 8; ++i) b8[i] = b[i];
    for (int it = 0; it < n_iter; ++it)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (SPARSE) c[t] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b, c[t], 0x4444, 0, 0);
            else c[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b8, c[t], 0, 0, 0);
        }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) s += c[t][0];
    if (s == 12345.f) out[0] = s;
}

static unsigned short bf(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }

int main() {
    float* d;
    const int N = 2 * 8 * 4 * 2 * 16;
    hipMalloc(&d, N * 4);
    probe<<<1, 64>>>(d);
    std::vector<float> o(N);
    hipMemcpy(o.data(), d, N * 4, hipMemcpyDeviceToHost);
    int n = 0;
    printf("A (h, slot j, position p) meets B (h, element e):\n");
    for (int hA = 0; hA < 2; ++hA)
        for (int j = 0; j < 8; ++j)
            for (int p = 0; p < 4; ++p) {
                printf("  A h=%d j=%d p=%d ->", hA, j, p);
                for (int hB = 0; hB < 2; ++hB)
                    for (int e = 0; e < 16; ++e, ++n)
                        if (o[n] != 0.f) printf(" B(h=%d,e=%d: %g)", hB, e, o[n]);
                printf("\n");
            }
    // hypothesis: B lane (n, h) element e -> K = 16 h + e; A lane (m, h) slot j = 2 g + i, position p -> K = 16 (g >> 1) + 8 h +
    //             4 (g & 1) + p (the pairing the probe prints), idx bits [2j+1 : 2j] (abid 0: low 16 bits); C as the dense 32x32 MFMA
    srand(1);
    std::vector<int> A(32 * 32, 0), B(32 * 32);
    for (int m = 0; m < 32; ++m)
        for (int g = 0; g < 8; ++g) {
            const int cnt = rand() % 3;
            int p0 = rand() % 4, p1 = rand() % 4;
            if (cnt >= 1) A[m * 32 + 4 * g + p0] = 1;
            if (cnt >= 2 && p1 != p0) A[m * 32 + 4 * g + p1] = 1;
        }
    for (int i = 0; i < 32 * 32; ++i) B[i] = rand() % 7 - 3;
    std::vector<unsigned short> av(64 * 8, 0), bv(64 * 16);
    std::vector<int> iv(64, 0);
    for (int l = 0; l < 64; ++l) {
        const int m = l & 31, h = l >> 5;
        for (int g = 0; g < 4; ++g) {
            // (found by the probe above) the lane's group g holds the logical Ks 16 (g >> 1) + 8 h + 4 (g & 1) + 0..3
            const int k0 = 16 * (g >> 1) + 8 * h + 4 * (g & 1);
            int slot = 2 * g, pos_used[2] = {0, 1}, k = 0;
            for (int p = 0; p < 4; ++p)
                if (A[m * 32 + k0 + p]) { av[l * 8 + slot + k] = bf(1.0f); pos_used[k] = p; ++k; }
            if (k == 1) pos_used[1] = (pos_used[0] + 1 + rand() % 3) & 3;     // (an unused slot: value 0, any other position)
            if (k == 2 && ORDER_SWAP && rand() % 2) {                          // (does the order of the two positions matter?)
                std::swap(pos_used[0], pos_used[1]);
            }
            iv[l] |= pos_used[0] << (2 * slot) | pos_used[1] << (2 * (slot + 1));
        }
        for (int e = 0; e < 16; ++e) bv[l * 16 + e] = bf((float)B[(16 * h + e) * 32 + m]);      // (n = l & 31)
    }
    void *da, *db, *di;
    float* dc;
    hipMalloc(&da, av.size() * 2); hipMalloc(&db, bv.size() * 2); hipMalloc(&di, 64 * 4); hipMalloc(&dc, 64 * 16 * 4);
    hipMemcpy(da, av.data(), av.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, bv.data(), bv.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(di, iv.data(), 64 * 4, hipMemcpyHostToDevice);
    for (int abid = 0; abid < 2; ++abid) {
        full<<<1, 64>>>((const bf16x8*)da, (const int*)di, (const bf16x16*)db, dc, abid);
        std::vector<float> c(64 * 16);
        hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
                float want = 0;
                for (int k = 0; k < 32; ++k) want += A[row * 32 + k] * B[k * 32 + col];
                if (c[l * 16 + i] != want) ++bad;
            }
        printf("hypothesis (abid %d): %d of 1024 elements differ\n", abid, bad);
    }
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int n_iter = 2000;
        for (int sp = 0; sp < 2; ++sp) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (sp) rate<true><<<1024, 256>>>(d, n_iter); else rate<false><<<1024, 256>>>(d, n_iter);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double n_inst = 1024.0 * 4 * n_iter * 4;                       // wave instructions
            const double flop = n_inst * 2.0 * 32 * 32 * (sp ? 32 : 16);          // (logical flop of the sparse form: K = 32)
            printf("%s: %.3f ms for %.0f wave instructions on 1024 SIMDs: %.1f ns per instruction and SIMD, %.0f TFLOP/s (logical)\n",
                   sp ? "v_smfmac_f32_32x32x32_bf16" : "v_mfma_f32_32x32x16_bf16  ", ms, n_inst, ms * 1e6 / (n_inst / 1024), flop / ms / 1e9);
        }
    }
    return 0;
}
