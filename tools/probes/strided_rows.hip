// Probe: what HBM rate does the K0 access pattern get on gfx950?
//
// The 1x1 projections (proj_stream_f16x3.hip, proj_dw_f16x3.hip) read channel-major fp32 [B][R][N] tensors as
// tiles of ALL R rows x a few positions: every row contributes one short segment (64 - 256 bytes), the rows are
// N*4 = 16 KB apart.  This probe reads the same tensor with exactly that pattern (no arithmetic beyond a sum, no
// LDS) for several segment lengths, tile orders and grid sizes, next to a plain linear read, and writes with the
// pattern of the y stores (4-byte stores, 32 / 64 lanes along a row).  Eight 54 MB buffers are rotated so that the
// 256 MB memory-side cache does not serve the reads.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/strided_rows.hip -o tools/probes/strided_rows && tools/probes/strided_rows
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int B = 8, R = 416, N = 4096;

// tile = image b, positions [n0, n0 + SEG): all R rows.  SEG floats per row segment; LPR = SEG / 4 lanes per row.
// ADJ: a workgroup walks over neighbouring tiles (T = wg * per + i) instead of tiles gridDim.x apart.
template <int SEG, bool ADJ>
__global__ __launch_bounds__(256) void read_tiles(const float* __restrict__ x, float* __restrict__ out) {
    constexpr int LPR = SEG / 4, RPP = 256 / LPR, PASSES = (R + RPP - 1) / RPP;
    const int tid = threadIdx.x, lr = tid / LPR, lq = tid % LPR;
    const int tiles_per_img = N / SEG, ntiles = B * tiles_per_img;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < per; ++i) {
        const int T = ADJ ? blockIdx.x * per + i : blockIdx.x + i * gridDim.x;
        if (T >= ntiles) break;
        const int b = T / tiles_per_img, n0 = (T - b * tiles_per_img) * SEG;
        const float* p = x + (size_t)b * R * N + n0 + lq * 4;
        constexpr int CH = PASSES < 16 ? PASSES : 16;          // loads in flight per thread
        for (int s0 = 0; s0 < PASSES; s0 += CH) {
            f32x4 v[CH];
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                const int r = (s0 + s) * RPP + lr;
                v[s] = r < R ? *reinterpret_cast<const f32x4*>(p + (size_t)r * N) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s = 0; s < CH; ++s) acc += v[s];
        }
    }
    out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ __launch_bounds__(256) void read_linear(const float* __restrict__ x, float* __restrict__ out, size_t n4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const f32x4*>(x)[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += reinterpret_cast<const f32x4*>(x)[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// y-store pattern: tile = 256 rows x SEG positions, lane = position (4-byte stores), a wave covers 64 / SEG rows
template <int SEG>
__global__ __launch_bounds__(256) void write_tiles(float* __restrict__ y, int rows) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tiles_per_img = N / SEG, ntiles = B * tiles_per_img;
    constexpr int RPI = 64 / SEG > 0 ? 64 / SEG : 1;        // rows per store instruction
    for (int T = blockIdx.x; T < ntiles; T += gridDim.x) {
        const int b = T / tiles_per_img, n0 = (T - b * tiles_per_img) * SEG;
        float* p = y + (size_t)b * rows * N + n0 + lane % SEG;
        for (int r = wave * RPI + lane / SEG; r < rows; r += 4 * RPI) p[(size_t)r * N] = (float)r;
    }
}

__global__ __launch_bounds__(256) void write_linear(float* __restrict__ y, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<f32x4*>(y)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}

int main() {
    constexpr int NB = 8;
    const size_t elems = (size_t)B * R * N, bytes = elems * 4;
    std::vector<float*> buf(NB);
    for (auto& p : buf) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); }
    float* out;
    CK(hipMalloc(&out, 4096 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double nbytes, auto&& launch) {
        for (int w = 0; w < NB; ++w) launch(buf[w]);             // warm-up (code load, TLB)
        hipEventRecord(e0, 0);
        const int reps = 4 * NB;
        for (int i = 0; i < reps; ++i) launch(buf[i % NB]);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::printf("%-44s %8.2f us  %7.0f GB/s\n", name, ms * 1e3 / reps, nbytes * reps / (ms * 1e-3) / 1e9);
        return 0;
    };
    char name[128];
    for (int grid : {256, 512, 1024}) {
        std::snprintf(name, sizeof name, "read linear                      grid %4d", grid);
        timeit(name, (double)bytes, [&](float* p) { hipLaunchKernelGGL(read_linear, dim3(grid), dim3(256), 0, 0, p, out, elems / 4); });
#define RD(SEG, ADJ)                                                                                             \
        std::snprintf(name, sizeof name, "read rows x %4d B, %s grid %4d", SEG * 4, ADJ ? "adjacent tiles," : "strided tiles, ", grid); \
        timeit(name, (double)bytes, [&](float* p) { hipLaunchKernelGGL((read_tiles<SEG, ADJ>), dim3(grid), dim3(256), 0, 0, p, out); });
        RD(16, false) RD(32, false) RD(64, false) RD(128, false) RD(256, false)
        RD(16, true) RD(32, true) RD(64, true)
#undef RD
    }
    const double wbytes = (double)B * 256 * N * 4;
    for (int grid : {256, 512, 1024}) {
        std::snprintf(name, sizeof name, "write linear                     grid %4d", grid);
        timeit(name, wbytes, [&](float* p) { hipLaunchKernelGGL(write_linear, dim3(grid), dim3(256), 0, 0, p, (size_t)B * 256 * N / 4); });
#define WR(SEG)                                                                                                  \
        std::snprintf(name, sizeof name, "write 256 rows x %4d B (4-byte stores) grid %4d", SEG * 4, grid);    \
        timeit(name, wbytes, [&](float* p) { hipLaunchKernelGGL(write_tiles<SEG>, dim3(grid), dim3(256), 0, 0, p, 256); });
        WR(32) WR(64)
#undef WR
    }
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    std::printf("ok\n");
    return 0;
}
