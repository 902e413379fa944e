// Probe: exact lane/element mapping of ds_read_b64_tr_b16 (gfx950), and of a b128 B-operand built from two of them.
//   hipcc --offload-arch=gfx950 -O3 tr16_probe.hip -o tr16_probe
// LDS holds value = its own element index (as a 16-bit integer).  Experiment 1: lane l passes the address of element
// 4*l (its "natural" 8-byte piece of a linear image) and the four returned elements are printed per lane.
// Experiment 2: lane l passes address of row (l&15)... to test the [4 rows][16 cols] reading.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(int* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    const int l = threadIdx.x;
    for (int i = l; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    int elem;
    if (mode == 0) elem = 4 * l;                               // linear: lane l -> elements 4l..4l+3
    else if (mode == 1) elem = (l & 15) * 64 + (l >> 4) * 4;   // 16 rows of 64 elements; lane group g reads cols 4g..4g+3
    else elem = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 256;   // per 16-lane group: 4 rows x 16 cols (row stride 64)
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

int main() {
    int* d; hipMalloc(&d, 64 * 4 * 4);
    std::vector<int> h(256);
    for (int mode = 0; mode < 3; ++mode) {
        probe<<<1, 64>>>(d, mode);
        hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
