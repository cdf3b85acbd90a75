// Probe (NOT part of the product): does the WIDTH of the pixel tile a persistent workgroup streams decide the HBM rate?  The channel
// GEMMs of csrc/conv1x1.hip read / write (rows x N) bf16 matrices in tiles of `rows` x 64 pixels = 128-byte runs per row, rows 2 MB
// apart.  This kernel copies such a matrix tile by tile (persistent grid, 16 B per lane) for tile widths of 64 ... 1024 pixels.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/tile_stream.hip -o tools/probes/libtile_stream.so
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int TW>      // pixels per tile
__global__ __launch_bounds__(256) void tile_copy(const uint4* __restrict__ in, uint4* __restrict__ out, int rows, long long N, int write) {
    constexpr int VPR = TW / 8;                       // 16-byte vectors per row of a tile
    const long long tiles = N / TW;
    const long long nv = N / 8;                       // vectors per matrix row
    for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint4* ip = in + t * VPR;
        uint4* op = out + t * VPR;
        for (int idx = threadIdx.x; idx < rows * VPR; idx += 256 * 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = idx + u * 256;
                const int r = i / VPR, c = i % VPR;
                v[u] = (i < rows * VPR) ? ip[(long long)r * nv + c] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = idx + u * 256;
                const int r = i / VPR, c = i % VPR;
                if (i < rows * VPR) {
                    if (write) op[(long long)r * nv + c] = v[u];
                    else if (v[u].x == 0x12345678u) op[0] = v[u];
                }
            }
        }
    }
}

extern "C" int mk_probe_tile_copy(int tw, const void* in, void* out, int rows, long long N, int write, int grid, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define GO(T) case T: hipLaunchKernelGGL((tile_copy<T>), dim3(grid), dim3(256), 0, s, (const uint4*)in, (uint4*)out, rows, N, write); break
    switch (tw) { GO(64); GO(128); GO(256); GO(512); GO(1024); default: return 1; }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
