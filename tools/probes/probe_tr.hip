// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which LDS element lands in which lane/slot).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    int byte_addr;
    if (mode == 0) byte_addr = l * 8;                                  // linear
    else if (mode == 1) byte_addr = (l & 15) * 64 + (l >> 4) * 8;      // each lane = its own 32-element row, group picks column block
    else byte_addr = (l & 3) * 8 + ((l >> 2) & 3) * 256 + (l >> 4) * 1024;  // 4 lanes per 32B row-chunk, rows 128 elements apart
    s16x4 v;
    uint32_t a = (uint32_t)(uintptr_t)lds + byte_addr;   // LDS address = low 32 bits of the generic pointer offset
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(byte_addr + (uint32_t)(uintptr_t)(&lds[0]) - (uint32_t)(uintptr_t)(&lds[0]))) : "memory");
    (void)a;
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (element index = byte/2):\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf(" lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]);
            if (l % 4 == 3) printf("\n");
        }
    }
    return 0;
}
