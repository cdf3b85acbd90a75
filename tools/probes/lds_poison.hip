// Probe library (NOT part of the product): kernels that leave a known pattern in the resources a later / co-resident
// workgroup inherits, to tell "victim reads something it never wrote" apart from "box property":
//   mk_probe_lds_poison   every workgroup fills its whole dynamic LDS allocation (up to 160 KB) with `pattern`, optionally spins
//                         for `spin` clock reads so that it stays co-resident with kernels of other streams / processes
//   mk_probe_vgpr_poison  every lane writes `pattern` to 256 vector registers and leaves (the next wave on the SIMD inherits them)
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/lds_poison.hip -o tools/probes/liblds_poison.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void lds_poison_kernel(uint32_t pattern, int words, long long spin, uint32_t* sink) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    uint32_t acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < spin) acc += lds[(threadIdx.x * 33 + acc) % words];
    if (acc == 0x12345678u && sink) sink[0] = acc;          // never true: keeps the reads alive
}

__global__ __launch_bounds__(256) void vgpr_poison_kernel(uint32_t pattern, uint32_t* sink) {
    uint32_t v[200];
#pragma unroll
    for (int i = 0; i < 200; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "s"(pattern + 0u));
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 200; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(v[i]));
    if (acc == 0x12345678u && sink) sink[threadIdx.x] = acc;
}

extern "C" int mk_probe_lds_poison(uint32_t pattern, int bytes, int blocks, int threads, long long spin, void* sink, void* stream) {
    static int set = 0;
    if (set < bytes) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
        set = bytes;
    }
    hipLaunchKernelGGL(lds_poison_kernel, dim3(blocks), dim3(threads), bytes, (hipStream_t)stream, pattern, bytes / 4, spin, (uint32_t*)sink);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int mk_probe_vgpr_poison(uint32_t pattern, int blocks, void* sink, void* stream) {
    hipLaunchKernelGGL(vgpr_poison_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pattern, (uint32_t*)sink);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
