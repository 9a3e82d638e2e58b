// How many 256-thread workgroups are co-resident on one CU of gfx950 for a given dynamic-LDS size?  (Round 5: the four-wave conv
// kernel asks for 2 x 80 KB = all 160 KB of a CU.)  Every workgroup bumps a counter keyed on (XCC, SE, CU) from the hardware id
// registers, spins ~20 us, and drops it again; the maximum seen per CU is the co-residency.
// build: hipcc --offload-arch=gfx950 -O3 -w -o tools/probes/lds_occupancy tools/probes/lds_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256, 2) void occ_kernel(int* cnt, int* maxc, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;    // HW_REG_XCC_ID
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const int key = (int)(((xcc * 8 + se) * 2 + sh) * 16 + cu);
    lds[threadIdx.x * 16] = (unsigned char)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) { const int c = atomicAdd(&cnt[key], 1) + 1; atomicMax(&maxc[key], c); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 40000ull) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) { atomicSub(&cnt[key], 1); if (lds[16] == 77) sink[0] = 1; }
}

int main() {
    int *cnt, *maxc; unsigned* sink;
    hipMalloc(&cnt, 4096 * 4); hipMalloc(&maxc, 4096 * 4); hipMalloc(&sink, 64);
    for (int kb : {32, 64, 72, 76, 78, 79, 80, 96}) {
        hipMemset(cnt, 0, 4096 * 4); hipMemset(maxc, 0, 4096 * 4);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&occ_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        hipLaunchKernelGGL(occ_kernel, dim3(4096), dim3(256), kb * 1024, 0, cnt, maxc, sink);
        hipError_t e = hipDeviceSynchronize();
        std::vector<int> h(4096);
        hipMemcpy(h.data(), maxc, 4096 * 4, hipMemcpyDeviceToHost);
        int used = 0, mx = 0, mn = 1 << 30;
        for (int v : h) if (v) { ++used; mx = std::max(mx, v); mn = std::min(mn, v); }
        printf("dynamic LDS %3d KB per 256-thread workgroup: %s, %d CU keys seen, co-resident workgroups per CU: min %d max %d\n", kb, hipGetErrorString(e), used, mn, mx);
    }
    return 0;
}
