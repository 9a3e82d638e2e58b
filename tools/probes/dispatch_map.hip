// Which CU / XCD does workgroup b land on, and when?  (tuning probe, not product code)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
    extern __shared__ char lds[];
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = hw; out[blockIdx.x * 4 + 1] = xcc; out[blockIdx.x * 4 + 2] = (unsigned)(t0 >> 8); out[blockIdx.x * 4 + 3] = (unsigned)(__builtin_amdgcn_s_memtime() >> 8); }
    lds[threadIdx.x] = 0;
}
int main() {
    const int nb = 1024;
    unsigned* d; hipMalloc(&d, nb * 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 70000, 0, d, 40); hipDeviceSynchronize(); }
    std::vector<unsigned> h(nb * 4); hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    unsigned tmin = ~0u; for (int b = 0; b < nb; ++b) tmin = h[b*4+2] < tmin ? h[b*4+2] : tmin;
    for (int b = 0; b < nb; b += (b < 40 ? 1 : 37)) {
        unsigned hw = h[b*4];
        printf("b %4d xcc %u se %u sh %u cu %2u simd %u wave %u  start %6u end %6u\n", b, h[b*4+1] & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15, h[b*4+2] - tmin, h[b*4+3] - tmin);
    }
    // co-residency: for the first 512 blocks, which pairs share (xcc, se, sh, cu)?
    int same_pair_b_b256 = 0, same_pair_b_b1 = 0, same_b_b8 = 0;
    auto key = [&](int b) { unsigned hw = h[b*4]; return ((h[b*4+1] & 15) << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15); };
    for (int b = 0; b < 256; ++b) { same_pair_b_b256 += key(b) == key(b + 256); same_pair_b_b1 += key(b) == key(b ^ 1); same_b_b8 += key(b) == key(b + 8); }
    printf("first round: key(b)==key(b+256): %d/256   key(b)==key(b^1): %d/256   key(b)==key(b+8): %d/256\n", same_pair_b_b256, same_pair_b_b1, same_b_b8);
    return 0;
}
