// Round 5, VERDICT r4 item 3, sub-idea 1: "feed the WEIGHT operand straight from L2 into VGPRs (weights pre-packed in fragment order: one
// global_load_dwordx4 per lane is one MFMA operand) and keep only the pixel operand in LDS".
// A stand-alone model of the bf16 256x256 tile's main loop (eight waves 2 x 4, wave tile 128 x 64: MT = 4, NT = 2; 32-deep chunks = 2 k-steps of
// v_mfma_f32_32x32x16_bf16; 4-deep ring; one barrier per chunk, DMA three chunks ahead), one persistent workgroup per CU, all data L2-resident:
//   lds      both operands staged through LDS by global_load_lds (what conv_planes_kernel does): per chunk 32 KB of DMA, 8 waves x 12 ds_read_b128
//   l2       pixel operand through LDS (16 KB of DMA, 8 ds_read_b128 per wave), weight fragments by global_load_dwordx4 from a fragment-ordered
//            array (1 KB contiguous per wave instruction), one chunk ahead in registers; the two waves that share a channel block load the
//            same fragments (2 x redundancy through the vector L1)
//   nob      the l2 loop with the weight loads removed (fragments constant): the ceiling of what removing the LDS reads can buy
// Prints the sustained MFMA rate of each.  build: hipcc --offload-arch=gfx950 -O3 -w -o tools/probes/weights_from_l2 tools/probes/weights_from_l2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef short bf16x8v __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, ROWB = 64, NS = 4, KS = 2;     // 32-deep chunks (64-byte rows), 4-deep ring: conv_planes_kernel<1, 256, 256, 2, 4, 4>
constexpr int WIN = 4;                                    // chunks of the per-CU source window (re-read cyclically: stays in L2)

template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: lds, 1: l2, 2: nob, 3: lds without any DMA (static stages), 4: lds with the weight READS removed (its DMA stays), 5: lds with the
// weight DMA removed (its reads stay, of a static stage), 6: nob without the pixel DMA (MFMAs + 8 pixel reads per wave only)
template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(const unsigned char* __restrict__ asrc, const unsigned char* __restrict__ bsrc,
                                                   const unsigned char* __restrict__ bfrag, float* out, int nchunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr bool BLDS = MODE == 0 || MODE == 3 || MODE == 4 || MODE == 5;      // stages hold the weight rows too
    constexpr bool ADMA = MODE != 3 && MODE != 6, BDMA = MODE == 0 || MODE == 4, BREAD = MODE == 0 || MODE == 3 || MODE == 5;
    constexpr int STAGE = (BLDS ? BM + BN : BM) * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wid >> 2, wn = wid & 3;
    const unsigned char* aw = asrc + (size_t)blockIdx.x * WIN * BM * ROWB;
    const unsigned char* bw = bsrc + (size_t)(blockIdx.x & 7) * WIN * BN * ROWB;       // (a layer's weights are shared by all CUs: 8 windows only)
    const unsigned char* bf = bfrag + (size_t)(blockIdx.x & 7) * WIN * BN * ROWB;       // (a layer's weights are shared: 8 windows only)
    // DMA: a wave instruction = 16 rows x 64 B; lane -> (row, physical slot), the swizzle (slot ^ (row >> 2) & 3) applied on the source side
    const int drow = lane >> 2, dls = ((lane & 3) ^ ((drow >> 2) & 3)) * 16;
    auto dma = [&](int kc, int st) {
        const int w = kc % WIN;
        if constexpr (ADMA)
#pragma unroll
        for (int i = 0; i < BM / 128; ++i) {                                         // 2 pieces per wave
            const int r0 = (wid * (BM / 128) + i) * 16;
            __builtin_amdgcn_global_load_lds(GPTR(aw + ((size_t)w * BM + r0 + drow) * ROWB + dls), LPTR(lds + st * STAGE + r0 * ROWB), 16, 0, 0);
        }
        if constexpr (BDMA) {
#pragma unroll
            for (int i = 0; i < BN / 128; ++i) {
                const int r0 = (wid * (BN / 128) + i) * 16;
                __builtin_amdgcn_global_load_lds(GPTR(bw + ((size_t)w * BN + r0 + drow) * ROWB + dls), LPTR(lds + st * STAGE + (BM + r0) * ROWB), 16, 0, 0);
            }
        }
    };
    constexpr int NDMA = (ADMA ? 2 : 0) + (BDMA ? 2 : 0);
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int aaddr[4], baddr[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) aaddr[j] = (wm * 128 + j * 32 + l31) * ROWB;
#pragma unroll
    for (int i = 0; i < 2; ++i) baddr[i] = (BM + wn * 64 + i * 32 + l31) * ROWB;
    const int sw = (l31 >> 2) & 3;
    u32x4 bcur[2][KS], bnext[2][KS];
    auto bload = [&](int kc, u32x4 (&dst)[2][KS]) {
        const int w = kc % WIN;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                dst[i][ks] = *reinterpret_cast<const u32x4*>(bf + ((((size_t)(wn * 2 + i) * WIN + w) * KS + ks) * 64 + lane) * 16);
    };
    if constexpr (MODE == 1) bload(0, bcur);
    if constexpr (!BREAD && MODE != 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bcur[i][ks] = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    }
    dma(0, 0);
    dma(1, 1);
    dma(2, 2);
    int st = 0;
    for (int kc = 0; kc < nchunks; ++kc) {
        // chunk kc's pieces have landed; those of chunks kc+1, kc+2 may be in flight -- l2: the weight fragments of chunk kc (issued between
        // the pieces of kc+1 and kc+2; loads complete in order) as well, so only chunk kc+2's pieces may be
        wait_vmcnt<MODE == 1 ? NDMA : 2 * NDMA>();
        __syncthreads();
        if constexpr (MODE == 1) bload(kc + 1, bnext);
        dma(kc + 3, (st + 3) % NS);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8v af[4], wf[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const bf16x8v*>(lds + st * STAGE + aaddr[j] + (((ks * 2 + lhi) ^ sw) * 16));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (BREAD) wf[i] = *reinterpret_cast<const bf16x8v*>(lds + st * STAGE + baddr[i] + (((ks * 2 + lhi) ^ sw) * 16));
                else wf[i] = __builtin_bit_cast(bf16x8v, bcur[i][ks]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bcur[i][ks] = bnext[i][ks];
        }
        st = st + 1 == NS ? 0 : st + 1;
    }
    wait_vmcnt<0>();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
void run(const char* name, const unsigned char* a, const unsigned char* b, const unsigned char* bfr, float* out, int ncu) {
    const int nchunks = 8000;
    const int ldsb = NS * ((MODE == 0 || MODE == 3 || MODE == 4 || MODE == 5) ? BM + BN : BM) * ROWB;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((loop_kernel<MODE>), dim3(ncu), dim3(512), ldsb, 0, a, b, bfr, out, 100);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop_kernel<MODE>), dim3(ncu), dim3(512), ldsb, 0, a, b, bfr, out, nchunks);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double fl = 2.0 * BM * BN * 32 * (double)nchunks * ncu;
    printf("%-22s LDS %6d B: %8.3f ms  %7.1f TFLOP/s  (%.0f ns per chunk and CU)\n", name, ldsb, best, fl / best / 1e9, best * 1e6 / nchunks);
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    const size_t win = (size_t)WIN * BM * ROWB;
    unsigned char *a, *b, *bfr; float* out;
    hipMalloc(&a, win * ncu); hipMalloc(&b, win * ncu); hipMalloc(&bfr, win * 8); hipMalloc(&out, (size_t)ncu * 512 * 4);
    {   // random finite bf16 operands (|v| in [2^-7, 2), both signs): the matrix pipes' power draw depends on the operand bits
        std::vector<unsigned short> h(win * ncu / 2);
        unsigned x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x80ffu) | (0x3c00u + (((x >> 8) & 3u) << 8))); }
        hipMemcpy(a, h.data(), win * ncu, hipMemcpyHostToDevice);
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x80ffu) | (0x3c00u + (((x >> 8) & 3u) << 8))); }
        hipMemcpy(b, h.data(), win * ncu, hipMemcpyHostToDevice);
        hipMemcpy(bfr, h.data(), win * 8, hipMemcpyHostToDevice);
    }
    printf("%d CUs; model of the bf16 256x256 eight-wave tile's main loop, %d-chunk windows per CU (%.1f MB total: L2 / MALL resident)\n", ncu, WIN,
           (2.0 * win * ncu + win * 8) / 1e6);
    run<0>("lds", a, b, bfr, out, ncu);
    run<1>("l2", a, b, bfr, out, ncu);
    run<2>("nob", a, b, bfr, out, ncu);
    run<0>("lds", a, b, bfr, out, ncu);
    run<1>("l2", a, b, bfr, out, ncu);
    run<3>("lds, no DMA", a, b, bfr, out, ncu);
    run<4>("lds, no weight reads", a, b, bfr, out, ncu);
    run<5>("lds, no weight DMA", a, b, bfr, out, ncu);
    run<6>("nob, no DMA", a, b, bfr, out, ncu);
    return 0;
}
