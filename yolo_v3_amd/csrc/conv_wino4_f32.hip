// Winograd F(4x4,3x3) form of a 3x3 / stride-1 / pad-1 convolution, exact-fp32 mode (round 6).
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A      over 4x4 OUTPUT tiles (6x6 input patches at stride 4)
//
// with the Toom-Cook points (0, 1, -1, 1/2, -2, inf):
//   B^T = [ 1 -1.5 -2   1.5  1   0 ]      A^T = [ 1  1  1  1    1  0 ]      G = [   1      0      0   ]
//         [ 0 -1    .5  2.5  1   0 ]            [ 0  1 -1  1/2 -2  0 ]          [  1/3    1/3    1/3  ]
//         [ 0  1  -2.5   .5  1   0 ]            [ 0  1  1  1/4  4  0 ]          [ -1/3    1/3   -1/3  ]
//         [ 0 -2  -1    2    1   0 ]            [ 0  1 -1  1/8 -8  1 ]          [ -16/15 -8/15  -4/15 ]
//         [ 0  .5 -1   -.5   1   0 ]                                             [  1/15  -2/15   4/15 ]
//         [ 0  1  -1.5 -2    1.5 1 ]                                             [   0      0      1   ]
// 36 multiplications per 16 outputs and channel pair instead of 144: 4x fewer matrix instructions than the direct form, 1.78x fewer than
// F(2x2,3x3).  Why THESE points: with the textbook set (0, +-1, +-2) the whole network on hostile data lands 5.3e-4 from an fp64
// evaluation (the fp32 oracle itself: 1.4e-4); with one reciprocal pair (1/2, -2) in place of (2, -2) it is 1.9e-4 with all 31
// eligible layers in this form -- 1.2e-5 on the headline data, where the direct form measures 1.2e-5 (tools/winograd_f32_gate.py,
// profiles/r06a_wino4_numerics_gate.txt).  A^T's entries are powers of two: the output transform multiplies exactly.
//
// Three launches' worth of work in two:
//   wino4_input_f32_kernel   V[36][T][C] = B^T d B of every tile (T = B * ceil(H/4) * ceil(W/4)), fp32: reads 4 B and writes 9 B per
//                            input element (F(2x2): 16 B);
//   conv_wino4_f32_kernel    36 T x cout x C GEMMs on v_mfma_f32_16x16x4_f32 whose K loop walks the transform positions; at the end of a
//                            position the 16x16 product blocks are folded into four ROW accumulators (A^T along the patch's columns), at the
//                            end of a patch row into the tile's sixteen outputs (A^T along its rows): 8 + 32 + 128 accumulator registers per
//                            wave, nothing but V, U and the finished outputs ever touches HBM.  Same BN / LeakyReLU / residual epilogue
//                            as the direct kernel.
// Workgroup = 64 output channels x 32 tiles on FOUR waves (16 channels x 32 tiles each: two 16x16 accumulator blocks sharing the weight
// fragment), <= 256 registers and 36 KB of LDS: TWO workgroups per CU, so that one's 16-output epilogue runs under the other's main loop.
// Operands reach LDS by DMA (global_load_lds_dwordx4: one wave instruction = 8 rows of 128 B = one 32-channel chunk of 8 tiles / filters),
// three stages of 12 KB, one barrier per chunk; rows are XOR-swizzled by 16-byte slot (slot ^ (row >> 1) & 7: conflict-free ds_read_b128
// over the 64 banks -- the weight image is stored swizzled by the packing kernel, the V rows are swizzled on the source side).
//
// Replaces reference darknet.py:43-44 (conv_bn_relu.forward) and :52-53 (res_layer.forward) for the 3x3 stride-1 layers.
#include <type_traits>
#include "yv3_common.h"

namespace {

#define W4F_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define W4F_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BNC = 64;                        // output channels per workgroup
constexpr int BMT = 32;                        // tiles per workgroup
constexpr int ROWB = 128;                      // bytes per LDS row: one 32-channel chunk of fp32
constexpr int U_BYTES = BNC * ROWB;            // 8 KB
constexpr int V_BYTES = BMT * ROWB;            // 4 KB
constexpr int STAGE = U_BYTES + V_BYTES;       // 12 KB
#define W4F_NSTAGE 4                             // (the main loop's stage indices are compile-time constants for FOUR stages)
#ifndef W4F_PRIO
#define W4F_PRIO 3                              // s_setprio(1) around a chunk's MFMAs: +2 % on the launches' time in the network (profiles/r06i_*)
#endif
#ifndef W4F_EXTRA_LDS
#define W4F_EXTRA_LDS 0                         // (measurement builds: pad the allocation to force one workgroup per CU)
#endif
#ifndef W4F_DMA_FIRST
#define W4F_DMA_FIRST 0                         // (1: the chunk's three requests right behind the barrier -- A/B builds)
#endif
constexpr int NSTAGE = W4F_NSTAGE;             // ring stages: requests run NSTAGE chunks ahead of the multiplication
// measurement builds only (-DYV3_MEASURE -DW4F_ABL=mask; results INVALID): 1 no epilogue, 2 no MFMAs, 4 no fragment reads, 8 no DMA,
// 16 no barrier, 32 no fold
#if defined(YV3_MEASURE) && defined(W4F_ABL)
constexpr int ABL = W4F_ABL;
#else
constexpr int ABL = 0;
#endif

struct Wino4Params {
    const float* v;          // [36][T][C]
    const float* u;          // packed image [cout/64][36][C/32][64][8 slots][4]
    const float* alpha;
    const float* beta;
    const float* res;
    float* y;
    int C, Cout, H, W, th, tw, T;
    int cchunks;             // C / 32
    int nblk_n;              // Cout / 64
    int act;
    long long pos_stride;    // T * C
    unsigned u_bytes, v_bytes;   // sizes of the two DMA sources (buffer descriptors)
    // even schedule (SK): workgroups [0, n_full) take one whole item (tile block x channel block) each, the others one of `parts` equal
    // ranges of patch rows of an item in [n_full, nitems) -- see conv_wino4_f32_kernel
    int nitems, n_full, parts;
    float* sk_parts;         // hand-over area: one part (W4F_PART_FLOATS) per tail workgroup ...
    int* sk_flags;           // ... and one flag each (zero between launches)
    int* status;             // yv3_conv_desc.flags (bit 1 <- a hand-over timed out), may be NULL
};
constexpr int W4F_MAX_TAIL_WG = 2 * YV3_WINO_SK_MAX_WG - 1;     // tail workgroups of an even launch (their parts + one part's room for the flags fill the area)
constexpr int W4F_PART_FLOATS = 4 * 32 * 256;   // floats of one part: four waves x sixteen outputs x two blocks x 64 lanes x 4 channels = 128 KB

template <int N> __device__ __forceinline__ void w4f_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__device__ __forceinline__ int w4f_swz(int row) { return (row >> 1) & 7; }

// out[l] = sum_k B^T[l][k] in[k]
__device__ __forceinline__ void bt_apply(const f32x4 (&d)[6], f32x4 (&o)[6]) {
    o[0] = d[0] - 1.5f * d[1] - 2.0f * d[2] + 1.5f * d[3] + d[4];
    o[1] = 0.5f * d[2] - d[1] + 2.5f * d[3] + d[4];
    o[2] = d[1] - 2.5f * d[2] + 0.5f * d[3] + d[4];
    o[3] = 2.0f * (d[3] - d[1]) - d[2] + d[4];
    o[4] = 0.5f * (d[1] - d[3]) - d[2] + d[4];
    o[5] = d[1] - 1.5f * d[2] - 2.0f * d[3] + 1.5f * d[4] + d[5];
}

// V = B^T d B: one thread = one tile x 4 channels (16-byte loads / stores; a wave covers 256 consecutive channels-of-tiles)
__global__ __launch_bounds__(256) void wino4_input_f32_kernel(const float* __restrict__ x, float* __restrict__ v,
                                                              int H, int W, int C, int th, int tw, long long T) {
    const int cg = C >> 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * cg) return;
    const long long t = i / cg;
    const int c = (int)(i - t * cg) * 4;
    const int b = (int)(t / (th * tw));
    const int rem = (int)(t - (long long)b * th * tw);
    const int ty = rem / tw, tx = rem - ty * tw;
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    f32x4 tr[6][6];                                        // tr[r][l] = sum_k B^T[l][k] d[r][k]  (along the patch's columns first)
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        f32x4 d[6];
        const int yy = y0 + r;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int xx = x0 + q;
            const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            d[q] = ok ? *reinterpret_cast<const f32x4*>(x + (((long long)b * H + yy) * W + xx) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        bt_apply(d, tr[r]);
    }
#pragma unroll
    for (int l = 0; l < 6; ++l) {                          // V[xi][l] = sum_r B^T[xi][r] tr[r][l]
        f32x4 col[6], o[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) col[r] = tr[r][l];
        bt_apply(col, o);
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) *reinterpret_cast<f32x4*>(v + ((long long)(xi * 6 + l) * T + t) * C + c) = o[xi];
    }
}

// U [cout][cin][36] fp32 (G g G^T, computed by the host in fp64) -> the kernel's LDS image, chunk by chunk:
// [cout/64][36 positions][cin/32][64 rows][8 physical slots][4 floats], physical slot = slot ^ swz(row)
__global__ __launch_bounds__(256) void wino4_pack_kernel(const float* __restrict__ u, float* __restrict__ out, int cout, int cin) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // one float4 of the image
    const long long n4 = (long long)cout * cin * 36 / 4;
    if (i >= n4) return;
    const int cch = cin / 32;
    const int ps = (int)(i & 7);
    const int row = (int)((i >> 3) & 63);
    long long r = i >> 9;
    const int cc = (int)(r % cch); r /= cch;
    const int pos = (int)(r % 36);
    const int nb = (int)(r / 36);
    const int slot = ps ^ w4f_swz(row);
    const int o = nb * 64 + row, c = cc * 32 + slot * 4;
    f32x4 q;
#pragma unroll
    for (int e = 0; e < 4; ++e) q[e] = u[((long long)o * cin + c + e) * 36 + pos];
    *reinterpret_cast<f32x4*>(out + i * 4) = q;
}

// A^T[j][nu] of the points above
__device__ __forceinline__ constexpr float at_coef(int j, int nu) {
    return nu == 0 ? (j == 0 ? 1.f : 0.f)
         : nu == 1 ? 1.f
         : nu == 2 ? ((j & 1) ? -1.f : 1.f)
         : nu == 3 ? (j == 0 ? 1.f : j == 1 ? 0.5f : j == 2 ? 0.25f : 0.125f)
         : nu == 4 ? (j == 0 ? 1.f : j == 1 ? -2.f : j == 2 ? 4.f : -8.f)
         : (j == 3 ? 1.f : 0.f);
}

// CCH2: two chunks per transform position (cin 64); else a multiple of four (cin % 128 == 0) -- either way every chunk's ring stage
// is known at compile time (four stages; a position starts at stage 0, or 0 / 2 alternately), so fragment reads and DMA targets
// are immediate offsets and the DMA sources scalar bases: NO vector-ALU instruction in the steady-state loop but the MFMAs.
template <bool CCH2, bool SK>
__global__ __launch_bounds__(256, 2) void conv_wino4_f32_kernel(const Wino4Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = 36 * p.cchunks;
    const int nkx = 6 * p.cchunks;                                        // chunks of one patch row (= one unit of work)
#if defined(YV3_MEASURE) && defined(W4F_DEPHASE)          // the second resident of every CU (workgroup b + 256) starts W4F_DEPHASE / 1000 of a main loop late
    if (blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < nk * W4F_DEPHASE / 5000; ++i) __builtin_amdgcn_s_sleep(108);       // one chunk of a paired main loop ~ 1385 cycles = 0.2 x 108 x 64
#endif

    // ---- this workgroup's work: patch rows [xa, xb) of one item (item = tile block * nblk_n + channel block).
    // Plain launches and the full rounds of an even launch: a whole item, items dealt to the XCDs in contiguous ranges.
    // Even schedule (SK), workgroups n_full ...: the launch's LAST, partial round of items would leave part of the chip idle for a whole
    // item time (784 items on 512 slots: 0.348 ms where 1.53 rounds of work are 0.29, profiles/r06s_wino4_steps.log) -- instead each of
    // those items is cut into `parts` equal ranges of patch rows, one workgroup each, so that the tail fills the slots in shorter rounds.
    // A range that does not hold row 0 accumulates Y over its rows only, leaves it in the hand-over area and is done; the workgroup with
    // row 0 adds the parts (in row order: a fixed summation order per shape) and runs the epilogue.  The parts of an item sit on ONE XCD
    // (workgroups b, b + 8, ...: one L2, dispatched back to back); the protocol is conv_planes.hip's stream-K one.
    int item, xa = 0, xb = 6, sk_t = 0;
    if (!SK || (int)blockIdx.x < p.n_full) item = yv3_xcd_remap(blockIdx.x, SK ? p.n_full : (int)gridDim.x);
    else {
        sk_t = blockIdx.x - p.n_full;
        const int nt = p.nitems - p.n_full;
        const int x = sk_t & 7, l = sk_t >> 3;
        const int i0 = (int)(((long long)nt * x) >> 3), i1 = (int)(((long long)nt * (x + 1)) >> 3);   // this XCD's tail items
        const int il = l / p.parts, part = l - il * p.parts;
        if (il >= i1 - i0) return;                                         // (the XCDs' shares differ by one item: whole workgroups leave)
        item = p.n_full + i0 + il;
        const int rows = 6 / p.parts;
        xa = part * rows; xb = xa + rows;
    }

    // ---- DMA sources: buffer loads (buffer_load_dwordx4 ... offen lds): a scalar byte offset that advances by scalar adds + a per-lane
    // byte offset that never changes within an item -- no vector-ALU address arithmetic in the loop.
    // Weight side: the packed image is the LDS image; wave w copies 1 KB pieces 2w and 2w+1 of a chunk's 8 KB.
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, (int)p.u_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.v), 0, (int)p.v_bytes, 0x00020000);
    unsigned usoff = 0;
    const unsigned uoff = lane * 16;
    // V side: wave w stages tile rows [8w, 8w+8): lane -> row lane/8, physical slot lane%8, source slot = physical ^ swz(row)
    unsigned vsoff = 0, voff = 0;
    int pf_c = 0;                                                         // channel offset of the chunk being requested
    int pf_k = 0, pf_end = 0;                                             // its chunk number, the end of this range of chunks
    // one chunk = three 1 KB pieces per wave; `dma_piece(i, stage)` requests piece i, `dma_advance()` moves on to the next chunk --
    // past the last chunk the requests repeat it (a constant number of requests per iteration keeps the vmcnt waits exact)
    auto dma_piece = [&](int i, unsigned char* stage) {
        if (ABL & 8) return;
        if (i == 0)      __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, W4F_LPTR(stage + (2 * wid) * 1024), 16, uoff, usoff, 0, 0);
        // (the instruction's immediate offset moves BOTH the memory address and the LDS address: piece 1 = piece 0's bases + 1024)
        else if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(urs, W4F_LPTR(stage + (2 * wid) * 1024), 16, uoff, usoff, 1024, 0);
        else             __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, W4F_LPTR(stage + U_BYTES + wid * 1024), 16, voff, vsoff, 0, 0);
    };
    auto dma_advance = [&]() {
        if (pf_k + 1 < pf_end) {
            ++pf_k;
            usoff += BNC * 32 * 4;
            vsoff += 128;
            pf_c += 32;
            if (pf_c == p.C) { pf_c = 0; vsoff += (unsigned)(p.pos_stride - p.C) * 4u; }
        }
    };

    // ---- fragment addresses: A operand = U rows (this wave's 16 channels), B operand = V rows (two blocks of 16 tiles)
    const int fr = lane & 15, fq = lane >> 4;
    const int urow = 16 * wid + fr;
    const int u_off0 = urow * ROWB + ((fq ^ w4f_swz(urow)) << 4);
    const int u_off1 = urow * ROWB + (((4 + fq) ^ w4f_swz(urow)) << 4);
    int v_off0[2], v_off1[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int r = 16 * b + fr;
        v_off0[b] = U_BYTES + r * ROWB + ((fq ^ w4f_swz(r)) << 4);
        v_off1[b] = U_BYTES + r * ROWB + (((4 + fq) ^ w4f_swz(r)) << 4);
    }

    f32x4 P[2], R[4][2], Y[4][4][2];
    // the fragments of the chunk being multiplied live in registers (F); the NEXT chunk's are read while its MFMAs run
    struct Frag { f32x4 ua, va0, va1, ub, vb0, vb1; };
    auto read_frag = [&](const unsigned char* st) {
        Frag f;
        if (ABL & 4) { f.ua = f.va0 = f.va1 = f.ub = f.vb0 = f.vb1 = f32x4{(float)lane, 1.f, 2.f, 3.f}; return f; }
        f.ua = *reinterpret_cast<const f32x4*>(st + u_off0);
        f.va0 = *reinterpret_cast<const f32x4*>(st + v_off0[0]);
        f.va1 = *reinterpret_cast<const f32x4*>(st + v_off0[1]);
        f.ub = *reinterpret_cast<const f32x4*>(st + u_off1);
        f.vb0 = *reinterpret_cast<const f32x4*>(st + v_off1[0]);
        f.vb1 = *reinterpret_cast<const f32x4*>(st + v_off1[1]);
        return f;
    };
    Frag F0, F1;
    using TrueT = std::integral_constant<bool, true>;
    using FalseT = std::integral_constant<bool, false>;

    // one chunk (its fragments already in registers): [wait for my pieces of the NEXT chunk] [barrier: the next chunk is complete, and
    // everybody has taken this chunk's fragments out of its stage] [request chunk + NSTAGE into that stage, read the next chunk's
    // fragments: both under this chunk's 16 MFMAs -- a request costs its wave 60-250 cycles of issue, a fragment read ~100 of latency
    // (DESIGN.md section 5); in front of the MFMAs they were half of the loop's time (profiles/r06e_wino4_ablations.txt)]
    // (F: this chunk's fragments, G: where the next chunk's go -- the caller alternates two register sets, so nothing is copied)
#if defined(YV3_MEASURE) && defined(W4F_TIMELINE)          // cycle split of one workgroup's main loop -> the first floats of y (results INVALID)
    unsigned long long tl_t = __builtin_amdgcn_s_memtime(), tl_sync = 0, tl_burst = 0, tl_fold = 0;
    const unsigned long long tl_entry = tl_t;
#define W4F_MARK(acc_) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_ += t_ - tl_t; tl_t = t_; } while (0)
#else
#define W4F_MARK(acc_) do {} while (0)
#endif
    // FIRST: the first chunk of a transform position -- its first MFMAs start from a zero accumulator (an inline constant: the product
    // blocks are never cleared by vector instructions, which cost a wave ~30 cycles each while its SIMD's other wave issues MFMAs)
    // S: the ring stage of THIS chunk (compile time)
    auto chunk = [&](auto first_c, auto stage_c, const Frag& F, Frag& G) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int S = decltype(stage_c)::value;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        W4F_MARK(tl_fold);
        w4f_wait_vmcnt<3 * (NSTAGE - 2)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): my fragment reads of THIS chunk (issued a chunk ago) are out of its stage
        if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        W4F_MARK(tl_sync);
        unsigned char* const fr_ = lds + S * STAGE;
        G = read_frag(lds + ((S + 1) % NSTAGE) * STAGE);
#if W4F_PRIO == 3
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (ABL & 2) { P[0][t] += F.ua[t] + F.va0[t]; P[1][t] += F.va1[t]; }
            else {
            P[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.ua[t], F.va0[t], (FIRST && t == 0) ? zero4 : P[0], 0, 0, 0);
            P[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.ua[t], F.va1[t], (FIRST && t == 0) ? zero4 : P[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t < 3) dma_piece(t, fr_);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (ABL & 2) { P[0][t] += F.ub[t] + F.vb0[t]; P[1][t] += F.vb1[t]; }
            else {
            P[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.ub[t], F.vb0[t], P[0], 0, 0, 0);
            P[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.ub[t], F.vb1[t], P[1], 0, 0, 0);
            }
        }
#if W4F_PRIO == 3
        __builtin_amdgcn_s_setprio(0);
#endif
        dma_advance();
        W4F_MARK(tl_burst);
    };

    auto position = [&](auto nu_c) {
        constexpr int nu = decltype(nu_c)::value;
        {
            using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
            using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
            if constexpr (CCH2) {                          // 12 chunks per patch row: positions start at stage 0, 2, 0, 2, 0, 2
                if constexpr ((nu & 1) == 0) { chunk(TrueT{}, S0{}, F0, F1); chunk(FalseT{}, S1{}, F1, F0); }
                else                         { chunk(TrueT{}, S2{}, F0, F1); chunk(FalseT{}, S3{}, F1, F0); }
            } else {
                chunk(TrueT{}, S0{}, F0, F1); chunk(FalseT{}, S1{}, F1, F0); chunk(FalseT{}, S2{}, F0, F1); chunk(FalseT{}, S3{}, F1, F0);
                for (int cc = 4; cc < p.cchunks; cc += 4) {
                    chunk(FalseT{}, S0{}, F0, F1); chunk(FalseT{}, S1{}, F1, F0); chunk(FalseT{}, S2{}, F0, F1); chunk(FalseT{}, S3{}, F1, F0);
                }
            }
            // end of position (xi, nu): R[j] += A^T[j][nu] * M  (the next position's first MFMAs restart M from zero)
            if (ABL & 32) return;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    constexpr float z = 0.f;
                    const float cf = at_coef(j, nu);
                    if (cf != z) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) R[j][b][e] = fmaf(P[b][e], cf, R[j][b][e]);
                    }
                }
            }
        }
    };
    {
        const int nb = item % p.nblk_n;
        const int n0 = nb * BNC;
        const int m0 = (item / p.nblk_n) * BMT;
        {
            const int k0 = xa * nkx;
            usoff = (unsigned)((((long long)nb * nk + k0) * (BNC * 32) + (2 * wid) * 256) * 4);
            const int vrow = 8 * wid + (lane >> 3);
            const int vt = min(m0 + vrow, p.T - 1);                      // (rows past the last tile re-read it; never stored)
            vsoff = (unsigned)m0 * (unsigned)p.C * 4u + (unsigned)(6 * xa) * (unsigned)p.pos_stride * 4u;
            voff = (unsigned)(vt - m0) * (unsigned)p.C * 4u + ((((unsigned)lane & 7u) ^ (unsigned)w4f_swz(vrow)) << 4);
            pf_c = 0; pf_k = k0; pf_end = xb * nkx;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            P[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                R[j][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) Y[i][j][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // ring fill: the range's chunks 0 .. NSTAGE-1 (a patch row starts at stage 0)
#pragma unroll
        for (int d = 0; d < NSTAGE; ++d) {
#pragma unroll
            for (int i = 0; i < 3; ++i) dma_piece(i, lds + d * STAGE);
            dma_advance();
        }
        w4f_wait_vmcnt<3 * (NSTAGE - 1)>();
        __builtin_amdgcn_s_barrier();
#if W4F_PRIO == 1                                          // static priority for one of a SIMD's two waves (by its hardware wave slot)
        if (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1) __builtin_amdgcn_s_setprio(2);
#elif W4F_PRIO == 2                                        // ... by dispatch round
        if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(2);
#endif
        F0 = read_frag(lds);
        for (int xi = xa; xi < xb; ++xi) {
            position(std::integral_constant<int, 0>{}); position(std::integral_constant<int, 1>{}); position(std::integral_constant<int, 2>{});
            position(std::integral_constant<int, 3>{}); position(std::integral_constant<int, 4>{}); position(std::integral_constant<int, 5>{});
            // end of patch row xi: Y[i][j] += A^T[i][xi] * R[j], R cleared (coefficients by value: xi is a run-time index)
            const float c1 = xi == 0 || xi == 5 ? 0.f : xi == 1 ? 1.f : xi == 2 ? -1.f : xi == 3 ? 0.5f : -2.f;
            const float cfi[4] = {xi == 5 ? 0.f : 1.f, c1, c1 * c1, xi == 5 ? 1.f : c1 * c1 * c1};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (cfi[i] == 0.f) continue;                  // (wave-uniform: rows 0 and 5 of the patch reach one output row each)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int e = 0; e < 4; ++e) Y[i][j][b][e] = fmaf(R[j][b][e], cfi[i], Y[i][j][b][e]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b) R[j][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        w4f_wait_vmcnt<0>();                                   // (the repeated requests of the last chunks: nothing may land in a retired workgroup's LDS)
#if defined(YV3_MEASURE) && defined(W4F_TIMELINE)
        const unsigned long long tl_loop_end = __builtin_amdgcn_s_memtime();
#endif

        if constexpr (SK) {
            if (xa == 0 && xb < 6) {
                // the item's row 0 is here: add what the item's other workgroups accumulated for their rows, in row order
                for (int k = 1; k < p.parts; ++k) {
                    const int partner = sk_t + 8 * k;
                    if (tid == 0) {
                        int n = 0, f;
                        while ((f = __hip_atomic_load(p.sk_flags + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++n < (1 << 22))
                            __builtin_amdgcn_s_sleep(2);
                        // never expected (the host reports it): the hand-over timed out, or the partner ran behind another L2
                        if ((f == 0 || f != 1 + (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15)) && p.status) atomicOr(p.status, 2);
                        __hip_atomic_store(p.sk_flags + partner, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __syncthreads();
                    const float* w = p.sk_parts + (size_t)partner * W4F_PART_FLOATS + wid * (32 * 256) + lane * 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {                   // one output row (8 KB per wave) at a time: 32 registers in flight, not 128
                        f32x4 t[4][2];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int b = 0; b < 2; ++b) t[j][b] = *reinterpret_cast<const f32x4*>(w + ((i * 4 + j) * 2 + b) * 256);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int b = 0; b < 2; ++b) Y[i][j][b] += t[j][b];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }

        // ---- epilogue: 16 outputs; output (i, j) of tile t goes to pixel (4 ty + i, 4 tx + j).
        // accumulator element e of block b: channel n0 + 16 wid + 4 (lane >> 4) + e, tile m0 + 16 b + (lane & 15): a lane holds 16 bytes of
        // consecutive channels per (output, block), a wave instruction writes 64-byte runs of 16 tiles; the four waves' runs of one tile make
        // up its 256 contiguous bytes (merged in L2).  Straight from the registers: no transpose, no barrier; the residual rows of the next
        // patch row are requested while this one is scaled and stored (through an LDS transpose with one barrier per output and the residual
        // requested per output, the epilogue was 25 % of the launch: profiles/r06e_wino4_ablations.txt).
        // SK, a range without the item's row 0 (RAW): the same code path leaves the partial Y in the hand-over area instead -- scale 1, shift 0,
        // no activation, no residual, part-relative addresses (one wave instruction = 1 KB) -- and then raises its flag.  (A separate block of
        // stores for this next to the gather above makes the register allocator spill 360 values around the epilogue.)
        const bool raw = SK && xa > 0;
        const int cw = n0 + 16 * wid + 4 * fq;                 // this lane's first channel
        const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 al = raw ? one4 : *reinterpret_cast<const f32x4*>(p.alpha + cw);
        const f32x4 be = raw ? zero4 : *reinterpret_cast<const f32x4*>(p.beta + cw);
        const bool leaky = !raw && p.act == YV3_ACT_LEAKY;
        const float* const res = raw ? nullptr : p.res;
        float* const ybase = raw ? p.sk_parts + (size_t)sk_t * W4F_PART_FLOATS + wid * (32 * 256) + lane * 4 : p.y;
        long long pix[2];                                      // pixel index of (b, 4 ty, 4 tx) of my two tiles, -1: no such tile
        int py[2], px[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int t = m0 + 16 * b + fr;
            if (t < p.T) {
                const int tt = p.th * p.tw;
                const int bi = t / tt, rem = t - bi * tt;
                const int ty = rem / p.tw, tx = rem - ty * p.tw;
                py[b] = 4 * ty; px[b] = 4 * tx;
                pix[b] = ((long long)bi * p.H + py[b]) * p.W + px[b];
            } else { pix[b] = -1; py[b] = px[b] = 0; }
        }
        auto out_off = [&](int i, int j, int b) -> long long {
            if (raw) return ((i * 4 + j) * 2 + b) * 256;
            const bool ok = pix[b] >= 0 && py[b] + i < p.H && px[b] + j < p.W;
            return ok ? (pix[b] + (long long)i * p.W + j) * p.Cout + cw : -1;
        };
        f32x4 rr[2][4][2];                                     // residual rows: [patch row parity][j][block]
        auto load_res = [&](int i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const long long off = out_off(i, j, b);
                    rr[i & 1][j][b] = off >= 0 ? *reinterpret_cast<const f32x4*>(res + off) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
        };
        if (res) load_res(0);
#pragma unroll
        for (int i = 0; i < ((ABL & 1) ? 1 : 4); ++i) {
            if (res && i + 1 < 4) load_res(i + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const long long off = out_off(i, j, b);
                    f32x4 q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = fmaf(Y[i][j][b][e], al[e], be[e]);
                        if (leaky) v = v > 0.f ? v : 0.1f * v;
                        q[e] = res ? v + rr[i & 1][j][b][e] : v;
                    }
                    if (off >= 0) *reinterpret_cast<f32x4*>(ybase + off) = q;
                }
        }
        if (raw) {
            // same XCD = same L2: once the stores are acknowledged (the vector L1 writes through) the partner can read them; the XCC id
            // travels with the flag and is checked by the reader (conv_planes.hip)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(p.sk_flags + sk_t, 1 + (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15),
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#if defined(YV3_MEASURE) && defined(W4F_TIMELINE)
        if (blockIdx.x == W4F_TIMELINE && lane == 0) {
            const unsigned long long tl_end = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0);
            float* dbg = p.y + wid * 8;
            dbg[0] = (float)tl_sync / nk; dbg[1] = (float)tl_burst / nk; dbg[2] = (float)tl_fold / nk; dbg[3] = (float)(tl_loop_end - tl_entry);
            dbg[4] = (float)(tl_end - tl_loop_end); dbg[5] = (float)nk;
        }
#endif
    }
}

}  // namespace

// U = G g G^T of a 3x3 filter bank, [cout][cin][6][6] fp32 (device) -> the GEMM stage's packed image (cout % 64 == 0, cin % 32 == 0)
extern "C" int yv3_pack_wino4_weight_f32(const float* u_oc66, float* packed, int cout, int cin, void* stream) {
    if (!u_oc66 || !packed || cout <= 0 || cin <= 0) return YV3_EINVAL;
    if (cout % 64 || (cin != 64 && cin % 128)) return YV3_ESHAPE;
    const long long n4 = (long long)cout * cin * 9;
    hipLaunchKernelGGL(wino4_pack_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u_oc66, packed, cout, cin);
    YV3_CHECK_LAUNCH();
    return 0;
}

// bytes of yv3_conv_desc.wino_ws for a B x H x W x cin input: V + the hand-over area of the even schedule (parts + flags: the LAST
// yv3_wino_sk_bytes() bytes of whatever buffer the caller passes, rounded down to 256 -- zero-filled once by the caller, like the
// F(2x2) stage's)
static size_t wino4_v_bytes(int B, int H, int W, int cin) { return (size_t)36 * B * ((H + 3) / 4) * ((W + 3) / 4) * cin * sizeof(float); }
extern "C" size_t yv3_wino4_workspace_bytes(int B, int H, int W, int cin) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0) return 0;
    return ((wino4_v_bytes(B, H, W, cin) + 255) & ~(size_t)255) + yv3_wino_sk_bytes();
}

// workgroups' worth of work of the GEMM stage = its items (the launch rule in conv_igemm_f32.hip counts them)
long long yv3_wino4_f32_workgroups(const yv3_conv_desc* d) {
    const long long T = (long long)d->B * ((d->H + 3) / 4) * ((d->W + 3) / 4);
    return ((T + BMT - 1) / BMT) * (d->cout / BNC);
}

// The even schedule of a launch: n_full whole-item workgroups + the other items cut into `parts` ranges of 6 / parts patch rows (parts = 1:
// one item per workgroup throughout).  Measured (tools/wino4_even_ab.py, profiles/r06w_wino4_even_parts_calibration.txt): cutting pays when
// it puts otherwise idle CUs to work -- a tail of few items behind full rounds (520 items: 512 + 8 x 6: 0.259 -> 0.209 ms), small batches
// (512->1024 @13x13, one image: 16 items, 0.173 -> 0.062 ms, where the direct kernel takes 0.111) -- and not when the tail already
// covers most CUs once (784 items: 272 x 3 parts 0.355 against 0.360 ms; 1352 items of 128 channels: slower): lone workgroups run their
// rows 2x faster than two per CU, and every part pays its ring fill, hand-over and flag.  The rule is that model, in chunk times:
//   rows per part x chunks per row x (1 while the parts leave one workgroup per CU, 2 up to two, 2 x rounds beyond) + 29 + 5 parts
// against the uncut tail; behind full rounds only cuts that stay at one workgroup per CU.  The smallest wins (checked against all 23
// measured shapes).  Not under YV3_OPT_WINO4_TILES (callers that share the GPU with other work, net.stream_k = False: a range with
// row 0 waits for its partners).  tune[1]: 1 never, 2 no full rounds (every item cut: measurements); tune[2]: parts forced.
static long long wino4_tail_cost(long long tail, int P, int nkx, int ncu, bool lone_only) {
    const long long W = tail * P;
    if (P > 1 && (8 * ((tail + 7) / 8) * P > W4F_MAX_TAIL_WG || (lone_only && W > ncu))) return -1;
    const long long f = W <= ncu ? 1 : W <= 2 * ncu ? 2 : 2 * ((W + 2 * ncu - 1) / (2 * ncu));
    return (6 / P) * nkx * f + (P > 1 ? 29 + 5 * P : 0);
}
static void wino4_schedule(const yv3_conv_desc* d, long long items, int* n_full, int* parts) {
    const int ncu = yv3_num_cu(), slots = 2 * ncu;
    *n_full = (int)items; *parts = 1;
    if ((d->options & YV3_OPT_WINO4_TILES) || d->tune[1] == 1 || items < 1) return;
    const long long full = d->tune[1] == 2 ? 0 : (items / slots) * slots;
    const long long tail = items - full;
    if (tail == 0) return;
    const int nkx = 6 * (d->cin / 32);
    int best = 1; long long best_cost = wino4_tail_cost(tail, 1, nkx, ncu, false);
    for (int P = 2; P <= 6; ++P) {
        if (6 % P) continue;
        const long long c = wino4_tail_cost(tail, P, nkx, ncu, full > 0);
        if (c >= 0 && c < best_cost) { best = P; best_cost = c; }
    }
    if ((d->tune[2] == 2 || d->tune[2] == 3 || d->tune[2] == 6) && wino4_tail_cost(tail, d->tune[2], nkx, ncu, false) >= 0) best = d->tune[2];
    if (best == 1) return;
    *n_full = (int)full; *parts = best;
}

// Is the F(4x4) form the fastest one of this launch?  From a number of items on, which depends on the channels (the direct kernel's
// competitiveness: it has 64 x 64 tiles for small launches) and on whether the even schedule may cut the items: >= 256 input channels:
// always (one 13x13 image, 16 items: 0.062 ms against the direct kernel's 0.111; 26x26: 0.053 / 0.059); 128: from 0.17 items per CU
// (44 items: 0.049 / 0.050; 24: 0.048 / 0.032); 64: from 0.39 (86 items: 0.044 / 0.039, 128: 0.044 / 0.048).  One item per workgroup
// only (YV3_OPT_WINO4_TILES): from 0.3 items per CU, the round-6 crossover (profiles/r06o_wino4_forms_by_batch.txt).
bool yv3_wino4_f32_pays(const yv3_conv_desc* d) {
    const long long items = yv3_wino4_f32_workgroups(d), ncu = yv3_num_cu();
    if ((d->options & YV3_OPT_WINO4_TILES) || d->tune[1] == 1) return items * 10 >= 3 * ncu;
    return d->cin >= 256 ? true : d->cin == 128 ? items * 100 >= 17 * ncu : items * 100 >= 39 * ncu;
}

int yv3_conv2d_wino4_f32(const yv3_conv_desc* d, hipStream_t s) {
    const int th = (d->H + 3) / 4, tw = (d->W + 3) / 4;
    const long long T = (long long)d->B * th * tw;
    if (T > 0x7fffffffLL || d->cout % BNC || d->cout_pad != d->cout || (d->cin != 64 && d->cin % 128) || d->k != 3 || d->stride != 1 || d->cin_up) return YV3_ESHAPE;
    if (!d->w_wino4) return YV3_EINVAL;
    if (!d->wino_ws || d->wino_ws_bytes < yv3_wino4_workspace_bytes(d->B, d->H, d->W, d->cin)) return YV3_EWORKSPACE;
    float* v = (float*)d->wino_ws;
    {
        const long long n = T * (d->cin >> 2);
        hipLaunchKernelGGL(wino4_input_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)d->x, v, d->H, d->W, d->cin, th, tw, T);
        YV3_CHECK_LAUNCH();
    }
    Wino4Params p;
    p.v = v; p.u = (const float*)d->w_wino4;
    p.alpha = d->alpha; p.beta = d->beta; p.res = (const float*)d->residual; p.y = (float*)d->y;
    p.C = d->cin; p.Cout = d->cout; p.H = d->H; p.W = d->W; p.th = th; p.tw = tw; p.T = (int)T;
    p.cchunks = d->cin / 32; p.nblk_n = d->cout / BNC; p.act = d->act;
    p.pos_stride = T * d->cin;
    if (!p.alpha) return YV3_EINVAL;
    const unsigned long long vb = 36ull * T * d->cin * 4, ub = 36ull * d->cout * d->cin * 4;
    if (vb > 0xffffffffull || ub > 0xffffffffull) return YV3_ESHAPE;        // (32-bit buffer offsets: V of at most 4 GB)
    p.u_bytes = (unsigned)ub; p.v_bytes = (unsigned)vb;
    const long long items = ((T + BMT - 1) / BMT) * p.nblk_n;
    if (items * 6 > 0x7fffffffLL) return YV3_ESHAPE;
    p.nitems = (int)items;
    wino4_schedule(d, items, &p.n_full, &p.parts);
    // hand-over area: the last yv3_wino_sk_bytes() of the buffer = 1024 parts of 128 KB; parts 0 .. 1022 are the tail workgroups', the last one
    // holds their flags
    char* area = (char*)d->wino_ws + ((d->wino_ws_bytes - yv3_wino_sk_bytes()) & ~(size_t)255);
    p.sk_parts = (float*)area;
    p.sk_flags = (int*)(area + (size_t)W4F_MAX_TAIL_WG * W4F_PART_FLOATS * sizeof(float));
    p.status = d->flags;
    const size_t lds = (size_t)NSTAGE * STAGE + W4F_EXTRA_LDS;
    if (p.parts > 1) {
        const long long tail = items - p.n_full;
        const dim3 grid((unsigned)(p.n_full + 8 * ((tail + 7) / 8) * p.parts));
        if (p.cchunks == 2) hipLaunchKernelGGL((conv_wino4_f32_kernel<true, true>), grid, dim3(256), lds, s, p);
        else                hipLaunchKernelGGL((conv_wino4_f32_kernel<false, true>), grid, dim3(256), lds, s, p);
    } else {
        const dim3 grid((unsigned)items);
        if (p.cchunks == 2) hipLaunchKernelGGL((conv_wino4_f32_kernel<true, false>), grid, dim3(256), lds, s, p);
        else                hipLaunchKernelGGL((conv_wino4_f32_kernel<false, false>), grid, dim3(256), lds, s, p);
    }
    YV3_CHECK_LAUNCH();
    return 0;
}
