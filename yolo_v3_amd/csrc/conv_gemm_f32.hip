// 1x1 convolution (stride 1, no residual, no upsample operand) of the exact-fp32 mode as a PERSISTENT, DMA-fed GEMM (round 6); the same
// kernel carries a 3x3 operand path (K3, below) that is tested but not dispatched.
//
//   y[m][n] = act( sum_k x[m][k] * w[n][k] * alpha[n] + beta[n] )        m = pixel (NHWC: a row of cin floats), n = output channel
//
// conv_igemm_f32.hip runs these layers on 64x64 tiles, four workgroups per CU, operands staged global -> registers -> LDS: 74-110 TFLOP/s
// of the 157 fp32 MFMA peak (profiles/r06z_layers.txt; 136-144 is what the instruction gives on random data).  With K = cin of only 8-32 chunks a tile's index arithmetic, first-load latency and
// epilogue are a third of its life, and the four residents of a CU go through those phases together.  Here:
//   * one workgroup per CU that walks its share of the tiles (XCD-contiguous ranges, n fastest: the tiles of one pixel block follow each
//     other behind one L2);
//   * operands reach LDS by DMA (buffer_load ... lds, 16 bytes per lane: one wave instruction = 8 rows of one 32-float chunk), rows
//     XOR-swizzled by 16-byte slot ON THE SOURCE SIDE (the weights stay in yv3_pack_conv_weight's plain [cout][K] layout), conflict-free
//     ds_read_b128 fragments -- the ring runs NST-1 chunks ahead of the multiplication ACROSS tile boundaries, so the next tile's first
//     chunks are in LDS before the current tile's epilogue starts;
//   * eight waves, 64x32 wave tiles (two 32x32 accumulator blocks sharing the weight fragment), one barrier per chunk, the next
//     chunk's fragments read under this chunk's 32 MFMAs;
//   * the epilogue stores straight from the accumulators by buffer stores (a wave instruction writes two rows x 128 contiguous bytes; rows
//     past the last pixel fall outside the descriptor), scale / shift from LDS: no loads in the in-order vmcnt queue, no barrier;
//   * whole rounds of the chip only: the rows of a last round that is at most half full go to conv_igemm_f32.hip's small tiles.
// Measured per layer incl. that rest launch (profiles/r06ae_*, r06am_*): 256->128 @52 0.122 -> 0.109 ms, 512->256 @26 0.107 -> 0.100,
// 1024->512 @13 0.108 -> 0.105, 128->64 @104 0.154 -> 0.118; two-lane headline +1.1 % (profiles/r06an_*).
// K order per output element = conv_igemm_f32.hip's (chunk by chunk, k pairs (8 kk + t, 8 kk + 4 + t)): bit-identical results.
//
// Replaces reference darknet.py:43-44 (conv_bn_relu.forward, kernel 1) for the layers the launch rule below takes.
#include "yv3_common.h"

namespace {

#define G1_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

struct Gemm1Params {
    const float* x;
    const float* w;
    const float* alpha;      // may be NULL (1.0)
    const float* beta;
    float* y;
    int M, K, N;
    int nch;                 // K / 32
    int ntn;                 // N tiles
    int ntiles;              // all tiles
    int act;
    unsigned x_bytes, w_bytes, y_bytes;
    // K3 (3x3, pad 1): input picture, output grid, stride; K = 9 cin, k = (kh, kw, cin) as conv_igemm_f32.hip
    int H, W, Cin, Ho, Wo, stride;
    int cch;                 // cin / 32
};

#ifndef G1_NST_
#define G1_NST_ 3
#endif
constexpr int G1_NST = G1_NST_;  // ring stages (4 -- 128 KB, 128x128 tile only -- measured: +-0.5 %, profiles/r06am_gemm1x1_ring_depth4_ab.txt)
template <int N> __device__ __forceinline__ void g1_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ int g1_swz(int row) { return (row >> 1) & 7; }

// WM x WN waves, wave tile 64 (pixels) x 32 (channels): workgroup tile BM = 64 WM, BN = 32 WN
// K3: 3x3 / pad 1 layer (any stride) as an implicit GEMM: the A row of output pixel (b, ho, wo) and chunk (kh, kw, c0) is the 128 contiguous
// bytes of input pixel (ho s + kh - 1, wo s + kw - 1) at channel c0 -- one request address per row and TAP (recomputed when the tap
// changes, every cin / 32 chunks), the channel chunk in the scalar offset; halo rows and rows past the last pixel get an offset outside the
// buffer descriptor: the hardware fills their LDS rows with zeros
template <int WM, int WN, bool K3>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_gemm1x1_f32_kernel(const Gemm1Params p) {
    constexpr int NW = WM * WN, BM = 64 * WM, BN = 32 * WN;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW, PER = PA + PB;       // DMA instructions per wave and chunk
    static_assert(PA * 8 * NW == BM && PB * 8 * NW == BN && PB >= 1, "every wave copies whole 8-row pieces");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;

    // ---- this workgroup's tiles: XCD x (= blockIdx % 8) owns the contiguous range [x T / 8, (x + 1) T / 8), its workgroups stride through it
    const int x8 = blockIdx.x & 7, loc = blockIdx.x >> 3, L = gridDim.x >> 3;
    const int t0 = (int)(((long long)p.ntiles * x8) >> 3), t1 = (int)(((long long)p.ntiles * (x8 + 1)) >> 3);
    int my = t0 + loc < t1 ? (t1 - t0 - loc + L - 1) / L : 0;              // tiles of this workgroup: t0 + loc + j L
    if (my == 0) return;

    // scale / shift of all N <= 1024 channels into LDS once (behind the ring): the epilogue must not put loads into the vector-memory queue --
    // vmcnt counts in issue order, and a wait for them would also wait for every operand request in flight
    float* const ab = reinterpret_cast<float*>(lds + G1_NST * STAGE);     // [2][N]
    for (int i = tid; i < p.N; i += 64 * NW) { ab[i] = p.alpha ? p.alpha[i] : 1.f; ab[p.N + i] = p.beta[i]; }

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);

    // ---- producer state: the chunk being requested = chunk pf_c of tile pf_t
    int pf_t = t0 + loc, pf_c = 0, pf_left = my;                            // (pf_left: tiles not yet fully requested)
    unsigned a_soff = 0, b_soff = 0;
    unsigned a_voff[PA], b_voff[PB];
    const int prow = lane >> 3, pslot = lane & 7;
    auto tile_origin = [&](int t, int& m0, int& n0) { n0 = (t % p.ntn) * BN; m0 = (t / p.ntn) * BM; };
    unsigned a_pix[K3 ? PA : 1];                                            // K3: byte offset of input pixel (hi0, wi0) = (ho s - 1, wo s - 1), modulo 2^32
    int a_hi[K3 ? PA : 1], a_wi[K3 ? PA : 1];                               //     hi0, wi0; hi0 = INT_MIN / 2: no such row (past the last pixel)
    int pf_kh = 0, pf_kw = 0, pf_cc = 0;                                    // K3: tap and channel chunk of the chunk being requested
    auto producer_tap = [&]() {                                             // K3: request addresses of tap (pf_kh, pf_kw)
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = 8 * (wid * PA + i) + prow;
            const bool ok = (unsigned)(a_hi[i] + pf_kh) < (unsigned)p.H && (unsigned)(a_wi[i] + pf_kw) < (unsigned)p.W;
            a_voff[i] = ok ? a_pix[i] + (unsigned)((pf_kh * p.W + pf_kw) * p.Cin * 4) + (unsigned)((pslot ^ g1_swz(r)) << 4) : 0xffffff00u;
        }
    };
    auto producer_tile = [&]() {                                            // bases of tile pf_t
        int m0, n0; tile_origin(pf_t, m0, n0);
        b_soff = (unsigned)n0 * (unsigned)p.K * 4u;
        if constexpr (K3) {
            a_soff = 0; pf_kh = pf_kw = pf_cc = 0;
            const int HoWo = p.Ho * p.Wo;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int m = m0 + 8 * (wid * PA + i) + prow;
                const int b = m / HoWo, rem = m - b * HoWo;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                a_hi[i] = m < p.M ? ho * p.stride - 1 : -0x40000000; a_wi[i] = wo * p.stride - 1;
                // ((b H + hi) W + wi) cin 4 of a valid tap is < 2^32 (host check); hi0 / wi0 = -1 make the base "negative" by at most
                // (W + 1) cin 4: base + tap offset in 32-bit wrap-around arithmetic
                a_pix[i] = (unsigned)((b * p.H + ho * p.stride - 1) * p.W + a_wi[i]) * (unsigned)p.Cin * 4u;
            }
            producer_tap();
        } else {
            a_soff = (unsigned)m0 * (unsigned)p.K * 4u;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int r = 8 * (wid * PA + i) + prow;
                const int rr = min(m0 + r, p.M - 1) - m0;                   // (rows past the last pixel re-read it; never stored)
                a_voff[i] = (unsigned)rr * (unsigned)p.K * 4u + (unsigned)((pslot ^ g1_swz(r)) << 4);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int r = 8 * (wid * PB + i) + prow;
        b_voff[i] = (unsigned)r * (unsigned)p.K * 4u + (unsigned)((pslot ^ g1_swz(r)) << 4);
    }
    producer_tile();
    auto request = [&](int stage) {                                         // PER requests; past the last chunk they repeat it (exact vmcnt counts)
        unsigned char* st = lds + stage * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const unsigned vo = a_voff[i];             // (an array element as the builtin's argument: the host pass silently drops the kernel's launch stub)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, G1_LPTR(st + (wid * PA + i) * 1024), 16, vo, a_soff, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const unsigned vo = b_voff[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, G1_LPTR(st + A_BYTES + (wid * PB + i) * 1024), 16, vo, b_soff, 0, 0);
        }
        if (pf_left > 0) {
            if (++pf_c < p.nch) {
                b_soff += 128;
                if constexpr (K3) {
                    if (++pf_cc < p.cch) a_soff += 128;
                    else { pf_cc = 0; a_soff = 0; if (++pf_kw == 3) { pf_kw = 0; ++pf_kh; } producer_tap(); }
                } else a_soff += 128;
            }
            else if (--pf_left > 0) { pf_c = 0; pf_t += L; producer_tile(); }
            else { pf_c = p.nch - 1; }                                       // the very last chunk stays the one repeated
        }
    };

    // ---- fragment addresses (bytes within a stage): lane -> row l31 of each 32-row block, 16-byte slot 2 kk + (lane >> 5)
    const int l31 = lane & 31, lhi = lane >> 5;
    int a_off[2][4], b_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wm * 64 + i * 32 + l31;
            a_off[i][kk] = r * 128 + (((2 * kk + lhi) ^ g1_swz(r)) << 4);
        }
        const int r = wn * 32 + l31;
        b_off[kk] = A_BYTES + r * 128 + (((2 * kk + lhi) ^ g1_swz(r)) << 4);
    }
    struct Frag { f32x4 a[2][4], b[4]; };
    auto read_frag = [&](int stage) {
        Frag f;
        const unsigned char* st = lds + stage * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f.a[0][kk] = *reinterpret_cast<const f32x4*>(st + a_off[0][kk]);
            f.a[1][kk] = *reinterpret_cast<const f32x4*>(st + a_off[1][kk]);
            f.b[kk] = *reinterpret_cast<const f32x4*>(st + b_off[kk]);
        }
        return f;
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // ring fill
#pragma unroll
    for (int d = 0; d < G1_NST; ++d) request(d);
    g1_wait_vmcnt<PER * (G1_NST - 1)>();
    __builtin_amdgcn_s_barrier();
    Frag F0 = read_frag(0), F1;
    int stage = 0;
    int t = t0 + loc, c = 0;                                                // consumer: chunk c of tile t

    // one chunk, its fragments (F) already in registers: [my pieces of the next chunk have landed] [my fragment reads of this chunk are out
    // of its stage] [barrier: the next chunk is complete, this chunk's stage is free] [request chunk + NST into it, read the next chunk's
    // fragments into G: both under this chunk's 32 MFMAs]; behind a tile's last chunk: the epilogue, straight from the accumulators
    // (C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)) -- the next tile's chunks are already in flight
#if defined(YV3_MEASURE) && defined(G1_TIMELINE)           // cycle split of one workgroup -> the first floats of y (results INVALID)
    unsigned long long tl_t = __builtin_amdgcn_s_memtime(), tl_wait = 0, tl_bar = 0, tl_burst = 0, tl_epi = 0;
    const unsigned long long tl_entry = tl_t;
#define G1_MARK(acc_) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_ += t_ - tl_t; tl_t = t_; } while (0)
#else
#define G1_MARK(acc_) do {} while (0)
#endif
    // The epilogue's 32 stores per lane enter the same in-order queue as the operand requests (gfx9: one counter for loads and stores) BEHIND
    // the requests the next two chunks wait for: those two waits allow 32 more operations in flight, so that the stores have two chunk
    // times to be acknowledged instead of stalling the first chunk of the next tile (17-38 k cycles per tile before: profiles/r06ab_*)
    int after_epi = 0;
    auto chunk = [&](const Frag& F, Frag& G) {
        G1_MARK(tl_epi);
        if (after_epi > 0) { --after_epi; g1_wait_vmcnt<PER * (G1_NST - 2) + 32>(); }
        else g1_wait_vmcnt<PER * (G1_NST - 2)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        G1_MARK(tl_wait);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        G1_MARK(tl_bar);
        const int nxt = stage + 1 == G1_NST ? 0 : stage + 1;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a[0][kk][q], F.b[kk][q], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a[1][kk][q], F.b[kk][q], acc[1], 0, 0, 0);
            }
            // the next chunk's twelve fragment reads and this stage's refill (~16 instructions, the requests stall their wave for 60-250
            // cycles each): behind the first eight MFMAs for waves 0-3, behind sixteen for their SIMD partners 4-7 -- when both waves of a
            // SIMD did this at the same point the matrix pipe sat idle for ~500 of a chunk's 4600 cycles (profiles/r06ad_*)
            if (kk == (wid < NW / 2 ? 0 : 1)) { __builtin_amdgcn_sched_barrier(0); G = read_frag(nxt); request(stage); __builtin_amdgcn_sched_barrier(0); }
        }
        __builtin_amdgcn_s_setprio(0);
        G1_MARK(tl_burst);
        stage = nxt;
        if (++c < p.nch) return;
        c = 0;
        int m0, n0; tile_origin(t, m0, n0);
        t += L;
        const int n = n0 + wn * 32 + l31;
        const float al = ab[n], be = ab[p.N + n];
        const bool leaky = p.act == YV3_ACT_LEAKY;
        after_epi = G1_NST - 1;
        // buffer stores: the row enters the per-lane offset, so rows past the last pixel fall outside the descriptor and are dropped by the
        // hardware (no compare / exec juggling); one vector add per store
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned vb = (unsigned)(((long long)(m0 + wm * 64 + i * 32 + 4 * lhi) * p.N + n) * 4);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = (e & 3) + 8 * (e >> 2);
                float v = fmaf(acc[i][e], al, be);
                if (leaky) v = fmaxf(v, 0.1f * v);                          // == v > 0 ? v : 0.1f * v for every v (signed zeros, NaN included)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, vb + (unsigned)(ro * p.N * 4), 0, 0);
                acc[i][e] = 0.f;
            }
        }
    };
    const int total = my * p.nch;
    int g = 0;
    for (; g + 1 < total; g += 2) { chunk(F0, F1); chunk(F1, F0); }
    if (g < total) chunk(F0, F1);
#if defined(YV3_MEASURE) && defined(G1_TIMELINE)
    if (blockIdx.x == G1_TIMELINE && lane == 0) {
        G1_MARK(tl_epi);
        __builtin_amdgcn_s_waitcnt(0);
        float* dbg = p.y + wid * 8;
        dbg[0] = (float)tl_wait / total; dbg[1] = (float)tl_bar / total; dbg[2] = (float)tl_burst / total; dbg[3] = (float)tl_epi / my;
        dbg[4] = (float)(tl_t - tl_entry); dbg[5] = (float)total; dbg[6] = (float)my;
    }
#endif
    g1_wait_vmcnt<0>();                                    // (the repeated requests of the last chunks: nothing may land in a retired workgroup's LDS)
}




}  // namespace

// Does the persistent GEMM take this fp32 descriptor?  Plain 1x1 / stride-1 layers and 3x3 layers (any stride; the caller has ruled the
// Winograd forms out) without residual or upsample operand whose tiles fill at least one round of the chip (tune[0] == 13: never, 14:
// whenever the shape fits)
static void gemm1_tiling(const yv3_conv_desc* d, int* bm, int* bn) { const bool wide = d->cout % 128 == 0; *bm = wide ? 128 : 256; *bn = wide ? 128 : 64; }
bool yv3_gemm1x1_f32_takes(const yv3_conv_desc* d) {
    // (3x3 layers: built, bit-identical and measured -- 113-122 TFLOP/s where the 128x128 eight-wave tiles give 116-124: the long K loops
    // of these layers amortise a tile's prologue and epilogue anyway, profiles/r06ag_gemm_k3_stride2_ab.txt -- taken with tune[0] == 14 only)
    const bool k1 = d->k == 1 && d->stride == 1, k3 = d->k == 3 && d->cout % 128 == 0 && d->tune[0] == 14;
    if (!(k1 || k3) || d->cin_up || d->residual || d->cin % 32 || d->cout % 64 || d->cout > 1024 || d->cout_pad != d->cout || d->tune[0] == 13) return false;
    const int pad = (d->k - 1) / 2;
    const long long Ho = (d->H + 2 * pad - d->k) / d->stride + 1, Wo = (d->W + 2 * pad - d->k) / d->stride + 1;
    const long long M = (long long)d->B * Ho * Wo;
    if ((long long)d->B * d->H * d->W * d->cin * 4 > 0xfffffe00LL || M * d->cout * 4 > 0xffffffffLL || (long long)d->cout * d->cin * d->k * d->k * 4 > 0xffffffffLL) return false;
    int bm, bn; gemm1_tiling(d, &bm, &bn);
    const long long tiles = ((M + bm - 1) / bm) * (d->cout / bn);
    return d->tune[0] == 14 || tiles >= (long long)yv3_num_cu();
}

// Pixel-row tiles that run on the GEMM: whole rounds of the chip, plus the last partial round when it is more than half full (measured,
// profiles/r06am_gemm1x1_ring_depth4_ab.txt: a rest of 0.28 / 0.33 rounds is cheaper on the small tiles, which fill the chip two to four to a
// CU -- 0.109 vs 0.111 ms, 0.106 vs 0.125; a rest of 0.64 rounds is cheaper as one more round here: 0.100 vs 0.103, 0.054 vs 0.057)
static long long gemm1_mtiles_here(const yv3_conv_desc* d, long long M, int bm, int bn) {
    const int ncu = yv3_num_cu(), ntn = d->cout / bn;
    const long long mt = (M + bm - 1) / bm;
    if (d->tune[1] == 3) return mt;                                        // (every row here -- measurements)
    const long long rounds = mt * ntn / ncu, rest = mt * ntn - rounds * ncu;
    return rounds >= 1 && rest > 0 && 2 * rest <= ncu ? rounds * ncu / ntn : mt;
}

// Kernel launches of a descriptor yv3_gemm1x1_f32_takes says yes to (1, or 2 with rows left for the small tiles)
int yv3_gemm1x1_f32_launches(const yv3_conv_desc* d) {
    const int pad = (d->k - 1) / 2;
    const long long Ho = (d->H + 2 * pad - d->k) / d->stride + 1, Wo = (d->W + 2 * pad - d->k) / d->stride + 1;
    const long long M = (long long)d->B * Ho * Wo;
    int bm, bn; gemm1_tiling(d, &bm, &bn);
    return gemm1_mtiles_here(d, M, bm, bn) * bm < M ? 2 : 1;
}

// Launches the GEMM on output pixels [0, *rows_done): whole rounds of the chip (one tile per CU and round) -- plus the last, partial round when
// it is more than half full.  The caller runs the remaining rows (< half a round of tiles) on conv_igemm_f32.hip's tiles, which fill
// the chip two to four to a CU and finish in a fraction of a round here: 1352 tiles = 5.28 rounds would cost 6 (same K order: same bits).
int yv3_conv2d_gemm1x1_f32(const yv3_conv_desc* d, hipStream_t s, long long* rows_done) {
    Gemm1Params p;
    p.x = (const float*)d->x; p.w = (const float*)d->w; p.alpha = d->alpha; p.beta = d->beta; p.y = (float*)d->y;
    const int pad = (d->k - 1) / 2;
    p.H = d->H; p.W = d->W; p.Cin = d->cin; p.stride = d->stride; p.cch = d->cin / 32;
    p.Ho = (d->H + 2 * pad - d->k) / d->stride + 1; p.Wo = (d->W + 2 * pad - d->k) / d->stride + 1;
    const long long M = (long long)d->B * p.Ho * p.Wo;
    if (M > 0x7fffffffLL) return YV3_ESHAPE;
    int bm, bn; gemm1_tiling(d, &bm, &bn);
    const int ntn = d->cout / bn;
    const int ncu = yv3_num_cu();
    const long long mt_run = gemm1_mtiles_here(d, M, bm, bn);
    const long long Mr = mt_run * bm < M ? mt_run * bm : M;
    *rows_done = Mr;
    p.M = (int)Mr; p.K = d->k * d->k * d->cin; p.N = d->cout; p.nch = p.K / 32; p.act = d->act;
    p.x_bytes = (unsigned)((long long)d->B * d->H * d->W * d->cin * 4); p.y_bytes = (unsigned)(Mr * d->cout * 4);
    p.w_bytes = (unsigned)((long long)d->cout * p.K * 4);
    p.ntn = ntn;
    const long long tiles = mt_run * ntn;
    p.ntiles = (int)tiles;
    const int grid = (int)(tiles < ncu ? (tiles + 7) / 8 * 8 : ncu);
    const size_t lds = (size_t)G1_NST * (bm + bn) * 128 + 2 * d->cout * sizeof(float);
    if (d->k == 3)      hipLaunchKernelGGL((conv_gemm1x1_f32_kernel<2, 4, true>), dim3(grid), dim3(512), lds, s, p);
    else if (bn == 128) hipLaunchKernelGGL((conv_gemm1x1_f32_kernel<2, 4, false>), dim3(grid), dim3(512), lds, s, p);
    else                hipLaunchKernelGGL((conv_gemm1x1_f32_kernel<4, 2, false>), dim3(grid), dim3(512), lds, s, p);
    YV3_CHECK_LAUNCH();
    return 0;
}
