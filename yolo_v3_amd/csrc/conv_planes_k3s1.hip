// 3x3 stride-1 convolution over bf16-plane tensors with KW-TAP REUSE of the activation tile.
//
// conv_planes_kernel re-fetches the pixel tile for each of the nine taps: per K chunk a 256x128 tile moves
// 48 KB of activations + 24 KB of weights from L2 to LDS, and with the matrix pipe at 6 bf16 MFMAs per
// product that DMA stream -- not the MFMAs -- bounds the loop.  For stride 1 and pad 1 the pixel needed by
// tile row r at tap kw is exactly the centre-tap pixel of tile row r + kw - 1 whenever both lie in the same
// image row, and is ZERO (padding) otherwise.  So this kernel stages ONE activation tile of BM+2 rows per
// (kh, 32-channel chunk) "super-chunk" and runs three K sub-chunks (kw = 0,1,2) out of it by reading the MFMA
// pixel fragments at row offsets 0/1/2, zeroing the fragments of lanes whose pixel sits on the left (kw=0) or
// right (kw=2) image border.  Activation DMA drops 3x; per sub-chunk the stream is 16 KB + 24 KB instead of 72 KB.
//
// LDS: activation stages double-buffered per super-chunk, weight stages double-buffered per sub-chunk
// (2*3*(BM+2)*64 + 2*3*BN*64 bytes = 145 KB for 256x128).  One raw s_barrier per sub-chunk, counted vmcnt,
// DMA pieces and the next k-step's fragment reads interleaved between MFMA units, same epilogue as
// conv_planes_kernel.  K order is unchanged (k = (kh*3+kw)*cin + c), so the packed weights are shared.
//
// Replaces reference darknet.py:43-44 / :52-53 for the 3x3 stride-1 layers (res_layer.conv2 and the odd
// layers of PreDetectionConvGroup): ~75 % of the network's FLOPs.
#include "conv_planes_common.h"

namespace {

template <int NP, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_planes_k3s1_kernel(const ConvParamsP p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int AR = BM + 2;                           // stage rows: tile rows -1 .. BM
    constexpr int A_PLANE = AR * ROWB, B_PLANE = BN * ROWB;
    constexpr int A_STAGE = NP * A_PLANE, B_STAGE = NP * B_PLANE;
    constexpr int AQ = BM / RPG / NW;                    // full 16-row DMA pieces per wave per plane
    constexpr int BROWS = BN / NW;
    static_assert(AQ >= 1 && AQ <= 2 && BM % (RPG * NW) == 0 && BROWS <= RPG, "tile/wave layout");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const a_lds = lds;                    // [2][NP][AR][64]
    unsigned char* const b_lds = lds + 2 * A_STAGE;      // [2][NP][BN][64]

    const int bid = yv3_xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (bid % p.ntiles) * BN;
    const int m0 = (bid / p.ntiles) * BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const bool tailw = wid == NW - 1;                    // this wave also stages the last two stage rows

    // ---- staging descriptors: stage row sr holds tile row sr-1 at the CENTRE tap (wi = wo) of input row ho-1+kh.
    // pieces q < AQ: sr = 16*(AQ*wid + q) + lane/4;  tail piece (q = AQ): sr = BM + lane/4 for lanes < 8.
    const int sslot = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;            // un-swizzled source slot (swz(sr) = (lane>>4)&3)
    long long aoff[AQ + 1];
    int ah0[AQ + 1];
    bool aok[AQ + 1];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int q = 0; q <= AQ; ++q) {
        const int sr = q < AQ ? RPG * (AQ * wid + q) + (lane >> 2) : BM + (lane >> 2);
        const int m = m0 + sr - 1;
        aok[q] = m >= 0 && m < p.M && (q < AQ || (tailw && lane < 8));
        const int mm = aok[q] ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        ah0[q] = ho - 1;
        aoff[q] = (((long long)b * p.H + (ho - 1)) * p.W + wo) * p.Cin + sslot;
    }
    const bool bact = lane < BROWS * SLOTS;
    const int brow = n0 % p.tb + BROWS * wid + (lane >> 2);
    const long long btile = (long long)(n0 / p.tb) * p.nk;
    const int bin = brow * PBK + (lane & 3) * 8;
    const int cchunks = p.Cin / PBK;
    const int nsuper = 3 * cchunks;                      // (kh, channel chunk) pairs

    // ---- DMA helpers
    const u16* ap[AQ + 1];
    long long aps[AQ + 1];
    unsigned char* adst = a_lds;
    auto a_prepare = [&](int S) {                        // S = kh * cchunks + cc
        const int kh = S / cchunks, cc = S - kh * cchunks;
        adst = a_lds + (S & 1) * A_STAGE;
#pragma unroll
        for (int q = 0; q <= AQ; ++q) {
            const bool ok = aok[q] && (unsigned)(ah0[q] + kh) < (unsigned)p.H;
            ap[q] = ok ? p.x + aoff[q] + (long long)kh * p.W * p.Cin + cc * PBK : g_zero_page;
            aps[q] = ok ? p.xs : 0;
        }
    };
    auto a_piece = [&](int q, int pl) {                  // q == AQ: the 2-row tail piece (8 lanes of the last wave)
        if (q < AQ) {
            __builtin_amdgcn_global_load_lds(GPTR(ap[q] + pl * aps[q]), LPTR(adst + pl * A_PLANE + (RPG * (AQ * wid + q)) * ROWB), 16, 0, 0);
        } else if (tailw) {
            if (lane < 8)
                __builtin_amdgcn_global_load_lds(GPTR(ap[AQ] + pl * aps[AQ]), LPTR(adst + pl * A_PLANE + BM * ROWB), 16, 0, 0);
        }
    };
    const u16* wbp = p.w;
    unsigned char* bdst = b_lds;
    auto b_prepare = [&](int s) {                        // s = S*3 + kw  ->  weight chunk (kh*3+kw)*cchunks + cc
        const int S = s / 3, kw = s - 3 * S;
        const int kh = S / cchunks, cc = S - kh * cchunks;
        const int kc = (kh * 3 + kw) * cchunks + cc;
        bdst = b_lds + (s & 1) * B_STAGE;
        wbp = p.w + ((btile + kc) * NP) * (long long)(p.tb * PBK) + bin;
    };
    auto b_piece = [&](int pl) {
        if (bact)
            __builtin_amdgcn_global_load_lds(GPTR(wbp + (long long)pl * (p.tb * PBK)), LPTR(bdst + pl * B_PLANE + wid * (BROWS * ROWB)), 16, 0, 0);
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    const int w_row = (wn * WTN + l31) * ROWB;
    const int w_sw = swz(l31);
    // image-border flags of this lane's pixels (one per m-tile): tap kw=0 reads wi = wo-1, kw=2 reads wo+1
    bool edge_l[MT], edge_r[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m0 + wm * WTM + j * 32 + l31;
        const int wo = m % p.Wo;
        edge_l[j] = wo == 0;
        edge_r[j] = wo == p.Wo - 1;
    }

    constexpr int KS = PBK / 16;
    constexpr int NF = (NT + MT) * NP;
    constexpr int NU = NT * MT;
    bf16x8v frag[KS][NF];
    const unsigned char* bst = b_lds;
    const unsigned char* ast = a_lds;
    int kwcur = 0;
    auto read_frag = [&](int ks, int f) {
        if (f < NT * NP) {
            const int fslot = ((ks * 2 + lhi) ^ w_sw) * 16;
            frag[ks][f] = *reinterpret_cast<const bf16x8v*>(bst + w_row + fslot + (f / NP) * 32 * ROWB + (f % NP) * B_PLANE);
        } else {
            const int g = f - NT * NP;
            const int srow = wm * WTM + (g / NP) * 32 + l31 + kwcur;      // stage row = tile row + 1 + (kw - 1)
            const int fslot = ((ks * 2 + lhi) ^ swz(srow)) * 16;
            frag[ks][f] = *reinterpret_cast<const bf16x8v*>(ast + srow * ROWB + fslot + (g % NP) * A_PLANE);
        }
    };

    // ---- prologue: activation super-chunk 0 and weight sub-chunk 0
    a_prepare(0);
#pragma unroll
    for (int q = 0; q <= AQ; ++q)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) a_piece(q, pl);
    b_prepare(0);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) b_piece(pl);

    int s = 0;
    for (int S = 0; S < nsuper; ++S) {
        const bool haveA = S + 1 < nsuper;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw, ++s) {
            // what may still be in flight: only the activation pieces of super-chunk S+1 issued in the PREVIOUS
            // sub-chunk (they were issued after that sub-chunk's weight pieces, so they are the youngest)
            if (kw == 0 || !haveA) wait_vmcnt<0>();
            else if (kw == 1) wait_vmcnt<NP>();                                   // kw=0 issued piece q=0 of every plane
            else { if (tailw) wait_vmcnt<(AQ - 1) * NP + NP>(); else wait_vmcnt<(AQ - 1) * NP>(); }
            __builtin_amdgcn_s_barrier();
            ast = a_lds + (S & 1) * A_STAGE;
            bst = b_lds + (s & 1) * B_STAGE;
            kwcur = kw;
            const bool moreB = s + 1 < p.nk;
#pragma unroll
            for (int f = 0; f < NF; ++f) read_frag(0, f);
            if (moreB) b_prepare(s + 1);
            if (haveA && kw == 0) a_prepare(S + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int gu = ks * NU + u;
                    if (ks + 1 < KS) {
#pragma unroll
                        for (int f = u * NF / NU; f < (u + 1) * NF / NU; ++f) read_frag(ks + 1, f);
                    }
                    // DMA ops of this sub-chunk, spread over its MFMA units: the NP weight pieces of sub-chunk s+1
                    // FIRST (so that the activation pieces are the youngest in flight, see the waits above), then
                    // this sub-chunk's share of super-chunk S+1: piece q=0 during kw=0, piece q=1 and the 2-row tail
                    // during kw=1, nothing during kw=2.
                    constexpr int NUN = KS * NU;
                    const int nops = kw == 0 ? 2 * NP : kw == 1 ? NP + (AQ > 1 ? NP : 0) + NP : NP;
#pragma unroll
                    for (int o = gu * nops / NUN; o < (gu + 1) * nops / NUN; ++o) {
                        if (o < NP) { if (moreB) b_piece(o); }
                        else if (haveA) {
                            const int o2 = o - NP;
                            if (kw == 0) a_piece(0, o2);
                            else if (AQ > 1 && o2 < NP) a_piece(AQ - 1, o2);
                            else a_piece(AQ, AQ > 1 ? o2 - NP : o2);
                        }
                    }
                    const int i = u / MT, j = u % MT;
                    bf16x8v* xf = &frag[ks][NT * NP + j * NP];
                    if (i == 0 && kw != 1) {                  // zero the pixels that fall off the image row
                        const bool z = kw == 0 ? edge_l[j] : edge_r[j];
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                            for (int e = 0; e < 8; ++e) xf[pl][e] = z ? (short)0 : xf[pl][e];
                    }
                    const bf16x8v* wf = &frag[ks][i * NP];
                    f32x16 c = acc[i][j];
                    c = mfma_unit<NP>(wf, xf, c);
                    acc[i][j] = c;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    epilogue_store<NP, BM, BN, WM, WN, false>(acc, p, lds, m0, n0, wid, lane);
}

template <int NP, int BM, int BN, int WM, int WN>
int launch_k3s1(const ConvParamsP& p, hipStream_t s) {
    const int mtiles = (p.M + BM - 1) / BM;
    const dim3 grid((unsigned)(mtiles * p.ntiles));
    const size_t pipe = (size_t)2 * NP * (BM + 2 + BN) * ROWB;
    const size_t epi = (size_t)WN * BM * (BN / WN + 4) * 4;
    const size_t lds = pipe > epi ? pipe : epi;
    hipLaunchKernelGGL((conv_planes_k3s1_kernel<NP, BM, BN, WM, WN>), grid, dim3(64 * WM * WN), lds, s, p);
    YV3_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Called by yv3_conv2d_planes for k=3, stride=1, no dual source, plane output.  Returns -100 when the shape is
// not covered (the caller then uses the generic kernel).
int yv3_conv2d_planes_k3s1(const ConvParamsP* pp, int np, int npad, long long M, hipStream_t s) {
    ConvParamsP p = *pp;
    if (p.Cin % PBK) return -100;
#define YV3_K3(BM_, BN_, WM_, WN_) (np == 3 ? launch_k3s1<3, BM_, BN_, WM_, WN_>(p, s) : np == 2 ? launch_k3s1<2, BM_, BN_, WM_, WN_>(p, s) : launch_k3s1<1, BM_, BN_, WM_, WN_>(p, s))
    if (npad % 128 == 0) {
        const long long blocks256 = ((M + 255) / 256) * (npad / 128);
        p.ntiles = npad / 128;
        if (blocks256 >= 512) return YV3_K3(256, 128, 4, 2);
        return YV3_K3(128, 128, 4, 2);
    }
    if (npad % 64 == 0) { p.ntiles = npad / 64; return YV3_K3(128, 64, 2, 2); }
    return -100;
#undef YV3_K3
}
