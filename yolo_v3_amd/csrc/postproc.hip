// Post-processing on the GPU: confidence filter, per-class greedy NMS, box geometry helpers.
//
// Replaces reference utils.postprocessing (utils.py:226-258), get_nms_detections
// (utils.py:148-202: a Python loop with one .item() per box = 73 % of the reference's
// post-processing time), get_raw_detections (utils.py:204-224), iou_vectorized
// (utils.py:98-119), bbox_iou (utils.py:122-146), bbox_cxcywh_to_x1y1x2y2 (boundingbox.py:25-29).
//
// Pipeline (all integer/compare work is bit-exact w.r.t. the reference on identical inputs):
//   filter   one wave per 64 rows: gather conf, skip rows with conf <= thr (cls <= 1 so the
//            product cannot pass), wave-cooperative score = cls*conf, max / first argmax,
//            append a 64-bit key per candidate:  cls<<52 | ~score_bits<<20 | row
//   rank     ascending key order == (class asc, score desc, row asc) == the reference's per-class
//            stable descending sort: keys are scattered into their class segments, then
//            rank = segment start + #smaller keys of the segment (compares through LDS tiles,
//            sum_c n_c^2 of them; deterministic), box records scattered in sorted order
//            (use_nms = 0: (row, class) order, rank = #smaller keys of the image)
//   mask     64x64 IOU tiles, one wave each, 64-bit ballots "j later, same class, IOU > thr"
//   scan     one wave per (image, class) segment, 64 boxes per step: the in-word greedy chain
//            runs on v_readlane'd mask words, kept rows are OR-ed into the removed set with
//            independent loads
//   compact  per image prefix sum of keep flags -> out[B,cap,7]
#include "yv3_common.h"
#include <type_traits>

namespace {

typedef unsigned long long u64;

constexpr int ROW_BITS = 20;                 // rows per image < 2^20
constexpr u64 ROW_MASK = (1ull << ROW_BITS) - 1;
constexpr int CLS_SHIFT = 52;                // 32 score bits in [20,52), class above

// Order-preserving map float -> uint32 for ALL finite floats (negative scores pass a negative obj_conf_thr):
// flip every bit of a negative number, only the sign bit of a non-negative one; ascending uint == ascending float.
__device__ inline uint32_t ord_bits(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ inline float ord_float(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ inline u64 make_key(int cls, float score, int row) {
    const uint32_t sb = ~ord_bits(score + 0.0f);               // (-0 -> +0: equal scores tie on the row, as torch.sort does); inverted: ascending key == descending score
    return ((u64)cls << CLS_SHIFT) | ((u64)sb << ROW_BITS) | (u64)row;
}
__device__ inline int key_cls(u64 k) { return (int)(k >> CLS_SHIFT); }
__device__ inline int key_row(u64 k) { return (int)(k & ROW_MASK); }
__device__ inline float key_score(u64 k) { return ord_float(~(uint32_t)(k >> ROW_BITS)); }

// Sort partitions.  The order is (class asc, score desc, row asc); the NMS path partitions the keys by class, splits every LARGE
// class into up to NB sub-partitions at splitter keys taken from a sample of the class (subpart_kernel), and only ranks inside
// a partition: sum over partitions of n^2 compares instead of sum over classes.  Partition order == key order by construction.
constexpr int NB = 32;                                                         // partition slots per class
__host__ __device__ inline int nbuckets(int C) { return C <= 256 ? NB : 1; }   // (per-image LDS tables are C * nbuckets ints)
// torch.max / torch.min propagate NaN; C fmaxf/fminf do not.
__device__ inline float tmax(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ inline float tmin(float a, float b) { return (a < b || a != a) ? a : b; }
__device__ inline float clamp0(float v) { return v < 0.f ? 0.f : v; }        // torch.clamp(min=0), NaN stays NaN

// reference utils.py:98-119 / 122-146, x1y1x2y2
__device__ inline float iou_xyxy(const f32x4 a, const f32x4 b) {
    const float iw = clamp0(tmin(a[2], b[2]) - tmax(a[0], b[0]));
    const float ih = clamp0(tmin(a[3], b[3]) - tmax(a[1], b[1]));
    const float inter = iw * ih;
    const float aa = (a[2] - a[0]) * (a[3] - a[1]);
    const float ab = (b[2] - b[0]) * (b[3] - b[1]);
    return inter / ((ab + aa) - inter);
}

// reference boundingbox.py:25-29
__device__ inline f32x4 to_xyxy(float cx, float cy, float w, float h) {
    f32x4 r;
    r[0] = cx - w / 2.f; r[1] = cy - h / 2.f; r[2] = cx + w / 2.f; r[3] = cy + h / 2.f;
    return r;
}

// ------------------------------------------------------------------------------ filter
struct CandView {
    u64* keys;         // [B][max_cand]
    int* segcnt;       // [B][C] candidates per class
};
__host__ __device__ inline size_t cand_keys_bytes(int B, int max_cand) { return ((size_t)B * max_cand * 8 + 255) & ~(size_t)255; }

__global__ void zero_kernel(int* a, int na, int* b, int nb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) a[i] = 0;
    if (i < nb) b[i] = 0;
}

// STAGED: a wave with many rows to score (a dense scene; eval mode, where nearly every row passes the objectness screen)
// copies its rows to LDS with coalesced loads, 32 rows at a time, and scores them one row per lane pair (each lane half of the
// classes, LDS stride 5 + C words: conflict-free for odd 5 + C): no load -> reduce chain per row, 80 instead of ~50 x 64
// instructions per 64 rows.  Same products, same first-index argmax, same candidates (their order in the key buffer is free:
// the keys are sorted afterwards).
constexpr int FILTER_STAGE_MIN = 12;             // rows of a wave that must be scored before staging pays
template <bool EVAL, bool STAGED>
__global__ __launch_bounds__(256) void filter_kernel(const float* __restrict__ dets, int N, int C, float thr, bool prob,
                                                     u64* keys, int* segcnt, int max_cand, int* counts) {
    extern __shared__ int hist[];                 // per-block class histogram: one global atomic per class per block
                                                  // (STAGED: followed by 4 x 32 x (5 + C) floats)
    const int b = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += 256) hist[c] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int row0 = wave * 64;
    const int attrib = 5 + C;
    const float* img = dets + (size_t)b * N * attrib;
    u64* kb = keys + (size_t)b * max_cand;
    float mybest = -INFINITY; int mycls = 0;          // lane L keeps the result of row row0+L (non-eval mode)
    if (row0 < N) {
        const int row = row0 + lane;
        float conf = 0.f;
        if (row < N) conf = img[(size_t)row * attrib + 4];
        // `prob`: the caller guarantees cls in [0,1] (sigmoid outputs) => cls*conf <= conf (rounding is
        // monotonic), so only rows with conf > thr can pass and the others are never read.
        u64 todo = __ballot(row < N && (!prob || conf > thr));
        if (STAGED && __popcll(todo) >= FILTER_STAGE_MIN) {
            float* stage = reinterpret_cast<float*>(hist + C) + (size_t)(threadIdx.x >> 6) * 32 * attrib;
            const int r = lane & 31, half = lane >> 5;
            const int ch = (C + 1) >> 1;
            const int c_lo = half ? ch : 0, c_hi = half ? C : ch;
            for (int ps = 0; ps < 2; ++ps) {
                const int prow0 = row0 + ps * 32;
                if (prow0 >= N || ((todo >> (ps * 32)) & 0xffffffffull) == 0) continue;
                const int nfl = min(32, N - prow0) * attrib;
                const float* src = img + (size_t)prow0 * attrib;
                __builtin_amdgcn_wave_barrier();                           // (the wave's previous pass is done with `stage`)
#pragma unroll 8
                for (int i = lane; i < nfl; i += 64) stage[i] = src[i];
                __builtin_amdgcn_wave_barrier();
                const bool active = (todo >> (ps * 32 + r)) & 1ull;       // (implies row < N)
                const float cf = __shfl(conf, ps * 32 + r);
                const float* pr = stage + r * attrib + 5;
                if (!EVAL) {
                    float best = -INFINITY; int bidx = 0x7fffffff; bool nan = false;
                    if (active)
                        for (int c = c_lo; c < c_hi; ++c) {
                            const float sc = pr[c] * cf;                   // utils.py:233
                            nan |= (sc != sc);
                            if (sc > best) { best = sc; bidx = c; }        // ascending c: first index wins
                        }
                    const float ob = __shfl_xor(best, 32);
                    const int oi = __shfl_xor(bidx, 32);
                    const bool on = __shfl_xor((int)nan, 32) != 0;
                    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
                    if (active && half == ps && !(nan || on)) { mybest = best; mycls = bidx; }   // torch.max propagates NaN -> not > thr
                } else {
                    int cnt = 0;
                    if (active)
                        for (int c = c_lo; c < c_hi; ++c) cnt += (pr[c] * cf > thr) ? 1 : 0;     // utils.py:238
                    int incl = cnt;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
                    const int total = __shfl(incl, 63);
                    if (total) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&counts[b], total);
                        int slot = __shfl(base, 0) + incl - cnt;
                        if (cnt)
                            for (int c = c_lo; c < c_hi; ++c) {
                                const float sc = pr[c] * cf;
                                if (sc > thr) {
                                    if (slot < max_cand) { kb[slot] = make_key(c, sc, prow0 + r); atomicAdd(&hist[c], 1); }
                                    ++slot;
                                }
                            }
                    }
                }
            }
            todo = 0;
        }
        if (!EVAL) {
            // R rows per round: their loads are independent and issued together (a dense scene is bound by the latency of the
            // row-after-row chain load -> reduce: 8 x 22743 rows, 21 k of them candidates, 99 -> ~35 us)
            constexpr int R = 4;
            while (todo) {
                int src[R]; float cfr[R]; const float* pr[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    src[r] = -1; cfr[r] = 0.f; pr[r] = img;
                    if (todo) {
                        src[r] = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        cfr[r] = __shfl(conf, src[r]);
                        pr[r] = img + (size_t)(row0 + src[r]) * attrib + 5;
                    }
                }
                float best[R]; int bidx[R]; bool nan[R];
#pragma unroll
                for (int r = 0; r < R; ++r) { best[r] = -INFINITY; bidx[r] = 0x7fffffff; nan[r] = false; }
                for (int c = lane; c < C; c += 64) {
                    float v[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) v[r] = src[r] >= 0 ? pr[r][c] : 0.f;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float s = v[r] * cfr[r];                     // utils.py:233
                        nan[r] |= (s != s);
                        if (s > best[r]) { best[r] = s; bidx[r] = c; }     // first index wins within a lane
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (src[r] < 0) continue;
                    float bs = best[r]; int bi = bidx[r];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        const float ob = __shfl_xor(bs, off);
                        const int oi = __shfl_xor(bi, off);
                        if (ob > bs || (ob == bs && oi < bi)) { bs = ob; bi = oi; }
                    }
                    const bool anynan = __any(nan[r]);                     // torch.max propagates NaN -> not > thr
                    if (lane == src[r] && !anynan) { mybest = bs; mycls = bi; }
                }
            }
        }
        while (EVAL && todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int r = row0 + src;
            const float cf = __shfl(conf, src);
            const float* p = img + (size_t)r * attrib + 5;
            for (int c0 = 0; c0 < C; c0 += 64) {
                const int c = c0 + lane;
                float s = -1.f;
                if (c < C) s = p[c] * cf;
                const bool pass = c < C && s > thr;                 // utils.py:238
                const u64 pm = __ballot(pass);
                if (pm) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&counts[b], __popcll(pm));
                    base = __shfl(base, 0);
                    if (pass) {
                        const int slot = base + __popcll(pm & ((1ull << lane) - 1));
                        if (slot < max_cand) {
                            kb[slot] = make_key(c, s, r);
                            atomicAdd(&hist[c], 1);
                        }
                    }
                }
            }
        }
    }
    if (!EVAL) {
        // ONE counter atomic per workgroup (a dense scene has 356 waves per image with candidates: their same-address atomics
        // queue up in L2): wave counts -> LDS, thread 0 reserves the block's range, every wave takes its slice
        __shared__ int wcnt[4], wbase;
        const int row = row0 + lane, wv = threadIdx.x >> 6;
        const bool pass = row0 < N && mybest > thr;                     // utils.py:243
        const u64 pm = __ballot(pass);
        if (lane == 0) wcnt[wv] = __popcll(pm);
        __syncthreads();
        const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (tot) {
            if (threadIdx.x == 0) wbase = atomicAdd(&counts[b], tot);
            __syncthreads();
            if (pass) {
                int base = wbase;
                for (int q = 0; q < wv; ++q) base += wcnt[q];
                const int slot = base + __popcll(pm & ((1ull << lane) - 1));
                if (slot < max_cand) {
                    kb[slot] = make_key(mycls, mybest, row);
                    atomicAdd(&hist[mycls], 1);
                }
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
        if (hist[c]) atomicAdd(&segcnt[b * C + c], hist[c]);
}

// ------------------------------------------------------------------------------ NMS workspace
struct NmsWs {
    u64* skey;          // [B][max_n] keys in sorted order
    f32x4* sbox;        // [B][max_n] x1y1x2y2
    float* sconf;       // [B][max_n]
    unsigned char* svalid;  // [B][max_n] self-IOU > thr
    int2* saux;         // [B][max_n] .x = bits of the area (x2-x1)*(y2-y1), .y = class | TAME_BIT
    u64* keepbits;      // [B][nw] bit p%64 of word p/64: sorted position p is kept
    int* segoff;        // [B][C * NB + 1] start of every (class, score bucket) partition; class c = [segoff[c * NB], segoff[(c + 1) * NB])
    u64* mask;          // [B][nw][max_n]: word wi of COLUMN j; bit k = row 64*wi+k is earlier than j, same class, IOU(row, j) > thr
    u64* pkey;          // [B][max_n] keys partitioned by class (unsorted inside a class)
    int nw;
};
constexpr int TAME_BIT = 0x10000;            // classes < 4096

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t nms_layout(NmsWs* ws, char* base, int B, int max_n, int C) {
    const size_t n = (size_t)B * max_n;
    const int nw = (max_n + 63) / 64;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    char* p0 = take(n * 8); char* p1 = take(n * 16); char* p2 = take(n * 4); char* p3 = take(n); char* p4 = take(n * 8);
    char* p5 = take((size_t)B * ((size_t)C * nbuckets(C) + 1) * 4); char* p6 = take(n * (size_t)nw * 8);
    char* p7 = take(n * 8); char* p8 = take((size_t)B * nw * 8);
    if (ws) {
        ws->skey = (u64*)p0; ws->sbox = (f32x4*)p1; ws->sconf = (float*)p2; ws->svalid = (unsigned char*)p3;
        ws->saux = (int2*)p4; ws->segoff = (int*)p5; ws->mask = (u64*)p6; ws->nw = nw;
        ws->pkey = (u64*)p7; ws->keepbits = (u64*)p8;
    }
    return off;
}

// A box the fast IOU test may see: finite, |coordinate| <= 1e18 (areas and their sums stay finite), x2 >= x1, y2 >= y1, area 0 or >= 1e-30.
__device__ inline bool box_tame(const f32x4 b) {
    const float L = 1e18f;
    const float a = (b[2] - b[0]) * (b[3] - b[1]);       // (a tiny positive area: products in the screen could go denormal)
    return fabsf(b[0]) <= L && fabsf(b[1]) <= L && fabsf(b[2]) <= L && fabsf(b[3]) <= L && b[2] >= b[0] && b[3] >= b[1] &&
           (a == 0.f || a >= 1e-30f);
}
__device__ inline int2 box_aux(const f32x4 b, int cls) {
    return make_int2(__float_as_int((b[2] - b[0]) * (b[3] - b[1])), cls | (box_tame(b) ? TAME_BIT : 0));
}

// Exclusive prefix sum of v[0..n) in LDS (in place), total -> v[n]; all threads of the workgroup call it.  `part` = LDS scratch
// of (blockDim.x / 64) ints.
__device__ inline void block_exclusive_scan(int* v, int n, int* part) {
    const int nthr = blockDim.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = nthr >> 6;
    const int chunk = (n + nthr - 1) / nthr;
    const int t0 = min(n, (int)threadIdx.x * chunk), t1 = min(n, t0 + chunk);
    int mysum = 0;
    for (int t = t0; t < t1; ++t) mysum += v[t];
    int incl = mysum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off); if (lane >= off) incl += u; }
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    int woff = 0, total = 0;
    for (int q = 0; q < nwv; ++q) { const int u = part[q]; if (q < wv) woff += u; total += u; }
    int run = woff + incl - mysum;
    for (int t = t0; t < t1; ++t) { const int u = v[t]; v[t] = run; run += u; }
    if (threadIdx.x == 0) v[n] = total;
    __syncthreads();
}

// Sorting by (class, score desc, row): scatter the keys into their class segments (any order inside a segment: the atomics only
// decide scratch positions), split large classes (subpart_kernel), then rank every key among the keys of ITS partition
// (rank_seg_kernel).  Keys are unique and the partition is monotone in the key, so segoff[partition] + (#smaller keys of the
// partition) is the key's rank among all keys -- the same permutation as an O(n^2) count over the image at sum_p n_p^2 compares.
// An image whose candidate list overflowed (reported to the host as an error) only has to stay inside its buffers.
// One workgroup per image: class counts -> LDS, exclusive prefix, then the scatter with LDS cursors (1024 threads: the scatter is a
// chain of load -> LDS atomic -> store per key, 21 k keys took 33 us with 256).  Writes segoff[b][c * nb + 0] = class start and
// segoff[b][c * nb + k] = class end for k >= 1 (slot 0 holds the whole class until subpart_kernel splits it), segoff[b][C * nb] = n.
// Also clears the image's keep words.
__global__ __launch_bounds__(1024) void segpart_kernel(const u64* __restrict__ keys, int max_cand, const int* __restrict__ counts,
                                                       const int* __restrict__ segcnt, NmsWs ws, int C, int max_n) {
    extern __shared__ int sh[];                    // [C + 1] class offsets, then [C] fill cursors
    __shared__ int part[16];
    const int nb = nbuckets(C), P = C * nb;
    int* off = sh;
    int* cur = sh + C + 1;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { off[c] = segcnt[b * C + c]; cur[c] = 0; }
    for (int w = threadIdx.x; w < ws.nw; w += blockDim.x) ws.keepbits[(size_t)b * ws.nw + w] = 0;
    __syncthreads();
    block_exclusive_scan(off, C, part);
    for (int q = threadIdx.x; q <= P; q += blockDim.x) {
        const int c = q / nb, k = q - c * nb;
        ws.segoff[(size_t)b * (P + 1) + q] = q == P ? off[C] : (k == 0 ? off[c] : off[c + 1]);
    }
    const int n = min(min(counts[b], max_cand), max_n);
    const u64* kb = keys + (size_t)b * max_cand;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const u64 k = kb[i];
        const int c = min(key_cls(k), C - 1);
        const int pos = off[c] + atomicAdd(&cur[c], 1);
        if (pos < max_n) ws.pkey[(size_t)b * max_n + pos] = k;
    }
}

// Split one large class segment (SUB_MIN < n_c <= SUB_MAX keys) into NB sub-partitions of about n_c / NB keys: the keys go to
// LDS, 256 of them (every (n_c / 256)-th: the scatter order is arbitrary, so this is a fair sample) are sorted by rank, every
// 8th is a splitter, every key finds its bucket = number of splitters <= key (binary search) and is written back to the class's
// own range of pkey, grouped by bucket; segoff[b][c * NB + k] = start of bucket k.  A class of a dense scene (5 k keys whose
// scores crowd into a narrow band) ranks in ~1/25 of the compares; fixed score buckets did not help such a class.
constexpr int SUB_MIN = 768, SUB_MAX = 16384, SUB_SAMPLE = 256;
__global__ __launch_bounds__(1024) void subpart_kernel(NmsWs ws, int C, int max_n) {
    extern __shared__ u64 kl[];                    // [n_c] the class's keys
    __shared__ u64 samp[SUB_SAMPLE], sorted[SUB_SAMPLE];
    __shared__ int cnt[NB + 1], cur[NB];
    __shared__ int cs[257];                        // class starts (only launched when nbuckets(C) == NB, i.e. C <= 256)
    __shared__ int pick;
    const int b = blockIdx.y;
    const int P = C * NB;
    int* so = ws.segoff + (size_t)b * (P + 1);
    for (int q = threadIdx.x; q <= C; q += blockDim.x) cs[q] = min(so[q * NB], max_n);
    // workgroup x of the image takes its x-th, (x + gridDim.x)-th, ... large class
    for (int g = blockIdx.x; ; g += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int found = -1, left = g;
            for (int q = 0; q < C && found < 0; ++q) {
                const int n_c = cs[q + 1] - cs[q];
                if (n_c > SUB_MIN && n_c <= SUB_MAX && left-- == 0) found = q;
            }
            pick = found;
        }
        if (threadIdx.x < NB) { cnt[threadIdx.x] = 0; cur[threadIdx.x] = 0; }
        __syncthreads();
        const int c = pick;
        if (c < 0) return;
        const int s = cs[c], n = cs[c + 1] - s;
        u64* kb = ws.pkey + (size_t)b * max_n + s;
        for (int i = threadIdx.x; i < n; i += blockDim.x) kl[i] = kb[i];
        __syncthreads();
        if (threadIdx.x < SUB_SAMPLE) samp[threadIdx.x] = kl[(int)(((long long)threadIdx.x * n) / SUB_SAMPLE)];
        __syncthreads();
        if (threadIdx.x < SUB_SAMPLE) {
            const u64 mine = samp[threadIdx.x];
            int r = 0;
            for (int q = 0; q < SUB_SAMPLE; ++q) r += samp[q] < mine ? 1 : 0;
            sorted[r] = mine;                      // (keys are unique: a permutation)
        }
        __syncthreads();
        // bucket(key) = #{k in 1..NB-1 : sorted[k * SUB_SAMPLE / NB] <= key}
        auto bucket = [&](u64 k) {
            int lo = 0, hi = NB - 1;               // largest j in [0, NB-1] with (j == 0 or splitter_j <= k)
            while (lo < hi) { const int m_ = (lo + hi + 1) >> 1; if (sorted[m_ * (SUB_SAMPLE / NB)] <= k) lo = m_; else hi = m_ - 1; }
            return lo;
        };
        for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(&cnt[bucket(kl[i])], 1);
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = 0;
            for (int k = 0; k < NB; ++k) { const int v = cnt[k]; cnt[k] = acc; acc += v; }
            cnt[NB] = acc;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const u64 k = kl[i];
            const int bk = bucket(k);
            kb[cnt[bk] + atomicAdd(&cur[bk], 1)] = k;
        }
        if (threadIdx.x < NB) so[c * NB + threadIdx.x] = s + cnt[threadIdx.x];
    }
}

// Rank inside the partitions + gather of the box records in sorted order.  Work item = (partition, 256-key slice of it), numbered
// partition by partition; a workgroup loops over items.  (Measured on the way here, profiles/r03_postproc_*: with whole classes as
// partitions -- 5 k keys -- the kernel took 115-122 us for 8 x 21 k candidates whatever the compare: the LDS broadcast read of a
// tile key returns 8 B to each of 64 lanes, and a lone wave per SIMD pays the full issue latency of compare -> add; four keys per
// thread or a borrow-chain compare made it slower.  Fewer compares was the fix.)
__global__ __launch_bounds__(256) void rank_seg_kernel(const float* __restrict__ dets, int N, int C, float nms_thr, NmsWs ws, int max_n) {
    __shared__ u64 tile[2][256];
    __shared__ int part[4];
    extern __shared__ int so[];                               // [P + 1] this image's partition offsets, then [P + 1] first work item of every partition
    const int P = C * nbuckets(C);
    int* first = so + P + 1;
    const int b = blockIdx.y;
    const int attrib = 5 + C;
    for (int c = threadIdx.x; c <= P; c += 256) so[c] = min(ws.segoff[(size_t)b * (P + 1) + c], max_n);
    __syncthreads();
    for (int c = threadIdx.x; c < P; c += 256) first[c] = (so[c + 1] - so[c] + 255) >> 8;
    __syncthreads();
    block_exclusive_scan(first, P, part);
    const int items = first[P];
    const u64* kb = ws.pkey + (size_t)b * max_n;
    for (int t = blockIdx.x; t < items; t += gridDim.x) {
        int lo = 0, hi = P - 1;                               // the partition whose slice range holds t: largest c with first[c] <= t
        while (lo < hi) { const int m_ = (lo + hi + 1) >> 1; if (first[m_] <= t) lo = m_; else hi = m_ - 1; }
        const int s = so[lo], e = so[lo + 1];
        const int i = s + (t - first[lo]) * 256 + threadIdx.x;
        const u64 mine = i < e ? kb[i] : ~0ull;
        int rank = 0;
        // the partition streams through a double-buffered LDS tile: tile k+1's global load is in flight under tile k's compares
        u64 kj = s + (int)threadIdx.x < e ? kb[s + threadIdx.x] : ~0ull;
        int buf = 0;
        __syncthreads();                                      // the previous item's last tile is dead
        for (int j0 = s; j0 < e; j0 += 256) {
            tile[buf][threadIdx.x] = kj;                       // (padding = ~0: never smaller)
            const int jn = j0 + 256 + threadIdx.x;
            kj = jn < e ? kb[jn] : ~0ull;
            __syncthreads();
            const u64* tl = tile[buf];
            const int lim = min(256, e - j0);
            if (lim == 256) {
#pragma unroll 16
                for (int q = 0; q < 256; ++q) rank += (tl[q] < mine) ? 1 : 0;
            } else {
#pragma unroll 4
                for (int q = 0; q < lim; ++q) rank += (tl[q] < mine) ? 1 : 0;
            }
            buf ^= 1;
        }
        if (i < e) {
            const int row = key_row(mine);
            const float* p = dets + ((size_t)b * N + min(row, N - 1)) * attrib;
            const f32x4 bx = to_xyxy(p[0], p[1], p[2], p[3]);               // utils.py:230
            const size_t o = (size_t)b * max_n + s + rank;
            ws.skey[o] = mine;
            ws.sbox[o] = bx;
            ws.sconf[o] = p[4];
            ws.svalid[o] = iou_xyxy(bx, bx) > nms_thr ? 1 : 0;              // diagonal of utils.py:177
            ws.saux[o] = box_aux(bx, key_cls(mine));
        }
    }
}

// rank-sort + gather over ALL keys of an image: (row, class) = torch.nonzero order (utils.py:204-224, use_nms = False; every
// candidate is kept).  The NMS path ranks inside class segments instead (segpart_kernel + rank_seg_kernel).
__global__ __launch_bounds__(256) void rank_raw_kernel(const float* __restrict__ dets, int N, int C,
                                                       const u64* __restrict__ keys, int max_cand, const int* __restrict__ counts,
                                                       NmsWs ws, int max_n) {
    __shared__ u64 tile[256];
    const int b = blockIdx.y;
    const int n = min(min(counts[b], max_cand), max_n);
    const u64* kb = keys + (size_t)b * max_cand;
    const int attrib = 5 + C;
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {
        const int i = i0 + threadIdx.x;
        u64 mine = ~0ull, mk = ~0ull;
        if (i < n) {
            mine = kb[i];
            mk = ((u64)key_row(mine) << 12) | (u64)key_cls(mine);
        }
        int rank = 0;
        for (int j0 = 0; j0 < n; j0 += 256) {
            const int j = j0 + threadIdx.x;
            u64 kj = ~0ull;
            if (j < n) { kj = kb[j]; kj = ((u64)key_row(kj) << 12) | (u64)key_cls(kj); }
            __syncthreads();
            tile[threadIdx.x] = kj;
            __syncthreads();
            const int lim = min(256, n - j0);
#pragma unroll 8
            for (int t = 0; t < lim; ++t) rank += (tile[t] < mk) ? 1 : 0;
        }
        if (i < n) {
            const int row = key_row(mine);
            const float* p = dets + ((size_t)b * N + row) * attrib;
            const f32x4 bx = to_xyxy(p[0], p[1], p[2], p[3]);               // utils.py:230
            const size_t o = (size_t)b * max_n + rank;
            ws.skey[o] = mine;
            ws.sbox[o] = bx;
            ws.sconf[o] = p[4];
        }
    }
}

// 64x64 IOU tiles -> bit masks, stored by COLUMN: bit k of mask[b][ti][j] = (row 64*ti+k < j) & same class & IOU(row, j) > thr,
// i.e. lane = the LATER box j, bits = the earlier boxes that would suppress it -- the layout the scan reads with coalesced,
// decision-independent loads.  Keys are sorted by class, so column tile tj only pairs with the row tiles from the one holding
// the start of its first box's class segment (tfirst) up to tj: the (ti, tj) work items of an image are numbered through a
// prefix sum over cnt[tj] = tj - tfirst[tj] + 1 (LDS) and every wave takes an equal, contiguous share of them.
//
// The compare.  Reference: iou = inter / ((area_b + area_a) - inter) in fp32, then `iou > thr` (utils.py:98-119,177).  For boxes
// that are `box_tame` the division is replaced by an EXACT equivalent: with u = union > 0 and q = inter / u, RN(q) > thr <=>
// q > mid, where mid = the midpoint of thr and the next float above it (q == mid is impossible: mid has 25 significant bits with
// the last one set, so mid * u is never a float) <=> inter > mid * u, and the product of a 25-bit and a 24-bit number is exact in
// double.  u == 0 only when both areas and inter are 0 (0/0 = NaN, not > thr; 0 > 0 false as well); tame boxes give
// inter <= min(areas) and u >= 0 by monotonicity of rounding.  min / max without NaN handling are exact for non-NaN inputs (a
// zero of the other sign changes no comparison).  Any other box (NaN, inf, huge, negative extent) or threshold takes the
// literal path, per 64x64 tile.
__device__ inline float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline float vmax0(float a) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a)); return r; }

__global__ __launch_bounds__(256) void mask_kernel(const int* __restrict__ counts, int max_cand, NmsWs ws, int max_n, int C,
                                                   float thr, double mid, float t2, int fast_ok) {
    __shared__ f32x4 rbox[4][64];
    __shared__ int2 raux[4][64];
    __shared__ int wpart[4];
    extern __shared__ int shm[];                         // [C + 1] segment offsets, [nt] tfirst, [nt + 1] prefix of cnt
    const int b = blockIdx.y;
    const int n = min(min(counts[b], max_cand), max_n);
    if (n <= 0) return;
    const int nt = (n + 63) >> 6;
    int* so = shm;
    int* tfirst = shm + C + 1;
    int* pre = tfirst + nt;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t base = (size_t)b * max_n;
    for (int c = threadIdx.x; c <= C; c += 256) so[c] = min(ws.segoff[(size_t)b * (C * nbuckets(C) + 1) + c * nbuckets(C)], n);     // class starts
    __syncthreads();
    // tfirst[t] = tile of the segment start of the class that position 64 t belongs to; pre = exclusive prefix of cnt
    const int chunk = (nt + 255) >> 8;
    const int t0 = threadIdx.x * chunk, t1 = min(nt, t0 + chunk);
    int mysum = 0;
    for (int t = t0; t < t1; ++t) {
        const int pos = t * 64;
        int lo = 0, hi = C - 1;                                    // largest c with so[c] <= pos
        while (lo < hi) { const int m_ = (lo + hi + 1) >> 1; if (so[m_] <= pos) lo = m_; else hi = m_ - 1; }
        const int tf = so[lo] >> 6;
        tfirst[t] = tf;
        mysum += t - tf + 1;
    }
    int incl = mysum;                                              // inclusive scan over the 256 threads
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
    if (lane == 63) wpart[wv] = incl;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < wv; ++q) woff += wpart[q];
    const int total = wpart[0] + wpart[1] + wpart[2] + wpart[3];
    int run = woff + incl - mysum;
    for (int t = t0; t < t1; ++t) { pre[t] = run; run += t - tfirst[t] + 1; }
    if (threadIdx.x == 0) pre[nt] = total;
    __syncthreads();

    const int nwaves = gridDim.x * 4;
    const int per = (total + nwaves - 1) / nwaves;
    int it = (blockIdx.x * 4 + wv) * per;
    const int it_end = min(total, it + per);
    if (it >= it_end) return;
    int tj;
    { int lo = 0, hi = nt - 1; while (lo < hi) { const int m_ = (lo + hi + 1) >> 1; if (pre[m_] <= it) lo = m_; else hi = m_ - 1; } tj = lo; }
    int ti = tfirst[tj] + (it - pre[tj]);
    const u64 lt = (1ull << lane) - 1ull;
    int cur_tj = -1;
    f32x4 bj = {0.f, 0.f, 0.f, 0.f}; float aj = 0.f, sj = 0.f; int cj = -2, cj0 = -2; bool tame_cols = true;
    for (; it < it_end; ++it) {
        const int j = tj * 64 + lane;
        if (cur_tj != tj) {
            cur_tj = tj;
            bj = f32x4{0.f, 0.f, 0.f, 0.f}; aj = 0.f; cj = -2;
            if (j < n) { bj = ws.sbox[base + j]; const int2 a = ws.saux[base + j]; aj = __int_as_float(a.x); cj = a.y; }
            tame_cols = __ballot(j < n && !(cj & TAME_BIT)) == 0;
            sj = t2 * aj;
            cj0 = __builtin_amdgcn_readfirstlane(cj);                 // (lane 0 of a column tile is always < n)
        }
        const int i = ti * 64 + lane;
        f32x4 bi = {0.f, 0.f, 0.f, 0.f}; int2 ai = make_int2(0, -1);
        if (i < n) { bi = ws.sbox[base + i]; ai = ws.saux[base + i]; }
        const bool fast = fast_ok && tame_cols && __ballot(i < n && !(ai.y & TAME_BIT)) == 0;
        // fast path: rows are staged with s_r = t2 * area_r (t2 slightly below thr / (1 + thr)) for the screen below; a column tile of
        // ONE class (nearly all tiles of a large class) stages s_r = +inf for rows of another class and needs no class compare per pair
        const bool one_cls = fast && __ballot(j < n && cj != cj0) == 0;
        if (fast) ai.x = __float_as_int((one_cls && i < n && ai.y != cj0) || i >= n ? INFINITY : t2 * __int_as_float(ai.x));
        __builtin_amdgcn_wave_barrier();                 // single wave owns rbox[wv]: LDS ops of one wave are in order
        rbox[wv][lane] = bi; raux[wv][lane] = ai;
        __builtin_amdgcn_wave_barrier();
        u64 m = 0;
        if (fast) {
            // Screen (no false negatives): a hit needs inter > thr * union, i.e. inter > thr / (1 + thr) * (area_j + area_r) up to a few
            // ulp; t2 is that factor reduced by 2^-18, so `inter' >= s_j + s_r` holds for every hit (inter' = inter whenever both
            // overlaps are positive; one clamp is enough to keep a disjoint pair's product <= 0).  11 (10) VALU instructions per pair;
            // the exact test below only runs for rows where some lane passes (~1 row in 10 of a dense scene).
            auto row_bits = [&](auto with_cls) {
                for (int g = 0; g < 8; ++g) {                // 8 rows per byte of the mask word (constant bit positions)
                    unsigned m8 = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f32x4 r = rbox[wv][g * 8 + q];
                        const int2 ra = raux[wv][g * 8 + q];
                        const float iwc = vmax0(vmin(r[2], bj[2]) - vmax(r[0], bj[0]));
                        const float ihr = vmin(r[3], bj[3]) - vmax(r[1], bj[1]);
                        bool maybe = iwc * ihr >= sj + __int_as_float(ra.x);
                        if (with_cls) maybe = maybe && ra.y == cj;
                        if (__builtin_amdgcn_ballot_w64(maybe)) {
                            const float inter = iwc * vmax0(ihr);
                            const float uni = (aj + (r[2] - r[0]) * (r[3] - r[1])) - inter;
                            const bool hit = maybe && ra.y == cj && ((double)inter > mid * (double)uni);
                            m8 |= hit ? (1u << q) : 0u;
                        }
                    }
                    m |= (u64)m8 << (8 * g);
                }
            };
            if (one_cls) row_bits(std::false_type{}); else row_bits(std::true_type{});
        } else {
            const int cls_j = cj & (TAME_BIT - 1);
#pragma unroll 4
            for (int k = 0; k < 64; ++k) {
                const int2 ra = raux[wv][k];
                const bool hit = cj >= 0 && ra.y >= 0 && ((ra.y & (TAME_BIT - 1)) == cls_j) && (iou_xyxy(rbox[wv][k], bj) > thr);
                m |= hit ? (1ull << k) : 0ull;
            }
        }
        if (ti == tj) m &= lt;                           // diagonal tile: earlier rows only
        if (j < n) ws.mask[((size_t)b * ws.nw + ti) * max_n + j] = m;
        if (++ti > tj) { ++tj; if (tj < nt) ti = tfirst[tj]; }
    }
}

// Greedy NMS of one (image, class) segment (reference utils.py:180-190): box j is kept iff it is valid and no KEPT earlier box of
// its class has IOU > thr with it.  One workgroup per segment, SCAN_WAVES waves taking the segment's 64-box words round-robin.
// For word w a wave loads the column masks of its 64 boxes against every earlier word of the segment -- addresses and data do not
// depend on any decision, so all of it is in flight long before it is needed -- ANDs each with that word's keep bits as they are
// published in LDS (`done` = number of finished words), and resolves the in-word dependencies as a fixed point on wave ballots:
// an undecided box dies if a kept earlier box hits it, is kept if no earlier undecided-or-kept box does; the lowest undecided box
// is always decided, typical words take 3-5 rounds instead of one step per kept box.  The only serial part per word is
// "last AND + fixed point + publish" (~0.3 us); the old kernel had two dependent global-memory round trips per word (2.1 us).
constexpr int SCAN_WAVES = 8;
// LDS traffic of one wave is issued and completed in order, and an LDS word written by one wave is visible to every wave of the
// workgroup: "keep word, then counter" on the writer and "counter, then keep word" on the reader need no hardware fence -- only the
// compiler must keep the program order (a generic workgroup fence would also wait for the wave's global loads in flight, which is
// exactly what the prefetch wants to avoid: measured 1.07 us per word with fences + a generic-address poll).
#define YV3_COMPILER_BARRIER() asm volatile("" ::: "memory")
__global__ __launch_bounds__(SCAN_WAVES * 64) void scan_kernel(const int* __restrict__ counts, int max_cand, NmsWs ws, int max_n, int C) {
    extern __shared__ u64 keepw[];         // [words of this segment] keep bits, published in word order
    __shared__ int done_s;                 // number of published words
    auto published = [&]() { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(&done_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); };
    const int b = blockIdx.y;
    const int n = min(min(counts[b], max_cand), max_n);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: w, wl, batch bounds live in SGPRs)
    const size_t base = (size_t)b * max_n;
    const u64* mb = ws.mask + (size_t)b * ws.nw * max_n;
    const u64 lt = (1ull << lane) - 1ull;
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        const int nb = nbuckets(C);
        int s0 = ws.segoff[(size_t)b * (C * nb + 1) + c * nb], s1 = ws.segoff[(size_t)b * (C * nb + 1) + (c + 1) * nb];
        s0 = __builtin_amdgcn_readfirstlane(min(s0, n)); s1 = __builtin_amdgcn_readfirstlane(min(s1, n));
        if (s1 <= s0) continue;                                              // (uniform over the workgroup)
        const int w_lo = s0 >> 6, w_hi = (s1 - 1) >> 6;
        __syncthreads();                                                     // the previous segment's words are dead
        if (threadIdx.x == 0) done_s = 0;
        __syncthreads();
        for (int w = w_lo + wv; w <= w_hi; w += SCAN_WAVES) {
            const int p = w * 64 + lane;
            const bool inseg = p >= s0 && p < s1;
            const bool valid = inseg && ws.svalid[base + p] != 0;
            const u64 diag = inseg ? (mb[(size_t)w * max_n + p] & lt) : 0ull;
            u64 rem = 0;
            // words [w_lo, w-1): final long before they are needed; BATCH column-mask loads in flight, the next batch behind them
            constexpr int BATCH = 8;
            const int wl = w - 1;                                            // the word whose keep bits arrive last
            const u64 xl = (inseg && wl >= w_lo) ? mb[(size_t)wl * max_n + p] : 0ull;
            if (wl > w_lo) {
                // two register sets in ping-pong: 2 x BATCH loads of this wave in flight at any time (a copy between the sets would
                // wait for the newer one: one memory latency per batch, 0.9 us per word on a 90-word segment)
                u64 xa[BATCH], xb[BATCH];
                auto fetch = [&](u64 (&x)[BATCH], int wi0) {
#pragma unroll
                    for (int q = 0; q < BATCH; ++q) x[q] = (inseg && wi0 + q < wl) ? mb[(size_t)(wi0 + q) * max_n + p] : 0ull;
                };
                auto consume = [&](const u64 (&x)[BATCH], int wi0) {
                    if (wi0 >= wl) return;
                    const int need = min(wl, wi0 + BATCH) - w_lo;            // words [w_lo, wi0 + BATCH) must be final
                    while (published() < need) __builtin_amdgcn_s_sleep(1);
                    YV3_COMPILER_BARRIER();
                    u64 kw[BATCH];                                           // (x[q] = 0 beyond wl: the clamped read is harmless, and unguarded
#pragma unroll                                                               //  LDS reads are issued together instead of one round trip each)
                    for (int q = 0; q < BATCH; ++q) kw[q] = keepw[min(wi0 + q, wl - 1) - w_lo];
#pragma unroll
                    for (int q = 0; q < BATCH; ++q) rem |= x[q] & kw[q];
                };
                fetch(xa, w_lo);
                for (int wi0 = w_lo; wi0 < wl; wi0 += 2 * BATCH) {
                    fetch(xb, wi0 + BATCH);
                    consume(xa, wi0);
                    fetch(xa, wi0 + 2 * BATCH);
                    consume(xb, wi0 + BATCH);
                }
            }
            if (wl >= w_lo) {                                                // the serial part starts here
                while (published() < w - w_lo) __builtin_amdgcn_s_sleep(1);
                YV3_COMPILER_BARRIER();
                rem |= xl & keepw[wl - w_lo];
            }
            // in-word fixed point (diag only has bits of earlier lanes)
            u64 undec = __ballot(valid && rem == 0ull);
            u64 kept = 0;
            while (undec) {
                const bool mine = (undec >> lane) & 1ull;
                const bool killed = mine && (diag & kept) != 0ull;
                const bool free_ = mine && !killed && (diag & undec) == 0ull;
                const u64 k2 = __ballot(free_), d2 = __ballot(killed);
                kept |= k2;
                undec &= ~(k2 | d2);
            }
            if (lane == 0) keepw[w - w_lo] = kept;
            YV3_COMPILER_BARRIER();
            if (lane == 0) __hip_atomic_store(&done_s, w - w_lo + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (words < w are final)
        }
        // keep words -> global, after the last word (an atomic inside the loop sits in front of the next word's loads in the
        // wave's in-order memory queue).  The first / last word can straddle two segments: OR; the others are this segment's alone.
        __syncthreads();
        for (int i = threadIdx.x; i <= w_hi - w_lo; i += SCAN_WAVES * 64) {
            const u64 kept = keepw[i];
            u64* dst = &ws.keepbits[(size_t)b * ws.nw + w_lo + i];
            if (i == 0 || i == w_hi - w_lo) { if (kept) atomicOr(dst, kept); } else *dst = kept;
        }
    }
}

// out[b][k] = x1,y1,x2,y2,conf,score,cls for the k-th kept position (utils.py:193-199).  One workgroup per 1024 sorted positions:
// its output offset is the popcount of the keep words before them.  `all`: every candidate is kept (use_nms = False).
__global__ __launch_bounds__(256) void compact_kernel(const int* __restrict__ counts, int max_cand, NmsWs ws, int max_n,
                                                      float* out, int cap, int* out_counts, int all) {
    __shared__ int red[4];
    __shared__ int wcnt[16];
    const int b = blockIdx.y;
    const int n = min(min(counts[b], max_cand), max_n);
    const int nwn = (n + 63) >> 6;                                         // words that can hold kept positions
    const int wq = blockIdx.x * 16;                                        // first word of this workgroup
    if (blockIdx.x != 0 && wq >= nwn) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64* kw = ws.keepbits + (size_t)b * ws.nw;
    auto word = [&](int w) -> u64 {
        if (w >= nwn) return 0ull;
        const int left = n - w * 64;
        const u64 live = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
        return (all ? ~0ull : kw[w]) & live;
    };
    // popcount of the words before wq (workgroup 0: of ALL words = the image's kept count)
    const int lim = blockIdx.x == 0 ? nwn : wq;
    int part = 0;
    for (int w = threadIdx.x; w < lim; w += 256) part += __popcll(word(w));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) red[wv] = part;
    if (threadIdx.x < 16) wcnt[threadIdx.x] = __popcll(word(wq + threadIdx.x));
    __syncthreads();
    const int before = red[0] + red[1] + red[2] + red[3];
    if (blockIdx.x == 0) { if (threadIdx.x == 0) out_counts[b] = before; }
    int pos0 = blockIdx.x == 0 ? 0 : before;
    const size_t base = (size_t)b * max_n;
    for (int r = 0; r < 4; ++r) {
        const int wl = r * 4 + wv;                                         // word of this wave in this round
        int pre = pos0;
        for (int q = 0; q < wl; ++q) pre += wcnt[q];
        const u64 kwv = word(wq + wl);
        const int p = (wq + wl) * 64 + lane;
        if ((kwv >> lane) & 1ull) {
            const int pos = pre + __popcll(kwv & ((1ull << lane) - 1ull));
            if (pos < cap) {
                const u64 key = ws.skey[base + p];
                const f32x4 bx = ws.sbox[base + p];
                float* o = out + ((size_t)b * cap + pos) * 7;
                o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3];
                o[4] = ws.sconf[base + p]; o[5] = key_score(key); o[6] = (float)key_cls(key);
            }
        }
    }
}

// ------------------------------------------------------------------------------ geometry helpers
__global__ void cxcywh_kernel(const float* in, float* out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f32x4 r = to_xyxy(in[i * 4 + 0], in[i * 4 + 1], in[i * 4 + 2], in[i * 4 + 3]);
    out[i * 4 + 0] = r[0]; out[i * 4 + 1] = r[1]; out[i * 4 + 2] = r[2]; out[i * 4 + 3] = r[3];
}

__global__ void iou_matrix_kernel(const float* b1, int n1, int ld1, const float* b2, int n2, int ld2, int mode, float* out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n2 || i >= n1) return;
    const float* p = b1 + (size_t)i * ld1;
    const float* q = b2 + (size_t)j * ld2;
    f32x4 a, b;
    if (mode == 1) { a = to_xyxy(p[0], p[1], p[2], p[3]); b = to_xyxy(q[0], q[1], q[2], q[3]); }
    else { a[0] = p[0]; a[1] = p[1]; a[2] = p[2]; a[3] = p[3]; b[0] = q[0]; b[1] = q[1]; b[2] = q[2]; b[3] = q[3]; }
    // utils.py:116 / :143: union = area(b1 row) broadcast + area(b2 row) - inter
    const float iw = clamp0(tmin(a[2], b[2]) - tmax(a[0], b[0]));
    const float ih = clamp0(tmin(a[3], b[3]) - tmax(a[1], b[1]));
    const float inter = iw * ih;
    const float aa = (a[2] - a[0]) * (a[3] - a[1]);
    const float ab = (b[2] - b[0]) * (b[3] - b[1]);
    out[(size_t)i * n2 + j] = inter / ((aa + ab) - inter);
}

}  // namespace

extern "C" size_t yv3_postproc_cand_bytes(int B, int max_cand, int num_class) {
    if (B <= 0 || max_cand <= 0 || num_class <= 0) return 0;
    return cand_keys_bytes(B, max_cand) + align256((size_t)B * num_class * 4);
}

extern "C" size_t yv3_postproc_nms_workspace_bytes(int B, int max_n, int num_class) {
    if (B <= 0 || max_n <= 0 || num_class <= 0) return 0;
    return nms_layout(nullptr, nullptr, B, max_n, num_class);
}

extern "C" int yv3_postproc_filter(const float* dets, int B, int N, int num_class, float conf_thr, int mode,
                                   void* cand, int max_cand, int* cand_counts, void* stream) {
    if (!dets || !cand || !cand_counts || B <= 0 || N <= 0 || num_class <= 0 || max_cand <= 0) return YV3_EINVAL;
    if (N > (1 << ROW_BITS) || num_class >= 4096) return YV3_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    u64* keys = (u64*)cand;
    int* segcnt = (int*)((char*)cand + cand_keys_bytes(B, max_cand));
    // counters are cleared by a kernel, not hipMemsetAsync: memset nodes captured into a hipGraph
    // were observed to write garbage from the second replay on (ROCm 7.2, gfx950)
    hipLaunchKernelGGL(zero_kernel, dim3(yv3_ceil_div((long long)B * num_class, 256)), dim3(256), 0, s,
                       cand_counts, B, segcnt, B * num_class);
    YV3_CHECK_LAUNCH();
    const dim3 grid((unsigned)yv3_ceil_div(N, 256), (unsigned)B);
    const bool prob = (mode & YV3_PP_PROB) != 0 && conf_thr >= 0.f;
    const size_t hl = (size_t)num_class * sizeof(int);
    const size_t stage = (size_t)4 * 32 * (5 + num_class) * sizeof(float);
    const bool staged = hl + stage <= 96 * 1024;                      // <= 183 classes; more: the unstaged kernel
    const bool ev = (mode & YV3_PP_EVAL) != 0;
#define YV3_FILTER(EV_, ST_) hipLaunchKernelGGL((filter_kernel<EV_, ST_>), grid, dim3(256), hl + (ST_ ? stage : 0), s, dets, N, num_class, \
                                                conf_thr, prob, keys, segcnt, max_cand, cand_counts)
    if (ev) { if (staged) YV3_FILTER(true, true); else YV3_FILTER(true, false); }
    else    { if (staged) YV3_FILTER(false, true); else YV3_FILTER(false, false); }
#undef YV3_FILTER
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_postproc_nms(const float* dets, int B, int N, int num_class, float nms_thr, int use_nms,
                                const void* cand, int max_cand, const int* cand_counts, int max_n,
                                float* out_boxes, int cap, int* out_counts,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (!dets || !cand || !cand_counts || !out_boxes || !out_counts || !workspace) return YV3_EINVAL;
    if (B <= 0 || N <= 0 || num_class <= 0 || max_cand <= 0 || cap <= 0 || max_n <= 0) return YV3_EINVAL;
    if (max_n > max_cand) max_n = max_cand;
    NmsWs ws;
    if (nms_layout(&ws, (char*)workspace, B, max_n, num_class) > workspace_bytes) return YV3_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const u64* keys = (const u64*)cand;
    const int* segcnt = (const int*)((const char*)cand + cand_keys_bytes(B, max_cand));

    const int rb = max_n < 256 * 64 ? yv3_ceil_div(max_n, 256) : 64;
    const int nw = ws.nw;
    if (use_nms) {
        const size_t mask_lds = (size_t)(2 * nw + num_class + 2) * sizeof(int), scan_lds = (size_t)nw * sizeof(u64);
        if (mask_lds > 128 * 1024 || scan_lds > 128 * 1024) return YV3_ESHAPE;        // > ~1 M candidates per image
        hipLaunchKernelGGL(segpart_kernel, dim3(B), dim3(1024), (size_t)(2 * num_class + 1) * sizeof(int), s,
                           keys, max_cand, cand_counts, segcnt, ws, num_class, max_n);
        YV3_CHECK_LAUNCH();
        if (nbuckets(num_class) == NB && max_n > SUB_MIN) {
            const int cap_n = max_n < SUB_MAX ? max_n : SUB_MAX;
            int big = max_n / SUB_MIN < num_class ? max_n / SUB_MIN : num_class;            // large classes an image can have
            if (big > 4) big = 4;                                                           // (a workgroup loops over its share of them)
            hipLaunchKernelGGL(subpart_kernel, dim3(big, B), dim3(1024), (size_t)cap_n * sizeof(u64), s, ws, num_class, max_n);
            YV3_CHECK_LAUNCH();
        }
        const int items = yv3_ceil_div(max_n, 256) + num_class * nbuckets(num_class);         // upper bound of (partition, slice) work items per image
        // ~2048 workgroups over the batch (each loops over its items): sparse images of a large batch need few of them (bs=64:
        // 32 per image, 0.12 ms for the NMS stage instead of 0.14 with 256), a dense small batch all of them (bs=8: 256, 0.88 vs 1.01 ms)
        int rgrid = 2048 / B; rgrid = rgrid < 32 ? 32 : rgrid > 256 ? 256 : rgrid; rgrid = items < rgrid ? items : rgrid;
        hipLaunchKernelGGL(rank_seg_kernel, dim3(rgrid, B), dim3(256), (size_t)(2 * num_class * nbuckets(num_class) + 2) * sizeof(int), s,
                           dets, N, num_class, nms_thr, ws, max_n);
        YV3_CHECK_LAUNCH();
        // mask: every wave takes an equal share of the image's (row tile, column tile) items; ~8192 workgroups over the batch
        long long mb = ((long long)nw * nw + 3) / 4;
        int mcap = 8192 / B; mcap = mcap < 32 ? 32 : mcap > 1024 ? 1024 : mcap;
        if (mb > mcap) mb = mcap;
        // exact division-free compare (see mask_kernel) for thresholds that are positive normal floats
        const int fast_ok = nms_thr >= 1.17549435e-38f && nms_thr < 1e30f;
        const double mid = fast_ok ? 0.5 * ((double)nms_thr + (double)nextafterf(nms_thr, INFINITY)) : 0.0;
        const float t2 = fast_ok ? (float)((double)nms_thr / (1.0 + (double)nms_thr) * (1.0 - 1.0 / 262144.0)) : 0.f;   // screen factor
        hipLaunchKernelGGL(mask_kernel, dim3((unsigned)mb, B), dim3(256), mask_lds, s,
                           cand_counts, max_cand, ws, max_n, num_class, nms_thr, mid, t2, fast_ok);
        YV3_CHECK_LAUNCH();
        // one workgroup per (image, class) segment for small batches; a large batch (sparse scenes in practice: most segments are
        // empty) loops ~2048 / B workgroups per image over the classes
        int sgrid = B <= 16 ? num_class : 2048 / B; sgrid = sgrid < 8 ? 8 : sgrid; sgrid = sgrid > num_class ? num_class : sgrid;
        hipLaunchKernelGGL(scan_kernel, dim3(sgrid, B), dim3(SCAN_WAVES * 64), scan_lds, s, cand_counts, max_cand, ws, max_n, num_class);
        YV3_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(rank_raw_kernel, dim3(rb, B), dim3(256), 0, s, dets, N, num_class, keys, max_cand, cand_counts, ws, max_n);
        YV3_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(compact_kernel, dim3(yv3_ceil_div(nw, 16), B), dim3(256), 0, s, cand_counts, max_cand, ws, max_n,
                       out_boxes, cap, out_counts, use_nms ? 0 : 1);
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_cxcywh_to_xyxy(const float* in, float* out, long long n, void* stream) {
    if (!in || !out || n < 0) return YV3_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(cxcywh_kernel, dim3(yv3_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n);
    YV3_CHECK_LAUNCH();
    return 0;
}

extern "C" int yv3_iou_matrix(const float* b1, int n1, int ld1, const float* b2, int n2, int ld2, int mode,
                              float* out, void* stream) {
    if (!b1 || !b2 || !out || n1 < 0 || n2 < 0 || ld1 < 4 || ld2 < 4) return YV3_EINVAL;
    if (n1 == 0 || n2 == 0) return 0;
    if (n1 > 65535) return YV3_ESHAPE;
    hipLaunchKernelGGL(iou_matrix_kernel, dim3(yv3_ceil_div(n2, 256), n1), dim3(256), 0, (hipStream_t)stream,
                       b1, n1, ld1, b2, n2, ld2, mode, out);
    YV3_CHECK_LAUNCH();
    return 0;
}
