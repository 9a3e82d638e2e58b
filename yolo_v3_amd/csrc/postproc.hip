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
    int* segcnt;       // [B][C]
};
__host__ __device__ inline size_t cand_keys_bytes(int B, int max_cand) { return ((size_t)B * max_cand * 8 + 255) & ~(size_t)255; }

__global__ void zero_kernel(int* a, int na, int* b, int nb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) a[i] = 0;
    if (i < nb) b[i] = 0;
}

template <bool EVAL>
__global__ __launch_bounds__(256) void filter_kernel(const float* __restrict__ dets, int N, int C, float thr, bool prob,
                                                     u64* keys, int* segcnt, int max_cand, int* counts) {
    extern __shared__ int hist[];                 // per-block class histogram: one global atomic per class per block
    const int b = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += 256) hist[c] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int row0 = wave * 64;
    const int attrib = 5 + C;
    const float* img = dets + (size_t)b * N * attrib;
    u64* kb = keys + (size_t)b * max_cand;
    if (row0 < N) {
        const int row = row0 + lane;
        float conf = 0.f;
        if (row < N) conf = img[(size_t)row * attrib + 4];
        // `prob`: the caller guarantees cls in [0,1] (sigmoid outputs) => cls*conf <= conf (rounding is
        // monotonic), so only rows with conf > thr can pass and the others are never read.
        u64 todo = __ballot(row < N && (!prob || conf > thr));
        float mybest = -INFINITY; int mycls = 0;      // lane L keeps the result of row row0+L (non-eval mode)
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int r = row0 + src;
            const float cf = __shfl(conf, src);
            const float* p = img + (size_t)r * attrib + 5;
            if (!EVAL) {
                float best = -INFINITY; int bidx = 0x7fffffff; bool nan = false;
                for (int c = lane; c < C; c += 64) {
                    const float s = p[c] * cf;                         // utils.py:233
                    nan |= (s != s);
                    if (s > best) { best = s; bidx = c; }              // first index wins within a lane
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const float ob = __shfl_xor(best, off);
                    const int oi = __shfl_xor(bidx, off);
                    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
                }
                nan = __any(nan);                                       // torch.max propagates NaN -> not > thr
                if (lane == src && !nan) { mybest = best; mycls = bidx; }
            } else {
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
            const bool pass = mybest > thr;                             // utils.py:243
            const u64 pm = __ballot(pass);
            if (pm) {                                                   // ONE counter atomic per wave
                int base = 0;
                if (lane == 0) base = atomicAdd(&counts[b], __popcll(pm));
                base = __shfl(base, 0);
                if (pass) {
                    const int slot = base + __popcll(pm & ((1ull << lane) - 1));
                    if (slot < max_cand) {
                        kb[slot] = make_key(mycls, mybest, row);
                        atomicAdd(&hist[mycls], 1);
                    }
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
    unsigned char* keep;    // [B][max_n]
    int* segoff;        // [B][C+1]
    u64* mask;          // [B][max_n][nw]
    u64* pkey;          // [B][max_n] keys partitioned by class (unsorted inside a class)
    int nw;
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t nms_layout(NmsWs* ws, char* base, int B, int max_n, int C) {
    const size_t n = (size_t)B * max_n;
    const int nw = (max_n + 63) / 64;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    char* p0 = take(n * 8); char* p1 = take(n * 16); char* p2 = take(n * 4); char* p3 = take(n); char* p4 = take(n);
    char* p5 = take((size_t)B * (C + 1) * 4); char* p6 = take(n * (size_t)nw * 8);
    char* p7 = take(n * 8);
    if (ws) {
        ws->skey = (u64*)p0; ws->sbox = (f32x4*)p1; ws->sconf = (float*)p2; ws->svalid = (unsigned char*)p3;
        ws->keep = (unsigned char*)p4; ws->segoff = (int*)p5; ws->mask = (u64*)p6; ws->nw = nw;
        ws->pkey = (u64*)p7;
    }
    return off;
}

// Sorting by (class, score desc, row) in two steps: scatter the keys into their class segments (any order inside a segment:
// the atomics only decide scratch positions), then rank every key among the keys of ITS segment.  Keys are unique and the
// class is the major key, so segoff[cls] + (#smaller keys of the class) is the key's rank among all keys -- the same
// permutation as rank_kernel's O(n^2) count at sum_c n_c^2 compares (dense scene, 21 k candidates in 37 classes: 13x fewer).
// An image whose candidate list overflowed (reported to the host as an error) only has to stay inside its buffers.
// One workgroup per image: class counts -> LDS, exclusive prefix (-> segoff[b][0..C], used by rank_seg / scan), then the
// scatter with LDS cursors.
__global__ __launch_bounds__(256) void segpart_kernel(const u64* __restrict__ keys, int max_cand, const int* __restrict__ counts,
                                                      const int* __restrict__ segcnt, NmsWs ws, int C, int max_n) {
    extern __shared__ int sh[];                    // [C + 1] segment offsets, then [C] fill cursors
    int* off = sh;
    int* cur = sh + C + 1;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) { off[c + 1] = segcnt[b * C + c]; cur[c] = 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c < C; ++c) { const int v = off[c + 1]; off[c] = acc; acc += v; }
        off[C] = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c <= C; c += 256) ws.segoff[b * (C + 1) + c] = off[c];
    const int n = min(min(counts[b], max_cand), max_n);
    const u64* kb = keys + (size_t)b * max_cand;
    for (int i = threadIdx.x; i < n; i += 256) {
        const u64 k = kb[i];
        const int c = min(key_cls(k), C - 1);
        const int pos = off[c] + atomicAdd(&cur[c], 1);
        if (pos < max_n) ws.pkey[(size_t)b * max_n + pos] = k;
    }
}

__global__ __launch_bounds__(256) void rank_seg_kernel(const float* __restrict__ dets, int N, int C, float nms_thr, NmsWs ws, int max_n) {
    __shared__ u64 tile[256];
    __shared__ int item[3];                                   // segment start, end, first key of this work item (-1: none left)
    extern __shared__ int so[];                               // [C + 1] this image's segment offsets
    const int b = blockIdx.y;
    const int attrib = 5 + C;
    for (int c = threadIdx.x; c <= C; c += 256) so[c] = ws.segoff[b * (C + 1) + c];
    const u64* kb = ws.pkey + (size_t)b * max_n;
    for (int t = blockIdx.x; ; t += gridDim.x) {
        // work item t of this image = (class segment, 256-key slice of it); slices are numbered segment by segment
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = 0, s = 0, e = 0, k0 = -1;
            for (int c = 0; c < C; ++c) {
                const int cs = min(so[c], max_n), ce = min(so[c + 1], max_n);
                const int nt = (ce - cs + 255) >> 8;
                if (t < acc + nt) { s = cs; e = ce; k0 = cs + (t - acc) * 256; break; }
                acc += nt;
            }
            item[0] = s; item[1] = e; item[2] = k0;
        }
        __syncthreads();
        const int s = item[0], e = item[1], k0 = item[2];
        if (k0 < 0) break;
        const int i = k0 + threadIdx.x;
        const u64 mine = i < e ? kb[i] : ~0ull;
        int rank = 0;
        for (int j0 = s; j0 < e; j0 += 256) {
            const int j = j0 + threadIdx.x;
            const u64 kj = j < e ? kb[j] : ~0ull;
            __syncthreads();
            tile[threadIdx.x] = kj;
            __syncthreads();
            const int lim = min(256, e - j0);
#pragma unroll 8
            for (int q = 0; q < lim; ++q) rank += (tile[q] < mine) ? 1 : 0;
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
            ws.keep[o] = 0;
        }
    }
}

// rank-sort + gather over ALL keys of an image.  Used with RAW = true: (row, class) = torch.nonzero order (utils.py:204-224);
// the NMS path ranks inside class segments instead (partition_kernel + rank_seg_kernel below).
template <bool RAW>
__global__ __launch_bounds__(256) void rank_kernel(const float* __restrict__ dets, int N, int C, float nms_thr,
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
            mk = RAW ? (((u64)key_row(mine) << 12) | (u64)key_cls(mine)) : mine;
        }
        int rank = 0;
        for (int j0 = 0; j0 < n; j0 += 256) {
            const int j = j0 + threadIdx.x;
            u64 kj = ~0ull;
            if (j < n) { kj = kb[j]; if (RAW) kj = ((u64)key_row(kj) << 12) | (u64)key_cls(kj); }
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
            ws.svalid[o] = RAW ? 1 : (iou_xyxy(bx, bx) > nms_thr ? 1 : 0);  // diagonal of utils.py:177
            ws.keep[o] = RAW ? 1 : 0;
        }
    }
}

// 64x64 IOU tiles -> bit masks.  bit j of mask[b][i][tj] = (j > i) & same class & IOU(i,j) > thr.
// One wave per tile; tiles whose class ranges are disjoint (keys are sorted by class) are skipped -- the first / last class of
// every 64-row tile is staged in LDS once per workgroup, so skipping costs an LDS read, not two dependent global loads
// (0.2 ms of a 21 k-candidate image went into those).
__global__ __launch_bounds__(256) void mask_kernel(const int* __restrict__ counts, int max_cand, NmsWs ws, int max_n, float thr, int lds_tiles) {
    __shared__ f32x4 cbox[4][64];
    __shared__ int ccls[4][64];
    extern __shared__ int tcls[];                        // [2][nt] first / last class of each tile (when nt <= lds_tiles)
    const int b = blockIdx.y;
    const int n = min(min(counts[b], max_cand), max_n);
    const int nt = (n + 63) >> 6;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t base = (size_t)b * max_n;
    const bool staged = nt <= lds_tiles;
    if (staged) {
        for (int t = threadIdx.x; t < nt; t += 256) {
            tcls[t] = key_cls(ws.skey[base + t * 64]);
            tcls[nt + t] = key_cls(ws.skey[base + min(t * 64 + 63, n - 1)]);
        }
        __syncthreads();
    }
    const long long ntile = (long long)nt * nt;
    for (long long t = (long long)blockIdx.x * 4 + wv; t < ntile; t += (long long)gridDim.x * 4) {
        const int ti = (int)(t / nt), tj = (int)(t - (long long)ti * nt);
        if (tj < ti) continue;
        // class ranges: rows [ti*64, ..] have classes >= first row's; disjoint from the columns' -> no pairs
        if (staged) { if (tcls[nt + ti] < tcls[tj]) continue; }
        else if (key_cls(ws.skey[base + min(ti * 64 + 63, n - 1)]) < key_cls(ws.skey[base + tj * 64])) continue;
        const int i = ti * 64 + lane, j = tj * 64 + lane;
        f32x4 bj = {0.f, 0.f, 0.f, 0.f}; int cj = -1;
        if (j < n) { bj = ws.sbox[base + j]; cj = key_cls(ws.skey[base + j]); }
        cbox[wv][lane] = bj; ccls[wv][lane] = cj;
        // single wave owns cbox[wv]: LDS ops of one wave are in order, no barrier needed
        __builtin_amdgcn_wave_barrier();
        if (i < n) {
            const f32x4 bi = ws.sbox[base + i];
            const int ci = key_cls(ws.skey[base + i]);
            u64 m = 0;
#pragma unroll 8
            for (int k = 0; k < 64; ++k) {
                const bool hit = (tj * 64 + k > i) && (ccls[wv][k] == ci) && (iou_xyxy(bi, cbox[wv][k]) > thr);
                m |= hit ? (1ull << k) : 0ull;
            }
            ws.mask[(base + i) * ws.nw + tj] = m;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Greedy scan of one (image, class) segment (reference utils.py:180-190), 64 boxes (one mask word) per
// step.  Wave 0 resolves the in-word chain on scalar registers (one iteration per KEPT box); then all four
// waves OR the kept rows' mask words into the removed set of the later words: wave v takes rows i = v mod 4,
// every lane issues its 16 row loads unconditionally (masked afterwards) so they pipeline instead of
// serialising on memory latency.
__global__ __launch_bounds__(256) void scan_kernel(const int* __restrict__ counts, int max_cand, NmsWs ws, int max_n, int C) {
    extern __shared__ u64 remv[];          // [nw] one bit per sorted position of the image, + [nw] = keep word
    const int b = blockIdx.y, c = blockIdx.x;
    const int n = min(min(counts[b], max_cand), max_n);
    int s0 = ws.segoff[b * (C + 1) + c], s1 = ws.segoff[b * (C + 1) + c + 1];
    s0 = min(s0, n); s1 = min(s1, n);
    if (s1 <= s0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w_lo = s0 >> 6, w_hi = (s1 - 1) >> 6;
    u64* keepw = remv + ws.nw;
    for (int w = w_lo + (int)threadIdx.x; w <= w_hi; w += 256) remv[w] = 0;
    __syncthreads();
    const size_t base = (size_t)b * max_n;
    for (int w = w_lo; w <= w_hi; ++w) {
        if (wv == 0) {
            const int p = w * 64 + lane;
            const bool inseg = p >= s0 && p < s1;
            u64 diag = 0; bool valid = false;
            if (inseg) { diag = ws.mask[(base + p) * ws.nw + w]; valid = ws.svalid[base + p] != 0; }
            // everything below is wave-uniform: keep it in SGPRs (readfirstlane / readlane, no LDS crossbar)
            const u64 rw = remv[w];
            u64 dead = (((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(rw >> 32))) << 32 |
                        (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)rw)) | ~__ballot(valid);
            u64 keepm = 0;
            u64 cand = ~dead;
            const int dlo = (int)diag, dhi = (int)(diag >> 32);
            while (cand) {                                   // one step per KEPT box, not per box
                const int i = __ffsll((long long)cand) - 1;
                keepm |= 1ull << i;
                const u64 di = ((u64)(unsigned)__builtin_amdgcn_readlane(dhi, i) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane(dlo, i);
                dead |= di;
                cand = ~dead & ~((2ull << i) - 1ull);        // live boxes after i
            }
            if (inseg) ws.keep[base + p] = (unsigned char)((keepm >> lane) & 1ull);
            if (lane == 0) keepw[0] = keepm;
        }
        __syncthreads();
        if (w < w_hi) {
            const u64 km = keepw[0];
            for (int w2 = w + 1 + lane; w2 <= w_hi; w2 += 64) {
                u64 acc = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int i = 4 * k + wv;
                    const int r = min(w * 64 + i, n - 1);                    // clamp: rows past the end are masked off
                    const u64 m = ws.mask[(base + r) * ws.nw + w2];
                    acc |= ((km >> i) & 1ull) ? m : 0ull;
                }
                if (acc) atomicOr(&remv[w2], acc);
            }
        }
        __syncthreads();
    }
}

// out[b][k] = x1,y1,x2,y2,conf,score,cls for the k-th kept position (utils.py:193-199)
__global__ __launch_bounds__(256) void compact_kernel(const int* __restrict__ counts, int max_cand, NmsWs ws, int max_n,
                                                      float* out, int cap, int* out_counts) {
    __shared__ int wsum[4];
    const int b = blockIdx.x;
    const int n = min(min(counts[b], max_cand), max_n);
    const size_t base = (size_t)b * max_n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int total = 0;
    for (int p0 = 0; p0 < n; p0 += 256) {
        const int p = p0 + threadIdx.x;
        const bool k = p < n && ws.keep[base + p];
        const u64 bal = __ballot(k);
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();
        int pre = 0;
        for (int q = 0; q < wv; ++q) pre += wsum[q];
        const int blocksum = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const int pos = total + pre + __popcll(bal & ((1ull << lane) - 1));
        if (k && pos < cap) {
            const u64 key = ws.skey[base + p];
            const f32x4 bx = ws.sbox[base + p];
            float* o = out + ((size_t)b * cap + pos) * 7;
            o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3];
            o[4] = ws.sconf[base + p]; o[5] = key_score(key); o[6] = (float)key_cls(key);
        }
        total += blocksum;
        __syncthreads();
    }
    if (threadIdx.x == 0) out_counts[b] = total;
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
    if (mode & YV3_PP_EVAL) hipLaunchKernelGGL(filter_kernel<true>, grid, dim3(256), hl, s, dets, N, num_class, conf_thr, prob, keys, segcnt, max_cand, cand_counts);
    else                    hipLaunchKernelGGL(filter_kernel<false>, grid, dim3(256), hl, s, dets, N, num_class, conf_thr, prob, keys, segcnt, max_cand, cand_counts);
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
    if (use_nms) {
        hipLaunchKernelGGL(segpart_kernel, dim3(B), dim3(256), (size_t)(2 * num_class + 1) * sizeof(int), s,
                           keys, max_cand, cand_counts, segcnt, ws, num_class, max_n);
        YV3_CHECK_LAUNCH();
        const int items = yv3_ceil_div(max_n, 256) + num_class;              // upper bound of (segment, slice) work items per image
        // ~2048 workgroups over the batch (each loops over its items): sparse images of a large batch need few of them (bs=64:
        // 32 per image, 0.12 ms for the NMS stage instead of 0.14 with 256), a dense small batch all of them (bs=8: 256, 0.88 vs 1.01 ms)
        int rgrid = 2048 / B; rgrid = rgrid < 32 ? 32 : rgrid > 256 ? 256 : rgrid; rgrid = items < rgrid ? items : rgrid;
        hipLaunchKernelGGL(rank_seg_kernel, dim3(rgrid, B), dim3(256), (size_t)(num_class + 1) * sizeof(int), s,
                           dets, N, num_class, nms_thr, ws, max_n);
    } else {
        hipLaunchKernelGGL(rank_kernel<true>, dim3(rb, B), dim3(256), 0, s, dets, N, num_class, nms_thr, keys, max_cand, cand_counts, ws, max_n);
    }
    YV3_CHECK_LAUNCH();
    if (use_nms) {
        const long long nt = (max_n + 63) / 64;
        long long mb = (nt * nt + 3) / 4;
        if (mb > 1024) mb = 1024;
        const int lds_tiles = nt <= 4096 ? (int)nt : 0;                       // 2 ints per tile, <= 32 KB
        hipLaunchKernelGGL(mask_kernel, dim3((unsigned)mb, B), dim3(256), (size_t)lds_tiles * 2 * sizeof(int), s,
                           cand_counts, max_cand, ws, max_n, nms_thr, lds_tiles);
        YV3_CHECK_LAUNCH();
        const size_t lds = (size_t)((max_n + 63) / 64 + 1) * 8;
        if (lds > 64 * 1024) return YV3_ESHAPE;
        hipLaunchKernelGGL(scan_kernel, dim3(num_class, B), dim3(256), lds, s, cand_counts, max_cand, ws, max_n, num_class);
        YV3_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(compact_kernel, dim3(B), dim3(256), 0, s, cand_counts, max_cand, ws, max_n, out_boxes, cap, out_counts);
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
