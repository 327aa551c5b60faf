// Neighbour sweep kernels: what PointCloud.plant_kdtree computes with one
// cKDTree.query(point, k+1) per point (pointCloudToolbox.py:81-85).
//
// Semantics reproduced: candidates are the float32-rounded coordinates
// (pct:74); squared distances are accumulated in fp64 as ((dx*dx + dy*dy) +
// dz*dz) with no FMA contraction, which is bit-for-bit what SciPy evaluates
// for 3-D data; the k+1 smallest are taken, result 0 is dropped (pct:84-85),
// and sqrt(d2) is rounded to float32 (pct:78).  Exact-distance ties are ordered
// by public index so the result does not depend on the cell order.
//
// Kernels (all hand-written for gfx950, 64-lane waves):
//   k_knn_fast   one wave = one work item (a cell and <= items_q of its owned queries).  The 27-cell stencil is staged
//                once into LDS (12 B per candidate, all global loads in flight together), the item's queries are
//                prefetched into registers.  Per PAIR of queries: float32 squared distances of all staged
//                candidates in packed arithmetic, a threshold that leaves k+1 .. 64 R of them (ballot counts,
//                secant steps), compaction of the survivors, exact fp64 keys for those only, ONE wave-wide
//                bitonic network in registers on 32-bit elements (DPP row operations, v_permlane16/32_swap,
//                v_med3_u32 compare-exchanges), proof checks, stores.  R = 1 holds k+1 <= 64, R = 2 k+1 <= 128.
//                Anything it cannot prove exact goes to the redo list (one counter increment per item).
//   k_knn_exact  one wave = one query of the redo list (or every query, for testing): candidates cube by cube
//                from global memory, (fp64 d2, public index) comparisons, shell-by-shell widening until the
//                searched cube guarantees the answer.
//   k_knn_brute  exhaustive sweep, wave per query: small clouds and the on-device cross-check.
//   k_export*    neighbour table (sorted space, owned rows) -> public (rows, k) index / distance arrays.
#include "pct_knn_sweep.h"

namespace {

// ---------------------------------------------------------------------------
// Fast sweep: wave = work item (one cell, <= items_q consecutive queries).
//
// Elements of the wave-wide network are single 32-bit integers
//     key << SLOT_BITS | payload
// payload = LDS slot of a staged stencil candidate (or, on the pre-selection path, the candidate's place in the
// compacted list of survivors, from which the slot is looked up afterwards); key = floor(d2 * scale) with the
// exact fp64 squared distance d2 and scale = 2^KEY_BITS / (2.3 cell^2), just above the largest squared distance the
// 27-cell stencil can VOUCH for (beyond it keys saturate).  The quantisation is a monotone map of the exact value, so
// wherever two keys differ the order is the exact order.  Equal keys among the first k+2 of the sorted selection are
// put in the exact order in place (order_equal_keys); a saturated (k+1)-th key, every query whose answer is
// not guaranteed to lie inside the stencil or inside what the float32 pre-selection kept, and whole items whose
// stencil does not fit the LDS staging area are appended to the redo list and done by k_knn_exact.  Unflagged
// results are therefore bit-identical to the exact path: the stored distance is recomputed in fp64 from the
// coordinates.  (DESIGN.md 4.2 walks through the steps.)
// ---------------------------------------------------------------------------
constexpr unsigned kPadElem = 0xFFFFFFFFu;

template <int R>
struct FastK {
    unsigned e[R];
};

// ---------------------------------------------------------------------------
// Sorting network of the fast sweep: bitonic merges in the "flip" form -- a merge of two ascending runs of
// SIZE/2 first compares element i with element i ^ (SIZE - 1), then runs the half-cleaners of strides
// SIZE/4 .. 1 -- in which EVERY compare-exchange leaves the smaller element at the lower index.  Which of the two
// a lane keeps therefore depends only on one bit of its lane id: six lane-constant words sel[j] = -(bit j of
// lane) serve all 21 (28) levels, and a level is  partner move + v_med3_u32  (med3(a, b, 0) = min,
// med3(a, b, ~0) = max) with no per-level mask in scalar registers.  Partner moves: DPP for xor 1, 2, 3, 7, 8, 15,
// two DPP moves for xor 4, v_permlane16/32_swap for xor 16 / 32 (the pair of results holds {own, partner} in
// lane-dependent order -- as a set that is all a compare-exchange needs), ds_bpermute for the two wide flips.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

struct SortLanes {
    unsigned sel[6];     // sel[j] = all ones if bit j of the lane id is set
    int a31, a63;        // byte addresses of lanes lane ^ 31, lane ^ 63 for ds_bpermute
};

__device__ __forceinline__ SortLanes make_sort_lanes() {
    SortLanes c;
    const int lane = lane_id();
#pragma unroll
    for (int j = 0; j < 6; ++j) c.sel[j] = (unsigned)__builtin_amdgcn_sbfe(lane, j, 1);
    c.a31 = (lane ^ 31) << 2;
    c.a63 = (lane ^ 63) << 2;
    return c;
}

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v / 2); }

// compare-exchange with the element STRIDE lanes away (STRIDE < 64), smaller one to the lower lane
template <int R, int STRIDE>
__device__ __forceinline__ void fast_stride(FastK<R>& t, const SortLanes& c) {
    const unsigned sel = c.sel[ilog2(STRIDE)];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if constexpr (STRIDE == 16) {
            const auto p = __builtin_amdgcn_permlane16_swap(t.e[r], t.e[r], false, false);
            t.e[r] = umed3(p[0], p[1], sel);
        } else if constexpr (STRIDE == 32) {
            const auto p = __builtin_amdgcn_permlane32_swap(t.e[r], t.e[r], false, false);
            t.e[r] = umed3(p[0], p[1], sel);
        } else {
            t.e[r] = umed3(t.e[r], (unsigned)lane_xor<STRIDE>((int)t.e[r]), sel);
        }
    }
}

// first step of a merge of SIZE elements: element i against element i ^ (SIZE - 1)
template <int R, int SIZE>
__device__ __forceinline__ void fast_flip(FastK<R>& t, const SortLanes& c) {
    if constexpr (SIZE <= 64) {
        const unsigned sel = c.sel[ilog2(SIZE) - 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            unsigned pk;
            if constexpr (SIZE == 2) pk = (unsigned)__builtin_amdgcn_mov_dpp((int)t.e[r], 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
            else if constexpr (SIZE == 4) pk = (unsigned)__builtin_amdgcn_mov_dpp((int)t.e[r], 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
            else if constexpr (SIZE == 8) pk = (unsigned)__builtin_amdgcn_mov_dpp((int)t.e[r], 0x141, 0xF, 0xF, true);  // row_half_mirror
            else if constexpr (SIZE == 16) pk = (unsigned)__builtin_amdgcn_mov_dpp((int)t.e[r], 0x140, 0xF, 0xF, true); // row_mirror
            else if constexpr (SIZE == 32) pk = (unsigned)__builtin_amdgcn_ds_bpermute(c.a31, (int)t.e[r]);
            else pk = (unsigned)__builtin_amdgcn_ds_bpermute(c.a63, (int)t.e[r]);
            t.e[r] = umed3(t.e[r], pk, sel);
        }
    } else {
        static_assert(SIZE == 128 && R == 2, "two registers per lane at most");
        const unsigned lo_rev = (unsigned)__builtin_amdgcn_ds_bpermute(c.a63, (int)t.e[0]);
        const unsigned hi_rev = (unsigned)__builtin_amdgcn_ds_bpermute(c.a63, (int)t.e[1]);
        t.e[0] = min(t.e[0], hi_rev);
        t.e[1] = max(t.e[1], lo_rev);
    }
}

template <int R, int STRIDE>
__device__ __forceinline__ void fast_strides(FastK<R>& t, const SortLanes& c) {
    if constexpr (STRIDE >= 1) {
        fast_stride<R, STRIDE>(t, c);
        fast_strides<R, STRIDE / 2>(t, c);
    }
}

// ascending sort of 64 R elements (element index = lane + 64 * register), starting from sorted runs of SIZE / 2
template <int R, int SIZE>
__device__ __forceinline__ void fast_sort_from(FastK<R>& t, const SortLanes& c) {
    fast_flip<R, SIZE>(t, c);
    fast_strides<R, SIZE / 4>(t, c);
    if constexpr (SIZE < 64 * R) fast_sort_from<R, SIZE * 2>(t, c);
}

// NSETS independent ascending sorts of 64 R elements each (set s = registers s R .. s R + R - 1), level by level side
// by side: the lane-level steps treat all NSETS R registers alike, only the 128-wide flip pairs registers per set
template <int R, int NSETS, int SIZE>
__device__ __forceinline__ void fast_sort_sets(FastK<NSETS * R>& t, const SortLanes& c) {
    if constexpr (SIZE <= 64) {
        fast_flip<NSETS * R, SIZE>(t, c);
    } else {
        static_assert(SIZE == 128 && R == 2, "two registers per set at most");
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            const unsigned lo_rev = (unsigned)__builtin_amdgcn_ds_bpermute(c.a63, (int)t.e[2 * s]);
            const unsigned hi_rev = (unsigned)__builtin_amdgcn_ds_bpermute(c.a63, (int)t.e[2 * s + 1]);
            t.e[2 * s] = min(t.e[2 * s], hi_rev);
            t.e[2 * s + 1] = max(t.e[2 * s + 1], lo_rev);
        }
    }
    fast_strides<NSETS * R, (SIZE / 4 < 32 ? SIZE / 4 : 32)>(t, c);
    if constexpr (SIZE < 64 * R) fast_sort_sets<R, NSETS, SIZE * 2>(t, c);
}

// ---------------------------------------------------------------------------
// Equal keys inside the sorted list: the quantised key cannot order those elements, the exact values can.
// On the reference's own generator output (theta x phi lattices, utils.py:883-914) every point has symmetric partners
// whose squared distances differ by float32 rounding noise only -- most queries meet at least one pair of equal keys
// among their first k+2 entries, and handing each of them to the wave-per-query exact sweep costs 10-50x the fast
// path.  Instead the list is repaired in place: an odd-even transposition over the sorted list in which two
// neighbours are compared -- by exact fp64 d2, then by public index, the total order of k_knn_exact -- ONLY when
// their keys are equal.  Elements with different keys never move, so every run of equal keys ends up in the exact
// order and everything proven on keys (the (k+1)-th key against the stencil radius, the pre-selection cut, eps)
// stays proven.  Runs are short (2, 4 or 8 symmetric partners): two or three passes and a quiet round.
//   exact_d2(payload)  fp64 squared distance of the element with that payload (LDS reads only; real elements only)
//   pos_of(payload)    its sorted position; may use cross-lane reads: called with every lane active
// Returns false when the list is still not in order after kOrderPasses (a long pile of equal keys): redo list.
// ---------------------------------------------------------------------------
constexpr int kOrderPasses = 36;

template <int R, int SLOT_BITS, class ExactD2, class PosOf>
__device__ __forceinline__ bool order_equal_keys(unsigned* e, const float4* __restrict__ pts, const ExactD2& exact_d2,
                                                 const PosOf& pos_of) {
    const int lane = lane_id();
    constexpr unsigned PAYLOAD = (1u << SLOT_BITS) - 1u;
    double d[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        d[r] = INFINITY;
        if (e[r] != kPadElem) d[r] = exact_d2(e[r] & PAYLOAD);
    }
    int quiet = 0;
#pragma unroll 1
    for (int pass = 0; pass < kOrderPasses; ++pass) {
        const int par = pass & 1;
        const int pl = ((lane - par) ^ 1) + par;      // partner lane: -1 / 64 = last / first lane of the neighbouring register
        const int addr = (pl & 63) << 2;
        const int dr = pl >> 6;                       // -1, 0, +1: register of the partner relative to mine
        unsigned be[R];
        int blo[R], bhi[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            be[r] = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)e[r]);
            blo[r] = __builtin_amdgcn_ds_bpermute(addr, __double2loint(d[r]));
            bhi[r] = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(d[r]));
        }
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int pr = r + dr;
            unsigned pe = kPadElem;
            double pd = INFINITY;
#pragma unroll
            for (int r2 = 0; r2 < R; ++r2)
                if (pr == r2) { pe = be[r2]; pd = __hiloint2double(bhi[r2], blo[r2]); }
            const bool same = pr >= 0 && pr < R && e[r] != kPadElem && pe != kPadElem && ((e[r] ^ pe) >> SLOT_BITS) == 0u;
            bool p_lt_m = pd < d[r], m_lt_p = d[r] < pd;
            const bool tie = same && pd == d[r];
            if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {       // exact tie somewhere: public indices decide
                const int my_pos = pos_of(e[r] & PAYLOAD), p_pos = pos_of(pe & PAYLOAD);
                if (tie) {
                    const int mp = pub_index(pts, my_pos), pp = pub_index(pts, p_pos);
                    p_lt_m = pp < mp;
                    m_lt_p = mp < pp;
                }
            }
            const bool take = same && (pl > lane ? p_lt_m : m_lt_p);     // the lower position keeps the smaller one
            if (take) { e[r] = pe; d[r] = pd; }
            any |= take;
        }
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) quiet = 0;
        else if (++quiet == 2) return true;
    }
    return false;
}

// waves per block of the fast sweep: chosen so that whole blocks fill the 160 KiB of LDS (waves are independent;
// R = 1: 6 blocks x 4 waves x 6.25 KiB, R = 2: 4 blocks x 4 waves x 9.5 KiB)
#ifndef PCT_FAST_WAVES
#define PCT_FAST_WAVES 4
#endif
template <int R> constexpr int kFastWaves = PCT_FAST_WAVES;

// EPS = the hybrid eps-ball query is on (candidates beyond eps do not count); without it every staged slot is a
// candidate and the per-batch eps compares and candidate counts drop out.
// PRE = float32 pre-selection (clouds whose query coordinates are the float32 tree coordinates): the threshold
// that cuts the staged candidates down to <= 64 R is searched on squared distances computed in packed float32
// (two candidates per instruction), and only the survivors get the exact fp64 distance and a key.  A float32
// squared distance of float32 points is within 5 * 2^-24 relative of the exact one (the difference of two
// floats is rounded once, then one product and two fused multiply-adds), so a candidate that was cut has an
// exact squared distance >= T (1 - 2^-20): the query is accepted only if its (k+1)-th exact key lies below that.
typedef float float2v __attribute__((ext_vector_type(2)));

// Level passes (pct_levels.hip) want to know WHY a row was not answered: 1 = the stencil cannot vouch for the answer
// (cells too small for this query), 2 = the stencil overflowed the staging area (cells too large), 3 = anything else;
// they get it, with the stencil population, in the row's slot of redo_m.  Plain sweeps append the bare row to the list.

// PAIR (with PRE, R = 1): two queries of the item per loop trip, their instruction streams side by side in the same
// basic blocks -- they share the LDS reads of the candidates, and each hides the other's dependency stalls.
template <int R, bool EPS, bool PRE, bool PAIR = false, bool Q64 = false, bool TREE = false>
__global__ __launch_bounds__(64 * kFastWaves<R>, (TREE ? (R == 1 ? (PCT_TREE_CAP <= 512 ? 6 : PCT_TREE_CAP <= 768 ? 4 : 3) : (PCT_TREE_CAP2 <= 768 ? 4 : 3)) : R == 1 ? (Q64 ? 5 : 6) : 4)) void k_knn_fast(KnnArgs a, const int2* __restrict__ items, int64_t n_items,
                                                                  int items_q, int* __restrict__ redo,
                                                                  int* __restrict__ redo_count) {
#ifndef PCT_STAGE_CAP2
#define PCT_STAGE_CAP2 768
#endif
    // (tree items: an octree level changes the population fourfold on a surface; segments whose stencil exceeds the
    // staging area are split further at build time (k_tree_refine), so the capacity trades refinement and trips to
    // the exact sweep against occupancy: 768 slots / 4 blocks per CU measured best -- 1/r^2 scan, 1 M points: fast
    // sweep 0.51 | 0.65 | 0.83 ms at 512 | 768 | 1024 slots, whole call 2.30 | 1.57 | 1.65 ms)
    constexpr int CAP = TREE ? (R == 1 ? PCT_TREE_CAP : PCT_TREE_CAP2) : R == 1 ? kStageCap : PCT_STAGE_CAP2;   // staged stencil candidates per wave
    // low bits of a network element: the staged slot of the candidate, or -- pre-selection -- its place in the
    // compacted list of survivors (6 / 7 bits; the slot is looked up in that list afterwards), which leaves three
    // more bits for the key and cuts key collisions eightfold
    constexpr int SLOT_BITS = PRE ? (R == 1 ? 6 : 7) : (R == 1 ? 9 : 10);
    constexpr int KEY_BITS = 32 - SLOT_BITS;
    constexpr int CAP_POW2 = 1024;          // slots < CAP <= 1024: masks a garbage list entry read for a padding element
    static_assert(CAP <= CAP_POW2, "staging capacity");
    static_assert(PRE || CAP <= (1 << SLOT_BITS), "slot field too narrow");
    __shared__ float s_cx[kFastWaves<R>][CAP];        // staged stencil, structure of arrays:
    __shared__ float s_cy[kFastWaves<R>][CAP];        // 12 B per candidate
    __shared__ float s_cz[kFastWaves<R>][CAP];
    __shared__ unsigned s_pend[kFastWaves<R>][64 * R];
    __shared__ unsigned short s_pend2[kFastWaves<R>][PAIR ? 64 * R : 2];    // survivors of the second query of a pair (16 bits: R = 2 stays at 4 blocks per CU)
    __shared__ int s_offc[kFastWaves<R>][16];          // sorted position - flat slot, per non-empty run

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int64_t item = (int64_t)blockIdx.x * kFastWaves<R> + w;
    if (item >= n_items) return;
    const SortLanes sort_dir = make_sort_lanes();

    pct_grid g = a.g;
    const int* __restrict__ cs = a.cell_start;
    const int2 it2 = items[item];
    int cx, cy, cz, qs, nq, row0;
    constexpr int NRUNS = TREE ? 27 : 9;           // ranges of the cloud the stencil is staged from
    int run_s = 0, run_len = 0;
    float* cand_x = s_cx[w];
    float* cand_y = s_cy[w];
    float* cand_z = s_cz[w];
    unsigned* pend = s_pend[w];
    int* offc = s_offc[w];
    if constexpr (TREE) {
        // item = {first query (Morton position = table row) | (queries - 1) << 26, segment}
        const int seg = __builtin_amdgcn_readfirstlane(it2.y);
        if (seg < 0) return;                       // an item of a segment that was split (pct_tree.hip: k_tree_refine)
        const unsigned packed = (unsigned)__builtin_amdgcn_readfirstlane(it2.x);
        qs = (int)(packed & 0x3ffffffu);
        nq = (int)(packed >> 26) + 1;
        row0 = qs;
        const int4 hd = a.tree_seg[seg];
        const int level = __builtin_amdgcn_readfirstlane(hd.x);
        cx = __builtin_amdgcn_readfirstlane(hd.y);
        cy = __builtin_amdgcn_readfirstlane(hd.z);
        cz = __builtin_amdgcn_readfirstlane(hd.w);
        // the grid of this level: edges scale by exact powers of two, so (x - o) * inv_cell - cx lies in [0, 1) for
        // every point the Morton code put into the cell
        g.cell = __builtin_ldexp(a.g.cell, level);
        g.inv_cell = __builtin_ldexp(a.g.inv_cell, -level);
        g.nx = g.ny = g.nz = 1 << (a.tree_bits - level);
        if (lane < 27) {
            const int2 r = a.tree_runs[(int64_t)seg * 27 + lane];
            run_s = r.x;
            run_len = r.y;
        }
    } else {
        const int cell = __builtin_amdgcn_readfirstlane(it2.x);
        const int chunk = __builtin_amdgcn_readfirstlane(it2.y);
        cx = cell % g.nx;
        cy = (cell / g.nx) % g.ny;
        cz = cell / (g.nx * g.ny);
        qs = cs[cell] + chunk * items_q;                       // owned points sit first in the cell
        const int qe = min(cs[cell] + a.cell_own[cell], qs + items_q);
        nq = qe - qs;
        row0 = a.own_start[cell] + chunk * items_q;            // neighbour-table row of query qs

        // ---- bounds of the 9 x-runs of the 27-cell stencil, fetched in parallel by lanes 0..8 (centre row first)
        if (lane < 9) {
            const int z = cz + kRowOrder[lane][0], y = cy + kRowOrder[lane][1];
            if (z >= 0 && z < g.nz && y >= 0 && y < g.ny) {
                const int row = (z * g.ny + y) * g.nx;
                run_s = cs[row + max(cx - 1, 0)];
                run_len = cs[row + min(cx + 1, g.nx - 1) + 1] - run_s;
            }
        }
    }
    // the item's own queries (<= items_q <= 64 consecutive sorted positions), one per lane
    float4 my_q = make_float4(0.f, 0.f, 0.f, 0.f);
    double4 my_qd = make_double4(0., 0., 0., 0.);
    if (lane < nq) {
        my_q = a.pts[qs + lane];
        if (a.ptsd) my_qd = a.ptsd[qs + lane];
    }
    float my_eq = 0.f;       // Q64: distance between the float64 query and its float32 rounding, rounded up
    if constexpr (Q64) {
        const double ex = my_qd.x - (double)my_q.x, ey = my_qd.y - (double)my_q.y, ez = my_qd.z - (double)my_q.z;
        my_eq = (float)sqrt((ex * ex + ey * ey) + ez * ez) * (1.0f + 0x1p-22f);
        if (!(my_eq >= 0.f)) my_eq = INFINITY;        // NaN cannot happen with finite inputs; be safe
    }
    // exclusive prefix of the run lengths over lanes 0..8 = first flat slot of every run; m = staged candidates
    int my_pre = 0, m = 0;
    {
        int acc = 0;
#pragma unroll
        for (int t = 0; t < NRUNS; ++t) {
            my_pre = lane == t ? acc : my_pre;
            acc += __builtin_amdgcn_readlane(run_len, t);
        }
        m = acc;
    }
    unsigned long long n_flush = 0, n_step = 0, n_redo = 0;
    // rows of this item that go to the redo list: collected here (bit = query of the item, two bits of reason) and
    // appended with ONE counter increment when the item is done -- a counter increment per query serialises at the
    // memory side (~88 per us on one address: 10^5 failing queries of a cloud of uneven density cost a millisecond)
    unsigned long long redo_mask = 0ull, redo_why_lo = 0ull, redo_why_hi = 0ull;
    const auto note_redo = [&](int row, int why) {
        const int q = row - row0;
        redo_mask |= 1ull << q;
        redo_why_lo |= (unsigned long long)(why & 1) << q;
        redo_why_hi |= (unsigned long long)((why >> 1) & 1) << q;
    };

    // (a pass of the density-adaptive sweep, no eps bound: a stencil that does not even hold k+1 points cannot answer
    // any of the item's queries -- every point is binned somewhere -- so the item is classified "cells too small"
    // without being swept)
    const bool hopeless = !EPS && a.row_done != nullptr && m < a.k + 1;
    // (tree items: the slot -> position code has four bits per slot, 16 non-empty ranges -- a surface meets about ten
    // of its 27 stencil cells; more is a volume, where these items pay as little as uniform cells do)
    const bool crowded = TREE && __popcll(__builtin_amdgcn_ballot_w64(lane < NRUNS && run_len > 0)) > 16;
    if (m > CAP || hopeless || crowded) {
        // stencil does not fit the staging area (dense cluster): the exact sweep takes the whole item
        if (a.row_done) {
            // passes of the density-adaptive sweep keep no list: reason and stencil population go to the row's own
            // slot (a cloud of very uneven density fails hundreds of thousands of one-query items per pass, and as
            // many increments of ONE counter serialise for milliseconds)
            if (lane < nq) a.redo_m[row0 + lane] = ((hopeless ? 1 : 2) << 29) | min(m, (1 << 29) - 1);
            return;
        }
        int base = 0;
        if (lane == 0) base = atomicAdd(redo_count, nq);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < nq) {
            redo[base + lane] = row0 + lane;
        }
        if (a.stats && lane == 0) {
            if (!hopeless) atomicAdd(&a.counters[1], 1ull);
            atomicAdd(&a.counters[4], (unsigned long long)nq);
        }
        return;
    }

    // ---- copy the runs as one flat range: all global loads of the item are in flight together.
    // Flat slot j belongs to the u-th non-empty run, u = (number of run starts <= j) - 1.  The run starts are
    // marked in a CAP-bit string in LDS (the list area is free here), so that a batch of 64 slots gets its u
    // from one 64-bit word and a masked bit count instead of eight compares; offc[u] = sorted position - flat
    // slot of run u.  u is also remembered in 4 bits per staged slot (run_code: this lane's slots lane,
    // 64 + lane, ...) for the store phase, which turns a slot back into a sorted position with one cross-lane
    // read and one LDS read.
    unsigned run_code[(CAP / 64 + 7) / 8];
#pragma unroll
    for (int i = 0; i < (CAP / 64 + 7) / 8; ++i) run_code[i] = 0u;
    {
        unsigned* bits = pend;
        static_assert(CAP / 32 <= 64 * R, "bit string does not fit the list area");
        if (lane < CAP / 32) bits[lane] = 0u;
        wave_lds_sync();
        const bool nonempty = lane < NRUNS && run_len > 0;
        const unsigned long long ne = __builtin_amdgcn_ballot_w64(nonempty);
        if (nonempty) {
            atomicOr(&bits[my_pre >> 5], 1u << (my_pre & 31));
            offc[__builtin_amdgcn_mbcnt_lo((unsigned)ne, 0)] = run_s - my_pre;     // lanes 0..26: low word only
        }
        wave_lds_sync();
        float4 tmp[CAP / 64];
        int ubase = -1;
#pragma unroll
        for (int b = 0; b < CAP / 64; ++b) {
            tmp[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b * 64 < m) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)bits[2 * b]);
                const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)bits[2 * b + 1]);
                const unsigned long long B = ((unsigned long long)hi << 32) | lo;
                const unsigned long long S = B >> 1;               // starts <= lane  =  starts of (B >> 1) below lane, + bit 0
                const int c0 = ubase + (int)(lo & 1u);
                const int u = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(S >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)S, (unsigned)c0));
                ubase += (int)__popcll(B);
                const int j = b * 64 + lane;
                run_code[b >> 3] |= (unsigned)u << (4 * (b & 7));
                if (j < m) tmp[b] = a.pts[j + offc[u]];
            }
        }
        wave_lds_sync();                      // the bit string is dead: the list area goes back to the queries
#pragma unroll
        for (int b = 0; b < CAP / 64; ++b) {
            const int j = b * 64 + lane;
            if (j < m) {
                cand_x[j] = tmp[b].x; cand_y[j] = tmp[b].y; cand_z[j] = tmp[b].z;
            } else if (PRE && (b & ~1) * 64 < m) {
                // the pre-selection works on pairs of batches: unused slots sit at +inf and never pass a threshold
                cand_x[j] = INFINITY; cand_y[j] = 0.f; cand_z[j] = 0.f;
            }
        }
    }
    wave_lds_sync();
    const int k = a.k;
    const double eps2 = EPS ? a.eps2 : (double)INFINITY;
    // Key range: the cube of 27 cells vouches for at most 1.5 cell edges around a query (min(gx + 1, 2 - gx) <= 1.5 per
    // axis, guaranteed_r2), so no accepted list holds a squared distance beyond 2.25 cell^2: candidates farther out
    // (the stencil reaches 12 cell^2) may share the saturated key -- if the (k+1)-th is among them the query was
    // beyond the guarantee anyway.  (Until round 2 the range was the stencil's 12.1 cell^2: keys 5x coarser, equal
    // keys 5x as frequent.  Queries at the rim of the grid, whose guarantee is unbounded on a side, can lose a
    // provable answer to the saturation check: exact sweep.)
    constexpr double kKeyRange = 2.3;
    const double scale = (double)(1u << KEY_BITS) / (kKeyRange * g.cell * g.cell);
    const unsigned key_max = (1u << KEY_BITS) - 1u;
    // Per query, the largest key the stencil can vouch for: lane l evaluates query l once per item (the radius
    // guaranteed by the 27-cell cube depends on where the query sits inside its cell).  floor() keeps it conservative.
    unsigned my_gkey;
    {
        const double lqx = a.ptsd ? my_qd.x : (double)my_q.x, lqy = a.ptsd ? my_qd.y : (double)my_q.y,
                     lqz = a.ptsd ? my_qd.z : (double)my_q.z;
        const double gx = (lqx - g.ox) * g.inv_cell - cx;
        const double gy = (lqy - g.oy) * g.inv_cell - cy;
        const double gz = (lqz - g.oz) * g.inv_cell - cz;
        // 0xFFFFFFFF only when nothing bounds the answer (the cube covers the grid and no points were left out):
        // that alone vouches for "fewer than k+1 points exist".  A query clamped into a boundary cell from far
        // outside the grid box can have a finite guarantee beyond the key range: keep it below the sentinel.
        const double g2 = fmin(guaranteed_r2(g, cx, cy, cz, gx, gy, gz, 1), limit_r2(g, cx, cy, cz, gx, gy, gz));
        my_gkey = g2 == INFINITY ? 0xFFFFFFFFu : (unsigned)fmin(g2 * scale, 4294967294.0);
    }
    // ceil(eps^2 * scale): the whole eps ball must be inside the guaranteed radius too
    const unsigned eps_key = EPS && eps2 < 1e300 ? (unsigned)fmin(ceil(eps2 * scale), 4294967295.0) : 0xFFFFFFFFu;

    constexpr int NB = CAP / 64;             // candidate registers per lane: slot = b * 64 + lane
    constexpr int LIST = 64 * R;             // capacity of the sorted list
    unsigned t_prev = 0;                     // threshold of the previous query of this item (0 = none yet)
    float t_prev_f = 0.f;                    // same for the float32 pre-selection
    const float cell2f = (float)(g.cell * g.cell);
    // eps^2 rounded up generously in float32: everything inside the eps ball passes the pre-selection, the exact
    // test follows on the survivors
    const float eps2a = EPS ? (float)fmin(eps2 * (1.0 + 0x1p-18), 3.0e38) : INFINITY;
    const double eps1 = EPS ? sqrt(eps2) * (1.0 + 0x1p-50) : 0.0;       // eps itself, rounded up

    if constexpr (PRE && PAIR) {
        unsigned short* pend_b = s_pend2[w];
        const auto push_redo = [&](int row, int why) { note_redo(row, why); };
        // smallest exact key a candidate cut by the float32 threshold T can have: its float32 d'^2 >= T means the
        // exact d'^2 >= T (1 - 2^-20) (arithmetic error of the packed evaluation); for a float64 query the exact
        // distance to the true query is at least d' - eq
        const auto cut_key = [&](float T, double eq) {
            double lo2 = (double)T * (1.0 - 0x1p-20);
            if constexpr (Q64) {
                // (sqrt(L) - eq)^2 >= L - 2 eq sqrt(L); an upper bound of the root is enough: float32 root, rounded up
                const double root_up = (double)__builtin_sqrtf(T) * (1.0 + 0x1p-21);
                lo2 = fmax(lo2 - 2.0 * eq * root_up, 0.0);
            }
            return (unsigned)fmin(lo2 * scale, 4294967294.0);
        };
        // The per-query body is compiled once per number of staged batch PAIRS in use (NBP: 128 slots each): the loops
        // over the batches are then straight code -- the "is this batch in use" tests were a scalar compare and a
        // branch per batch, per loop, per query, on a kernel whose scalar unit is as busy as its vector units.  The
        // smallest variant also serves the items with fewer pairs and keeps the tests (GUARD).
        const auto pair_loop = [&](auto NBP_, auto GUARD_) {
            constexpr int NBP = decltype(NBP_)::value, NBU = 2 * NBP;
            constexpr bool GUARD = decltype(GUARD_)::value;
            for (int qi = 0; qi < nq; qi += 2) {
                const bool live_b = qi + 1 < nq;             // an odd tail runs its last query twice, the copy is discarded
                const int qj = live_b ? qi + 1 : qi;
                const int row_a = row0 + qi, row_b = row0 + qj;
                float ax, ay, az, bx, by, bz;
                if constexpr (!Q64) {
                    ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.x), qi));
                    ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.y), qi));
                    az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.z), qi));
                    bx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.x), qj));
                    by = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.y), qj));
                    bz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.z), qj));
                }
                // Float64 cloud (Q64): the candidates are the float32-rounded points (the reference's tree data, pct:74) and
                // my_q is the query ROUNDED to float32, so the pre-selection measures distances to a point that lies
                // eq = |q64 - q32| away from the true query: every bound taken from it moves by eq (triangle inequality).
                double qax, qay, qaz, qbx, qby, qbz, eq_a = 0.0, eq_b = 0.0;
                if constexpr (Q64) {
                    const auto rl = [&](double v, int l) {
                        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
                    };
                    qax = rl(my_qd.x, qi); qay = rl(my_qd.y, qi); qaz = rl(my_qd.z, qi);
                    qbx = rl(my_qd.x, qj); qby = rl(my_qd.y, qj); qbz = rl(my_qd.z, qj);
                    ax = (float)qax; ay = (float)qay; az = (float)qaz;         // == the float32 record of the point (k_pack_f64)
                    bx = (float)qbx; by = (float)qby; bz = (float)qbz;
                    eq_a = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_eq), qi));
                    eq_b = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_eq), qj));
                } else {
                    qax = (double)ax; qay = (double)ay; qaz = (double)az; qbx = (double)bx; qby = (double)by; qbz = (double)bz;
                }
                // ---- float32 squared distances of ALL staged candidates to both queries (one set of LDS reads) --------
                float ap_a[NBU], ap_b[NBU];
#pragma unroll
                for (int p2 = 0; p2 < NBP; ++p2) {
                    ap_a[2 * p2] = ap_a[2 * p2 + 1] = INFINITY;
                    ap_b[2 * p2] = ap_b[2 * p2 + 1] = INFINITY;
                    if (!GUARD || p2 * 128 < m) {
                        const int sa = p2 * 128 + lane, sb = sa + 64;
                        const float2v vx = {cand_x[sa], cand_x[sb]}, vy = {cand_y[sa], cand_y[sb]}, vz = {cand_z[sa], cand_z[sb]};
                        {
                            const float2v dx = vx - ax, dy = vy - ay, dz = vz - az;
                            float2v d = dx * dx;
                            d = __builtin_elementwise_fma(dy, dy, d);
                            d = __builtin_elementwise_fma(dz, dz, d);
                            ap_a[2 * p2] = d.x;
                            ap_a[2 * p2 + 1] = d.y;
                        }
                        {
                            const float2v dx = vx - bx, dy = vy - by, dz = vz - bz;
                            float2v d = dx * dx;
                            d = __builtin_elementwise_fma(dy, dy, d);
                            d = __builtin_elementwise_fma(dz, dz, d);
                            ap_b[2 * p2] = d.x;
                            ap_b[2 * p2 + 1] = d.y;
                        }
                        n_step += 4;
                    }
                }
                // ---- thresholds: k+1 <= #(d < T) <= LIST for each query, never beyond the eps ball ----------------------
                float T_a = EPS ? eps2a : INFINITY, T_b = T_a;
                if constexpr (EPS && Q64) {          // exact d < eps  =>  d' < eps + eq
                    const double ea = eps1 + eq_a, eb = eps1 + eq_b;
                    T_a = (float)fmin(ea * ea * (1.0 + 0x1p-18), 3.0e38);
                    T_b = (float)fmin(eb * eb * (1.0 + 0x1p-18), 3.0e38);
                }
                int tot_a = m, tot_b = m;
                if constexpr (EPS) {
                    tot_a = tot_b = 0;
#pragma unroll
                    for (int b = 0; b < NBU; ++b)
                        if (!GUARD || (b & ~1) * 64 < m) {
                            tot_a += (int)__popcll(__builtin_amdgcn_ballot_w64(ap_a[b] < T_a));
                            tot_b += (int)__popcll(__builtin_amdgcn_ballot_w64(ap_b[b] < T_b));
                        }
                }
                int cnt_a = tot_a, cnt_b = tot_b;
                bool ok_a = true, ok_b = live_b;              // still on the fast path
                unsigned bkey_a = 0xFFFFFFFFu, bkey_b = 0xFFFFFFFFu;
                const bool need_a = tot_a > LIST, need_b = live_b && tot_b > LIST;
                if (need_a || need_b) {
                    const float target = 0.5f * (float)(k + 1 + LIST);
                    float t0 = t_prev_f > 0.f ? t_prev_f : cell2f;
                    if (!(t0 < T_a)) t0 = 0.5f * T_a;
                    float lo_a = 0.f, hi_a = T_a, t_a = t0, lo_b = 0.f, hi_b = T_b, t_b = t0;
                    bool go_a = need_a, go_b = need_b, found_a = !need_a, found_b = !need_b;
#pragma unroll 1
                    for (int trial = 0; trial < 16 && (go_a || go_b); ++trial) {
                        int c_a = 0, c_b = 0;
#pragma unroll
                        for (int b = 0; b < NBU; ++b)
                            if (!GUARD || (b & ~1) * 64 < m) {
                                c_a += (int)__popcll(__builtin_amdgcn_ballot_w64(ap_a[b] < t_a));
                                c_b += (int)__popcll(__builtin_amdgcn_ballot_w64(ap_b[b] < t_b));
                            }
                        if (go_a) {
                            if (c_a >= k + 1 && c_a <= LIST) { T_a = t_a; cnt_a = c_a; found_a = true; go_a = false; }
                            else {
                                if (c_a < k + 1) lo_a = t_a; else hi_a = t_a;
                                float nt = c_a > 0 ? t_a * target * __builtin_amdgcn_rcpf((float)c_a) : 4.f * t_a;
                                if (!(nt > lo_a && nt < hi_a)) nt = hi_a < INFINITY ? 0.5f * (lo_a + hi_a) : 2.f * lo_a;
                                nt = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nt)));
                                if (!(nt > lo_a && nt < hi_a)) go_a = false; else t_a = nt;   // no float left between
                            }
                        }
                        if (go_b) {
                            if (c_b >= k + 1 && c_b <= LIST) { T_b = t_b; cnt_b = c_b; found_b = true; go_b = false; }
                            else {
                                if (c_b < k + 1) lo_b = t_b; else hi_b = t_b;
                                float nt = c_b > 0 ? t_b * target * __builtin_amdgcn_rcpf((float)c_b) : 4.f * t_b;
                                if (!(nt > lo_b && nt < hi_b)) nt = hi_b < INFINITY ? 0.5f * (lo_b + hi_b) : 2.f * lo_b;
                                nt = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nt)));
                                if (!(nt > lo_b && nt < hi_b)) go_b = false; else t_b = nt;
                            }
                        }
                    }
                    if (need_a) {
                        ok_a = found_a && T_a >= 1e-30f;
                        if (ok_a) { t_prev_f = T_a; bkey_a = cut_key(T_a, eq_a); }
                    }
                    if (need_b) {
                        ok_b = ok_b && found_b && T_b >= 1e-30f;
                        if (ok_b) { t_prev_f = T_b; bkey_b = cut_key(T_b, eq_b); }
                    }
                    if (!ok_a) { push_redo(row_a, 3); T_a = 0.f; cnt_a = 0; }        // nothing passes, nothing is stored
                    if (!ok_b) { if (live_b) push_redo(row_b, 3); T_b = 0.f; cnt_b = 0; }
                    if (!ok_a && !ok_b) continue;
                }
                if (!live_b) { T_b = 0.f; cnt_b = 0; }
                // ---- compact the slots of the survivors of both queries, then exact keys for them only ------------------
                FastK<2 * R> both;                 // set 0 = query a (registers 0 .. R-1), set 1 = query b
                float out_d[2 * R];                // float32 distance / sorted position of survivor lane + 64 r of each set
                int out_p[2 * R];
                const auto slot_to_pos = [&](int j) {
                    unsigned code = (unsigned)__builtin_amdgcn_ds_bpermute((j & 63) << 2, (int)run_code[0]);
                    if constexpr ((CAP / 64 + 7) / 8 > 1) {
                        const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute((j & 63) << 2, (int)run_code[1]);
                        code = (j >> 9) ? hi : code;
                    }
                    return j + offc[(code >> ((((unsigned)j >> 6) & 7u) << 2)) & 15u];
                };
                {
                    int base_a = 0, base_b = 0;
                    wave_lds_sync();
#pragma unroll
                    for (int b = 0; b < NBU; ++b) {
                        if (!GUARD || (b & ~1) * 64 < m) {
                            const bool pa = ap_a[b] < T_a, pb = ap_b[b] < T_b;
                            const unsigned long long ma = __builtin_amdgcn_ballot_w64(pa), mb = __builtin_amdgcn_ballot_w64(pb);
                            if (pa) pend[base_a + __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0))] = (unsigned)(b * 64 + lane);
                            if (pb) pend_b[base_b + __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0))] = (unsigned short)(b * 64 + lane);
                            base_a += (int)__popcll(ma);
                            base_b += (int)__popcll(mb);
                        }
                    }
                    wave_lds_sync();
                    // Survivor i's exact distance and sorted position are worked out here, by the lane that holds its
                    // coordinates anyway, and parked in that lane (out_d / out_p); after the sort the lane that ends up
                    // with list entry i fetches them with one cross-lane read each instead of recomputing them.
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int i = lane + 64 * r;
                        unsigned e_a = kPadElem, e_b = kPadElem;
                        const int ja = (int)pend[i] & (CAP_POW2 - 1), jb = (int)pend_b[i] & (CAP_POW2 - 1);   // stale beyond cnt: masked, unused
                        out_p[r] = slot_to_pos(ja);           // cross-lane reads inside: every lane active here
                        out_p[R + r] = slot_to_pos(jb);
                        out_d[r] = out_d[R + r] = INFINITY;
                        if (i < cnt_a) {
                            const double dx = (double)cand_x[ja] - qax, dy = (double)cand_y[ja] - qay, dz = (double)cand_z[ja] - qaz;
                            const double d2 = (dx * dx + dy * dy) + dz * dz;
                            out_d[r] = (float)sqrt(d2);
                            if (!EPS || d2 < eps2) e_a = (min((unsigned)(d2 * scale), key_max - 1u) << SLOT_BITS) | (unsigned)i;
                        }
                        if (i < cnt_b) {
                            const double dx = (double)cand_x[jb] - qbx, dy = (double)cand_y[jb] - qby, dz = (double)cand_z[jb] - qbz;
                            const double d2 = (dx * dx + dy * dy) + dz * dz;
                            out_d[R + r] = (float)sqrt(d2);
                            if (!EPS || d2 < eps2) e_b = (min((unsigned)(d2 * scale), key_max - 1u) << SLOT_BITS) | (unsigned)i;
                        }
                        both.e[r] = e_a;
                        both.e[R + r] = e_b;
                    }
                    wave_lds_sync();
                    fast_sort_sets<R, 2, 2>(both, sort_dir);
                    n_flush += 2;
                }
                // ---- proof obligations per query (see the single-query path below) -----------------------------------------
                bool amb_a = false, amb_b = false, sparse_a = false, sparse_b = false;
                bool col_a = false, col_b = false;        // equal keys among the first k+2 entries
                {
                    unsigned tau_a, tau_b;         // element k of each list = the (k+1)-th nearest (padding if fewer exist)
                    {
                        const int sl = k >> 6, src = k & 63;
                        unsigned va = both.e[0], vb = both.e[R];
#pragma unroll
                        for (int r = 1; r < R; ++r)
                            if (sl == r) { va = both.e[r]; vb = both.e[R + r]; }
                        tau_a = (unsigned)__builtin_amdgcn_readlane((int)va, src);
                        tau_b = (unsigned)__builtin_amdgcn_readlane((int)vb, src);
                    }
                    const unsigned g_a = (unsigned)__builtin_amdgcn_readlane((int)my_gkey, qi);
                    const unsigned g_b = (unsigned)__builtin_amdgcn_readlane((int)my_gkey, qj);
                    const unsigned tk_a = tau_a >> SLOT_BITS, tk_b = tau_b >> SLOT_BITS;
                    const unsigned need_ka = min(tau_a == kPadElem ? 0xFFFFFFFFu : tk_a + 1u, eps_key);
                    const unsigned need_kb = min(tau_b == kPadElem ? 0xFFFFFFFFu : tk_b + 1u, eps_key);
                    sparse_a = need_ka > g_a;
                    sparse_b = need_kb > g_b;
                    amb_a |= need_ka > min(g_a, bkey_a);
                    amb_b |= need_kb > min(g_b, bkey_b);
                    amb_a |= tau_a != kPadElem && tk_a >= key_max - 1u;
                    amb_b |= tau_b != kPadElem && tk_b >= key_max - 1u;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        unsigned up_a = __shfl_down(both.e[r], 1), up_b = __shfl_down(both.e[R + r], 1);     // element i+1
                        if (r + 1 < R) {
                            const unsigned na = (unsigned)__builtin_amdgcn_readlane((int)both.e[r + 1 < R ? r + 1 : r], 0);
                            const unsigned nb = (unsigned)__builtin_amdgcn_readlane((int)both.e[R + (r + 1 < R ? r + 1 : r)], 0);
                            if (lane == 63) { up_a = na; up_b = nb; }
                        } else if (lane == 63) {
                            up_a = kPadElem;
                            up_b = kPadElem;
                        }
                        const int i = lane + 64 * r;
                        col_a |= i <= k && both.e[r] != kPadElem && up_a != kPadElem && ((both.e[r] ^ up_a) >> SLOT_BITS) == 0u;
                        col_b |= i <= k && both.e[R + r] != kPadElem && up_b != kPadElem && ((both.e[R + r] ^ up_b) >> SLOT_BITS) == 0u;
                    }
                }
                if (ok_a && __ballot(amb_a) != 0ull) { push_redo(row_a, sparse_a ? 1 : 3); ok_a = false; }
                if (ok_b && __ballot(amb_b) != 0ull) { push_redo(row_b, sparse_b ? 1 : 3); ok_b = false; }
                // equal keys: ordered here by the exact values (order_equal_keys), not by the exact sweep
                const auto pos_of_set = [&](unsigned at, int set) {
                    int p = __builtin_amdgcn_ds_bpermute((int)(at & 63u) << 2, out_p[set * R]);
#pragma unroll
                    for (int r2 = 1; r2 < R; ++r2) {
                        const int p2 = __builtin_amdgcn_ds_bpermute((int)(at & 63u) << 2, out_p[set * R + r2]);
                        if ((int)(at >> 6) == r2) p = p2;
                    }
                    return p;
                };
                // (the query is fetched from its lane again: keeping the six coordinates of the pair alive across the sort
                // for this rare branch would cost the common path scalar registers it does not have)
                const auto query_of = [&](int ql, double& x, double& y, double& z) {
                    if constexpr (Q64) {
                        x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_qd.x), ql), __builtin_amdgcn_readlane(__double2loint(my_qd.x), ql));
                        y = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_qd.y), ql), __builtin_amdgcn_readlane(__double2loint(my_qd.y), ql));
                        z = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_qd.z), ql), __builtin_amdgcn_readlane(__double2loint(my_qd.z), ql));
                    } else {
                        x = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.x), ql));
                        y = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.y), ql));
                        z = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.z), ql));
                    }
                };
                if (ok_a && __ballot(col_a) != 0ull) {
                    double ux, uy, uz;
                    query_of(qi, ux, uy, uz);
                    const bool done = order_equal_keys<R, SLOT_BITS>(&both.e[0], a.pts,
                        [&](unsigned at) {
                            const int j = (int)pend[at] & (CAP_POW2 - 1);
                            const double dx = (double)cand_x[j] - ux, dy = (double)cand_y[j] - uy, dz = (double)cand_z[j] - uz;
                            return (dx * dx + dy * dy) + dz * dz;
                        },
                        [&](unsigned at) { return pos_of_set(at, 0); });
                    if (!done) { push_redo(row_a, 3); ok_a = false; }
                }
                if (ok_b && __ballot(col_b) != 0ull) {
                    double ux, uy, uz;
                    query_of(qj, ux, uy, uz);
                    const bool done = order_equal_keys<R, SLOT_BITS>(&both.e[R], a.pts,
                        [&](unsigned at) {
                            const int j = (int)pend_b[at] & (CAP_POW2 - 1);
                            const double dx = (double)cand_x[j] - ux, dy = (double)cand_y[j] - uy, dz = (double)cand_z[j] - uz;
                            return (dx * dx + dy * dy) + dz * dz;
                        },
                        [&](unsigned at) { return pos_of_set(at, 1); });
                    if (!done) { push_redo(row_b, 3); ok_b = false; }
                }
                // ---- store: slot -> sorted position (cross-lane reads with every lane active), exact distance ----------------
#pragma unroll
                for (int set = 0; set < 2; ++set) {
                    const bool ok = set == 0 ? ok_a : ok_b;
                    const int row = set == 0 ? row_a : row_b;
                    int found = 0;
                    char* const prow = (char*)(a.nbr_pos + (int64_t)row * a.pitch);      // uniform: scalar base + lane offset
                    char* const drow = (char*)(a.nbr_dist + (int64_t)row * a.pitch);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int i = lane + 64 * r;
                        const unsigned e = both.e[set * R + r];
                        const bool real = e != kPadElem;
                        const unsigned at = e & ((1u << SLOT_BITS) - 1u);       // survivor index: lane at & 63, register at >> 6
                        float dist = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(at & 63u) << 2, __float_as_int(out_d[set * R])));
                        int pos = __builtin_amdgcn_ds_bpermute((int)(at & 63u) << 2, out_p[set * R]);
#pragma unroll
                        for (int r2 = 1; r2 < R; ++r2) {
                            const float d2nd = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(at & 63u) << 2, __float_as_int(out_d[set * R + r2])));
                            const int p2nd = __builtin_amdgcn_ds_bpermute((int)(at & 63u) << 2, out_p[set * R + r2]);
                            if ((int)(at >> 6) == r2) { dist = d2nd; pos = p2nd; }
                        }
                        if (ok && i >= 1 && i <= k) {
                            const unsigned off = (unsigned)(i - 1) * 4u;
                            *(int*)(prow + off) = real ? pos : -1;
                            *(float*)(drow + off) = real ? dist : INFINITY;
                            found += real;
                        }
                    }
                    if (ok) {
                        if (a.nbr_cnt) {
                            for (int o = 32; o > 0; o >>= 1) found += __shfl_xor(found, o);
                            if (lane == 0) a.nbr_cnt[row] = found;
                        }
                        if (a.row_done && lane == 0) a.row_done[row] = 1;
                    }
                }
            }
        };
        {
            using std::integral_constant;
            constexpr int PAIRS = NB / 2, LOW = PAIRS / 2;          // R = 1: 4 pairs, variants 2 | 3 | 4; R = 2: 6 pairs, 3 | 4 | 5 | 6
            const int nbp = (m + 127) >> 7;
#ifdef PCT_NO_NBP                                                   // tuning aid: one guarded body as before
            if (nbp >= 0) pair_loop(integral_constant<int, PAIRS>{}, integral_constant<bool, true>{});
            else
#endif
            if (nbp <= LOW) pair_loop(integral_constant<int, LOW>{}, integral_constant<bool, true>{});
            else if (nbp == LOW + 1) pair_loop(integral_constant<int, LOW + 1>{}, integral_constant<bool, false>{});
            else if (PAIRS > LOW + 2 && nbp == LOW + 2) pair_loop(integral_constant<int, (PAIRS > LOW + 2 ? LOW + 2 : PAIRS)>{}, integral_constant<bool, false>{});
            // (PAIRS > LOW + 3 -- the 1024-slot staging area of the tree items: the last variant also serves counts it is
            // not cut for, so it keeps the "is this batch in use" tests; unused batches are not initialised)
            else pair_loop(integral_constant<int, PAIRS>{}, integral_constant<bool, (PAIRS > LOW + 3)>{});
        }
    } else
    for (int qi = 0; qi < nq; ++qi) {
        const int row = row0 + qi;
        double qx, qy, qz;
        if (a.ptsd) {
            qx = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_qd.x), qi), __builtin_amdgcn_readlane(__double2loint(my_qd.x), qi));
            qy = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_qd.y), qi), __builtin_amdgcn_readlane(__double2loint(my_qd.y), qi));
            qz = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(my_qd.z), qi), __builtin_amdgcn_readlane(__double2loint(my_qd.z), qi));
        } else {
            qx = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.x), qi));
            qy = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.y), qi));
            qz = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.z), qi));
        }

        FastK<R> best;
        bool amb = false;                    // per-lane: something this kernel cannot prove exact
        bool col = false;                    // per-lane: equal keys among the first k+2 entries
        unsigned bkey = 0xFFFFFFFFu;         // exact keys of the candidates the pre-selection cut are >= bkey
        if constexpr (PRE) {
            const float fqx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.x), qi));
            const float fqy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.y), qi));
            const float fqz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.z), qi));
            // ---- float32 squared distances of ALL staged candidates, two batches per packed instruction ---------
            float appr[NB];
#pragma unroll
            for (int p2 = 0; p2 < NB / 2; ++p2) {
                appr[2 * p2] = INFINITY;
                appr[2 * p2 + 1] = INFINITY;
                if (p2 * 128 < m) {
                    const int sa = p2 * 128 + lane, sb = sa + 64;
                    const float2v vx = {cand_x[sa], cand_x[sb]}, vy = {cand_y[sa], cand_y[sb]}, vz = {cand_z[sa], cand_z[sb]};
                    const float2v dx = vx - fqx, dy = vy - fqy, dz = vz - fqz;
                    float2v d = dx * dx;
                    d = __builtin_elementwise_fma(dy, dy, d);
                    d = __builtin_elementwise_fma(dz, dz, d);
                    appr[2 * p2] = d.x;
                    appr[2 * p2 + 1] = d.y;
                    n_step += 2;
                }
            }
            // ---- threshold T with k+1 <= #(appr < T) <= LIST; never beyond the eps ball -----------------------------
            float T = EPS ? eps2a : INFINITY;
            int total = m;
            if constexpr (EPS) {
                total = 0;
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if ((b & ~1) * 64 < m) total += (int)__popcll(__builtin_amdgcn_ballot_w64(appr[b] < T));
            }
            int cnt = total;
            if (total > LIST) {
                float lo = 0.f, hi = T;                        // count(lo) < k+1 ; count(hi) > LIST
                float t = t_prev_f > 0.f ? t_prev_f : cell2f;  // first guess: one cell edge
                if (!(t < hi)) t = 0.5f * hi;
                const float target = 0.5f * (float)(k + 1 + LIST);
                bool found = false;
#pragma unroll 1
                for (int trial = 0; trial < 16; ++trial) {
                    int c = 0;
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        if ((b & ~1) * 64 < m) c += (int)__popcll(__builtin_amdgcn_ballot_w64(appr[b] < t));
                    if (c >= k + 1 && c <= LIST) { T = t; cnt = c; found = true; break; }
                    if (c < k + 1) lo = t; else hi = t;
                    float nt = c > 0 ? t * target * __builtin_amdgcn_rcpf((float)c) : 4.f * t;
                    if (!(nt > lo && nt < hi)) nt = hi < INFINITY ? 0.5f * (lo + hi) : 2.f * lo;
                    nt = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nt)));   // uniform by construction
                    if (!(nt > lo && nt < hi)) break;          // no float left between: a pile of equal distances
                    t = nt;
                }
                if (!found || !(T >= 1e-30f)) {                // no usable threshold: the exact sweep takes the query
                    note_redo(row, 3);
                    continue;
                }
                t_prev_f = T;
                bkey = (unsigned)fmin((double)T * (1.0 - 0x1p-20) * scale, 4294967294.0);
            }
            // ---- compact the slots of the survivors, then exact keys for them only ---------------------------------
            {
                int base = 0;
                wave_lds_sync();
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if ((b & ~1) * 64 < m) {
                        const bool pass = appr[b] < T;
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
                        if (pass) {
                            const int at = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                            pend[at] = (unsigned)(b * 64 + lane);                       // at < cnt <= LIST
                        }
                        base += (int)__popcll(mask);
                    }
                }
                wave_lds_sync();
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = lane + 64 * r;
                    unsigned e = kPadElem;
                    if (i < cnt) {
                        const int j = (int)pend[i];
                        const double dx = (double)cand_x[j] - qx, dy = (double)cand_y[j] - qy, dz = (double)cand_z[j] - qz;
                        const double d2 = (dx * dx + dy * dy) + dz * dz;
                        if (!EPS || d2 < eps2) e = (min((unsigned)(d2 * scale), key_max - 1u) << SLOT_BITS) | (unsigned)i;
                    }
                    best.e[r] = e;
                }
                wave_lds_sync();
                fast_sort_from<R, 2>(best, sort_dir);          // ascending
                ++n_flush;
            }
        } else {
            // ---- keys of ALL staged candidates, in registers (0xFFFFFFFF = not a candidate) ----------------------
            unsigned key[NB];
            int total = EPS ? 0 : m;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                key[b] = 0xFFFFFFFFu;
                if (b * 64 < m) {
                    const int slot = b * 64 + lane;
                    if (slot < m) {
                        const double dx = (double)cand_x[slot] - qx, dy = (double)cand_y[slot] - qy, dz = (double)cand_z[slot] - qz;
                        const double d2 = (dx * dx + dy * dy) + dz * dz;
                        if (!EPS || d2 < eps2) key[b] = min((unsigned)(d2 * scale), key_max - 1u);   // key_max itself: padding only
                    }
                    if constexpr (EPS) total += (int)__popcll(__builtin_amdgcn_ballot_w64(key[b] != 0xFFFFFFFFu));
                    ++n_step;
                }
            }

            // ---- threshold T with k+1 <= #(key < T) <= LIST: a few ballot-count trials.  Counts grow about linearly in
            // d^2 (= in the key) on a surface, so a secant step from the previous query's threshold usually lands at once.
            unsigned T = key_max + 1u;           // "everything"
            int cnt = total;
            if (total > LIST) {
                unsigned lo = 0u, hi = key_max + 1u;          // count(lo) < k+1 ; count(hi) > LIST
                unsigned t = t_prev ? t_prev : (unsigned)((double)(1u << KEY_BITS) / kKeyRange);   // first guess: one cell edge
                const float target = 0.5f * (float)(k + 1 + LIST);
                bool found = false;
#pragma unroll 1
                for (int trial = 0; trial < 16; ++trial) {
                    int c = 0;
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        if (b * 64 < m) c += (int)__popcll(__builtin_amdgcn_ballot_w64(key[b] < t));
                    if (c >= k + 1 && c <= LIST) { T = t; cnt = c; found = true; break; }
                    if (c < k + 1) lo = t; else hi = t;
                    if (hi - lo <= 1u) break;                  // a pile of equal keys straddles the window
                    const float guess = (float)t * target * __builtin_amdgcn_rcpf((float)(c > 0 ? c : 1));   // a guess: 1 ulp is plenty
                    unsigned nt = guess >= 4294967040.f ? hi : (unsigned)guess;
                    if (c == 0) nt = t * 4u > t ? t * 4u : hi;
                    if (nt <= lo || nt >= hi) nt = lo + (hi - lo) / 2u;
                    t = nt;
                }
                if (!found) {                                   // no usable threshold: the exact sweep takes the query
                    note_redo(row, 3);
                    continue;
                }
            }
            t_prev = T <= key_max ? T : t_prev;

            // ---- compact the selected candidates (all of them when there are <= LIST) and sort them ONCE ----------
            {
                int base = 0;
                wave_lds_sync();
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (b * 64 < m) {
                        const bool pass = key[b] < T;
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
                        if (pass) {
                            const int at = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                            pend[at] = (key[b] << SLOT_BITS) | (unsigned)(b * 64 + lane);      // at < cnt <= LIST
                        }
                        base += (int)__popcll(mask);
                    }
                }
                wave_lds_sync();
                const int have = cnt < LIST ? cnt : LIST;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = lane + 64 * r;
                    best.e[r] = i < have ? pend[i] : kPadElem;
                }
                wave_lds_sync();
                fast_sort_from<R, 2>(best, sort_dir);          // ascending
                ++n_flush;
            }
        }
        unsigned tau;                        // element k of the list = the (k+1)-th nearest (padding if fewer exist)
        {
            const int sl = k >> 6, src = k & 63;
            unsigned v = best.e[0];
#pragma unroll
            for (int r = 1; r < R; ++r)
                if (sl == r) v = best.e[r];
            tau = (unsigned)__builtin_amdgcn_readlane((int)v, src);
        }

        // ---- is every point closer than the (k+1)-th best inside the stencil?  (key rounded up; all in key units)
        bool sparse = false;
        {
            const unsigned gkey = (unsigned)__builtin_amdgcn_readlane((int)my_gkey, qi);
            const unsigned tkey = tau >> SLOT_BITS;
            const unsigned need = min(tau == kPadElem ? 0xFFFFFFFFu : tkey + 1u, eps_key);
            sparse = need > gkey;
            amb |= need > min(gkey, bkey);
            // a saturated key (a point clamped into a boundary cell from outside the grid box) says nothing
            // about the true distance
            amb |= tau != kPadElem && tkey >= key_max - 1u;
        }
        // ---- neighbours with equal keys inside the first k+2 entries: order not proven
        {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                unsigned up = __shfl_down(best.e[r], 1);                     // element i+1 for lanes 0..62
                if (r + 1 < R) {
                    const unsigned first_next = (unsigned)__builtin_amdgcn_readlane((int)best.e[r + 1 < R ? r + 1 : r], 0);
                    if (lane == 63) up = first_next;
                } else if (lane == 63) {
                    up = kPadElem;
                }
                const int i = lane + 64 * r;
                col |= i <= k && best.e[r] != kPadElem && up != kPadElem && ((best.e[r] ^ up) >> SLOT_BITS) == 0u;
            }
        }
        if (__ballot(amb) != 0ull) {
            note_redo(row, sparse ? 1 : 3);
            continue;
        }
        // sorted position of staged slot j = j + offset of its run (cross-lane reads: every lane active)
        const auto slot_pos = [&](int j) {
            unsigned code = (unsigned)__builtin_amdgcn_ds_bpermute((j & 63) << 2, (int)run_code[0]);
            if constexpr ((CAP / 64 + 7) / 8 > 1) {
                const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute((j & 63) << 2, (int)run_code[1]);
                code = (j >> 9) ? hi : code;
            }
            return j + offc[(code >> ((((unsigned)j >> 6) & 7u) << 2)) & 15u];
        };
        if (__ballot(col) != 0ull) {      // equal keys: ordered here by the exact values, not by the exact sweep
            const auto slot_of = [&](unsigned at) {
                int j = (int)at;
                if constexpr (PRE) j = (int)pend[j] & (CAP_POW2 - 1);
                return j;
            };
            const bool done = order_equal_keys<R, SLOT_BITS>(&best.e[0], a.pts,
                [&](unsigned at) {
                    const int j = slot_of(at);
                    const double dx = (double)cand_x[j] - qx, dy = (double)cand_y[j] - qy, dz = (double)cand_z[j] - qz;
                    return (dx * dx + dy * dy) + dz * dz;
                },
                [&](unsigned at) { return slot_pos(slot_of(at) & (CAP_POW2 - 1)); });
            if (!done) {
                note_redo(row, 3);
                continue;
            }
        }

        // ---- store: exact fp64 distance re-derived from the coordinates -------
        int found = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane + 64 * r;
            const unsigned e = best.e[r];
            const bool real = e != kPadElem;
            int j = (int)(e & ((1u << SLOT_BITS) - 1u));             // staged slot (anything for padding)
            if constexpr (PRE) j = (int)pend[j] & (CAP_POW2 - 1);        // ... via the survivors' list (still intact)
            // sorted position of slot j = j + offset of its run.  The cross-lane reads need every lane active:
            // they stay outside the divergent part.
            unsigned code = (unsigned)__builtin_amdgcn_ds_bpermute((j & 63) << 2, (int)run_code[0]);
            if constexpr ((CAP / 64 + 7) / 8 > 1) {
                const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute((j & 63) << 2, (int)run_code[1]);
                code = (j >> 9) ? hi : code;
            }
            const unsigned t = (code >> ((((unsigned)j >> 6) & 7u) << 2)) & 15u;       // index of the slot's run
            const int pos_real = j + offc[t];
            if (i >= 1 && i <= k) {
                int pos = -1;
                float dist = INFINITY;
                if (real) {
                    const double dx = (double)cand_x[j] - qx, dy = (double)cand_y[j] - qy, dz = (double)cand_z[j] - qz;
                    dist = (float)sqrt((dx * dx + dy * dy) + dz * dz);
                    pos = pos_real;
                }
                a.nbr_pos[(int64_t)row * a.pitch + (i - 1)] = pos;
                a.nbr_dist[(int64_t)row * a.pitch + (i - 1)] = dist;
                found += real;
            }
        }
        if (a.nbr_cnt) {
            for (int o = 32; o > 0; o >>= 1) found += __shfl_xor(found, o);
            if (lane == 0) a.nbr_cnt[row] = found;
        }
        if (a.row_done && lane == 0) a.row_done[row] = 1;
    }
    if (redo_mask) {
        const int cnt = (int)__popcll(redo_mask);
        if (a.row_done) {
            if ((redo_mask >> lane) & 1ull) {
                const int why = (int)((redo_why_lo >> lane) & 1ull) | ((int)((redo_why_hi >> lane) & 1ull) << 1);
                a.redo_m[row0 + lane] = (why << 29) | min(m, (1 << 29) - 1);
            }
        } else {
            int base = 0;
            if (lane == 0) base = atomicAdd(redo_count, cnt);
            base = __builtin_amdgcn_readfirstlane(base);
            if ((redo_mask >> lane) & 1ull) redo[base + (int)__popcll(redo_mask & ((1ull << lane) - 1ull))] = row0 + lane;
        }
        n_redo += (unsigned long long)cnt;
    }
    // statistics are opt-in: ~10^5 waves adding to the same words serialise at the memory side
    if (a.stats && lane == 0) {
        atomicAdd(&a.counters[2], n_flush);
        atomicAdd(&a.counters[3], n_step);
        if (n_redo) atomicAdd(&a.counters[4], n_redo);
    }
}

// ---------------------------------------------------------------------------
// k_knn_pair: the fast sweep for the case the headline workload is -- one list register (k + 1 <= 64), a float32 cloud,
// the uniform cell list, a plain sweep (no level pass) -- written for the SCALAR unit as much as for the vector units.
// Same algorithm, same proofs and bit-identical rows as k_knn_fast<1, EPS, true, true>; what differs:
//   * the kernel argument holds only what this kernel reads, the work item's cell coordinates come from two
//     multiplications (host-side magic numbers) instead of three integer divisions, per-item bases replace the
//     per-query 64-bit row arithmetic, and the opt-in statistics do not live in the loop: no scalar register spills
//     (k_knn_fast: 31 at the 106-register cap);
//   * the threshold search of the two queries of a pair runs in lanes 0 and 1 of the same vector instructions
//     (one secant step serves both) and leaves the loop with one ballot;
//   * the compaction of the survivors has no divergent region: a lane without a survivor writes to a spare slot
//     (k_knn_fast: s_and_saveexec / s_or exec and a branch per batch and query);
//   * exact keys, distances and positions of both queries are worked out for all 64 lanes in one basic block (the two
//     fp64 chains interleave; lanes beyond the survivor count are set to padding afterwards);
//   * the two sorting networks are ONE hand-scheduled assembly block (pct_sort_pair.inc, tools/gen_sort_asm.py): the
//     sets alternate instruction by instruction, so the wait states of every DPP read are the other set's work
//     (10 s_nop per pair instead of 40, 82 VALU instead of 99, both ds_bpermute of a flip in flight together);
//   * DIST = false (the fused curvature call, whose fit never reads distances) leaves out the correctly rounded
//     float32(sqrt(fp64)) and the second table: pct_get_neighbors derives the same bits from the positions on demand.
// Staged batches are used in pairs (128 slots); the body is compiled per number of pairs in use, without guards
// (1 | 2 | 3 | 4): slots of a staged pair beyond the stencil's population sit at +inf.
// ---------------------------------------------------------------------------
#ifndef PCT_SORT_INC
#define PCT_SORT_INC "pct_sort_pair.inc"
#endif
#include PCT_SORT_INC

__device__ __forceinline__ void sort_pair_asm(unsigned& ea, unsigned& eb, const SortLanes& c) {
    unsigned ta, tb;
    asm volatile(PCT_SORT_PAIR_ASM
                 : [ea] "+v"(ea), [eb] "+v"(eb), [ta] "=&v"(ta), [tb] "=&v"(tb)
                 : [sel0] "v"(c.sel[0]), [sel1] "v"(c.sel[1]), [sel2] "v"(c.sel[2]), [sel3] "v"(c.sel[3]), [sel4] "v"(c.sel[4]),
                   [sel5] "v"(c.sel[5]), [a31] "v"(c.a31), [a63] "v"(c.a63));
    ea = PCT_SORT_PAIR_RESULT_A;
    eb = PCT_SORT_PAIR_RESULT_B;
}

struct PairArgs {
    const float4* pts;        // cell-sorted candidate records {x, y, z, public index}
    const double4* ptsd;      // Q64: the native float64 coordinates in the same order (queries; the candidates stay float32, pct:74)
    const int* cell_start;
    const int* cell_own;
    const int* own_start;
    const int2* items;        // work items {cell, chunk of items_q queries}
    int* nbr_pos;
    float* nbr_dist;          // unused when DIST = false
    int* nbr_cnt;             // EPS only
    int* redo;
    int* redo_count;
    unsigned long long* counters;
    int n_items, items_q;
    int items_per_xcd;        // blocks are dealt to the 8 XCDs in turn: block b takes item (b % 8) * items_per_xcd + b / 8, so that
                              // the items one XCD's L2 serves at a time are neighbours in cell order (they share most of their stencils)
    int k, pitch;
    int stats;
    unsigned magic_x, magic_xy;      // cell -> (cx, cy, cz) by multiplication: q = (x * magic) >> shift, exact for x < 2^30
    int shift_x, shift_xy;
    double eps2;
    pct_grid g;
    // TREE (the hierarchical cell list, pct_tree.hip): an item is a run of queries of one segment; g = the finest level's grid
    const int4* tree_seg;     // per segment {level, cx, cy, cz}
    const int2* tree_runs;    // per segment 27 x {first position, points}, centre cell first
    int tree_bits;
};

constexpr int kPairCap = kStageCap;
static_assert((kPairCap & (kPairCap - 1)) == 0 && kPairCap % 128 == 0 && kPairCap <= 512, "staging capacity of k_knn_pair");
static_assert(PCT_TREE_CAP % 128 == 0 && PCT_TREE_CAP <= 1024, "staging capacity of k_knn_pair on the hierarchical cell list");
template <int CAP>
struct PairLdsT {
    float cx[CAP], cy[CAP], cz[CAP];                     // staged stencil, 12 B per candidate
    unsigned pend[64 + 4];                               // staged slots of the survivors of query a; [64]: the spare slot
    unsigned short pend_b[64 + 8];                       // ... of query b
    int offc[16];                                        // sorted position - flat slot, per non-empty run
};
using PairLds = PairLdsT<kPairCap>;

#ifndef PCT_PAIR_WAVES
#define PCT_PAIR_WAVES 1
#endif
// waves (= work items) per block; they share nothing but the launch.  One: a finished wave's slot and LDS go to the
// next block at once (items differ in queries and in staged batches: with four waves per block the fastest three
// waited for the slowest, 4.6 of 6 wave slots per SIMD filled; 0.392 -> 0.379 ms)
constexpr int kPairWaves = PCT_PAIR_WAVES;

// Q64: a float64 cloud -- the candidates are the float32-rounded points (the reference's tree data, pct:74), a query is the
// native float64 point (pct:83): the float32 pre-selection measures from the query ROUNDED to float32, a point
// eq = |q64 - q32| away from the true one, and every bound taken from it moves by eq (see k_knn_fast); exact keys and
// distances use the float64 query.
// TREE: the work items of the hierarchical cell list (round 3; k_knn_fast<1, .., TREE> until then) -- an item is a run of
// queries of one octree segment, its stencil the 27 ranges of the Morton-ordered cloud the build recorded, its grid the
// segment's level; 768 staged slots (what the build refines segments for), slot ids of 10 + 4 bits.
template <bool EPS, bool DIST, bool Q64 = false, bool TREE = false>
__global__ __launch_bounds__(64 * kPairWaves, (TREE ? 4 : Q64 ? 5 : 6)) void k_knn_pair(PairArgs a) {
    constexpr int CAP = TREE ? PCT_TREE_CAP : kPairCap, LIST = 64, SLOT_BITS = 6, KEY_BITS = 32 - SLOT_BITS;
    constexpr int SB = TREE ? 10 : 9;                    // bits of a staged slot inside a slot id (the run index sits above)
    constexpr unsigned kIdMask = (1u << (SB + 4)) - 1u;
    __shared__ PairLdsT<CAP> s_lds[kPairWaves];
    const int w = kPairWaves == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int blk = (int)blockIdx.x * kPairWaves + w;
    const int item = a.items_per_xcd ? (blk & 7) * a.items_per_xcd + (blk >> 3) : blk;
    if (item >= a.n_items || (a.items_per_xcd && (blk >> 3) >= a.items_per_xcd)) return;
    PairLdsT<CAP>& L = s_lds[w];
    const SortLanes sort_dir = make_sort_lanes();
    const int* __restrict__ cs = a.cell_start;

    const int2 it2 = a.items[item];
    constexpr int NRUNS = TREE ? 27 : 9;                          // ranges of the cloud the stencil is staged from
    int cx, cy, cz, qs, nq, row0;
    int run_s = 0, run_len = 0;
    pct_grid g_lvl;                                               // TREE: the grid of the item's level
    if constexpr (TREE) {
        // item = {first query (Morton position = table row) | (queries - 1) << 26, segment}
        const int seg = __builtin_amdgcn_readfirstlane(it2.y);
        if (seg < 0) return;                                      // an item of a segment that was split (k_tree_refine)
        const unsigned packed = (unsigned)__builtin_amdgcn_readfirstlane(it2.x);
        qs = (int)(packed & 0x3ffffffu);
        nq = (int)(packed >> 26) + 1;
        row0 = qs;
        const int4 hd = a.tree_seg[seg];
        const int level = __builtin_amdgcn_readfirstlane(hd.x);
        cx = __builtin_amdgcn_readfirstlane(hd.y);
        cy = __builtin_amdgcn_readfirstlane(hd.z);
        cz = __builtin_amdgcn_readfirstlane(hd.w);
        // edges scale by exact powers of two, so (x - o) * inv_cell - cx lies in [0, 1) for every point of the cell
        g_lvl = a.g;
        g_lvl.cell = __builtin_ldexp(a.g.cell, level);
        g_lvl.inv_cell = __builtin_ldexp(a.g.inv_cell, -level);
        g_lvl.nx = g_lvl.ny = g_lvl.nz = 1 << (a.tree_bits - level);
        if (lane < 27) {
            const int2 r = a.tree_runs[(int64_t)seg * 27 + lane];
            run_s = r.x;
            run_len = r.y;
        }
    } else {
        const int cell = __builtin_amdgcn_readfirstlane(it2.x);
        const int chunk = __builtin_amdgcn_readfirstlane(it2.y);
        const int nx = a.g.nx, ny = a.g.ny, nz = a.g.nz;
        cz = (int)(((unsigned long long)(unsigned)cell * a.magic_xy) >> a.shift_xy);
        const int rem = cell - cz * (nx * ny);
        cy = (int)(((unsigned long long)(unsigned)rem * a.magic_x) >> a.shift_x);
        cx = rem - cy * nx;
        const int c0 = cs[cell];
        qs = c0 + chunk * a.items_q;                              // owned points sit first in the cell
        nq = min(c0 + a.cell_own[cell], qs + a.items_q) - qs;
        row0 = a.own_start[cell] + chunk * a.items_q;             // neighbour-table row of query qs

        // ---- bounds of the 9 x-runs of the 27-cell stencil, fetched in parallel by lanes 0..8 (centre row first)
        if (lane < 9) {
            const int z = cz + kRowOrder[lane][0], y = cy + kRowOrder[lane][1];
            if (z >= 0 && z < nz && y >= 0 && y < ny) {
                const int row = (z * ny + y) * nx;
                run_s = cs[row + max(cx - 1, 0)];
                run_len = cs[row + min(cx + 1, nx - 1) + 1] - run_s;
            }
        }
    }
    const pct_grid& G = TREE ? g_lvl : a.g;
    // the item's own queries (<= items_q <= 64 consecutive sorted positions), one per lane
    float4 my_q = make_float4(0.f, 0.f, 0.f, 0.f);
    double my_qx = 0., my_qy = 0., my_qz = 0.;
    float my_eq = 0.f;       // Q64: distance between the float64 query and its float32 rounding, rounded up
    if (lane < nq) {
        my_q = a.pts[qs + lane];
        if constexpr (Q64) {
            const double4 qd = a.ptsd[qs + lane];
            my_qx = qd.x; my_qy = qd.y; my_qz = qd.z;
        }
    }
    if constexpr (Q64) {
        const double ex = my_qx - (double)my_q.x, ey = my_qy - (double)my_q.y, ez = my_qz - (double)my_q.z;
        my_eq = (float)sqrt((ex * ex + ey * ey) + ez * ez) * (1.0f + 0x1p-22f);
        if (!(my_eq >= 0.f)) my_eq = INFINITY;
    }
    // exclusive prefix of the run lengths over lanes 0..8 = first flat slot of every run; m = staged candidates
    int my_pre = 0, m = 0;
    {
        int acc = 0;
#pragma unroll
        for (int t = 0; t < NRUNS; ++t) {
            my_pre = lane == t ? acc : my_pre;
            acc += __builtin_amdgcn_readlane(run_len, t);
        }
        m = acc;
    }
    // (tree items: four bits of run index per slot id, 16 non-empty ranges -- a surface meets about ten of its 27 stencil
    // cells; more is a volume, where these items pay as little as uniform cells do)
    const bool crowded = TREE && __popcll(__builtin_amdgcn_ballot_w64(lane < NRUNS && run_len > 0)) > 16;
    if (m > CAP || crowded) {
        // stencil does not fit the staging area (dense cluster): the exact sweep takes the whole item
        int base = 0;
        if (lane == 0) base = atomicAdd(a.redo_count, nq);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < nq) a.redo[base + lane] = row0 + lane;
        if (a.stats && lane == 0) {
            atomicAdd(&a.counters[1], 1ull);
            atomicAdd(&a.counters[4], (unsigned long long)nq);
        }
        return;
    }

    // ---- copy the runs as one flat range (see k_knn_fast): run starts as a bit string in the list area; the run index
    // u of this lane's slot of batch b rides in the slot id itself (slotx[b] = slot | u << 9: what the compaction
    // writes into the survivors' lists), offc[u] = sorted position - flat slot of run u
    unsigned slotx[CAP / 64];
    {
        unsigned* bits = L.pend;
        if (lane < CAP / 32) bits[lane] = 0u;
        wave_lds_sync();
        const bool nonempty = lane < NRUNS && run_len > 0;
        const unsigned long long ne = __builtin_amdgcn_ballot_w64(nonempty);
        if (nonempty) {
            atomicOr(&bits[my_pre >> 5], 1u << (my_pre & 31));
            L.offc[__builtin_amdgcn_mbcnt_lo((unsigned)ne, 0)] = run_s - my_pre;
        }
        wave_lds_sync();
        float4 tmp[CAP / 64];
        int ubase = -1;
#pragma unroll
        for (int b = 0; b < CAP / 64; ++b) {
            tmp[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            slotx[b] = (unsigned)(b * 64 + lane);
            if (b * 64 < m) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)bits[2 * b]);
                const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)bits[2 * b + 1]);
                const unsigned long long B = ((unsigned long long)hi << 32) | lo;
                const unsigned long long S = B >> 1;               // starts <= lane  =  starts of (B >> 1) below lane, + bit 0
                const int s0 = ubase + (int)(lo & 1u);
                const int u = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(S >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)S, (unsigned)s0));
                ubase += (int)__popcll(B);
                const int j = b * 64 + lane;
                slotx[b] |= (unsigned)u << SB;
                if (j < m) tmp[b] = a.pts[j + L.offc[u]];
            }
        }
        wave_lds_sync();                      // the bit string is dead: the list area goes back to the queries
#pragma unroll
        for (int b = 0; b < CAP / 64; ++b) {
            const int j = b * 64 + lane;
            if (j < m) {
                L.cx[j] = tmp[b].x; L.cy[j] = tmp[b].y; L.cz[j] = tmp[b].z;
            } else if ((b & ~1) * 64 < m) {
                L.cx[j] = INFINITY; L.cy[j] = 0.f; L.cz[j] = 0.f;     // unused slot of a staged pair: passes no threshold
            }
        }
    }
    wave_lds_sync();

    const int k = a.k;
    const double eps2 = EPS ? a.eps2 : (double)INFINITY;
    constexpr double kKeyRange = 2.3;                     // what the 27-cell cube can vouch for, in cell^2 (k_knn_fast)
    const double edge = G.cell;
    const double scale = (double)(1u << KEY_BITS) / (kKeyRange * edge * edge);
    constexpr unsigned key_max = (1u << KEY_BITS) - 1u;
    unsigned my_gkey;                                     // per query (lane l = query l): the largest key the stencil vouches for
    {
        const double lqx = Q64 ? my_qx : (double)my_q.x, lqy = Q64 ? my_qy : (double)my_q.y, lqz = Q64 ? my_qz : (double)my_q.z;
        const double gx = (lqx - G.ox) * G.inv_cell - cx;
        const double gy = (lqy - G.oy) * G.inv_cell - cy;
        const double gz = (lqz - G.oz) * G.inv_cell - cz;
        const double g2 = fmin(guaranteed_r2(G, cx, cy, cz, gx, gy, gz, 1), limit_r2(G, cx, cy, cz, gx, gy, gz));
        my_gkey = g2 == INFINITY ? 0xFFFFFFFFu : (unsigned)fmin(g2 * scale, 4294967294.0);
    }
    const unsigned eps_key = EPS && eps2 < 1e300 ? (unsigned)fmin(ceil(eps2 * scale), 4294967295.0) : 0xFFFFFFFFu;
    const float cell2f = (float)(edge * edge);
    const float eps2a = EPS ? (float)fmin(eps2 * (1.0 + 0x1p-18), 3.0e38) : INFINITY;
    const double eps1 = EPS ? sqrt(eps2) * (1.0 + 0x1p-50) : 0.0;       // eps itself, rounded up (Q64)
    float t_prev_f = 0.f;                                 // threshold of the previous query of this item (0 = none yet)
    unsigned long long redo_mask = 0ull;                  // queries of this item the exact sweep has to take

    // table rows of this item: one 64-bit base per item, 32-bit offsets per query and lane (list entry i -> column i - 1)
    char* const pos_item = (char*)(a.nbr_pos + (int64_t)row0 * a.pitch);
    char* const dist_item = DIST ? (char*)(a.nbr_dist + (int64_t)row0 * a.pitch) : nullptr;
    const unsigned lane_off = (unsigned)(lane - 1) * 4u;
    const unsigned pitch4 = (unsigned)a.pitch * 4u;
    const bool col_lane = lane >= 1 && lane <= k;         // lanes whose list entry is a table column
    const unsigned long long first_k1 = (2ull << k) - 1ull;          // lanes 0 .. k: the entries whose order matters

    static_assert(CAP <= (1 << SB) && SB + 4 <= 16, "slot ids: SB bits of slot, 4 bits of run index, 16-bit survivor list of query b");
    const auto slot_of = [](unsigned id) { return TREE ? min((int)(id & ((1u << SB) - 1u)), CAP - 1) : (int)(id & (unsigned)(CAP - 1)); };

    const auto pair_loop = [&](auto NBP_) {
        constexpr int NBP = decltype(NBP_)::value, NBU = 2 * NBP;
        for (int qi = 0; qi < nq; qi += 2) {
            const bool live_b = qi + 1 < nq;             // an odd tail runs its last query twice, the copy is discarded
            const int qj = live_b ? qi + 1 : qi;
            const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.x), qi));
            const float ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.y), qi));
            const float az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.z), qi));
            const float bx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.x), qj));
            const float by = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.y), qj));
            const float bz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.z), qj));
            // the queries the exact keys measure from: the float32 record widened, or (Q64) the native coordinates
            const auto rl64 = [&](double v, int l) {
                return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
            };
            const double qax = Q64 ? rl64(my_qx, qi) : (double)ax, qay = Q64 ? rl64(my_qy, qi) : (double)ay, qaz = Q64 ? rl64(my_qz, qi) : (double)az;
            const double qbx = Q64 ? rl64(my_qx, qj) : (double)bx, qby = Q64 ? rl64(my_qy, qj) : (double)by, qbz = Q64 ? rl64(my_qz, qj) : (double)bz;
            // ---- float32 squared distances of ALL staged candidates to both queries (one set of LDS reads) --------
            float ap_a[NBU], ap_b[NBU];
#pragma unroll
            for (int p2 = 0; p2 < NBP; ++p2) {
                const int sa = p2 * 128 + lane, sb = sa + 64;
                const float2v vx = {L.cx[sa], L.cx[sb]}, vy = {L.cy[sa], L.cy[sb]}, vz = {L.cz[sa], L.cz[sb]};
                {
                    const float2v dx = vx - ax, dy = vy - ay, dz = vz - az;
                    float2v d = dx * dx;
                    d = __builtin_elementwise_fma(dy, dy, d);
                    d = __builtin_elementwise_fma(dz, dz, d);
                    ap_a[2 * p2] = d.x;
                    ap_a[2 * p2 + 1] = d.y;
                }
                {
                    const float2v dx = vx - bx, dy = vy - by, dz = vz - bz;
                    float2v d = dx * dx;
                    d = __builtin_elementwise_fma(dy, dy, d);
                    d = __builtin_elementwise_fma(dz, dz, d);
                    ap_b[2 * p2] = d.x;
                    ap_b[2 * p2 + 1] = d.y;
                }
            }
            // ---- thresholds: k+1 <= #(d < T) <= LIST for each query, never beyond the eps ball.  Lane 0 searches for
            // query a, lane 1 for query b: the counts are wave-wide ballots, the secant arithmetic is per lane.
            // +inf without eps; Q64: exact d < eps  =>  d' < eps + eq, per query (lane 0: a, lanes >= 1: b)
            float T_init = eps2a;
            float Ti_a = eps2a, Ti_b = eps2a;
            const float eq_a = Q64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_eq), qi)) : 0.f;
            const float eq_b = Q64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_eq), qj)) : 0.f;
            const float v_eq = lane == 0 ? eq_a : eq_b;
            if constexpr (EPS && Q64) {
                const double ee = eps1 + (double)v_eq;
                T_init = (float)fmin(ee * ee * (1.0 + 0x1p-18), 3.0e38);
                Ti_a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(T_init), 0));
                Ti_b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(T_init), 1));
            }
            int tot_a = m, tot_b = m;
            if constexpr (EPS) {
                tot_a = tot_b = 0;
#pragma unroll
                for (int b = 0; b < NBU; ++b) {
                    tot_a += (int)__popcll(__builtin_amdgcn_ballot_w64(ap_a[b] < Ti_a));
                    tot_b += (int)__popcll(__builtin_amdgcn_ballot_w64(ap_b[b] < Ti_b));
                }
            }
            const bool need_a = tot_a > LIST, need_b = live_b && tot_b > LIST;
            float T_a = Ti_a, T_b = Ti_b;
            int cnt_a = tot_a, cnt_b = tot_b;
            bool ok_a = true, ok_b = live_b;
            unsigned bkey_a = 0xFFFFFFFFu, bkey_b = 0xFFFFFFFFu;     // exact keys of the candidates the pre-selection cut are >= bkey
#ifdef PCT_ABL_NO_TRIAL
            T_a = T_b = 0.33f * cell2f; cnt_a = cnt_b = 57;
            if (false)
#endif
            if (need_a || need_b) {
                const float target = 0.5f * (float)(k + 1 + LIST);
                float t0 = t_prev_f > 0.f ? t_prev_f : cell2f;
                if (!(t0 < T_init)) t0 = 0.5f * T_init;
                const bool mine = lane == 0 ? need_a : need_b;         // (lanes >= 2 mirror lane 1; nobody reads them)
                float v_t = t0, v_lo = 0.f, v_hi = T_init, v_T = T_init;
                int v_cnt = lane == 0 ? tot_a : tot_b;
                bool go = mine, found = !mine;
#pragma unroll 1
                for (int trial = 0; trial < 16; ++trial) {
                    const float ta = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v_t), 0));
                    const float tb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v_t), 1));
                    int c_a = 0, c_b = 0;
#pragma unroll
                    for (int b = 0; b < NBU; ++b) {
                        c_a += (int)__popcll(__builtin_amdgcn_ballot_w64(ap_a[b] < ta));
                        c_b += (int)__popcll(__builtin_amdgcn_ballot_w64(ap_b[b] < tb));
                    }
                    const int c = lane == 0 ? c_a : c_b;
                    const bool in = go && (unsigned)(c - (k + 1)) <= (unsigned)(LIST - (k + 1));
                    v_T = in ? v_t : v_T;
                    v_cnt = in ? c : v_cnt;
                    found = found || in;
                    go = go && !in;
                    if ((__builtin_amdgcn_ballot_w64(go) & 3ull) == 0ull) break;      // both thresholds found: no secant step
                    // secant step for the lanes still searching (count ~ linear in d^2 on a surface); c = 0 gives +inf,
                    // which the interval test below turns into a doubling / a bisection
                    const bool below = c < k + 1;
                    v_lo = go && below ? v_t : v_lo;
                    v_hi = go && !below ? v_t : v_hi;
                    float nt = v_t * target * __builtin_amdgcn_rcpf((float)c);
                    if (!(nt > v_lo && nt < v_hi)) nt = v_hi < INFINITY ? 0.5f * (v_lo + v_hi) : 2.f * v_lo;
                    go = go && nt > v_lo && nt < v_hi;         // no float left between: a pile of equal distances
                    v_t = go ? nt : v_t;
                    if ((__builtin_amdgcn_ballot_w64(go) & 3ull) == 0ull) break;
                }
                const unsigned fm = (unsigned)__builtin_amdgcn_ballot_w64(found);
                T_a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v_T), 0));
                T_b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v_T), 1));
                cnt_a = __builtin_amdgcn_readlane(v_cnt, 0);
                cnt_b = __builtin_amdgcn_readlane(v_cnt, 1);
                // smallest exact key a candidate cut by the float32 threshold T can have: its float32 d^2 >= T means the
                // exact d^2 >= T (1 - 2^-20) (arithmetic error of the packed evaluation)
                double lo2 = (double)v_T * (1.0 - 0x1p-20);
                if constexpr (Q64) {
                    // (sqrt(L) - eq)^2 >= L - 2 eq sqrt(L); an upper bound of the root is enough: float32 root, rounded up
                    const double root_up = (double)__builtin_sqrtf(v_T) * (1.0 + 0x1p-21);
                    lo2 = fmax(lo2 - 2.0 * (double)v_eq * root_up, 0.0);
                }
                const unsigned v_bkey = (unsigned)fmin(lo2 * scale, 4294967294.0);
                ok_a = (fm & 1u) != 0u && (!need_a || T_a >= 1e-30f);
                ok_b = live_b && (fm & 2u) != 0u && (!need_b || T_b >= 1e-30f);
                if (need_a && ok_a) { t_prev_f = T_a; bkey_a = (unsigned)__builtin_amdgcn_readlane((int)v_bkey, 0); }
                if (need_b && ok_b) { t_prev_f = T_b; bkey_b = (unsigned)__builtin_amdgcn_readlane((int)v_bkey, 1); }
                if (!ok_a) { redo_mask |= 1ull << qi; T_a = 0.f; cnt_a = 0; }        // nothing passes, nothing is stored
                if (!ok_b) { if (live_b) redo_mask |= 1ull << qj; T_b = 0.f; cnt_b = 0; }
                if (!ok_a && !ok_b) continue;
            }
            if (!live_b) { T_b = 0.f; cnt_b = 0; }
            // ---- compact the slots of the survivors of both queries.  Per batch and query: one compare (the pass mask
            // goes to a scalar pair), two v_mbcnt for the rank among the survivors, one v_lshl_add for the LDS address --
            // four vector instructions -- and the write under exec = mask; the running list address and exec are
            // scalar work (the scalar unit has the room: the kernel is bound by vector issue, 4 cycles per instruction).
            // Hand-placed: on gfx940-class parts a VALU read of an SGPR needs two wait states after the VALU write of
            // it; the two queries' instructions fill each other's.
            {
                unsigned wr_a = (unsigned)(uintptr_t)&L.pend[0], wr_b = (unsigned)(uintptr_t)&L.pend_b[0];
                const unsigned long long all = __builtin_amdgcn_read_exec();
                wave_lds_sync();
#ifndef PCT_ABL_NO_COMPACT
#pragma unroll
                for (int b = 0; b < NBU; ++b) {
                    unsigned ra, rb, ca, cb;
                    const unsigned slot = slotx[b];
                    asm volatile(
                        "v_cmp_gt_f32 vcc, %[ta], %[apa]\n"
                        "v_cmp_gt_f32 s[96:97], %[tb], %[apb]\n"
                        "s_bcnt1_i32_b64 %[ca], vcc\n"
                        "v_mbcnt_lo_u32_b32 %[ra], vcc_lo, 0\n"
                        "s_bcnt1_i32_b64 %[cb], s[96:97]\n"
                        "v_mbcnt_lo_u32_b32 %[rb], s96, 0\n"
                        "v_mbcnt_hi_u32_b32 %[ra], vcc_hi, %[ra]\n"
                        "v_mbcnt_hi_u32_b32 %[rb], s97, %[rb]\n"
                        "v_lshl_add_u32 %[ra], %[ra], 2, %[wra]\n"
                        "v_lshl_add_u32 %[rb], %[rb], 1, %[wrb]\n"
                        "s_mov_b64 exec, vcc\n"
                        "ds_write_b32 %[ra], %[slot]\n"
                        "s_mov_b64 exec, s[96:97]\n"
                        "ds_write_b16 %[rb], %[slot]\n"
                        "s_mov_b64 exec, %[all]\n"
                        "s_lshl2_add_u32 %[wra], %[ca], %[wra]\n"
                        "s_lshl1_add_u32 %[wrb], %[cb], %[wrb]\n"
                        : [ra] "=&v"(ra), [rb] "=&v"(rb), [ca] "=&s"(ca), [cb] "=&s"(cb), [wra] "+s"(wr_a), [wrb] "+s"(wr_b)
                        : [ta] "v"(T_a), [tb] "v"(T_b), [apa] "v"(ap_a[b]), [apb] "v"(ap_b[b]), [slot] "v"(slot), [all] "s"(all)
                        : "vcc", "scc", "s96", "s97", "memory");
                }
#endif
                wave_lds_sync();
            }
            // ---- exact keys for the survivors only.  Survivor `lane` of each query: staged slot -> coordinates -> fp64
            // ((dx^2 + dy^2) + dz^2) (no FMA: SciPy's value) -> key, distance, sorted position; worked out by every lane
            // (a stale list entry is masked into the staging area and gives a garbage value nobody uses).
            const unsigned sxa = L.pend[lane] & kIdMask, sxb = (unsigned)L.pend_b[lane] & kIdMask;      // slot | run << SB
            const int ja = slot_of(sxa), jb = slot_of(sxb);
            const int out_pa = ja + L.offc[sxa >> SB], out_pb = jb + L.offc[sxb >> SB];                 // sorted positions
            float out_da = 0.f, out_db = 0.f;
            unsigned e_a, e_b;
#ifdef PCT_ABL_NO_KEYS
            e_a = ((unsigned)ja << 6) | (unsigned)lane; e_b = ((unsigned)jb << 6) | (unsigned)lane;
            if (false)
#endif
            {
                const double dxa = (double)L.cx[ja] - qax, dya = (double)L.cy[ja] - qay, dza = (double)L.cz[ja] - qaz;
                const double dxb = (double)L.cx[jb] - qbx, dyb = (double)L.cy[jb] - qby, dzb = (double)L.cz[jb] - qbz;
                const double d2a = (dxa * dxa + dya * dya) + dza * dza;
                const double d2b = (dxb * dxb + dyb * dyb) + dzb * dzb;
                if constexpr (DIST) {
                    out_da = (float)sqrt(d2a);
                    out_db = (float)sqrt(d2b);
                }
                const unsigned ka = (min((unsigned)(d2a * scale), key_max - 1u) << SLOT_BITS) | (unsigned)lane;
                const unsigned kb = (min((unsigned)(d2b * scale), key_max - 1u) << SLOT_BITS) | (unsigned)lane;
                e_a = lane < cnt_a && (!EPS || d2a < eps2) ? ka : kPadElem;
                e_b = lane < cnt_b && (!EPS || d2b < eps2) ? kb : kPadElem;
            }
            wave_lds_sync();
#ifndef PCT_ABL_NO_SORT
            sort_pair_asm(e_a, e_b, sort_dir);
#endif
            // ---- proof obligations per query (all in key units, see k_knn_fast) ------------------------------------
            const unsigned tau_a = (unsigned)__builtin_amdgcn_readlane((int)e_a, k);      // the (k+1)-th nearest (padding if fewer exist)
            const unsigned tau_b = (unsigned)__builtin_amdgcn_readlane((int)e_b, k);
            const unsigned g_a = (unsigned)__builtin_amdgcn_readlane((int)my_gkey, qi);
            const unsigned g_b = (unsigned)__builtin_amdgcn_readlane((int)my_gkey, qj);
            const unsigned tk_a = tau_a >> SLOT_BITS, tk_b = tau_b >> SLOT_BITS;
            const unsigned need_ka = min(tau_a == kPadElem ? 0xFFFFFFFFu : tk_a + 1u, eps_key);
            const unsigned need_kb = min(tau_b == kPadElem ? 0xFFFFFFFFu : tk_b + 1u, eps_key);
            const bool amb_a = need_ka > min(g_a, bkey_a) || (tau_a != kPadElem && tk_a >= key_max - 1u);
            const bool amb_b = need_kb > min(g_b, bkey_b) || (tau_b != kPadElem && tk_b >= key_max - 1u);
#ifndef PCT_ABL_NO_CHECK
            if (ok_a && amb_a) { redo_mask |= 1ull << qi; ok_a = false; }
            if (ok_b && amb_b) { redo_mask |= 1ull << qj; ok_b = false; }
#endif
            // equal keys among the first k+2 entries: ordered here by the exact values (order_equal_keys).  Detection:
            // element i ^ element i+1 (one v_xor with a wave_shl:1 operand per set) below 64 <=> same key
            {
                unsigned xa, xb;
                asm("v_xor_b32_dpp %0, %2, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                    "v_xor_b32_dpp %1, %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                    "s_nop 0"
                    : "=&v"(xa), "=&v"(xb) : "v"(e_a), "v"(e_b));
#ifdef PCT_ABL_NO_CHECK
                const unsigned long long cm_any = 0ull;
#else
                const unsigned long long cm_any = __builtin_amdgcn_ballot_w64(min(xa, xb) < 64u) & first_k1;
#endif
                unsigned long long cm_a = 0ull, cm_b = 0ull;
                if (__builtin_expect(cm_any != 0ull, 0)) {
                    cm_a = __builtin_amdgcn_ballot_w64(xa < 64u && e_a != kPadElem) & first_k1;
                    cm_b = __builtin_amdgcn_ballot_w64(xb < 64u && e_b != kPadElem) & first_k1;
                }
                if (__builtin_expect((cm_a | cm_b) != 0ull, 0)) {
                    if (ok_a && cm_a != 0ull) {
                        const double ux = qax, uy = qay, uz = qaz;
                        const bool done = order_equal_keys<1, SLOT_BITS>(&e_a, a.pts,
                            [&](unsigned at) {
                                const int j = slot_of(L.pend[at]);
                                const double dx = (double)L.cx[j] - ux, dy = (double)L.cy[j] - uy, dz = (double)L.cz[j] - uz;
                                return (dx * dx + dy * dy) + dz * dz;
                            },
                            [&](unsigned at) { return __builtin_amdgcn_ds_bpermute((int)(at & 63u) << 2, out_pa); });
                        if (!done) { redo_mask |= 1ull << qi; ok_a = false; }
                    }
                    if (ok_b && cm_b != 0ull) {
                        const double ux = qbx, uy = qby, uz = qbz;
                        const bool done = order_equal_keys<1, SLOT_BITS>(&e_b, a.pts,
                            [&](unsigned at) {
                                const int j = slot_of((unsigned)L.pend_b[at]);
                                const double dx = (double)L.cx[j] - ux, dy = (double)L.cy[j] - uy, dz = (double)L.cz[j] - uz;
                                return (dx * dx + dy * dy) + dz * dz;
                            },
                            [&](unsigned at) { return __builtin_amdgcn_ds_bpermute((int)(at & 63u) << 2, out_pb); });
                        if (!done) { redo_mask |= 1ull << qj; ok_b = false; }
                    }
                }
            }
            // ---- store: the lane that holds list entry i fetches position (and distance) of survivor e & 63 ----------
            {
                const unsigned off_a = lane_off + (unsigned)qi * pitch4, off_b = lane_off + (unsigned)qj * pitch4;
                const bool real_a = e_a != kPadElem, real_b = e_b != kPadElem;
                const int at_a = (int)(e_a << 2), at_b = (int)(e_b << 2);        // ds_bpermute reads lane (address >> 2) & 63: the survivor index
                const int pos_a = __builtin_amdgcn_ds_bpermute(at_a, out_pa), pos_b = __builtin_amdgcn_ds_bpermute(at_b, out_pb);
                float dist_a = 0.f, dist_b = 0.f;
                if constexpr (DIST) {
                    dist_a = __int_as_float(__builtin_amdgcn_ds_bpermute(at_a, __float_as_int(out_da)));
                    dist_b = __int_as_float(__builtin_amdgcn_ds_bpermute(at_b, __float_as_int(out_db)));
                }
#ifdef PCT_ABL_NO_STORE
                if (pos_a == 0x7fffffff && pos_b == 0x7ffffff1)
#endif
                if (ok_a && col_lane) {
                    *(int*)(pos_item + off_a) = real_a ? pos_a : -1;
                    if constexpr (DIST) *(float*)(dist_item + off_a) = real_a ? dist_a : INFINITY;
                }
#ifdef PCT_ABL_NO_STORE
                if (pos_a == 0x7fffffff && pos_b == 0x7ffffff1)
#endif
                if (ok_b && col_lane) {
                    *(int*)(pos_item + off_b) = real_b ? pos_b : -1;
                    if constexpr (DIST) *(float*)(dist_item + off_b) = real_b ? dist_b : INFINITY;
                }
                if constexpr (EPS) {
                    const int f_a = (int)__popcll(__builtin_amdgcn_ballot_w64(real_a && col_lane));
                    const int f_b = (int)__popcll(__builtin_amdgcn_ballot_w64(real_b && col_lane));
                    if (lane == 0) {
                        if (ok_a) a.nbr_cnt[row0 + qi] = f_a;
                        if (ok_b) a.nbr_cnt[row0 + qj] = f_b;
                    }
                }
            }
        }
    };
    {
        using std::integral_constant;
        const int nbp = (m + 127) >> 7;
        if (nbp <= 1) pair_loop(integral_constant<int, 1>{});
        else if (nbp == 2) pair_loop(integral_constant<int, 2>{});
        else if (nbp == 3) pair_loop(integral_constant<int, (CAP >= 384 ? 3 : 1)>{});
        else if (nbp == 4) pair_loop(integral_constant<int, (CAP >= 512 ? 4 : 1)>{});
        else if (nbp == 5) pair_loop(integral_constant<int, (CAP >= 640 ? 5 : 1)>{});
        else pair_loop(integral_constant<int, (CAP >= 768 ? 6 : 1)>{});
    }
#if defined(PCT_ABL_NO_SORT) || defined(PCT_ABL_NO_COMPACT) || defined(PCT_ABL_NO_KEYS) || defined(PCT_ABL_NO_TRIAL) || defined(PCT_ABL_NO_STORE) || defined(PCT_ABL_NO_CHECK)
    redo_mask = 0ull;          // timing experiments: nothing goes to the exact sweep
#endif
    if (redo_mask) {
        const int cnt = (int)__popcll(redo_mask);
        int base = 0;
        if (lane == 0) base = atomicAdd(a.redo_count, cnt);
        base = __builtin_amdgcn_readfirstlane(base);
        if ((redo_mask >> lane) & 1ull) a.redo[base + (int)__popcll(redo_mask & ((1ull << lane) - 1ull))] = row0 + lane;
        if (a.stats && lane == 0) atomicAdd(&a.counters[4], (unsigned long long)cnt);
    }
}

// ---------------------------------------------------------------------------
// k_knn_duo: k_knn_pair's scheme for rows of 65 .. 128 entries (k = 64 .. 127; BASELINE configs[4] asks for k = 80) --
// a float32 cloud, the uniform cell list, a plain sweep.  ONE query per loop trip; its list is two registers per lane
// (element = lane + 64 * register), and the two registers take the roles the two queries of a pair play in
// k_knn_pair: the compaction handles two staged batches per block of instructions, the exact keys of survivors
// `lane` and `lane + 64` are two interleaved fp64 chains, and the sorting network (pct_sort_duo.inc, the same
// generator) sorts the two halves side by side and then merges them (element i against 127 - i, strides 32 .. 1).
// Same proofs, same bit-identical rows as k_knn_fast<2, EPS, true, true>: what could not be proven goes to
// the redo list.  Positions (and distances) of the survivors wait in LDS for the sorted order.
// ---------------------------------------------------------------------------
#include "pct_sort_duo.inc"

__device__ __forceinline__ void sort_duo_asm(unsigned& ea, unsigned& eb, const SortLanes& c) {
    unsigned ta, tb;
    asm volatile(PCT_SORT_DUO_ASM
                 : [ea] "+v"(ea), [eb] "+v"(eb), [ta] "=&v"(ta), [tb] "=&v"(tb)
                 : [sel0] "v"(c.sel[0]), [sel1] "v"(c.sel[1]), [sel2] "v"(c.sel[2]), [sel3] "v"(c.sel[3]), [sel4] "v"(c.sel[4]),
                   [sel5] "v"(c.sel[5]), [a31] "v"(c.a31), [a63] "v"(c.a63));
    const unsigned lo = PCT_SORT_DUO_RESULT_A, hi = PCT_SORT_DUO_RESULT_B;
    ea = lo;
    eb = hi;
}

// Staging capacity: cells are sized for 0.35 (k + 1) points, a surface's 27-cell stencil then holds 12 - 14 cells' worth --
// 400 - 470 candidates at k = 64 .. 80: 512 slots sent 8 % of the items (k = 64) to 30 % (k = 80) to the exact sweep,
// 768 slots (4 waves per SIMD with the 16-bit survivor list) send a handful.
#ifndef PCT_DUO_CAP
#define PCT_DUO_CAP 768
#endif
constexpr int kDuoCap = PCT_DUO_CAP;
template <bool DIST, int CAP>
struct DuoLds {
    float cx[CAP], cy[CAP], cz[CAP];                     // staged stencil, 12 B per candidate
    unsigned short pend[128 + 8];                        // staged slot (| run << 10) of survivor s; (first: the run-start bit string)
    int pay_p[128];                                      // sorted position of survivor s
    float pay_d[DIST ? 128 : 1];                         // its float32 distance
    int offc[16];                                        // sorted position - flat slot, per non-empty run
};

// Q64: a float64 cloud, as in k_knn_pair -- float32-rounded candidates, native float64 queries, every bound taken from the
// float32 pre-selection widened by eq = |q64 - q32|.
// TREE: the items of the hierarchical cell list (as in k_knn_pair), 1024 staged slots -- what pct_tree.hip refines
// segments for when two list registers are in use.
template <bool EPS, bool DIST, bool Q64 = false, bool TREE = false>
__global__ __launch_bounds__(64, ((TREE ? PCT_TREE_CAP2 : PCT_DUO_CAP) <= 768 ? 4 : 3)) void k_knn_duo(PairArgs a) {
    constexpr int CAP = TREE ? PCT_TREE_CAP2 : kDuoCap, LIST = 128, SLOT_BITS = 7, KEY_BITS = 32 - SLOT_BITS;
    static_assert(CAP % 128 == 0 && CAP <= 1024, "slot ids: 10 bits of slot, 4 bits of run index");
    __shared__ DuoLds<DIST, CAP> L;
    const int lane = lane_id();
    const int item = a.items_per_xcd ? ((int)blockIdx.x & 7) * a.items_per_xcd + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    if (item >= a.n_items) return;
    const SortLanes sort_dir = make_sort_lanes();
    const int* __restrict__ cs = a.cell_start;

    // ---- the work item, its stencil and its queries: as in k_knn_pair -----------------------------------------------
    const int2 it2 = a.items[item];
    constexpr int NRUNS = TREE ? 27 : 9;
    int cx, cy, cz, qs, nq, row0;
    int run_s = 0, run_len = 0;
    pct_grid g_lvl;                                               // TREE: the grid of the item's level
    if constexpr (TREE) {
        const int seg = __builtin_amdgcn_readfirstlane(it2.y);
        if (seg < 0) return;                                      // an item of a segment that was split (k_tree_refine)
        const unsigned packed = (unsigned)__builtin_amdgcn_readfirstlane(it2.x);
        qs = (int)(packed & 0x3ffffffu);
        nq = (int)(packed >> 26) + 1;
        row0 = qs;
        const int4 hd = a.tree_seg[seg];
        const int level = __builtin_amdgcn_readfirstlane(hd.x);
        cx = __builtin_amdgcn_readfirstlane(hd.y);
        cy = __builtin_amdgcn_readfirstlane(hd.z);
        cz = __builtin_amdgcn_readfirstlane(hd.w);
        g_lvl = a.g;
        g_lvl.cell = __builtin_ldexp(a.g.cell, level);
        g_lvl.inv_cell = __builtin_ldexp(a.g.inv_cell, -level);
        g_lvl.nx = g_lvl.ny = g_lvl.nz = 1 << (a.tree_bits - level);
        if (lane < 27) {
            const int2 r = a.tree_runs[(int64_t)seg * 27 + lane];
            run_s = r.x;
            run_len = r.y;
        }
    } else {
        const int cell = __builtin_amdgcn_readfirstlane(it2.x);
        const int chunk = __builtin_amdgcn_readfirstlane(it2.y);
        const int nx = a.g.nx, ny = a.g.ny, nz = a.g.nz;
        cz = (int)(((unsigned long long)(unsigned)cell * a.magic_xy) >> a.shift_xy);
        const int rem = cell - cz * (nx * ny);
        cy = (int)(((unsigned long long)(unsigned)rem * a.magic_x) >> a.shift_x);
        cx = rem - cy * nx;
        const int c0 = cs[cell];
        qs = c0 + chunk * a.items_q;
        nq = min(c0 + a.cell_own[cell], qs + a.items_q) - qs;
        row0 = a.own_start[cell] + chunk * a.items_q;
        if (lane < 9) {
            const int z = cz + kRowOrder[lane][0], y = cy + kRowOrder[lane][1];
            if (z >= 0 && z < nz && y >= 0 && y < ny) {
                const int row = (z * ny + y) * nx;
                run_s = cs[row + max(cx - 1, 0)];
                run_len = cs[row + min(cx + 1, nx - 1) + 1] - run_s;
            }
        }
    }
    const pct_grid& G = TREE ? g_lvl : a.g;
    float4 my_q = make_float4(0.f, 0.f, 0.f, 0.f);
    double my_qx = 0., my_qy = 0., my_qz = 0.;
    float my_eq = 0.f;       // Q64: distance between the float64 query and its float32 rounding, rounded up
    if (lane < nq) {
        my_q = a.pts[qs + lane];
        if constexpr (Q64) {
            const double4 qd = a.ptsd[qs + lane];
            my_qx = qd.x; my_qy = qd.y; my_qz = qd.z;
        }
    }
    if constexpr (Q64) {
        const double ex = my_qx - (double)my_q.x, ey = my_qy - (double)my_q.y, ez = my_qz - (double)my_q.z;
        my_eq = (float)sqrt((ex * ex + ey * ey) + ez * ez) * (1.0f + 0x1p-22f);
        if (!(my_eq >= 0.f)) my_eq = INFINITY;
    }
    int my_pre = 0, m = 0;
    {
        int acc = 0;
#pragma unroll
        for (int t = 0; t < NRUNS; ++t) {
            my_pre = lane == t ? acc : my_pre;
            acc += __builtin_amdgcn_readlane(run_len, t);
        }
        m = acc;
    }
    const bool crowded = TREE && __popcll(__builtin_amdgcn_ballot_w64(lane < NRUNS && run_len > 0)) > 16;      // (four bits of run index)
    if (m > CAP || crowded) {
        int base = 0;
        if (lane == 0) base = atomicAdd(a.redo_count, nq);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < nq) a.redo[base + lane] = row0 + lane;
        if (a.stats && lane == 0) {
            atomicAdd(&a.counters[1], 1ull);
            atomicAdd(&a.counters[4], (unsigned long long)nq);
        }
        return;
    }
    unsigned slotx[CAP / 64];
    {
        unsigned* bits = (unsigned*)L.pend;
        static_assert(sizeof(L.pend) >= CAP / 8, "the run-start bit string lives in the survivor list");
        if (lane < CAP / 32) bits[lane] = 0u;
        wave_lds_sync();
        const bool nonempty = lane < NRUNS && run_len > 0;
        const unsigned long long ne = __builtin_amdgcn_ballot_w64(nonempty);
        if (nonempty) {
            atomicOr(&bits[my_pre >> 5], 1u << (my_pre & 31));
            L.offc[__builtin_amdgcn_mbcnt_lo((unsigned)ne, 0)] = run_s - my_pre;
        }
        wave_lds_sync();
        float4 tmp[CAP / 64];
        int ubase = -1;
#pragma unroll
        for (int b = 0; b < CAP / 64; ++b) {
            tmp[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            slotx[b] = (unsigned)(b * 64 + lane);
            if (b * 64 < m) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)bits[2 * b]);
                const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)bits[2 * b + 1]);
                const unsigned long long B = ((unsigned long long)hi << 32) | lo;
                const unsigned long long S = B >> 1;
                const int s0 = ubase + (int)(lo & 1u);
                const int u = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(S >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)S, (unsigned)s0));
                ubase += (int)__popcll(B);
                const int j = b * 64 + lane;
                slotx[b] |= (unsigned)u << 10;
                if (j < m) tmp[b] = a.pts[j + L.offc[u]];
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int b = 0; b < CAP / 64; ++b) {
            const int j = b * 64 + lane;
            if (j < m) {
                L.cx[j] = tmp[b].x; L.cy[j] = tmp[b].y; L.cz[j] = tmp[b].z;
            } else if ((b & ~1) * 64 < m) {
                L.cx[j] = INFINITY; L.cy[j] = 0.f; L.cz[j] = 0.f;
            }
        }
    }
    wave_lds_sync();

    const int k = a.k;
    const double eps2 = EPS ? a.eps2 : (double)INFINITY;
    constexpr double kKeyRange = 2.3;
    const double edge = G.cell;
    const double scale = (double)(1u << KEY_BITS) / (kKeyRange * edge * edge);
    constexpr unsigned key_max = (1u << KEY_BITS) - 1u;
    unsigned my_gkey;
    {
        const double lqx = Q64 ? my_qx : (double)my_q.x, lqy = Q64 ? my_qy : (double)my_q.y, lqz = Q64 ? my_qz : (double)my_q.z;
        const double gx = (lqx - G.ox) * G.inv_cell - cx;
        const double gy = (lqy - G.oy) * G.inv_cell - cy;
        const double gz = (lqz - G.oz) * G.inv_cell - cz;
        const double g2 = fmin(guaranteed_r2(G, cx, cy, cz, gx, gy, gz, 1), limit_r2(G, cx, cy, cz, gx, gy, gz));
        my_gkey = g2 == INFINITY ? 0xFFFFFFFFu : (unsigned)fmin(g2 * scale, 4294967294.0);
    }
    const unsigned eps_key = EPS && eps2 < 1e300 ? (unsigned)fmin(ceil(eps2 * scale), 4294967295.0) : 0xFFFFFFFFu;
    const float cell2f = (float)(edge * edge);
    const float eps2a = EPS ? (float)fmin(eps2 * (1.0 + 0x1p-18), 3.0e38) : INFINITY;
    const double eps1 = EPS ? sqrt(eps2) * (1.0 + 0x1p-50) : 0.0;       // eps itself, rounded up (Q64)
    float t_prev_f = 0.f;
    unsigned long long redo_mask = 0ull;

    char* const pos_item = (char*)(a.nbr_pos + (int64_t)row0 * a.pitch);
    char* const dist_item = DIST ? (char*)(a.nbr_dist + (int64_t)row0 * a.pitch) : nullptr;
    const unsigned pitch4 = (unsigned)a.pitch * 4u;
    // list entry i = lane + 64 r  ->  table column i - 1
    const unsigned lane_off0 = (unsigned)(lane - 1) * 4u, lane_off1 = (unsigned)(lane + 63) * 4u;
    const bool col0 = lane >= 1 && lane <= k, col1 = lane + 64 <= k;
    // entries 0 .. k are the ones whose order matters: lanes 0 .. k of register 0, lanes 0 .. k - 64 of register 1
    const unsigned long long order_lo = k >= 63 ? ~0ull : (2ull << k) - 1ull;
    const unsigned long long order_hi = k < 64 ? 0ull : k - 64 >= 63 ? ~0ull : (2ull << (k - 64)) - 1ull;
    const unsigned short* pend_hi = &L.pend[64];

    const auto query_loop = [&](auto NBP_) {
        constexpr int NBP = decltype(NBP_)::value, NBU = 2 * NBP;
        for (int qi = 0; qi < nq; ++qi) {
            const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.x), qi));
            const float ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.y), qi));
            const float az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_q.z), qi));
            // the query the exact keys measure from: the float32 record widened, or (Q64) the native coordinates
            const auto rl64 = [&](double v, int l) {
                return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
            };
            const double qx = Q64 ? rl64(my_qx, qi) : (double)ax, qy = Q64 ? rl64(my_qy, qi) : (double)ay, qz = Q64 ? rl64(my_qz, qi) : (double)az;
            const float eq = Q64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_eq), qi)) : 0.f;
            // ---- float32 squared distances of all staged candidates (packed: two batches per instruction) ----------
            float ap[NBU];
#pragma unroll
            for (int p2 = 0; p2 < NBP; ++p2) {
                const int sa = p2 * 128 + lane, sb = sa + 64;
                const float2v vx = {L.cx[sa], L.cx[sb]}, vy = {L.cy[sa], L.cy[sb]}, vz = {L.cz[sa], L.cz[sb]};
                const float2v dx = vx - ax, dy = vy - ay, dz = vz - az;
                float2v d = dx * dx;
                d = __builtin_elementwise_fma(dy, dy, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                ap[2 * p2] = d.x;
                ap[2 * p2 + 1] = d.y;
            }
            // ---- threshold: k+1 <= #(d < T) <= LIST, never beyond the eps ball (wave-uniform search) -----------------
            // +inf without eps; Q64: exact d < eps  =>  d' < eps + eq
            float T_init = eps2a;
            if constexpr (EPS && Q64) {
                const double ee = eps1 + (double)eq;
                T_init = (float)fmin(ee * ee * (1.0 + 0x1p-18), 3.0e38);
            }
            int tot = m;
            if constexpr (EPS) {
                tot = 0;
#pragma unroll
                for (int b = 0; b < NBU; ++b) tot += (int)__popcll(__builtin_amdgcn_ballot_w64(ap[b] < T_init));
            }
            const bool need = tot > LIST;
            float T = T_init;
            int cnt = tot;
            unsigned bkey = 0xFFFFFFFFu;          // exact keys of the candidates the pre-selection cut are >= bkey
            if (need) {
                const float target = 0.5f * (float)(k + 1 + LIST);
                float t = t_prev_f > 0.f ? t_prev_f : cell2f;
                if (!(t < T_init)) t = 0.5f * T_init;
                float lo = 0.f, hi = T_init;
                bool found = false;
#pragma unroll 1
                for (int trial = 0; trial < 16; ++trial) {
                    int c = 0;
#pragma unroll
                    for (int b = 0; b < NBU; ++b) c += (int)__popcll(__builtin_amdgcn_ballot_w64(ap[b] < t));
                    if ((unsigned)(c - (k + 1)) <= (unsigned)(LIST - (k + 1))) { T = t; cnt = c; found = true; break; }
                    const bool below = c < k + 1;
                    lo = below ? t : lo;
                    hi = below ? hi : t;
                    float nt = t * target * __builtin_amdgcn_rcpf((float)c);
                    if (!(nt > lo && nt < hi)) nt = hi < INFINITY ? 0.5f * (lo + hi) : 2.f * lo;
                    if (__builtin_amdgcn_ballot_w64(!(nt > lo && nt < hi)) != 0ull) break;      // no float left between: a pile of equal distances
                    t = nt;
                }
                if (!found || __builtin_amdgcn_ballot_w64(!(T >= 1e-30f)) != 0ull) { redo_mask |= 1ull << qi; continue; }
                t_prev_f = T;
                // smallest exact key a candidate cut by the float32 threshold can have (k_knn_pair)
                double lo2 = (double)T * (1.0 - 0x1p-20);
                if constexpr (Q64) {
                    // (sqrt(L) - eq)^2 >= L - 2 eq sqrt(L); an upper bound of the root is enough: float32 root, rounded up
                    const double root_up = (double)__builtin_sqrtf(T) * (1.0 + 0x1p-21);
                    lo2 = fmax(lo2 - 2.0 * (double)eq * root_up, 0.0);
                }
                bkey = (unsigned)fmin(lo2 * scale, 4294967294.0);
            }
            // ---- compact the staged slots of the survivors, two batches per block of instructions (k_knn_pair's
            // hand-placed sequence; here both batches append to the same list)
            {
                unsigned wr = (unsigned)(uintptr_t)&L.pend[0], wr1;
                const unsigned long long all = __builtin_amdgcn_read_exec();
                wave_lds_sync();
#pragma unroll
                for (int b = 0; b < NBU; b += 2) {
                    unsigned r0, r1, n0, n1;
                    asm volatile(
                        "v_cmp_gt_f32 vcc, %[t], %[ap0]\n"
                        "v_cmp_gt_f32 s[96:97], %[t], %[ap1]\n"
                        "s_bcnt1_i32_b64 %[n0], vcc\n"
                        "v_mbcnt_lo_u32_b32 %[r0], vcc_lo, 0\n"
                        "s_bcnt1_i32_b64 %[n1], s[96:97]\n"
                        "v_mbcnt_lo_u32_b32 %[r1], s96, 0\n"
                        "v_mbcnt_hi_u32_b32 %[r0], vcc_hi, %[r0]\n"
                        "v_mbcnt_hi_u32_b32 %[r1], s97, %[r1]\n"
                        "s_lshl1_add_u32 %[wr1], %[n0], %[wr]\n"
                        "v_lshl_add_u32 %[r0], %[r0], 1, %[wr]\n"
                        "v_lshl_add_u32 %[r1], %[r1], 1, %[wr1]\n"
                        "s_mov_b64 exec, vcc\n"
                        "ds_write_b16 %[r0], %[slot0]\n"
                        "s_mov_b64 exec, s[96:97]\n"
                        "ds_write_b16 %[r1], %[slot1]\n"
                        "s_mov_b64 exec, %[all]\n"
                        "s_lshl1_add_u32 %[wr], %[n1], %[wr1]\n"
                        : [r0] "=&v"(r0), [r1] "=&v"(r1), [n0] "=&s"(n0), [n1] "=&s"(n1), [wr] "+s"(wr), [wr1] "=&s"(wr1)
                        : [t] "v"(T), [ap0] "v"(ap[b]), [ap1] "v"(ap[b + 1]), [slot0] "v"(slotx[b]), [slot1] "v"(slotx[b + 1]), [all] "s"(all)
                        : "vcc", "scc", "s96", "s97", "memory");
                }
                wave_lds_sync();
            }
            // ---- exact keys of survivors `lane` and `lane + 64` (two interleaved fp64 chains); a stale list entry is
            // masked into the staging area and gives a garbage value nobody uses
            const unsigned sx0 = (unsigned)L.pend[lane] & 0x3FFFu, sx1 = (unsigned)pend_hi[lane] & 0x3FFFu;      // slot | run << 10
            const int j0 = min((int)(sx0 & 1023u), CAP - 1), j1 = min((int)(sx1 & 1023u), CAP - 1);
            unsigned e[2];
            {
                const double dx0 = (double)L.cx[j0] - qx, dy0 = (double)L.cy[j0] - qy, dz0 = (double)L.cz[j0] - qz;
                const double dx1 = (double)L.cx[j1] - qx, dy1 = (double)L.cy[j1] - qy, dz1 = (double)L.cz[j1] - qz;
                const double d20 = (dx0 * dx0 + dy0 * dy0) + dz0 * dz0;
                const double d21 = (dx1 * dx1 + dy1 * dy1) + dz1 * dz1;
                L.pay_p[lane] = j0 + L.offc[sx0 >> 10];
                L.pay_p[lane + 64] = j1 + L.offc[sx1 >> 10];
                if constexpr (DIST) {
                    L.pay_d[lane] = (float)sqrt(d20);
                    L.pay_d[lane + 64] = (float)sqrt(d21);
                }
                const unsigned k0 = (min((unsigned)(d20 * scale), key_max - 1u) << SLOT_BITS) | (unsigned)lane;
                const unsigned k1 = (min((unsigned)(d21 * scale), key_max - 1u) << SLOT_BITS) | (unsigned)(lane + 64);
                e[0] = lane < cnt && (!EPS || d20 < eps2) ? k0 : kPadElem;
                e[1] = lane + 64 < cnt && (!EPS || d21 < eps2) ? k1 : kPadElem;
            }
            wave_lds_sync();
            sort_duo_asm(e[0], e[1], sort_dir);
            // ---- proof obligations (key units, see k_knn_fast) ----------------------------------------------------
            const unsigned tau = (unsigned)__builtin_amdgcn_readlane((int)(k < 64 ? e[0] : e[1]), k & 63);     // the (k+1)-th nearest (padding if fewer exist)
            const unsigned gk = (unsigned)__builtin_amdgcn_readlane((int)my_gkey, qi);
            const unsigned tk = tau >> SLOT_BITS;
            const unsigned need_k = min(tau == kPadElem ? 0xFFFFFFFFu : tk + 1u, eps_key);
            if (need_k > min(gk, bkey) || (tau != kPadElem && tk >= key_max - 1u)) { redo_mask |= 1ull << qi; continue; }
            // equal keys among the first k + 2 entries: ordered here by the exact values (order_equal_keys).  Detection:
            // element i ^ element i + 1 below 2^SLOT_BITS <=> same key
            {
                unsigned n0, n1;
                asm("s_nop 1\n"
                    "v_mov_b32_dpp %0, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                    "v_mov_b32_dpp %1, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                    "s_nop 0"
                    : "=&v"(n0), "=&v"(n1) : "v"(e[0]), "v"(e[1]));
                n0 = lane == 63 ? (unsigned)__builtin_amdgcn_readlane((int)e[1], 0) : n0;        // entry 64 follows entry 63
                const bool same0 = ((e[0] ^ n0) >> SLOT_BITS) == 0u && e[0] != kPadElem && n0 != kPadElem;
                const bool same1 = ((e[1] ^ n1) >> SLOT_BITS) == 0u && e[1] != kPadElem && n1 != kPadElem && lane < 63;
                const unsigned long long cm = (__builtin_amdgcn_ballot_w64(same0) & order_lo) | (__builtin_amdgcn_ballot_w64(same1) & order_hi);
                if (__builtin_expect(cm != 0ull, 0)) {
                    const bool done = order_equal_keys<2, SLOT_BITS>(e, a.pts,
                        [&](unsigned at) {
                            const int j = min((int)L.pend[at] & 1023, CAP - 1);
                            const double dx = (double)L.cx[j] - qx, dy = (double)L.cy[j] - qy, dz = (double)L.cz[j] - qz;
                            return (dx * dx + dy * dy) + dz * dz;
                        },
                        [&](unsigned at) { return L.pay_p[at]; });
                    if (!done) { redo_mask |= 1ull << qi; continue; }
                }
            }
            // ---- store: the lane that holds list entry i looks up position (and distance) of survivor e & 127 --------
            {
                const unsigned off0 = lane_off0 + (unsigned)qi * pitch4, off1 = lane_off1 + (unsigned)qi * pitch4;
                const bool real0 = e[0] != kPadElem, real1 = e[1] != kPadElem;
                const int s0 = (int)(e[0] & 127u), s1 = (int)(e[1] & 127u);
                const int pos0 = L.pay_p[s0], pos1 = L.pay_p[s1];
                if (col0) *(int*)(pos_item + off0) = real0 ? pos0 : -1;
                if (col1) *(int*)(pos_item + off1) = real1 ? pos1 : -1;
                if constexpr (DIST) {
                    const float d0 = L.pay_d[s0], d1 = L.pay_d[s1];
                    if (col0) *(float*)(dist_item + off0) = real0 ? d0 : INFINITY;
                    if (col1) *(float*)(dist_item + off1) = real1 ? d1 : INFINITY;
                }
                if constexpr (EPS) {
                    const int f = (int)__popcll(__builtin_amdgcn_ballot_w64(real0 && col0)) + (int)__popcll(__builtin_amdgcn_ballot_w64(real1 && col1));
                    if (lane == 0) a.nbr_cnt[row0 + qi] = f;
                }
            }
            wave_lds_sync();          // the payload arrays are free for the next query
        }
    };
    {
        using std::integral_constant;
        const int nbp = (m + 127) >> 7;
        if (nbp <= 1) query_loop(integral_constant<int, 1>{});
        else if (nbp == 2) query_loop(integral_constant<int, 2>{});
        else if (nbp == 3) query_loop(integral_constant<int, 3>{});
        else if (nbp == 4) query_loop(integral_constant<int, 4>{});
        else if (nbp == 5) query_loop(integral_constant<int, 5>{});
        else if (nbp == 6) query_loop(integral_constant<int, 6>{});
        else if (nbp == 7) query_loop(integral_constant<int, (CAP >= 896 ? 7 : 1)>{});
        else query_loop(integral_constant<int, (CAP >= 1024 ? 8 : 1)>{});
    }
    if (redo_mask) {
        const int cnt = (int)__popcll(redo_mask);
        int base = 0;
        if (lane == 0) base = atomicAdd(a.redo_count, cnt);
        base = __builtin_amdgcn_readfirstlane(base);
        if ((redo_mask >> lane) & 1ull) a.redo[base + (int)__popcll(redo_mask & ((1ull << lane) - 1ull))] = row0 + lane;
        if (a.stats && lane == 0) atomicAdd(&a.counters[4], (unsigned long long)cnt);
    }
}

// cKDTree.query for caller-supplied points (pct_query_points): the exhaustive sweep with the query read from a
// separate array and every element of the list stored (nothing is "the point itself" here).
__global__ __launch_bounds__(256) void k_plain_records(const float* __restrict__ xyz, int64_t n, float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], __int_as_float((int)i));
}

template <int R>
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_query_points(const float4* __restrict__ pts, int n, const double* __restrict__ q_xyz,
                                                                      int64_t m, int k, double eps2, int* __restrict__ idx_out,
                                                                      double* __restrict__ dist_out) {
    __shared__ double s_pend_d[kWavesPerBlock][64 * R + 64];
    __shared__ int s_pend_p[kWavesPerBlock][64 * R + 64];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int64_t q = (int64_t)blockIdx.x * kWavesPerBlock + w;
    if (q >= m) return;
    Sweep<R> sw;
    sw.k = k - 1;                        // the list keeps elements 0 .. sw.k: the k nearest
    sw.eps2 = eps2;
    sw.pts = pts;
    sw.pend_d = s_pend_d[w];
    sw.pend_p = s_pend_p[w];
    sw.qx = q_xyz[3 * q]; sw.qy = q_xyz[3 * q + 1]; sw.qz = q_xyz[3 * q + 2];
    sw.reset();
    for (int base = 0;; base += 64) {
        const bool have = base < n;
        if (have) {
            const int pos = base + lane;
            const bool valid = pos < n;
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) c = pts[pos];
            sw.consider(c, pos, valid);
            if (sw.npend < 64 * R) continue;
        }
        if (sw.npend > 0 || sw.empty) sw.flush();
        if (!have) break;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        if (i < k) {
            const bool real = sw.best.p[r] != INT_MAX;
            idx_out[q * k + i] = real ? sw.best.p[r] : n;
            dist_out[q * k + i] = real ? sqrt(sw.best.d[r]) : (double)INFINITY;
        }
    }
}

// The float32 distance of a table entry, from the two records: the sweep's own expression -- fp64 ((dx^2 + dy^2) +
// dz^2) without contraction, correctly rounded root, one rounding to float32 (pct:78) -- so a table written without
// distances (the fused curvature call) yields the very bits the sweep would have stored.
__device__ __forceinline__ float table_distance(const float4* pts, const double4* ptsd, int qpos, const float4 c) {
    double qx, qy, qz;                 // the query as the sweep measured from it: native float64 coordinates where the cloud has them (pct:83)
    if (ptsd) { const double4 q = ptsd[qpos]; qx = q.x; qy = q.y; qz = q.z; }
    else { const float4 q = pts[qpos]; qx = (double)q.x; qy = (double)q.y; qz = (double)q.z; }
    const double dx = (double)c.x - qx, dy = (double)c.y - qy, dz = (double)c.z - qz;
    return (float)sqrt((dx * dx + dy * dy) + dz * dz);
}

// neighbour table -> public (rows,k) arrays for public rows [begin,end).  owned_pos == nullptr: the table came
// from the exhaustive sweep (row = public index - q_begin, entries = public indices).
__global__ __launch_bounds__(256) void k_export(const float4* __restrict__ pts, const double4* __restrict__ ptsd, const int* __restrict__ owned_pos, int q_begin,
                                                const int* __restrict__ nbr_pos, const float* __restrict__ nbr_dist,
                                                const int* __restrict__ nbr_cnt, int64_t n, int64_t n_rows, int k, int pitch,
                                                int64_t begin, int64_t end, int* __restrict__ idx_out,
                                                float* __restrict__ dist_out, int* __restrict__ cnt_out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / k;
    const int j = (int)(t - row * k);
    if (row >= n_rows) return;
    const int pub = owned_pos ? __float_as_int(pts[owned_pos[row]].w) : (int)row + q_begin;
    if (pub < begin || pub >= end) return;
    const int64_t o = (int64_t)(pub - begin) * k + j;
    const int pos = nbr_pos[row * pitch + j];
    const float4 c = pos < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : pts[pos];
    if (idx_out) idx_out[o] = pos < 0 ? (int)n : __float_as_int(c.w);
    if (dist_out) dist_out[o] = nbr_dist ? nbr_dist[row * pitch + j] : pos < 0 ? INFINITY : table_distance(pts, ptsd, owned_pos[row], c);
    if (cnt_out && j == 0) cnt_out[pub - begin] = nbr_cnt ? nbr_cnt[row] : k;
}

// the same for an explicit list of public rows (one block per listed row)
__global__ __launch_bounds__(128) void k_export_rows(const float4* __restrict__ pts, const double4* __restrict__ ptsd, const int* __restrict__ row_of, const int* __restrict__ owned_pos, int q_begin,
                                                     const int* __restrict__ nbr_pos, const float* __restrict__ nbr_dist,
                                                     const int* __restrict__ nbr_cnt, int64_t n, int k, int pitch,
                                                     const int64_t* __restrict__ rows, int* __restrict__ idx_out,
                                                     float* __restrict__ dist_out, int* __restrict__ cnt_out) {
    const int64_t r = blockIdx.x;
    const int64_t pub = rows[r];
    const int64_t row = row_of ? row_of[pub - q_begin] : pub - q_begin;
    for (int j = threadIdx.x; j < k; j += 128) {
        const int pos = nbr_pos[row * pitch + j];
        const float4 c = pos < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : pts[pos];
        if (idx_out) idx_out[r * k + j] = pos < 0 ? (int)n : __float_as_int(c.w);
        if (dist_out) dist_out[r * k + j] = nbr_dist ? nbr_dist[row * pitch + j] : pos < 0 ? INFINITY : table_distance(pts, ptsd, owned_pos[row], c);
    }
    if (cnt_out && threadIdx.x == 0) cnt_out[r] = nbr_cnt ? nbr_cnt[row] : k;
}

// lane_xor<S>() against the generic shuffle, every S (hardware self-test)
__global__ __launch_bounds__(64) void k_selftest(int* fails) {
    const int lane = lane_id();
    const int v = lane * 7919 + 13;
    int bad = 0;
    bad += lane_xor<1>(v) != __shfl_xor(v, 1);
    bad += lane_xor<2>(v) != __shfl_xor(v, 2);
    bad += lane_xor<4>(v) != __shfl_xor(v, 4);
    bad += lane_xor<8>(v) != __shfl_xor(v, 8);
    bad += lane_xor<16>(v) != __shfl_xor(v, 16);
    bad += lane_xor<32>(v) != __shfl_xor(v, 32);
    if (bad) atomicAdd(fails, bad);
}

// One thread per work item: population and non-empty cells of its 27-cell stencil (pct_item_census).
__global__ __launch_bounds__(256) void k_item_census(const int2* __restrict__ items, int64_t n_items, int items_q,
                                                     const int* __restrict__ cs, const int* __restrict__ cell_own, pct_grid g, int k,
                                                     int cap, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long sh[4][4];
    unsigned long long v[4] = {0, 0, 0, 0};
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < n_items; it += (int64_t)gridDim.x * 256) {
        const int2 e = items[it];
        const int cell = e.x;
        const int cx = cell % g.nx, cy = (cell / g.nx) % g.ny, cz = cell / (g.nx * g.ny);
        const int nq = min(items_q, cell_own[cell] - e.y * items_q);
        int m = 0, occupied = 0;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy) {
                const int z = cz + dz, y = cy + dy;
                if (z < 0 || z >= g.nz || y < 0 || y >= g.ny) continue;
                const int row = (z * g.ny + y) * g.nx;
                int prev = cs[row + max(cx - 1, 0)];
                for (int x = max(cx - 1, 0); x <= min(cx + 1, g.nx - 1); ++x) {
                    const int next = cs[row + x + 1];
                    occupied += next > prev;
                    m += next - prev;
                    prev = next;
                }
            }
        v[0] += (unsigned)nq;
        // the stencil vouches for about one cell edge around the query: on a surface that disc holds ~pi/9 of the
        // stencil's population, so a stencil below ~2.5 (k+1) points will mostly fail the proof ("short")
        if (m > cap) v[1] += (unsigned)nq;
        else if (2 * m < 5 * (k + 1)) v[2] += (unsigned)nq;
        else v[3] += (unsigned long long)nq * (unsigned)occupied;
    }
    for (int j = 0; j < 4; ++j) {
        for (int o = 32; o > 0; o >>= 1) v[j] += __shfl_xor(v[j], o);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][j] = v[j];
    }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&out[threadIdx.x], sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

KnnArgs make_args(pct_ctx* ctx, int32_t k, double eps, bool grid) {
    KnnArgs a = {};
    a.pts = (const float4*)(grid ? ctx->sorted4.p : ctx->pts4.p);
    a.ptsd = ctx->has_f64 ? (const double4*)(grid ? ctx->sorted4d.p : ctx->pts4d.p) : nullptr;
    a.cell_start = (const int*)ctx->cell_cnt.p;
    a.cell_own = (const int*)ctx->cell_own.p;
    a.own_start = (const int*)ctx->own_start.p;
    a.owned_pos = (const int*)ctx->owned_pos.p;
    a.n_owned = ctx->own_flag ? ctx->own_count : ctx->q_end - ctx->q_begin;
    a.occ = (const int*)ctx->occ.p;
    a.n_occ = ctx->n_occ;
    a.n = ctx->n;
    a.g = ctx->grid;
    a.k = k;
    a.pitch = (k + 3) & ~3;
    a.eps2 = eps > 0 ? eps * eps : INFINITY;
    a.q_begin = (int)ctx->q_begin;
    a.q_end = (int)ctx->q_end;
    a.nbr_pos = (int*)ctx->nbr_pos.p;
    a.nbr_dist = (float*)ctx->nbr_dist.p;
    a.nbr_cnt = eps > 0 ? (int*)ctx->nbr_cnt.p : nullptr;
    a.row_done = ctx->own_flag || ctx->level_mode ? (int*)ctx->row_done.p : nullptr;
    a.redo_m = a.row_done ? (int*)ctx->redo_m.p : nullptr;
    a.counters = (unsigned long long*)ctx->counters.p;
    a.stats = ctx->collect_stats ? 1 : 0;
    return a;
}

int reserve_table(pct_ctx* ctx, int32_t k, double eps, bool with_dist = true) {
    ctx->nbr_pitch = (k + 3) & ~3;                        // 16-byte aligned rows (the fit kernel reads int4)
    const size_t rows = (size_t)(ctx->own_flag ? ctx->own_count : ctx->q_end - ctx->q_begin);     // one row per owned query
    PCT_TRY(pct_reserve(ctx, &ctx->nbr_pos, rows * ctx->nbr_pitch * sizeof(int)));
    if (with_dist) PCT_TRY(pct_reserve(ctx, &ctx->nbr_dist, rows * ctx->nbr_pitch * sizeof(float)));
    ctx->dist_valid = with_dist;
    if (eps > 0) PCT_TRY(pct_reserve(ctx, &ctx->nbr_cnt, rows * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->counters, 64 * sizeof(unsigned long long)));
    if (!ctx->counters_clean) PCT_HIP(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
    ctx->counters_clean = false;
    return PCT_OK;
}

}  // namespace

// phase 0: fast sweep + exact sweep of what it flagged (or, exact_only, the exact sweep of every query);
// phase 1: fast sweep only, the flagged rows stay in ctx->redo (level passes); phase 2: exact sweep of ctx->redo
int pct_launch_knn_grid(pct_ctx* ctx, int32_t k, double eps, bool exact_only, int phase) {
    const int64_t n_rows = ctx->own_flag ? ctx->own_count : ctx->q_end - ctx->q_begin;
    // The plain sweep of a float32 cloud with one list register goes to the scalar-lean kernel (k_knn_pair); there the
    // fused curvature call, whose fit never reads distances, writes no distance table (pct_get_neighbors derives the
    // same bits from the positions when asked).
    const double c2_ = ctx->grid.cell * ctx->grid.cell;                // (the float32 pre-selection's range, see f32_ok below)
    const bool f32_ok_ = c2_ > 1e-30 && c2_ < 1e30 && (!(eps > 0) || eps * eps > 1e-36);
    // (float64 clouds take it too unless the rounding distance of the queries is not small against a cell edge --
    // coordinates so large that float32 resolves them barely finer than the cells: see q64_ok below)
    const pct_grid& gg_ = ctx->grid;
    const double far_ = fmax(fmax(fabs(gg_.ox), fabs(gg_.ox + gg_.nx * gg_.cell)),
                             fmax(fmax(fabs(gg_.oy), fabs(gg_.oy + gg_.ny * gg_.cell)), fmax(fabs(gg_.oz), fabs(gg_.oz + gg_.nz * gg_.cell))));
    const bool q64_ok_ = !ctx->has_f64 || far_ * 0x1p-23 < gg_.cell * 0x1p-7;
    const bool pair_kernel = !exact_only && phase == 0 && q64_ok_ && f32_ok_ && !ctx->own_flag && !ctx->level_mode && ctx->n_items > 0 &&
                             k + 1 <= pct_fast_r1_max() && ctx->n_items < ((int64_t)1 << 31) - 8 && !pct_getenv("PCT_NO_PAIR") &&
                             !pct_getenv("PCT_NO_PAIR_KERNEL");
    // rows of 62 .. 128 entries: the same scheme with two list registers (k_knn_duo)
    const bool duo_kernel = !exact_only && phase == 0 && q64_ok_ && f32_ok_ && !ctx->own_flag && !ctx->level_mode && ctx->n_items > 0 &&
                            k + 1 > pct_fast_r1_max() && k + 1 <= 128 && ctx->n_items < ((int64_t)1 << 31) - 8 &&
                            !pct_getenv("PCT_NO_PAIR") && !pct_getenv("PCT_NO_DUO_KERNEL");
    const bool skip_dist = (pair_kernel || duo_kernel) && ctx->skip_dist_req && !pct_getenv("PCT_KEEP_DIST");
    if (phase != 2) {
        PCT_TRY(reserve_table(ctx, k, eps, !skip_dist));
        PCT_TRY(pct_reserve(ctx, &ctx->redo, ((size_t)n_rows + 16) * sizeof(int)));
        if (ctx->level_mode) PCT_TRY(pct_reserve(ctx, &ctx->redo_m, ((size_t)n_rows + 16) * sizeof(int)));
    }
    KnnArgs a = make_args(ctx, k, eps, true);
    if (!ctx->dist_valid) a.nbr_dist = nullptr;
    int* redo_count = (int*)ctx->counters.p + 14;            // counters buffer: 8 x u64, last int pair reserved
    int* redo = (int*)ctx->redo.p;
    const dim3 block(64 * kWavesPerBlock);
    if (!exact_only && phase != 2 && ctx->n_items > 0) {
        const dim3 grid1((unsigned)((ctx->n_items + kFastWaves<1> - 1) / kFastWaves<1>)), block1(64 * kFastWaves<1>);
        const dim3 grid2((unsigned)((ctx->n_items + kFastWaves<2> - 1) / kFastWaves<2>)), block2(64 * kFastWaves<2>);
        const int2* items = (const int2*)ctx->occ.p;
        // The float32 pre-selection squares coordinate differences of up to three cell edges: outside this window
        // they overflow (or the eps ball's radius underflows) and every candidate would fail the threshold test,
        // so such clouds take the variant that keys every candidate in float64.
        const double c2 = ctx->grid.cell * ctx->grid.cell;
        const bool f32_ok = c2 > 1e-30 && c2 < 1e30 && (!(eps > 0) || eps * eps > 1e-36);
        // Float64 clouds take the paired pre-selecting variant too (Q64: bounds widened by the rounding distance of the
        // query) unless that distance is not small against a cell edge -- coordinates so large that float32 resolves
        // them barely finer than the cells: there every query would be sent to the exact sweep.
        const pct_grid& gg = ctx->grid;
        const double far = fmax(fmax(fabs(gg.ox), fabs(gg.ox + gg.nx * gg.cell)),
                                fmax(fmax(fabs(gg.oy), fabs(gg.oy + gg.ny * gg.cell)), fmax(fabs(gg.oz), fabs(gg.oz + gg.nz * gg.cell))));
        const bool q64_ok = ctx->has_f64 && f32_ok && far * 0x1p-23 < gg.cell * 0x1p-7 && !ctx->level_mode;
        const bool e = eps > 0, pre = !ctx->has_f64 && f32_ok, r1 = k + 1 <= pct_fast_r1_max();
#define PCT_FAST(R_, E_, P_, GRID_, BLOCK_) \
    PCT_LAUNCH((k_knn_fast<R_, E_, P_>), GRID_, BLOCK_, 0, ctx->stream, a, items, ctx->n_items, ctx->items_q, redo, redo_count)
        const bool no_pair = pct_getenv("PCT_NO_PAIR") != nullptr;                 // tuning aid (read per call: tests flip it)
#define PCT_FAST_PAIR(R_, E_, GRID_, BLOCK_) \
    PCT_LAUNCH((k_knn_fast<R_, E_, true, true>), GRID_, BLOCK_, 0, ctx->stream, a, items, ctx->n_items, ctx->items_q, redo, redo_count)
#define PCT_FAST_PAIR64(R_, E_, GRID_, BLOCK_) \
    PCT_LAUNCH((k_knn_fast<R_, E_, true, true, true>), GRID_, BLOCK_, 0, ctx->stream, a, items, ctx->n_items, ctx->items_q, redo, redo_count)
        // the plain sweep of a float32 cloud with one list register: the scalar-lean kernel (k_knn_pair)
        if (pair_kernel || duo_kernel) {
            PairArgs pa = {};
            pa.pts = a.pts; pa.ptsd = a.ptsd; pa.cell_start = a.cell_start; pa.cell_own = a.cell_own; pa.own_start = a.own_start;
            pa.items = items;
            pa.nbr_pos = a.nbr_pos; pa.nbr_dist = a.nbr_dist; pa.nbr_cnt = a.nbr_cnt;
            pa.redo = redo; pa.redo_count = redo_count; pa.counters = a.counters;
            pa.n_items = (int)ctx->n_items; pa.items_q = ctx->items_q;
            pa.k = a.k; pa.pitch = a.pitch; pa.stats = a.stats; pa.eps2 = a.eps2; pa.g = a.g;
            // x / d = (x * magic) >> shift for every x < 2^30 (cell ids): shift = 30 + ceil(log2 d), magic = ceil(2^shift / d) < 2^32
            const auto magic = [](unsigned d, unsigned* mg, int* sh) {
                int l = 0;
                while ((1ull << l) < d) ++l;
                *sh = 30 + l;
                *mg = (unsigned)(((1ull << *sh) + d - 1) / d);
            };
            magic((unsigned)a.g.nx, &pa.magic_x, &pa.shift_x);
            magic((unsigned)a.g.nx * (unsigned)a.g.ny, &pa.magic_xy, &pa.shift_xy);
            pa.items_per_xcd = (int)((ctx->n_items + 7) / 8);
            if (pct_getenv("PCT_NO_XCD_MAP")) pa.items_per_xcd = 0;
            const int64_t n_blk = pa.items_per_xcd ? (int64_t)pa.items_per_xcd * 8 : ctx->n_items;
            const dim3 gridp((unsigned)((n_blk + kPairWaves - 1) / kPairWaves)), blockp(64 * kPairWaves);
            if (duo_kernel) {
                const dim3 gridd((unsigned)n_blk), blockd(64);
                if (ctx->has_f64) {
                    if (e && skip_dist) PCT_LAUNCH((k_knn_duo<true, false, true>), gridd, blockd, 0, ctx->stream, pa);
                    else if (e) PCT_LAUNCH((k_knn_duo<true, true, true>), gridd, blockd, 0, ctx->stream, pa);
                    else if (skip_dist) PCT_LAUNCH((k_knn_duo<false, false, true>), gridd, blockd, 0, ctx->stream, pa);
                    else PCT_LAUNCH((k_knn_duo<false, true, true>), gridd, blockd, 0, ctx->stream, pa);
                } else
                if (e && skip_dist) PCT_LAUNCH((k_knn_duo<true, false>), gridd, blockd, 0, ctx->stream, pa);
                else if (e) PCT_LAUNCH((k_knn_duo<true, true>), gridd, blockd, 0, ctx->stream, pa);
                else if (skip_dist) PCT_LAUNCH((k_knn_duo<false, false>), gridd, blockd, 0, ctx->stream, pa);
                else PCT_LAUNCH((k_knn_duo<false, true>), gridd, blockd, 0, ctx->stream, pa);
            } else
            if (ctx->has_f64) {
                if (e && skip_dist) PCT_LAUNCH((k_knn_pair<true, false, true>), gridp, blockp, 0, ctx->stream, pa);
                else if (e) PCT_LAUNCH((k_knn_pair<true, true, true>), gridp, blockp, 0, ctx->stream, pa);
                else if (skip_dist) PCT_LAUNCH((k_knn_pair<false, false, true>), gridp, blockp, 0, ctx->stream, pa);
                else PCT_LAUNCH((k_knn_pair<false, true, true>), gridp, blockp, 0, ctx->stream, pa);
            } else
            if (e && skip_dist) PCT_LAUNCH((k_knn_pair<true, false>), gridp, blockp, 0, ctx->stream, pa);
            else if (e) PCT_LAUNCH((k_knn_pair<true, true>), gridp, blockp, 0, ctx->stream, pa);
            else if (skip_dist) PCT_LAUNCH((k_knn_pair<false, false>), gridp, blockp, 0, ctx->stream, pa);
            else PCT_LAUNCH((k_knn_pair<false, true>), gridp, blockp, 0, ctx->stream, pa);
        } else
        if (q64_ok && !no_pair) {
            if (r1 && !e) PCT_FAST_PAIR64(1, false, grid1, block1);
            else if (r1) PCT_FAST_PAIR64(1, true, grid1, block1);
            else if (!e) PCT_FAST_PAIR64(2, false, grid2, block2);
            else PCT_FAST_PAIR64(2, true, grid2, block2);
        } else
        if (r1 && !e && pre && !no_pair) PCT_FAST_PAIR(1, false, grid1, block1);
        else if (r1 && e && pre && !no_pair) PCT_FAST_PAIR(1, true, grid1, block1);
        else if (!r1 && !e && pre && !no_pair) PCT_FAST_PAIR(2, false, grid2, block2);
        else if (!r1 && e && pre && !no_pair) PCT_FAST_PAIR(2, true, grid2, block2);
        else if (r1 && !e && pre) PCT_FAST(1, false, true, grid1, block1);
        else if (r1 && e && pre) PCT_FAST(1, true, true, grid1, block1);
        else if (r1 && !e) PCT_FAST(1, false, false, grid1, block1);
        else if (r1) PCT_FAST(1, true, false, grid1, block1);
        else if (!e && pre) PCT_FAST(2, false, true, grid2, block2);
        else if (e && pre) PCT_FAST(2, true, true, grid2, block2);
        else if (!e) PCT_FAST(2, false, false, grid2, block2);
        else PCT_FAST(2, true, false, grid2, block2);
#undef PCT_FAST
#undef PCT_FAST_PAIR
#undef PCT_FAST_PAIR64
        PCT_HIP(ctx, hipGetLastError());
    }
    PCT_HIP(ctx, hipEventRecord(ctx->ev[7], ctx->stream));      // end of the dominant kernel
    // exact pass: the flagged queries (device-side count, fixed grid) or, for testing, every query
    if (phase != 1) {
        const int64_t waves = exact_only ? n_rows : 32768;   // one query per wave for typical redo counts
        const int blocks = (int)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
        const int* list = exact_only ? nullptr : redo;
        if (k + 1 <= 64)
            PCT_LAUNCH(k_knn_exact<1>, dim3(blocks), block, 0, ctx->stream, a, list, (const int*)redo_count);
        else if (k + 1 <= 128)
            PCT_LAUNCH(k_knn_exact<2>, dim3(blocks), block, 0, ctx->stream, a, list, (const int*)redo_count);
        else
            PCT_TRY(pct_launch_knn_exact_wide(ctx, a, blocks, list, (const int*)redo_count));
        PCT_HIP(ctx, hipGetLastError());
    }
    ctx->knn_sorted_space = true;
    return PCT_OK;
}

// Neighbour sweep on the hierarchical cell list (pct_build_tree): fast sweep over its work items, then the exact sweep
// on the same structure for what the fast one flagged.  The table is in Morton order (a sorted space like the uniform
// list's: sorted4, owned_pos = identity, row_of).
int pct_launch_knn_tree(pct_ctx* ctx, int32_t k, double eps) {
    if (ctx->level_mode || ctx->own_flag || ctx->q_begin != 0 || ctx->q_end != ctx->n)
        return pct_fail(ctx, PCT_ERR_INVALID, "the tree sweep takes whole clouds");
    const int64_t n_rows = ctx->n;
    // one list register: the scalar-lean kernel's TREE instantiation (k_knn_pair); it can leave the distance table out
    // for the fused call, like the uniform list's
    const bool lean_ok = ctx->n_items > 0 && ctx->n_items < ((int64_t)1 << 31) - 8 && !pct_getenv("PCT_NO_PAIR") &&
                         !pct_getenv("PCT_TREE_EXACT_ONLY");
    const bool pair_tree = lean_ok && k + 1 <= pct_fast_r1_max() && !pct_getenv("PCT_NO_PAIR_KERNEL");
    const bool duo_tree = lean_ok && k + 1 > pct_fast_r1_max() && k + 1 <= 128 && !pct_getenv("PCT_NO_DUO_KERNEL");     // two list registers: k_knn_duo
    const bool skip_dist = (pair_tree || duo_tree) && ctx->skip_dist_req && !pct_getenv("PCT_KEEP_DIST");
    PCT_TRY(reserve_table(ctx, k, eps, !skip_dist));
    PCT_TRY(pct_reserve(ctx, &ctx->redo, ((size_t)n_rows + 16) * sizeof(int)));
    KnnArgs a = make_args(ctx, k, eps, true);
    if (!ctx->dist_valid) a.nbr_dist = nullptr;
    a.tree_seg = (const int4*)ctx->tree_seg.p;
    a.tree_runs = (const int2*)ctx->tree_runs.p;
    a.tree_bits = ctx->tree_bits;
    a.tree_codes = (const unsigned long long*)ctx->tree_codes.p + ctx->n;     // second half: the sorted codes
    a.tree_lvl = (const unsigned char*)ctx->tree_lvl.p;
    a.tree_bucket = (const int*)ctx->tree_bucket.p;
    int* redo_count = (int*)ctx->counters.p + 14;
    int* redo = (int*)ctx->redo.p;
    const int2* items = (const int2*)ctx->occ.p;
    const bool r1 = k + 1 <= pct_fast_r1_max();
    const bool exact_only = pct_getenv("PCT_TREE_EXACT_ONLY") != nullptr;        // testing: every query through the exact sweep
    if (ctx->n_items > 0 && !exact_only) {
        const dim3 grid1((unsigned)((ctx->n_items + kFastWaves<1> - 1) / kFastWaves<1>)), block1(64 * kFastWaves<1>);
        const dim3 grid2((unsigned)((ctx->n_items + kFastWaves<2> - 1) / kFastWaves<2>)), block2(64 * kFastWaves<2>);
        const bool e = eps > 0;
        // float64 clouds: the variant whose bounds are widened by the distance between a query and its float32 rounding
        // (Q64); where that distance is not small against the item's cells the proofs fail and the exact sweep answers
#define PCT_TREE(R_, E_, Q_, GRID_, BLOCK_) \
    PCT_LAUNCH((k_knn_fast<R_, E_, true, true, Q_, true>), GRID_, BLOCK_, 0, ctx->stream, a, items, ctx->n_items, ctx->items_q, redo, redo_count)
        if (pair_tree || duo_tree) {
            PairArgs pa = {};
            pa.pts = a.pts; pa.ptsd = a.ptsd; pa.cell_start = a.cell_start;
            pa.items = items;
            pa.nbr_pos = a.nbr_pos; pa.nbr_dist = a.nbr_dist; pa.nbr_cnt = a.nbr_cnt;
            pa.redo = redo; pa.redo_count = redo_count; pa.counters = a.counters;
            pa.n_items = (int)ctx->n_items; pa.items_q = ctx->items_q;
            pa.k = a.k; pa.pitch = a.pitch; pa.stats = a.stats; pa.eps2 = a.eps2; pa.g = a.g;
            pa.tree_seg = a.tree_seg; pa.tree_runs = a.tree_runs; pa.tree_bits = a.tree_bits;
            pa.items_per_xcd = pct_getenv("PCT_NO_XCD_MAP") ? 0 : (int)((ctx->n_items + 7) / 8);
            const int64_t n_blk = pa.items_per_xcd ? (int64_t)pa.items_per_xcd * 8 : ctx->n_items;
            const dim3 gridp((unsigned)((n_blk + kPairWaves - 1) / kPairWaves)), blockp(64 * kPairWaves);
#define PCT_PAIR_TREE(E_, D_, Q_)                                                                       \
    do {                                                                                                \
        if (duo_tree) PCT_LAUNCH((k_knn_duo<E_, D_, Q_, true>), dim3((unsigned)n_blk), dim3(64), 0, ctx->stream, pa); \
        else PCT_LAUNCH((k_knn_pair<E_, D_, Q_, true>), gridp, blockp, 0, ctx->stream, pa);           \
    } while (0)
            if (ctx->has_f64) {
                if (e && skip_dist) PCT_PAIR_TREE(true, false, true);
                else if (e) PCT_PAIR_TREE(true, true, true);
                else if (skip_dist) PCT_PAIR_TREE(false, false, true);
                else PCT_PAIR_TREE(false, true, true);
            } else if (e && skip_dist) PCT_PAIR_TREE(true, false, false);
            else if (e) PCT_PAIR_TREE(true, true, false);
            else if (skip_dist) PCT_PAIR_TREE(false, false, false);
            else PCT_PAIR_TREE(false, true, false);
#undef PCT_PAIR_TREE
        } else
        if (ctx->has_f64) {
            if (r1 && !e) PCT_TREE(1, false, true, grid1, block1);
            else if (r1) PCT_TREE(1, true, true, grid1, block1);
            else if (!e) PCT_TREE(2, false, true, grid2, block2);
            else PCT_TREE(2, true, true, grid2, block2);
        } else if (r1 && !e) PCT_TREE(1, false, false, grid1, block1);
        else if (r1) PCT_TREE(1, true, false, grid1, block1);
        else if (!e) PCT_TREE(2, false, false, grid2, block2);
        else PCT_TREE(2, true, false, grid2, block2);
#undef PCT_TREE
        PCT_HIP(ctx, hipGetLastError());
    }
    PCT_HIP(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    const int blocks = (32768 + kWavesPerBlock - 1) / kWavesPerBlock;        // device-side count, fixed grid
    const int* list = exact_only ? nullptr : redo;
    if (k + 1 <= 64)
        PCT_LAUNCH(k_knn_exact_tree<1>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, ctx->stream, a, list, (const int*)redo_count);
    else
        PCT_LAUNCH(k_knn_exact_tree<2>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, ctx->stream, a, list, (const int*)redo_count);
    PCT_HIP(ctx, hipGetLastError());
    ctx->knn_sorted_space = true;
    return PCT_OK;
}

int pct_item_census(pct_ctx* ctx, int32_t k, unsigned long long out4[4]) {
    PCT_TRY(pct_reserve(ctx, &ctx->counters, 64 * sizeof(unsigned long long)));
    PCT_HIP(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
    ctx->counters_clean = false;
    const int cap = k + 1 <= pct_fast_r1_max() ? kStageCap : PCT_STAGE_CAP2_HOST;
    const int blocks = (int)((ctx->n_items + 255) / 256 < 1024 ? (ctx->n_items + 255) / 256 : 1024);
    if (blocks > 0) {
        PCT_LAUNCH(k_item_census, dim3(blocks), dim3(256), 0, ctx->stream, (const int2*)ctx->occ.p, ctx->n_items, ctx->items_q,
                           (const int*)ctx->cell_cnt.p, (const int*)ctx->cell_own.p, ctx->grid, k, cap, (unsigned long long*)ctx->counters.p);
        PCT_HIP(ctx, hipGetLastError());
    }
    PCT_HIP(ctx, hipMemcpyAsync(ctx->pin + 2112, ctx->counters.p, 32, hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out4, ctx->pin + 2112, 32);
    return PCT_OK;
}

int pct_launch_knn_brute(pct_ctx* ctx, int32_t k, double eps) {
    PCT_TRY(reserve_table(ctx, k, eps));
    KnnArgs a = make_args(ctx, k, eps, false);
    const int64_t nq = ctx->q_end - ctx->q_begin;
    const int blocks = (int)((nq + kWavesPerBlock - 1) / kWavesPerBlock);
    if (blocks > 0) {
        if (k + 1 <= 64)
            PCT_LAUNCH(k_knn_brute<1>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, ctx->stream, a);
        else if (k + 1 <= 128)
            PCT_LAUNCH(k_knn_brute<2>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, ctx->stream, a);
        else
            PCT_TRY(pct_launch_knn_brute_wide(ctx, a, blocks));
        PCT_HIP(ctx, hipGetLastError());
    }
    ctx->knn_sorted_space = false;
    return PCT_OK;
}

int pct_launch_export_neighbors(pct_ctx* ctx, int64_t begin, int64_t end, int32_t* d_idx, float* d_dist,
                                int32_t* d_cnt) {
    const bool sorted = ctx->knn_sorted_space;
    const int64_t n_rows = ctx->q_end - ctx->q_begin;
    const int64_t total = n_rows * ctx->k;
    const int blocks = (int)((total + 255) / 256);
    PCT_LAUNCH(k_export, dim3(blocks), dim3(256), 0, ctx->stream,
                       (const float4*)(sorted ? ctx->sorted4.p : ctx->pts4.p), sorted && ctx->has_f64 ? (const double4*)ctx->sorted4d.p : nullptr,
                       sorted ? (const int*)ctx->owned_pos.p : nullptr,
                       (int)ctx->q_begin, (const int*)ctx->nbr_pos.p, ctx->dist_valid ? (const float*)ctx->nbr_dist.p : nullptr,
                       ctx->eps > 0 ? (const int*)ctx->nbr_cnt.p : nullptr, ctx->n, n_rows, ctx->k, ctx->nbr_pitch, begin, end,
                       d_idx, d_dist, d_cnt);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

int pct_launch_export_rows(pct_ctx* ctx, const int64_t* d_rows, int64_t n_rows, int32_t* d_idx, float* d_dist, int32_t* d_cnt) {
    const bool sorted = ctx->knn_sorted_space;
    PCT_LAUNCH(k_export_rows, dim3((unsigned)n_rows), dim3(128), 0, ctx->stream,
                       (const float4*)(sorted ? ctx->sorted4.p : ctx->pts4.p), sorted && ctx->has_f64 ? (const double4*)ctx->sorted4d.p : nullptr,
                       sorted ? (const int*)ctx->row_of.p : nullptr,
                       sorted ? (const int*)ctx->owned_pos.p : nullptr,
                       (int)ctx->q_begin, (const int*)ctx->nbr_pos.p, ctx->dist_valid ? (const float*)ctx->nbr_dist.p : nullptr,
                       ctx->eps > 0 ? (const int*)ctx->nbr_cnt.p : nullptr, ctx->n, ctx->k, ctx->nbr_pitch, d_rows, d_idx,
                       d_dist, d_cnt);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

// {x, y, z, index} records of every point in public order, kept apart from the sweep's own arrays (a side query or
// a diagnostics fit must not disturb a resident table)
int pct_ensure_plain_records(pct_ctx* ctx) {
    if (!ctx->qpts4_valid) {
        PCT_TRY(pct_reserve(ctx, &ctx->qpts4, (size_t)ctx->n * sizeof(float4)));
        PCT_LAUNCH(k_plain_records, dim3((unsigned)((ctx->n + 255) / 256)), dim3(256), 0, ctx->stream,
                           ctx->xyz_view, ctx->n, (float4*)ctx->qpts4.p);
        PCT_HIP(ctx, hipGetLastError());
        ctx->qpts4_valid = true;
    }
    return PCT_OK;
}

int pct_launch_query_points(pct_ctx* ctx, const double* d_q, int64_t m, int32_t k, double eps, int32_t* d_idx, double* d_dist) {
    PCT_TRY(pct_ensure_plain_records(ctx));
    const double eps2 = eps > 0 ? eps * eps : (double)INFINITY;
    const int blocks = (int)((m + kWavesPerBlock - 1) / kWavesPerBlock);
    if (k <= 64)
        PCT_LAUNCH(k_query_points<1>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, ctx->stream,
                           (const float4*)ctx->qpts4.p, (int)ctx->n, d_q, m, k, eps2, d_idx, d_dist);
    else
        PCT_LAUNCH(k_query_points<2>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, ctx->stream,
                           (const float4*)ctx->qpts4.p, (int)ctx->n, d_q, m, k, eps2, d_idx, d_dist);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

int pct_launch_selftest(pct_ctx* ctx, int* d_fails) {
    PCT_LAUNCH(k_selftest, dim3(1), dim3(64), 0, ctx->stream, d_fails);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}
