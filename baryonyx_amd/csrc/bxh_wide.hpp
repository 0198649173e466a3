// baryonyx_amd/csrc/bxh_wide.hpp -- the row update for instances whose columns
// do not fit the staged kernel of bxh_kernels.hpp: columns of any length
// (thousands of entries), any number of rows (pi does not fit LDS), rows of
// tens of thousands of entries -- the shape of BASELINE.json configs[3]
// (synthetic SCP 100k x 1M at 2 %).
//
// Same algorithm, same order of every floating-point sum as the other kernels
// (compute_update_row / push_and_compute_update_row of the 01 and 101 solvers,
// itm-solver-inequalities-01.cpp:208-296, -equalities-01.cpp:184-270,
// -equalities-101.cpp:208-307, -inequalities-101.cpp:250-347; affect,
// itm-common.hpp:382-467), so the results are bit-identical to the oracle's.
//
// One row update here is a stream, not a latency chain: the r columns of the
// row hold sum(c_j) entries (4e7 at the XL shape: 480 MB of P and row
// indices), so G workgroups of an instance split the row's columns (any
// workgroup can take any column: x and P live in global memory, pi in a private
// copy per workgroup), every wave streams 32 columns at a time in 16-entry
// pieces through its own LDS slots (coalesced 16-byte loads: a full 128-byte
// line of P per column and instruction), and one lane per column and
// accumulator adds the staged entries strictly in CSC order.  Per row the
// workgroups meet twice: once to exchange an 8-word record of their part of
// the row (class counts, extreme keys, tie flag: everything select_variables
// and affect need when no two reduced costs are equal), once when their P
// stores of affect() have landed.  A row with a tie (or an interior rank) is
// published in full and sorted by every workgroup alike.
#pragma once

#include "bxh_kernels.hpp"

#include <type_traits>

namespace bxh {

// -DBXH_WPROF (diagnostic builds only, tools/debug/wprof_c4.py): shader-clock ticks of the phases
// of a long tied row, accumulated by thread 0 of an instance's first workgroup in st.prof[inst][0..15]
#ifdef BXH_WPROF
#define BXH_WSTAMP(k_)                                                         \
    do {                                                                       \
        if (tid == 0 && grp == 0 && st.prof) {                                 \
            const unsigned long long now_ = (unsigned long long)clock64();    \
            st.prof[(size_t)inst * 32 + (k_)] += now_ - wprof_t;               \
            wprof_t = now_;                                                    \
        }                                                                      \
    } while (0)
#define BXH_RSTAMP(k_)                                                         \
    do {                                                                       \
        if (tid == 0 && grp == 0 && st.prof) {                                 \
            const unsigned long long now_ = (unsigned long long)clock64();    \
            st.prof[(size_t)inst * 32 + (k_)] += now_ - rprof_t;               \
            rprof_t = now_;                                                    \
        }                                                                      \
    } while (0)
#else
#define BXH_WSTAMP(k_)                                                         \
    do {                                                                       \
    } while (0)
#define BXH_RSTAMP(k_)                                                         \
    do {                                                                       \
    } while (0)
#endif
// MinStd::shuffle at this kernel's call sites: inlined (the nondeterminism met in
// round 5); -DBXH_SHUFFLE_CALL brings the calls back (tools/debug/hunt.sh: the control build)
#ifdef BXH_SHUFFLE_CALL
#define BXH_WIDE_SHUFFLE shuffle
#else
#define BXH_WIDE_SHUFFLE shuffle_inl
#endif
constexpr int WIDE_T = 512;      // threads of a workgroup
constexpr int WIDE_TILE = 32;    // columns a wave streams together
constexpr int WIDE_SMALL = 1024; // rows up to this length are sorted in LDS
constexpr int WIDE_LCG_POW = 32770; // entries of DevProblem::lcg_pow (bxh_create uploads that many)
constexpr int WIDE_ISLOT = 80;   // bytes of an index slot (16 int32 + pad)
constexpr int WIDE_TKEYS = 2048; // tied keys of a row the fast path takes (more: the row is sorted)
constexpr int WIDE_TRUN = 1024;  // longest run the fast path splits at the selected position
constexpr int WIDE_TDL = 64 + WIDE_TKEYS; // ints of WideParams::wdl per instance

template<typename Float>
struct WideGeom;
template<>
struct WideGeom<double>
{
    static constexpr int PL = 4;      // P load instructions per piece round
    static constexpr int PCOLS = 8;   // columns one of them covers
    static constexpr int PPARTS = 8;  // 16-byte parts of a 16-entry piece
    static constexpr int PSLOT = 144; // bytes of a P slot (16 entries + pad)
};
template<>
struct WideGeom<float>
{
    static constexpr int PL = 2;
    static constexpr int PCOLS = 16;
    static constexpr int PPARTS = 4;
    static constexpr int PSLOT = 80;
};

struct WideParams
{
    int G;                     // workgroups per instance
    int chmax;                 // longest part of a row one workgroup takes (multiple of 32)
    unsigned hmask;            // slots of the tie table - 1
    int n2big;                 // next_pow2(rmax) when rmax > WIDE_SMALL, else 0
    int ring;                  // pi entries of the LDS window (a power of two)
    float rows_per_piece;      // m / (mean 16-entry pieces of a column): the window's pace
    const int2* col_desc;      // [n] {first padded CSC slot, padded length}
    void* piw;                 // [batch][G][m+1] Float: pi, one copy per workgroup
    unsigned long long* whash; // [batch][hmask+1] keys of the row (0 = empty)
    unsigned long long* wrec;  // [batch][G][8] the workgroups' records of the row
    unsigned* wcnt;            // [batch][4] arrival counters, then [1] error word
    unsigned long long* vbuf;  // [batch][rmax] reduced costs of a row that is sorted
    unsigned long long* bkey;  // [batch][G][n2big] sort scratch of long rows
    int* bid;                  // [batch][G + 2][n2big] (copy G: sorted chunks, G + 1: the sorted row; shared)
    int* vflag32;              // [batch][m] violated flags of the end-of-sweep scan
    unsigned* xt;              // [n] x of all instances, one bit per instance (batch <= 32), or null
    int own_sort;              // 1: a long tied row is sorted by every workgroup on its own (BXH_WIDE_OWN_SORT)
    int skip_scan;             // 1: the sweep kernel leaves compute_violated_constraints to
                               //    wide_flags_all_kernel (one pass over A for all instances)
    unsigned* wcount;          // [batch][hmask+1] duplicates of the key in that slot of whash (0: none)
    int* wdl;                  // [batch][WIDE_TDL] the row's tied keys: [0] how many, [1] flags,
                               //    [8 + g] members of the split run in workgroup g, [64 + i] slot of tied key i
    int tied_fast;             // 1: a tied row whose selection needs no sorted order takes the fast path
};

__host__ __device__ inline int
wide_stage_bytes(int fsize)
{
    return WIDE_TILE * ((fsize == 8 ? 144 : 80) + WIDE_ISLOT);
}

struct WideLds
{
    int off_misc, off_part, off_rec, off_val, off_pown, off_hslot, off_union, union_bytes, off_ring,
      total;
};

__host__ __device__ inline WideLds
wide_lds(int G, int chmax, int fsize, int ring)
{
    auto al = [](int v) { return (v + 15) & ~15; };
    WideLds p;
    int o = 0;
    // first: the LDS-DMA refill addresses the ring through M0
    p.off_ring = o; // pi of the rows [lo, lo + ring) by row % ring, + one zero entry (padding)
    o = al(o + (ring + 1) * fsize);
    p.off_misc = o;
    o = al(o + 128 * 4);
    p.off_part = o;
    o = al(o + 80 * 8);
    p.off_rec = o;
    o = al(o + G * 8 * 8);
    p.off_val = o;
    o = al(o + chmax * fsize);
    p.off_pown = o;
    o = al(o + chmax * fsize);
    p.off_hslot = o;
    o = al(o + chmax * 4);
    p.off_union = o;
    p.union_bytes = (WIDE_T / 64) * wide_stage_bytes(fsize);
    o = al(o + p.union_bytes);
    p.total = o;
    return p;
}

// ---------------------------------------------------------------------------
// The sort of a long row (more than WIDE_SMALL elements) with ties.
//
// bitonic_sort() over (key, id) pairs in global memory costs a barrier and a
// round of global loads / stores per compare-exchange step (120 steps for 32768
// pairs).  Here every step whose partner distance is below the chunk length C
// runs on a chunk held in LDS (keys in `lk`, ids in `li`, C pairs): the pairs go
// through global memory only for the few steps with a longer distance.  Same
// network, same result: the unique order "by key, then by row-local id".
__device__ inline void
wide_sort_big(unsigned long long* gkey, int* gid, int n2, unsigned long long* lk, int* li, int C)
{
    const int tid = threadIdx.x, T = blockDim.x;
    auto cmpx = [&](unsigned long long& ka, int& ia, unsigned long long& kb, int& ib, bool asc) {
        const bool gt = (ka > kb) || (ka == kb && ia > ib);
        if (gt == asc) {
            const unsigned long long tk = ka;
            ka = kb;
            kb = tk;
            const int ti = ia;
            ia = ib;
            ib = ti;
        }
    };
    // steps j = jtop .. 1 of stage k on the chunk that starts at pair c0 (in LDS)
    auto local_steps = [&](int c0, int k, int jtop) {
        for (int j = jtop; j > 0; j >>= 1) {
            for (int t = tid; t < (C >> 1); t += T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool asc = ((c0 + i) & k) == 0;
                unsigned long long ka = lk[i], kb = lk[l];
                int ia = li[i], ib = li[l];
                const unsigned long long ka0 = ka;
                const int ia0 = ia;
                cmpx(ka, ia, kb, ib, asc);
                if (ka != ka0 || ia != ia0) {
                    lk[i] = ka;
                    li[i] = ia;
                    lk[l] = kb;
                    li[l] = ib;
                }
            }
            __syncthreads();
        }
    };
    auto load_chunk = [&](int c0) {
        for (int i = tid; i < C; i += T) {
            lk[i] = gkey[c0 + i];
            li[i] = gid[c0 + i];
        }
        __syncthreads();
    };
    auto store_chunk = [&](int c0) {
        for (int i = tid; i < C; i += T) {
            gkey[c0 + i] = lk[i];
            gid[c0 + i] = li[i];
        }
        __syncthreads();
    };
    // stages k <= C: whole chunks
    for (int c0 = 0; c0 < n2; c0 += C) {
        load_chunk(c0);
        for (int k = 2; k <= C; k <<= 1)
            local_steps(c0, k, k >> 1);
        store_chunk(c0);
    }
    // stages k > C: the long distances in global memory, the rest in LDS again
    for (int k = C << 1; k <= n2; k <<= 1) {
        for (int j = k >> 1; j >= C; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool asc = (i & k) == 0;
                unsigned long long ka = gkey[i], kb = gkey[l];
                int ia = gid[i], ib = gid[l];
                const unsigned long long ka0 = ka;
                const int ia0 = ia;
                cmpx(ka, ia, kb, ib, asc);
                if (ka != ka0 || ia != ia0) {
                    gkey[i] = ka;
                    gid[i] = ia;
                    gkey[l] = kb;
                    gid[l] = ib;
                }
            }
            __syncthreads();
        }
        for (int c0 = 0; c0 < n2; c0 += C) {
            load_chunk(c0);
            local_steps(c0, k, C >> 1);
            store_chunk(c0);
        }
    }
}

__device__ inline bool
wide_sync(unsigned* cnt, unsigned target, unsigned* err, int* dead_s);

// The same order ("by key, then by row-local id") worked out by the G
// workgroups of the instance together, every one a share: the row is cut into
// chunks of C pairs (a power of two; chunk c is workgroup c % G's), each chunk is
// sorted in LDS and published (ckey / cid, shared by the instance); then every
// element's final position is its position in its own chunk plus, for every
// other chunk -- staged in LDS in turn -- the number of pairs below it (a binary
// search: the pairs are distinct), and it is written there (fkey / fid, shared).
// Every workgroup then takes a copy of the ids (oid: the shuffle of the tied runs
// permutes it); the sorted keys stay where they are (fkey) and are read through L2.  Two meetings of the instance's workgroups at the
// slow-row counter.  Returns false when a partner was lost.
template<typename Float>
__device__ inline bool
wide_sort_coop(const unsigned long long* vbuf, int r, int n2, bool maximize, int grp, int G,
               unsigned long long* ckey, int* cid, unsigned long long* fkey, int* fid,
               int* oid, unsigned long long* lk, int* li, int C,
               unsigned* cnt_sl, unsigned& nslow, unsigned* err, int* dead_s,
               unsigned long long* wprof = nullptr)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const int NC = n2 / C;
    constexpr int MAXU = 4; // C <= 4 * T
#ifdef BXH_WPROF
    unsigned long long wt_ = (unsigned long long)clock64();
#define BXH_WSORT_STAMP(k_)                                                    \
    do {                                                                       \
        if (wprof) {                                                           \
            const unsigned long long now_ = (unsigned long long)clock64();    \
            wprof[(k_)] += now_ - wt_;                                         \
            wt_ = now_;                                                        \
        }                                                                      \
    } while (0)
#else
#define BXH_WSORT_STAMP(k_)                                                    \
    do {                                                                       \
    } while (0)
#endif
    auto ld64 = [](const unsigned long long* p_) -> unsigned long long {
        return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto ld32 = [](const int* p_) -> int {
        return __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // ---- own chunks: sorted in LDS, published ----------------------------------
    for (int c = grp; c < NC; c += G) {
        const int c0 = c * C;
        for (int t = tid; t < C; t += T) {
            const int i = c0 + t;
            unsigned long long key = ~0ull;
            if (i < r)
                key = order_key<Float>((Float)__longlong_as_double((long long)ld64(&vbuf[i])), maximize);
            lk[t] = key;
            li[t] = i;
        }
        __syncthreads();
        for (int k = 2; k <= C; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (C >> 1); t += T) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int l = i | j;
                    const bool asc = (i & k) == 0;
                    const unsigned long long ka = lk[i], kb = lk[l];
                    const int ia = li[i], ib = li[l];
                    const bool gt = (ka > kb) || (ka == kb && ia > ib);
                    if (gt == asc) {
                        lk[i] = kb;
                        li[i] = ib;
                        lk[l] = ka;
                        li[l] = ia;
                    }
                }
                // pair t = ... 64 w + lane works on the positions 128 w ... 128 w + 127 at every
                // distance up to 64: the steps j = 64 ... 1 of a stage touch nothing another wave
                // touches, and a wave's LDS accesses complete in order
                if (j > 64 || j == 1)
                    __syncthreads();
                else {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
        for (int t = tid; t < C; t += T) {
            __hip_atomic_store(&ckey[c0 + t], lk[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&cid[c0 + t], li[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    BXH_WSORT_STAMP(8); // own chunks sorted and published
    if (!wide_sync(cnt_sl, (unsigned)G * (nslow + 1u), err, dead_s))
        return false;
    ++nslow;
    BXH_WSORT_STAMP(9); // first meeting
    // ---- final positions of the own chunks' pairs ---------------------------------
    for (int c = grp; c < NC; c += G) {
        const int c0 = c * C;
        unsigned long long mk[MAXU];
        int mi[MAXU], rank[MAXU];
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
            const int t = tid + u * T;
            mk[u] = ~0ull;
            mi[u] = 0x7fffffff;
            rank[u] = t;
            if (t < C) {
                mk[u] = ld64(&ckey[c0 + t]);
                mi[u] = ld32(&cid[c0 + t]);
            }
        }
        // (chunk x holds the row-local ids x * C ... x * C + C - 1: all of them below this
        // chunk's when x < c, all above when x > c -- so "pairs below (key, id)" is "keys <= key"
        // in a chunk in front and "keys < key" in one behind: the search needs the keys only)
        for (int x = 0; x < NC; ++x) {
            if (x == c)
                continue;
            __syncthreads(); // (the chunk staged before has been searched by every thread)
            for (int t = tid; t < C; t += T)
                lk[t] = ld64(&ckey[x * C + t]);
            __syncthreads();
            const bool front = x < c;
            int pos[MAXU];
#pragma unroll
            for (int u = 0; u < MAXU; ++u)
                pos[u] = 0;
            for (int step = C >> 1; step > 0; step >>= 1) {
#pragma unroll
                for (int u = 0; u < MAXU; ++u) {
                    const unsigned long long kq = lk[pos[u] + step - 1];
                    if (kq < mk[u] || (front && kq == mk[u]))
                        pos[u] += step;
                }
            }
#pragma unroll
            for (int u = 0; u < MAXU; ++u) {
                const unsigned long long kq = lk[pos[u]];
                if (kq < mk[u] || (front && kq == mk[u]))
                    pos[u] += 1;
                rank[u] += pos[u];
            }
        }
#pragma unroll
        for (int u = 0; u < MAXU; ++u)
            if (tid + u * T < C) {
                __hip_atomic_store(&fkey[rank[u]], mk[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&fid[rank[u]], mi[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    }
    BXH_WSORT_STAMP(10); // final positions
    if (!wide_sync(cnt_sl, (unsigned)G * (nslow + 1u), err, dead_s))
        return false;
    ++nslow;
    BXH_WSORT_STAMP(11); // second meeting
    // ---- a copy of the sorted row for this workgroup --------------------------------
    // (the ids only, and the row's own r of them: the tied runs' shuffles permute them.  The keys
    // are read where they are, through L2 -- the caller's bkey is fkey.  The shared arrays are
    // written again by the next long row that is sorted, behind that sort's first meeting: every
    // workgroup is through with this row by then)
    for (int i = tid; i < r; i += T)
        oid[i] = ld32(&fid[i]);
    __syncthreads();
    BXH_WSORT_STAMP(12); // the copy
    return true;
}

// std::shuffle(a, a + len, g) by ONE lane, g = minstd_rand0 in state `state`, for len <= 46340
// (libstdc++'s two-swaps-per-draw branch, <bits/stl_algo.h>; uniform_int_distribution's
// down-scaling with rejection, <bits/uniform_int_dist.h>): MinStd::shuffle in 32-bit
// arithmetic -- every range here is below 2^31, and a 64-bit division costs a lane ten times a
// 32-bit one.  Returns the engine's state behind the shuffle.
__device__ inline unsigned
lane_run_shuffle(int* a, unsigned len, unsigned state)
{
    if (len < 2)
        return state;
    auto draw = [&](unsigned urange) -> unsigned { // uniform_int_distribution{0, urange}(g)
        const unsigned uerange = urange + 1u;
        const unsigned scaling = 2147483645u / uerange;
        const unsigned past = uerange * scaling;
        unsigned ret;
        do {
            state = (unsigned)(((unsigned long long)state * 16807ull) % 2147483647ull);
            ret = state - 1u;
        } while (ret >= past);
        return ret / scaling;
    };
    auto swp = [&](unsigned i, unsigned j) {
        const int t = a[i];
        a[i] = a[j];
        a[j] = t;
    };
    unsigned i = 1;
    if ((len % 2u) == 0u) {
        swp(i, draw(1u));
        ++i;
    }
    while (i != len) {
        const unsigned b1 = i + 2u;
        const unsigned x = draw((i + 1u) * b1 - 1u);
        swp(i, x / b1);
        ++i;
        swp(i, x % b1);
        ++i;
    }
    return state;
}

// std::shuffle of ONE run of equal keys (ids[0 .. L) in LDS) by all threads:
// the engine is a linear congruential one, so every position works out its own
// swap partner from draw number `1 + d` (libstdc++'s pairwise scheme, see
// parallel_tie_shuffle), and the swaps are applied by tracing every final
// position back through them.  Returns false (nothing changed) when a draw would
// have been rejected -- the caller then shuffles the run serially.  jt / out:
// L ints of LDS each.  *state is advanced by L / 2 draws.  Needs L / 2 + 1 <
// pow_n (entries of lcg_pow).
__device__ inline bool
wide_run_shuffle(int* ids, int L, unsigned* state, const unsigned* lcg_pow, int* jt, int* out,
                 int* flag_s)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const unsigned st0 = *state;
    const bool even = (L & 1) == 0;
    if (tid == 0)
        *flag_s = 0;
    __syncthreads();
    bool reject = false;
    for (int i = tid; i < L; i += T) {
        int jv = 0;
        if (i >= 1) {
            unsigned urange;
            int d, half = 0;
            unsigned b1 = 2;
            if (even && i == 1) {
                d = 0;
                urange = 1u;
            } else {
                const int i0 = even ? (i & ~1) : (((i - 1) & ~1) + 1); // first index of the pair
                d = even ? (i >> 1) : ((i - 1) >> 1);
                half = i - i0;
                b1 = (unsigned)i0 + 2u;
                urange = ((unsigned)i0 + 1u) * b1 - 1u;
            }
            const unsigned urngrange = 2147483645u;
            const unsigned uerange = urange + 1u;
            const unsigned scaling = urngrange / uerange;
            const unsigned past = uerange * scaling;
            const unsigned raw = lcg_jump(lcg_pow, st0, d + 1) - 1u;
            if (raw >= past)
                reject = true;
            const unsigned x = raw / scaling;
            jv = (even && i == 1) ? (int)x : (half == 0 ? (int)(x / b1) : (int)(x % b1));
        }
        jt[i] = jv;
    }
    if (reject)
        *flag_s = 1;
    __syncthreads();
    if (*flag_s)
        return false;
    for (int p = tid; p < L; p += T) {
        int pos = p;
        for (int i = L - 1; i >= 1; --i) {
            const int j = jt[i];
            pos = pos == i ? j : (pos == j ? i : pos);
        }
        out[p] = ids[pos];
    }
    __syncthreads();
    for (int p = tid; p < L; p += T)
        ids[p] = out[p];
    *state = lcg_jump(lcg_pow, st0, L >> 1);
    __syncthreads();
    return true;
}

// ---------------------------------------------------------------------------
// The streaming loads of the column sums.  They are asynchronous and must stay
// in flight across other code (a piece ahead), which no compiler-visible
// register may do: hipcc knows nothing of a load issued from asm and is free to
// copy its destination before the data has landed (it did: v_mov of all sixteen
// gather destinations in front of the wait).  So the loads land in ACCUMULATOR
// registers named literally -- on gfx950 AGPRs come out of the same file as the
// VGPRs, a kernel without MFMA never touches them otherwise, and listing them as
// clobbers makes the kernel descriptor allocate them -- and are moved on by the
// very asm statement that waits for them:
//   a[24 K : 24 K + 23]  staging loads of one 16-entry piece (K = piece % 3: three
//             pieces in flight per wave): P (16 registers fp64 / 8 fp32,
//             device-scope sc1 loads: P is written by the other workgroups of
//             the instance) and row indices (8 registers, read-only data),
//             ds_write_b128 straight from the AGPRs.
template<typename Float>
struct WideAsm;

template<>
struct WideAsm<double>
{
    __device__ static inline void issue0(const char* p0, const char* p1, const char* p2, const char* p3, const char* i0, const char* i1)
    {
        asm volatile("global_load_dwordx4 a[0:3], %0, off sc1\n"
                     "global_load_dwordx4 a[4:7], %1, off sc1\n"
                     "global_load_dwordx4 a[8:11], %2, off sc1\n"
                     "global_load_dwordx4 a[12:15], %3, off sc1\n"
                     "global_load_dwordx4 a[16:19], %4, off\n"
                     "global_load_dwordx4 a[20:23], %5, off\n"
                     :
                     : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(i0), "v"(i1)
                     : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
    }
    __device__ static inline void land0_w0(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(0)\n"
                     "ds_write_b128 %0, a[0:3]\n"
                     "ds_write_b128 %0, a[4:7] offset:1152\n"
                     "ds_write_b128 %0, a[8:11] offset:2304\n"
                     "ds_write_b128 %0, a[12:15] offset:3456\n"
                     "ds_write_b128 %1, a[16:19]\n"
                     "ds_write_b128 %1, a[20:23] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
    }
    __device__ static inline void land0_w6(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(6)\n"
                     "ds_write_b128 %0, a[0:3]\n"
                     "ds_write_b128 %0, a[4:7] offset:1152\n"
                     "ds_write_b128 %0, a[8:11] offset:2304\n"
                     "ds_write_b128 %0, a[12:15] offset:3456\n"
                     "ds_write_b128 %1, a[16:19]\n"
                     "ds_write_b128 %1, a[20:23] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
    }
    __device__ static inline void land0_w12(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(12)\n"
                     "ds_write_b128 %0, a[0:3]\n"
                     "ds_write_b128 %0, a[4:7] offset:1152\n"
                     "ds_write_b128 %0, a[8:11] offset:2304\n"
                     "ds_write_b128 %0, a[12:15] offset:3456\n"
                     "ds_write_b128 %1, a[16:19]\n"
                     "ds_write_b128 %1, a[20:23] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
    }
    __device__ static inline void issue1(const char* p0, const char* p1, const char* p2, const char* p3, const char* i0, const char* i1)
    {
        asm volatile("global_load_dwordx4 a[24:27], %0, off sc1\n"
                     "global_load_dwordx4 a[28:31], %1, off sc1\n"
                     "global_load_dwordx4 a[32:35], %2, off sc1\n"
                     "global_load_dwordx4 a[36:39], %3, off sc1\n"
                     "global_load_dwordx4 a[40:43], %4, off\n"
                     "global_load_dwordx4 a[44:47], %5, off\n"
                     :
                     : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(i0), "v"(i1)
                     : "memory", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    }
    __device__ static inline void land1_w0(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(0)\n"
                     "ds_write_b128 %0, a[24:27]\n"
                     "ds_write_b128 %0, a[28:31] offset:1152\n"
                     "ds_write_b128 %0, a[32:35] offset:2304\n"
                     "ds_write_b128 %0, a[36:39] offset:3456\n"
                     "ds_write_b128 %1, a[40:43]\n"
                     "ds_write_b128 %1, a[44:47] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    }
    __device__ static inline void land1_w6(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(6)\n"
                     "ds_write_b128 %0, a[24:27]\n"
                     "ds_write_b128 %0, a[28:31] offset:1152\n"
                     "ds_write_b128 %0, a[32:35] offset:2304\n"
                     "ds_write_b128 %0, a[36:39] offset:3456\n"
                     "ds_write_b128 %1, a[40:43]\n"
                     "ds_write_b128 %1, a[44:47] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    }
    __device__ static inline void land1_w12(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(12)\n"
                     "ds_write_b128 %0, a[24:27]\n"
                     "ds_write_b128 %0, a[28:31] offset:1152\n"
                     "ds_write_b128 %0, a[32:35] offset:2304\n"
                     "ds_write_b128 %0, a[36:39] offset:3456\n"
                     "ds_write_b128 %1, a[40:43]\n"
                     "ds_write_b128 %1, a[44:47] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    }
    __device__ static inline void issue2(const char* p0, const char* p1, const char* p2, const char* p3, const char* i0, const char* i1)
    {
        asm volatile("global_load_dwordx4 a[48:51], %0, off sc1\n"
                     "global_load_dwordx4 a[52:55], %1, off sc1\n"
                     "global_load_dwordx4 a[56:59], %2, off sc1\n"
                     "global_load_dwordx4 a[60:63], %3, off sc1\n"
                     "global_load_dwordx4 a[64:67], %4, off\n"
                     "global_load_dwordx4 a[68:71], %5, off\n"
                     :
                     : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(i0), "v"(i1)
                     : "memory", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71");
    }
    __device__ static inline void land2_w0(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(0)\n"
                     "ds_write_b128 %0, a[48:51]\n"
                     "ds_write_b128 %0, a[52:55] offset:1152\n"
                     "ds_write_b128 %0, a[56:59] offset:2304\n"
                     "ds_write_b128 %0, a[60:63] offset:3456\n"
                     "ds_write_b128 %1, a[64:67]\n"
                     "ds_write_b128 %1, a[68:71] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71");
    }
    __device__ static inline void land2_w6(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(6)\n"
                     "ds_write_b128 %0, a[48:51]\n"
                     "ds_write_b128 %0, a[52:55] offset:1152\n"
                     "ds_write_b128 %0, a[56:59] offset:2304\n"
                     "ds_write_b128 %0, a[60:63] offset:3456\n"
                     "ds_write_b128 %1, a[64:67]\n"
                     "ds_write_b128 %1, a[68:71] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71");
    }
    __device__ static inline void land2_w12(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(12)\n"
                     "ds_write_b128 %0, a[48:51]\n"
                     "ds_write_b128 %0, a[52:55] offset:1152\n"
                     "ds_write_b128 %0, a[56:59] offset:2304\n"
                     "ds_write_b128 %0, a[60:63] offset:3456\n"
                     "ds_write_b128 %1, a[64:67]\n"
                     "ds_write_b128 %1, a[68:71] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71");
    }
    // staging loads of a piece into register set K (= piece % 3)
    template<int K>
    __device__ static inline void issue(const char* p0, const char* p1, const char* p2, const char* p3,
                                        const char* i0, const char* i1)
    {
        if constexpr (K == 0)
            issue0(p0, p1, p2, p3, i0, i1);
        else if constexpr (K == 1)
            issue1(p0, p1, p2, p3, i0, i1);
        else
            issue2(p0, p1, p2, p3, i0, i1);
    }
    // the loads of set K have landed (`ahead` younger pieces may stay in flight:
    // vmcnt counts in issue order): into the wave's LDS slots
    template<int K>
    __device__ static inline void land(int ahead, unsigned lds_p, unsigned lds_i)
    {
        if constexpr (K == 0) {
            if (ahead >= 2)
                land0_w12(lds_p, lds_i);
            else if (ahead == 1)
                land0_w6(lds_p, lds_i);
            else
                land0_w0(lds_p, lds_i);
        }
        else if constexpr (K == 1) {
            if (ahead >= 2)
                land1_w12(lds_p, lds_i);
            else if (ahead == 1)
                land1_w6(lds_p, lds_i);
            else
                land1_w0(lds_p, lds_i);
        }
        else if constexpr (K == 2) {
            if (ahead >= 2)
                land2_w12(lds_p, lds_i);
            else if (ahead == 1)
                land2_w6(lds_p, lds_i);
            else
                land2_w0(lds_p, lds_i);
        }
    }
};

template<>
struct WideAsm<float>
{
    __device__ static inline void issue0(const char* p0, const char* p1, const char*, const char*, const char* i0, const char* i1)
    {
        asm volatile("global_load_dwordx4 a[0:3], %0, off sc1\n"
                     "global_load_dwordx4 a[4:7], %1, off sc1\n"
                     "global_load_dwordx4 a[16:19], %2, off\n"
                     "global_load_dwordx4 a[20:23], %3, off\n"
                     :
                     : "v"(p0), "v"(p1), "v"(i0), "v"(i1)
                     : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
    }
    __device__ static inline void land0_w0(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(0)\n"
                     "ds_write_b128 %0, a[0:3]\n"
                     "ds_write_b128 %0, a[4:7] offset:1280\n"
                     "ds_write_b128 %1, a[16:19]\n"
                     "ds_write_b128 %1, a[20:23] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
    }
    __device__ static inline void land0_w4(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(4)\n"
                     "ds_write_b128 %0, a[0:3]\n"
                     "ds_write_b128 %0, a[4:7] offset:1280\n"
                     "ds_write_b128 %1, a[16:19]\n"
                     "ds_write_b128 %1, a[20:23] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
    }
    __device__ static inline void land0_w8(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(8)\n"
                     "ds_write_b128 %0, a[0:3]\n"
                     "ds_write_b128 %0, a[4:7] offset:1280\n"
                     "ds_write_b128 %1, a[16:19]\n"
                     "ds_write_b128 %1, a[20:23] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
    }
    __device__ static inline void issue1(const char* p0, const char* p1, const char*, const char*, const char* i0, const char* i1)
    {
        asm volatile("global_load_dwordx4 a[24:27], %0, off sc1\n"
                     "global_load_dwordx4 a[28:31], %1, off sc1\n"
                     "global_load_dwordx4 a[40:43], %2, off\n"
                     "global_load_dwordx4 a[44:47], %3, off\n"
                     :
                     : "v"(p0), "v"(p1), "v"(i0), "v"(i1)
                     : "memory", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    }
    __device__ static inline void land1_w0(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(0)\n"
                     "ds_write_b128 %0, a[24:27]\n"
                     "ds_write_b128 %0, a[28:31] offset:1280\n"
                     "ds_write_b128 %1, a[40:43]\n"
                     "ds_write_b128 %1, a[44:47] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    }
    __device__ static inline void land1_w4(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(4)\n"
                     "ds_write_b128 %0, a[24:27]\n"
                     "ds_write_b128 %0, a[28:31] offset:1280\n"
                     "ds_write_b128 %1, a[40:43]\n"
                     "ds_write_b128 %1, a[44:47] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    }
    __device__ static inline void land1_w8(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(8)\n"
                     "ds_write_b128 %0, a[24:27]\n"
                     "ds_write_b128 %0, a[28:31] offset:1280\n"
                     "ds_write_b128 %1, a[40:43]\n"
                     "ds_write_b128 %1, a[44:47] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    }
    __device__ static inline void issue2(const char* p0, const char* p1, const char*, const char*, const char* i0, const char* i1)
    {
        asm volatile("global_load_dwordx4 a[48:51], %0, off sc1\n"
                     "global_load_dwordx4 a[52:55], %1, off sc1\n"
                     "global_load_dwordx4 a[64:67], %2, off\n"
                     "global_load_dwordx4 a[68:71], %3, off\n"
                     :
                     : "v"(p0), "v"(p1), "v"(i0), "v"(i1)
                     : "memory", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71");
    }
    __device__ static inline void land2_w0(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(0)\n"
                     "ds_write_b128 %0, a[48:51]\n"
                     "ds_write_b128 %0, a[52:55] offset:1280\n"
                     "ds_write_b128 %1, a[64:67]\n"
                     "ds_write_b128 %1, a[68:71] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71");
    }
    __device__ static inline void land2_w4(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(4)\n"
                     "ds_write_b128 %0, a[48:51]\n"
                     "ds_write_b128 %0, a[52:55] offset:1280\n"
                     "ds_write_b128 %1, a[64:67]\n"
                     "ds_write_b128 %1, a[68:71] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71");
    }
    __device__ static inline void land2_w8(unsigned lds_p, unsigned lds_i)
    {
        asm volatile("s_waitcnt vmcnt(8)\n"
                     "ds_write_b128 %0, a[48:51]\n"
                     "ds_write_b128 %0, a[52:55] offset:1280\n"
                     "ds_write_b128 %1, a[64:67]\n"
                     "ds_write_b128 %1, a[68:71] offset:1280\n"
                     :
                     : "v"(lds_p), "v"(lds_i)
                     : "memory", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71");
    }
    // staging loads of a piece into register set K (= piece % 3)
    template<int K>
    __device__ static inline void issue(const char* p0, const char* p1, const char* p2, const char* p3,
                                        const char* i0, const char* i1)
    {
        if constexpr (K == 0)
            issue0(p0, p1, nullptr, nullptr, i0, i1);
        else if constexpr (K == 1)
            issue1(p0, p1, nullptr, nullptr, i0, i1);
        else
            issue2(p0, p1, nullptr, nullptr, i0, i1);
        (void)p2;
        (void)p3;
    }
    // the loads of set K have landed (`ahead` younger pieces may stay in flight:
    // vmcnt counts in issue order): into the wave's LDS slots
    template<int K>
    __device__ static inline void land(int ahead, unsigned lds_p, unsigned lds_i)
    {
        if constexpr (K == 0) {
            if (ahead >= 2)
                land0_w8(lds_p, lds_i);
            else if (ahead == 1)
                land0_w4(lds_p, lds_i);
            else
                land0_w0(lds_p, lds_i);
        }
        else if constexpr (K == 1) {
            if (ahead >= 2)
                land1_w8(lds_p, lds_i);
            else if (ahead == 1)
                land1_w4(lds_p, lds_i);
            else
                land1_w0(lds_p, lds_i);
        }
        else if constexpr (K == 2) {
            if (ahead >= 2)
                land2_w8(lds_p, lds_i);
            else if (ahead == 1)
                land2_w4(lds_p, lds_i);
            else
                land2_w0(lds_p, lds_i);
        }
    }
};

// Insert a key of the row into the instance's table (open addressing, one
// 64-bit compare-and-swap per probe, executed at the memory side: coherent for
// workgroups on any XCD).  Returns true when an equal key is already there.
// *slot: where this key went (to clear it after the row), -1 if nowhere.
// *dup: the slot that holds the equal key, -1 unless one was found.
__device__ inline bool
wide_hash_insert(unsigned long long* tab, unsigned mask, unsigned long long key, int* slot, int* dup)
{
    *slot = -1;
    *dup = -1;
    if (key == 0ull) // (only a NaN maps to 0, the empty mark: let the sort path have the row)
        return true;
    unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 38) & mask;
    for (unsigned probes = 0; probes <= mask; ++probes) {
        const unsigned long long old = atomicCAS(&tab[h], 0ull, key);
        if (old == 0ull) {
            *slot = (int)h;
            return false;
        }
        if (old == key) {
            *dup = (int)h;
            return true;
        }
        h = (h + 1) & mask;
    }
    return true; // table full (cannot happen: 4 slots per row entry)
}

// minstd_rand0's draws of std::shuffle over `len` elements (2 <= len <= 46 340: the pairwise
// scheme) from `state`, without the swaps: the engine's state behind them.  See lane_run_shuffle.
__device__ inline unsigned
lane_run_dry(unsigned len, unsigned state)
{
    if (len < 2)
        return state;
    auto draw = [&](unsigned urange) {
        const unsigned uerange = urange + 1u;
        const unsigned scaling = 2147483645u / uerange;
        const unsigned past = uerange * scaling;
        unsigned ret;
        do {
            state = (unsigned)(((unsigned long long)state * 16807ull) % 2147483647ull);
            ret = state - 1u;
        } while (ret >= past);
    };
    unsigned i = 1;
    if ((len % 2u) == 0u) {
        draw(1u);
        ++i;
    }
    while (i != len) {
        draw((i + 1u) * (i + 2u) - 1u);
        i += 2u;
    }
    return state;
}

// Bitwise OR of `v` over the workgroup (__syncthreads_or is the OR of predicates: one bit).
// slot: a word of LDS scratch.  All threads, uniform.
__device__ inline int
wide_or_bits(int v, int* slot)
{
    if (threadIdx.x == 0)
        *slot = 0;
    __syncthreads();
    if (v)
        atomicOr(slot, v);
    __syncthreads();
    const int all = *slot;
    __syncthreads(); // (every thread has read it: the word may be written again)
    return all;
}

// slots of the workgroup's LDS scratch words (misc_s) beyond those of bxh_kernels.hpp (MS_*)
enum
{
    WM_DEAD = 96, // a partner timed out: leave the sweep
    WM_TNP = 97,  // longest column (in pieces) of this workgroup's part of the row
    WM_TAKE = 100, // long tied row: where thread 0's serial shuffle takes over (a run beyond LDS)
    WM_FLAG = 101  // ... a draw of a run's shuffle was rejected
};

// One round of "where do the shuffles of a row's tied runs stand": the runs qb ... nruns - 1
// (lengths rl_len[], in key order) start at the engine state `st_row`, a run of L elements takes
// L / 2 draws unless one is rejected.  rl_base[q] = draws in front of run q, counted from run
// qb; *total = draws of all of them; returns the first run with a rejected draw (nruns: none),
// or -1 when there are more draws than lcg_pow has powers (nothing is known then).  Draw g is
// thread g % T's: its engine state follows from the one T draws earlier by a constant
// multiplier.  A draw over a range of `uerange` values is rejected only above
// 2147483645 - uerange: with no run beyond 512 elements (has_long false) one comparison
// against the widest range clears all but one draw or so per row, and only those get the
// search for their run, their own range and the division.  All threads, uniform; cand: 256
// ints of LDS; misc_s: WM_TAKE, WM_FLAG, MS_WAVES + 32.
__device__ inline int
wide_first_rejection(const int* rl_len, int* rl_base, int qb, int nruns, bool has_long, unsigned st_row,
                     const unsigned* lcg_pow, int* cand, int* misc_s, int* total)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (wave == 0) { // a running sum of the draws
        int acc = 0;
        for (int q0 = qb; q0 < nruns; q0 += 64) {
            const int q = q0 + lane;
            const int v = q < nruns ? (rl_len[q] >> 1) : 0;
            int incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o);
                if (lane >= o)
                    incl += up;
            }
            if (q < nruns)
                rl_base[q] = acc + incl - v;
            acc += __shfl(incl, 63);
        }
        if (lane == 0) {
            misc_s[WM_TAKE] = nruns;   // first run with a rejected draw
            misc_s[WM_FLAG] = acc;     // draws of the runs from qb on
            misc_s[MS_WAVES + 32] = 0; // draws that need a closer look
        }
    }
    __syncthreads();
    const int total_draws = misc_s[WM_FLAG];
    *total = total_draws;
    if (total_draws + 1 >= WIDE_LCG_POW) {
        __syncthreads(); // (misc_s may be written again)
        return -1;
    }
    const unsigned aT = lcg_pow[T];
    unsigned sg = tid < total_draws ? lcg_jump(lcg_pow, st_row, tid + 1) : 1u;
    constexpr int CAND = 256;
    const unsigned widest = has_long ? 2147483645u : 513u * 514u;
    for (int g = tid; g < total_draws; g += T) {
        if (sg - 1u > 2147483645u - widest) {
            const int slot = atomicAdd(&misc_s[MS_WAVES + 32], 1);
            if (slot < CAND)
                cand[slot] = g;
        }
        unsigned long long x = (unsigned long long)aT * sg;
        x = (x & 0x7fffffffull) + (x >> 31);
        x = (x & 0x7fffffffull) + (x >> 31);
        sg = (unsigned)(x >= 0x7fffffffull ? x - 0x7fffffffull : x);
    }
    __syncthreads();
    const int ncand = misc_s[MS_WAVES + 32];
    const bool all = ncand > CAND; // (too many to list: every draw closely)
    for (int e = tid; e < (all ? total_draws : ncand); e += T) {
        const int g = all ? e : cand[e];
        int lo = qb, hi = nruns - 1; // last run whose first draw is <= g
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rl_base[mid] <= g)
                lo = mid;
            else
                hi = mid - 1;
        }
        const int q = lo;
        const int L = rl_len[q], d = g - rl_base[q];
        const bool even = (L & 1) == 0;
        unsigned urange;
        if (even && d == 0)
            urange = 1u;
        else {
            const unsigned i0 = even ? 2u * (unsigned)d : 2u * (unsigned)d + 1u;
            urange = (i0 + 1u) * (i0 + 2u) - 1u;
        }
        const unsigned uerange = urange + 1u;
        const unsigned raw = lcg_jump(lcg_pow, st_row, g + 1) - 1u;
        const unsigned scaling = 2147483645u / uerange;
        if (raw >= uerange * scaling)
            atomicMin(&misc_s[WM_TAKE], q);
    }
    __syncthreads();
    const int first_rej = misc_s[WM_TAKE];
    __syncthreads(); // (every thread has read it: misc_s may be written again)
    return first_rej;
}

// The engine's way through the shuffles of a row's tied runs when only their lengths are known
// (rl_len[0 .. nruns), in key order, none beyond 46 340): rl_state[q] = the state in front of
// run q, *state_out = the state behind the last.  False when it cannot be told (see above).
__device__ inline bool
wide_runs_dry(const int* rl_len, int* rl_base, unsigned* rl_state, int nruns, bool has_long, unsigned st0,
              const unsigned* lcg_pow, int* cand, int* misc_s, unsigned* state_out)
{
    const int tid = threadIdx.x, T = blockDim.x;
    int qb = 0;
    unsigned st_row = st0;
    while (qb < nruns) {
        int total_draws = 0;
        const int first_rej =
          wide_first_rejection(rl_len, rl_base, qb, nruns, has_long, st_row, lcg_pow, cand, misc_s, &total_draws);
        if (first_rej < 0)
            return false;
        for (int q = qb + tid; q < nruns && q <= first_rej; q += T)
            rl_state[q] = lcg_jump(lcg_pow, st_row, rl_base[q]);
        if (first_rej >= nruns) {
            st_row = lcg_jump(lcg_pow, st_row, total_draws);
            qb = nruns;
        } else {
            // the run with the rejected draw: its draws one by one
            if (tid == 0)
                misc_s[MS_RNG] = (int)lane_run_dry((unsigned)rl_len[first_rej],
                                                   lcg_jump(lcg_pow, st_row, rl_base[first_rej]));
            __syncthreads();
            st_row = (unsigned)misc_s[MS_RNG];
            qb = first_rej + 1;
            __syncthreads();
        }
    }
    __syncthreads();
    *state_out = st_row;
    return true;
}

// Arrival of this workgroup at one of the instance's counters and wait for
// all G of them: every wave first waits for its own stores, one lane adds and
// polls (bounded: a missing partner sets the error word), the others wait at
// the barrier.  Returns false when the wait was given up.
__device__ inline bool
wide_sync(unsigned* cnt, unsigned target, unsigned* err, int* dead_s)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023) == 0 &&
                (spins > 8000000L ||
                 __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *dead_s = 1;
                break;
            }
        }
    }
    __syncthreads();
    return *dead_s == 0;
}

// One sweep of every instance: compute_order::run / push_and_run
// (itm-common.hpp:682-880) for the orders none / reversing / random_sorting.
template<typename Float, bool PM1>
__global__ __launch_bounds__(WIDE_T) void
sweep_kernel_wide(DevProblem pb, DevState st, DevSweep a, WideParams wp)
{
#if defined(BXH_DEV_ONLY_NP) && !defined(BXH_DEV_KEEP_WIDE)
    return; // (development builds of another kernel: see sweep_kernel_v2)
#endif

    using Geo = WideGeom<Float>;
    extern __shared__ __align__(16) unsigned char smem[];
    const int G = wp.G;
    int inst, grp;
    {
        const int b = blockIdx.x, ninst = gridDim.x / G;
        if ((ninst & 7) == 0) { // partners on one XCD under round-robin dispatch (speed only)
            inst = (b & 7) + 8 * ((b >> 3) / G);
            grp = (b >> 3) % G;
        } else {
            inst = b / G;
            grp = b % G;
        }
    }
    if (a.active && !a.active[inst])
        return;

    const int tid = threadIdx.x;
    constexpr int T = WIDE_T;
    const int lane = tid & 63, wave = tid >> 6;
    const WideLds lp = wide_lds(G, wp.chmax, (int)sizeof(Float), wp.ring);
#ifdef BXH_LDS_POISON // (diagnostic, tools/debug/hunt.sh: whatever reads LDS before writing it gets this pattern)
    for (int i = tid; i < lp.total / 4; i += T)
        ((unsigned*)smem)[i] = BXH_LDS_POISON;
    __syncthreads();
#endif
    Float* ring_s = (Float*)(smem + lp.off_ring);
    int* misc_s = (int*)(smem + lp.off_misc);
    Float* d_s = (Float*)(misc_s + MS_D);
    unsigned long long* part_s = (unsigned long long*)(smem + lp.off_part);
    unsigned long long* rec_s = (unsigned long long*)(smem + lp.off_rec);
    Float* val_s = (Float*)(smem + lp.off_val);
    Float* pown_s = (Float*)(smem + lp.off_pown);
    int* hslot_s = (int*)(smem + lp.off_hslot);
    unsigned char* uni = smem + lp.off_union;
    // sort path of short rows: the staging slots are idle then
    Float* val_all = (Float*)uni;
    const int n2s = WIDE_SMALL;
    unsigned long long* key_s = (unsigned long long*)(uni + WIDE_SMALL * sizeof(Float));
    int* id_s = (int*)((unsigned char*)key_s + n2s * 8);
    unsigned long long* key_run = (unsigned long long*)((unsigned char*)id_s + n2s * 4);
    int* id_run = (int*)((unsigned char*)key_run + n2s * 8);

    const bool maximize = pb.maximize != 0;
    const int kind = pb.kind;
    const int m = pb.m;

    Float* P = (Float*)st.P + (size_t)inst * (size_t)pb.nnzp;
    Float* pig = (Float*)st.pi + (size_t)inst * m;
    Float* piw = (Float*)wp.piw + ((size_t)inst * G + grp) * (size_t)(m + 1);
    unsigned* xg = st.x + (size_t)inst * (2 * pb.nwords);
    Float* staleg = (Float*)st.rstale + (size_t)inst * (pb.rmax + 1);
    unsigned long long* htab = wp.whash + (size_t)inst * ((size_t)wp.hmask + 1);
    unsigned* hcount = wp.wcount + (size_t)inst * ((size_t)wp.hmask + 1);
    int* tdl = wp.wdl + (size_t)inst * WIDE_TDL;
    unsigned long long* recg = wp.wrec + (size_t)inst * G * 8;
    unsigned long long* vbuf = wp.vbuf + (size_t)inst * pb.rmax;
    unsigned* cnt_s1 = wp.wcnt + (size_t)inst * 4;
    unsigned* cnt_s2 = cnt_s1 + 1;
    unsigned* cnt_sl = cnt_s1 + 2;
    unsigned* cnt_end = cnt_s1 + 3;
    unsigned* err_word = wp.wcnt + (size_t)(gridDim.x / G) * 4;
    const Float* cost = (const Float*)pb.cost;

    const int ord = order_of(a, inst);
    int* vlist = st.vlist + (size_t)inst * st.lcap;
    int* vrows = st.vwork + ((size_t)inst * G + grp) * st.lcap;

    const Float kappa = (Float)a.kappa[inst];
    const Float kfac = kappa / ((Float)1 - kappa);
    const Float delta = (Float)a.delta;
    const Float theta = (Float)a.theta;
    const Float obj_amp = (Float)a.obj_amp;
    const bool push = a.is_push != 0;

    // private copy of pi (a workgroup only ever reads its own copy: plain,
    // L1-cached loads in the column sums)
    for (int i = tid; i < m; i += T)
        piw[i] = pig[i];
    if (tid == 0)
        piw[m] = 0; // read through the padding of the columns

    MinStd rng;
    rng.s = st.status[inst].rng;
    const int nrows = st.vcount[inst];
    double alg_bytes = 0.0; // thread 0 of workgroup 0

    for (int i = tid; i < nrows; i += T)
        vrows[i] = vlist[i];
    if (tid == 0) {
        misc_s[MS_PICH] = 0;
        misc_s[WM_DEAD] = 0;
        ring_s[wp.ring] = 0; // what the padding of a column reads
    }
    __syncthreads();
    if (tid == 0) {
        if (ord == 2) // random_sorting (itm-common.hpp:708-714 / :817-823), every copy alike
            rng.BXH_WIDE_SHUFFLE(vrows, (unsigned)nrows);
        misc_s[MS_RNG] = (int)rng.s;
    }
    __syncthreads();
    rng.s = (unsigned)misc_s[MS_RNG];

    unsigned nslow = 0;
    bool dead = false;
    // byte address of the dynamic LDS segment in the LDS address space (ds_write from asm)
    const unsigned lds_base =
      (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned char* stg_p = uni + (size_t)wave * wide_stage_bytes((int)sizeof(Float));
    unsigned char* stg_i = stg_p + WIDE_TILE * Geo::PSLOT;

    for (int it = 0; it < nrows && !dead; ++it) {
        const int k = (ord == 1) ? vrows[nrows - 1 - it] : vrows[it];
        const int rs = pb.row_ptr[k];
        const int r = pb.row_ptr[k + 1] - rs;
        // this workgroup's part of the row: entries [start, start + cnt)
        const int ch = (((r + G - 1) / G) + WIDE_TILE - 1) & ~(WIDE_TILE - 1);
        const int start = min(r, grp * ch);
        const int cnt = min(ch, r - start);
        const int bk_min = pb.bmin[k], bk_max = pb.bmax[k];
        const int bk_neg = PM1 ? pb.bneg[k] : 0;

#ifdef BXH_WPROF
        unsigned long long rprof_t = (unsigned long long)clock64(); // phases of every row (slots 7, 23, 27 - 31)
#endif
        // ---- decrease_preference: P[row k] *= theta, kept for affect --------
        // (the P stores of the previous row's affect have landed: the second
        // meeting of that row)
        for (int l = tid; l < cnt; l += T) {
            const int pe = pb.row_pos[rs + start + l];
            Float p = __hip_atomic_load(&P[pe], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p = p * theta;
            __hip_atomic_store(&P[pe], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pown_s[l] = p;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        BXH_RSTAMP(27); // decrease_preference
        // ---- compute_reduced_costs: this workgroup's columns, 32 per wave ---
        // The eight waves take eight tiles at a time (a tile set) and walk them
        // in step, piece by piece: the rows inside a column ascend, so at piece
        // t every column of the set is near row (t + 1/2) * rows_per_piece, and
        // a window of `ring` pi entries in LDS, moved along by all waves
        // together, serves nearly every entry (the others: a gather from the
        // workgroup's copy in global memory).
        const int ntile = (cnt + WIDE_TILE - 1) / WIDE_TILE;
        const int nset = (ntile + T / 64 - 1) / (T / 64);
        // longest column of this workgroup's part, in pieces (uniform over the workgroup)
        if (tid == 0)
            misc_s[WM_TNP] = 0;
        __syncthreads();
        {
            int mynp = 0;
            for (int l = tid; l < cnt; l += T)
                mynp = max(mynp, wp.col_desc[pb.row_col[rs + start + l]].y >> 4);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                mynp = max(mynp, __shfl_xor(mynp, o, 64));
            if (lane == 0 && mynp > 0)
                atomicMax(&misc_s[WM_TNP], mynp);
        }
        __syncthreads();
        const int tnp_row = misc_s[WM_TNP];
        // The window: the ring holds four quarters of Q rows; three are valid
        // (q0 .. q0 + 2, centred on the rows of the piece at hand), the fourth is
        // being refilled by LDS-DMA while the waves stream on.
        const int Q = wp.ring >> 2;
        const int nq = (m + Q - 1) / Q;
        const unsigned rmask = (unsigned)wp.ring - 1u;
        constexpr int NL = Geo::PL + 2; // staging loads of one piece
        // Round t of tnp_row: every column is at the fraction t / tnp_row of its
        // entries (a column of np pieces works on piece t * np / tnp_row and rests
        // in the rounds where that does not move), i.e. -- the rows inside a
        // column being a sorted sample of 0 .. m - 1 -- near row (t + 1/2) m / tnp_row.
        const float rows_per_round = (float)m / (float)max(tnp_row, 1);
        auto quarter_of = [&](int round) -> int {
            const float centre = ((float)round + 0.5f) * rows_per_round;
            const int q = (int)(centre / (float)Q - 1.5f);
            return max(0, min(q, nq - 3));
        };
        // (piece of a column of np pieces at round t: t * np / tnp_row, tracked
        // without a division: acc += np, a carry past tnp_row moves the piece on)
        auto fill_sync = [&](int qa, int qb) { // quarters [qa, qb) by every thread (nothing in flight)
            for (int q = qa; q < qb; ++q)
                for (int i = tid; i < Q; i += T) {
                    const int rr = q * Q + i;
                    ring_s[(unsigned)rr & rmask] = rr < m ? piw[rr] : (Float)0;
                }
        };
        auto fill_dma = [&](int q) { // quarter q, asynchronously: 1 KB per wave and instruction
            const int chunks = Q * (int)sizeof(Float) / 1024;
            for (int ch = wave; ch < chunks; ch += T / 64) {
                const size_t row0 = (size_t)q * Q + (size_t)ch * (1024 / sizeof(Float));
                const char* src = (const char*)(piw + row0) + lane * 16;
                const unsigned dst = __builtin_amdgcn_readfirstlane(
                  lds_base + (unsigned)lp.off_ring +
                  (unsigned)(((unsigned)row0 & rmask) * (unsigned)sizeof(Float)));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n"
                             "s_mov_b32 m0, %2\n"
                             "s_nop 0\n"
                             "global_load_lds_dwordx4 %1, off\n"
                             "s_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(src), "s"(dst)
                             : "memory");
            }
        };
        for (int set = 0; set < nset; ++set) {
            const int tile = set * (T / 64) + wave;
            const int c = lane & 31;
            const bool prole = lane < 32;
            const int l = tile * WIDE_TILE + c;
            const bool valid = tile < ntile && l < cnt;
            int cs = 0, np = 0;
            Float cj = 0;
            bool negf = false;
            if (valid) {
                const int e = rs + start + l;
                const int j = pb.row_col[e];
                const int2 cd = wp.col_desc[j];
                cs = cd.x;
                np = cd.y >> 4;
                cj = cost[j];
                if (PM1)
                    negf = pb.fac_csr[e] < 0;
            }
            int tnp = np;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1)
                tnp = max(tnp, __shfl_xor(tnp, o, 64));
            tnp = __builtin_amdgcn_readfirstlane(tnp); // (uniform: scalar loop control)
            // the columns this lane loads for: their first slot and piece count
            const char* pbase[Geo::PL];
            int pnp[Geo::PL];
#pragma unroll
            for (int i = 0; i < Geo::PL; ++i) {
                const int ci = i * Geo::PCOLS + lane / Geo::PPARTS;
                const int ccs = __shfl(cs, ci, 64);
                pnp[i] = max(1, __shfl(np, ci, 64));
                pbase[i] = (const char*)P + (size_t)ccs * sizeof(Float) + (lane % Geo::PPARTS) * 16;
            }
            const char* ibase[2];
            int inp[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ci = i * 16 + (lane >> 2);
                const int ccs = __shfl(cs, ci, 64);
                inp[i] = max(1, __shfl(np, ci, 64));
                ibase[i] = (const char*)pb.col_row + (size_t)ccs * 4 + (lane & 3) * 16;
            }
            Float sum = 0; // lanes 0-31: sum_a_p of column c, lanes 32-63: sum_a_pi
            const unsigned lds_p = lds_base + (unsigned)(stg_p - smem) +
                                   (unsigned)((lane / Geo::PPARTS) * Geo::PSLOT + (lane % Geo::PPARTS) * 16);
            const unsigned lds_i = lds_base + (unsigned)(stg_i - smem) +
                                   (unsigned)((lane >> 2) * WIDE_ISLOT + (lane & 3) * 16);

            // the window at piece 0: the whole ring, by every thread (the set before
            // has drained: nothing of any wave is in flight)
            int q0 = quarter_of(0);
            fill_sync(q0, q0 + 4);
            __syncthreads();
            int since_adv = 3; // pieces since the last refill was issued (3: none pending)

            // Software pipeline, three pieces ahead: piece p lives in register set
            // p % 3 from its issue (behind the landing of piece p - 3) to its landing.
            // staging loads of round `rnd` into register set K: every column's
            // piece of that round (a column that rests is loaded again: its 8 or 4
            // lanes could be masked, but a wave-instruction costs the same)
            int pacc[Geo::PL], iacc[2]; // (rounds are issued in order, each once)
#pragma unroll
            for (int i = 0; i < Geo::PL; ++i)
                pacc[i] = 0;
            iacc[0] = iacc[1] = 0;
            auto issue_round = [&](auto kc, int) {
                constexpr int K = decltype(kc)::value;
                WideAsm<Float>::template issue<K>(pbase[0], pbase[1], pbase[Geo::PL - 2],
                                                  pbase[Geo::PL - 1], ibase[0], ibase[1]);
#pragma unroll
                for (int i = 0; i < Geo::PL; ++i) {
                    pacc[i] += pnp[i];
                    if (pacc[i] >= tnp_row) {
                        pacc[i] -= tnp_row;
                        pbase[i] += 16 * sizeof(Float);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    iacc[i] += inp[i];
                    if (iacc[i] >= tnp_row) {
                        iacc[i] -= tnp_row;
                        ibase[i] += 64;
                    }
                }
            };
            int oacc = 0; // this lane's own column
            const bool streaming = tnp > 0; // this wave has a tile in this set
            auto body = [&](auto kc, int piece) { // `piece`: the round
                constexpr int K = decltype(kc)::value;
                // staging sets of this wave in flight (one per round, to the last round)
                const int inflight = streaming ? min(3, tnp_row - piece) : 0;
                // ---- the window moves on?  (uniform over the workgroup)
                const int qn = quarter_of(piece);
                if (qn != q0) {
                    // the refill issued at the last move has landed (it is older than
                    // all but min(since_adv, inflight) of the sets in flight) ...
                    const int younger = min(min(since_adv, 2), inflight);
                    if (younger >= 2)
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
                    else if (younger == 1)
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads(); // ... in every wave, and nobody reads quarter q0 any more
                    if (qn == q0 + 1) {
                        fill_dma(q0 + 4); // into q0's slots; valid from the NEXT move on
                        since_adv = 0;
                    } else { // (columns far apart from the matrix average: start over)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
                        fill_sync(qn, qn + 4);
                        __syncthreads();
                        since_adv = 3;
                    }
                    q0 = qn;
                }
                ++since_adv;
                if (!streaming)
                    return; // (no tile of this set: the wave only keeps the barriers)
                const int lo = q0 * Q;
                WideAsm<Float>::template land<K>(inflight - 1, lds_p, lds_i);
                // (the slots are this wave's own: no barrier, but the compiler must
                // keep the other lanes' writes in front of this lane's reads)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // this column moves on in this round?
                oacc += np;
                const bool act = oacc >= tnp_row;
                if (act)
                    oacc -= tnp_row;
                const int* ip = (const int*)(stg_i + c * WIDE_ISLOT);
                int ri[16];
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    ri[t] = ip[t];
                if (piece + 3 < tnp_row)
                    issue_round(kc, piece + 3);
                if (prole) {
                    if (act) {
                        const Float* sp = (const Float*)(stg_p + c * Geo::PSLOT);
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const Float p = sp[t];
                            if (PM1) {
                                const Float fa = (ri[t] & 0x40000000) ? (Float)-1 : (Float)1;
                                sum += fa * p;
                            } else
                                sum += p;
                        }
                    }
                } else {
                    Float piv[16];
                    unsigned out = 0; // entries whose row is outside the three valid quarters
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int rr = ri[t] & 0x3fffffff;
                        const bool pad = rr == m;
                        const bool in = pad || (unsigned)(rr - lo) < (unsigned)(3 * Q);
                        piv[t] = ring_s[pad ? (unsigned)wp.ring : ((unsigned)rr & rmask)];
                        out |= (in || !act) ? 0u : (1u << t);
                    }
                    if (out) { // (rare: a column far ahead of or behind the others)
#pragma unroll
                        for (int t = 0; t < 16; ++t)
                            if (out & (1u << t))
                                piv[t] = piw[ri[t] & 0x3fffffff];
                    }
                    if (act) {
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            if (PM1) {
                                const Float fa = (ri[t] & 0x40000000) ? (Float)-1 : (Float)1;
                                sum += fa * piv[t];
                            } else
                                sum += piv[t];
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier(); // reads done before the next piece is staged
            };
            if (streaming) {
                issue_round(std::integral_constant<int, 0>{}, 0);
                if (tnp_row > 1)
                    issue_round(std::integral_constant<int, 1>{}, 1);
                if (tnp_row > 2)
                    issue_round(std::integral_constant<int, 2>{}, 2);
            }
            for (int p0 = 0; p0 < tnp_row; p0 += 3) {
                body(std::integral_constant<int, 0>{}, p0);
                if (p0 + 1 < tnp_row)
                    body(std::integral_constant<int, 1>{}, p0 + 1);
                if (p0 + 2 < tnp_row)
                    body(std::integral_constant<int, 2>{}, p0 + 2);
            }
            // a refill may still be on its way: the next set refills the ring itself
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const Float sum_a_pi = __shfl(sum, c + 32, 64);
            if (prole && valid) {
                Float v = cj - sum_a_pi - sum;
                if (PM1 && negf)
                    v = -v;
                if (push) // the 01 solvers index the cost with the LOCAL id (quirk)
                    v += obj_amp * (PM1 ? cj : cost[start + l]);
                val_s[l] = v;
            }
        }
        if (tid < 80)
            part_s[tid] = (tid == 66 || tid == 69) ? ~0ull : 0ull;
        if (tid == 0)
            misc_s[MS_WAVES + 34] = 0; // tied keys this workgroup lists
        __syncthreads();

        BXH_RSTAMP(28); // the column sums
        // ---- this workgroup's record of the row ----------------------------
        const unsigned long long K0 = order_key<Float>((Float)0, maximize);
        {
            unsigned long long A = ~0ull, A2 = ~0ull, B = ~0ull, C = ~0ull, D = ~0ull;
            int cN = 0, cZ = 0;
            bool tie = false;
            for (int l = tid; l < cnt; l += T) {
                const unsigned long long key = order_key<Float>(val_s[l], maximize);
                int slot, dup;
                tie |= wide_hash_insert(htab, wp.hmask, key, &slot, &dup);
                hslot_s[l] = slot;
                if (wp.tied_fast) {
                    // the duplicates of a key are counted where the key sits, and the first of
                    // them lists the key: what a tied row needs instead of its sorted order
                    // (listed in LDS first -- the staging slots are idle -- and then with ONE
                    // addition per workgroup to the instance's counter: a thousand additions to one
                    // address cost a tied row more than everything else here)
                    if (dup >= 0) {
                        if (atomicAdd(&hcount[dup], 1u) == 0u)
                            ((int*)uni)[atomicAdd(&misc_s[MS_WAVES + 34], 1)] = dup;
                    } else if (slot < 0)
                        atomicOr(&tdl[1], 1); // (a NaN: the row is sorted)
                }
                if (key < A) {
                    A2 = A;
                    A = key;
                } else if (key < A2)
                    A2 = key;
                B = umin64(B, ~key);
                if (key < K0) {
                    ++cN;
                    C = umin64(C, ~key);
                } else if (key == K0)
                    ++cZ;
                else
                    D = umin64(D, key);
            }
            const unsigned long long ownA = A;
            wave_umin64x4(A, B, C, D);
            unsigned long long E = ownA == A ? A2 : ownA;
            E = wave_umin64(E);
            cN = wave_sum(cN);
            cZ = wave_sum(cZ);
            const bool wtie = __any(tie);
            if (lane == 0) {
                atomicMax(&part_s[64], ~B); // largest key
                atomicMax(&part_s[65], ~C); // largest key of class N
                atomicMin(&part_s[66], D);  // smallest key of class S
                atomicMin(&part_s[69], A);  // smallest key
                part_s[wave * 2] = A;
                part_s[wave * 2 + 1] = E;
                atomicAdd(&part_s[67], ((unsigned long long)(unsigned)cN << 32) | (unsigned)cZ);
                if (wtie)
                    atomicOr(&part_s[68], 1ull);
            }
        }
        __syncthreads();
        if (wp.tied_fast && misc_s[MS_WAVES + 34] > 0) {
            const int nloc = misc_s[MS_WAVES + 34];
            if (tid == 0)
                misc_s[MS_WAVES + 35] = atomicAdd(&tdl[0], nloc);
            __syncthreads();
            const int at0 = misc_s[MS_WAVES + 35];
            for (int i = tid; i < nloc; i += T)
                if (at0 + i < WIDE_TKEYS)
                    __hip_atomic_store(&tdl[64 + at0 + i], ((const int*)uni)[i], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < 8) {
            unsigned long long v;
            const unsigned long long a1 = part_s[69];
            if (tid == 0)
                v = a1;
            else if (tid == 1) { // second smallest key of this part
                v = ~0ull;
                for (int w = 0; w < T / 64; ++w) {
                    const unsigned long long w1 = part_s[w * 2], w2 = part_s[w * 2 + 1];
                    v = umin64(v, w1 == a1 ? w2 : w1);
                }
            } else if (tid == 2)
                v = part_s[64];
            else if (tid == 3)
                v = part_s[65];
            else if (tid == 4)
                v = part_s[66];
            else if (tid == 5)
                v = part_s[67];
            else if (tid == 6)
                v = part_s[68];
            else
                v = (unsigned long long)cnt;
            __hip_atomic_store(&recg[grp * 8 + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        BXH_RSTAMP(29); // keys into the tie table, the record
        // ---- first meeting of the row: every part is summed ----------------
        if (!wide_sync(cnt_s1, (unsigned)G * (unsigned)(it + 1), err_word, misc_s + WM_DEAD)) {
            dead = true;
            break;
        }
        for (int i = tid; i < G * 8; i += T)
            rec_s[i] = __hip_atomic_load(&recg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();

        BXH_RSTAMP(30); // first meeting
        // ---- selection without sorting: the same scalar code in every thread
        // of every workgroup (see sweep_kernel_v2).  A row with equal reduced costs
        // takes it too (round 5) when the tie table tells all that the sorted order
        // would be used for: calculator_sort shuffles every run of equal values
        // (itm-common.hpp:139-147), which moves the engine by draws that follow from the
        // run LENGTHS in key order alone, and the order inside a run only matters for the
        // one run that the selected position splits.  The tied keys and their counts
        // (a few hundred in a row of 20 000 integer-cost columns) come out of the table,
        // are put in order in LDS, the engine is taken through the runs without a swap
        // (wide_runs_dry), and the split run -- if there is one -- is shuffled for real:
        // its members are told apart by their ordinal in the row (a count per workgroup,
        // one more meeting).  Anything else (a selected position whose key the records
        // do not give, more than WIDE_TKEYS tied keys, a run beyond 46 340) is sorted.
        bool fast = false;
        int f_mode = 0, f_nN = 0, f_nZ = 0;
        unsigned long long f_thr = 0ull;
        Float f_d = 0, f_newpi = 0;
        bool f_pich = false;
        bool row_tied = false;
        bool f_split = false; // the run of f_thr goes on behind the selected position
        int f_split_q = 0, f_split_off = 0, f_split_len = 0;
        // LDS of the tied keys: the staging slots (idle outside the column sums), 256 ints of the pi window
        int tk_cap = min(WIDE_TKEYS, (lp.union_bytes - 8 * WIDE_TRUN - 4 * wp.chmax) / 20);
        tk_cap = tk_cap >= 64 && wp.ring * (int)sizeof(Float) >= 1024 ? next_pow2(tk_cap + 1) / 2 : 0; // a power of two
        unsigned long long* tk_key = (unsigned long long*)uni;            // [tk_cap] in key order
        int* tk_len = (int*)(uni + (size_t)tk_cap * 8);                   // [tk_cap] elements of the run
        int* tk_base = tk_len + tk_cap;                                   // [tk_cap] (wide_first_rejection)
        unsigned* tk_state = (unsigned*)(tk_base + tk_cap);               // [tk_cap] engine in front of the run
        int* tk_list = (int*)(tk_state + tk_cap);                         // [WIDE_TRUN] the split run, shuffled
        int* tk_inv = tk_list + WIDE_TRUN;                                // [WIDE_TRUN] position of a member
        int* tk_ord = tk_inv + WIDE_TRUN;                                 // [chmax] ordinal of an own member
        int* tk_cand = (int*)ring_s;                                      // [256]
        for (int g = 0; g < G; ++g)
            row_tied |= rec_s[g * 8 + 6] != 0ull; // (a row of one NaN counts: its flag is to be taken back)
        if (r >= 2) {
            unsigned long long g1 = ~0ull, gx = 0ull, gxN = 0ull, gnS = ~0ull;
            int nN = 0, nZ = 0;
            bool anytie = false;
            for (int g = 0; g < G; ++g) {
                g1 = umin64(g1, rec_s[g * 8]);
                gx = umax64(gx, rec_s[g * 8 + 2]);
                gxN = umax64(gxN, rec_s[g * 8 + 3]);
                gnS = umin64(gnS, rec_s[g * 8 + 4]);
                nN += (int)(rec_s[g * 8 + 5] >> 32);
                nZ += (int)(rec_s[g * 8 + 5] & 0xffffffffu);
                anytie |= rec_s[g * 8 + 6] != 0ull;
            }
            auto second_smallest = [&]() -> unsigned long long {
                unsigned long long v = ~0ull;
                for (int g = 0; g < G; ++g) {
                    const unsigned long long a1 = rec_s[g * 8], a2 = rec_s[g * 8 + 1];
                    v = umin64(v, a1 == g1 ? a2 : a1);
                }
                return v;
            };
            bool ok = !anytie;
            const unsigned saved_rng = rng.s;
            int ntk = 0; // tied keys of the row
            if (anytie && wp.tied_fast) {
                // ---- the tied keys in order, the engine through their runs --------
#ifdef BXH_WPROF
                unsigned long long wprof_t = (unsigned long long)clock64();
#endif
                ntk = __hip_atomic_load(&tdl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int tflags = __hip_atomic_load(&tdl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = tflags == 0 && ntk >= 1 && ntk <= tk_cap;
                if (ok) {
                    // in key order: a bitonic network over the keys and their counts (the keys are distinct)
                    const int n2t = next_pow2(ntk);
                    for (int t = tid; t < n2t; t += T) {
                        unsigned long long key = ~0ull;
                        int len = 0;
                        if (t < ntk) {
                            const int sl = __hip_atomic_load(&tdl[64 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            key = __hip_atomic_load(&htab[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            len = 1 + (int)__hip_atomic_load(&hcount[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        tk_key[t] = key;
                        tk_len[t] = len;
                    }
                    __syncthreads();
                    for (int kk = 2; kk <= n2t; kk <<= 1)
                        for (int j = kk >> 1; j > 0; j >>= 1) {
                            for (int t = tid; t < (n2t >> 1); t += T) {
                                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                                const int l = i | j;
                                const unsigned long long ka = tk_key[i], kb = tk_key[l];
                                if ((ka > kb) == ((i & kk) == 0)) {
                                    const int la = tk_len[i], lb = tk_len[l];
                                    tk_key[i] = kb;
                                    tk_key[l] = ka;
                                    tk_len[i] = lb;
                                    tk_len[l] = la;
                                }
                            }
                            __syncthreads();
                        }
                    int kinds = 0;
                    for (int t = tid; t < ntk; t += T) {
                        const int len = tk_len[t];
                        kinds |= (len > 512 ? 1 : 0) | (len > 46340 ? 2 : 0);
                    }
                    kinds = wide_or_bits(kinds, misc_s + MS_WAVES + 36);
                    unsigned st_end = rng.s;
                    ok = !(kinds & 2);
                    if (ok)
                        ok = wide_runs_dry(tk_len, tk_base, tk_state, ntk, (kinds & 1) != 0, rng.s, pb.lcg_pow,
                                           tk_cand, misc_s, &st_end);
                    if (ok)
                        rng.s = st_end; // behind the shuffles; select_variables draws next
                }
                BXH_WSTAMP(13); // the tied keys in order, the engine through their runs
            }
            // elements of the run of `key` (1: not a tied key)
            auto run_of = [&](unsigned long long key, int& q) -> int {
                int lo = 0, hi = ntk;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (tk_key[mid] < key)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                q = lo;
                return (lo < ntk && tk_key[lo] == key) ? tk_len[lo] : 1;
            };
            int sel = 0;
            unsigned long long k_sel = 0ull, k_sel1 = 0ull;
            if (ok) {
                const int c_size = bk_neg;
                int bkmin = bk_min, bkmax = bk_max;
                if (kind == 0 || kind == 2 || bkmin == bkmax) {
                    sel = min(bkmin + c_size, r) - 1;
                } else {
                    if (kind == 1) {
                        bkmin = min(bkmin, r);
                        bkmax = min(bkmax, r);
                    } else {
                        bkmin += c_size;
                        bkmax = min(bkmax + c_size, r);
                    }
                    sel = bkmax - 1;
                    int cur = bkmin;
                    if (cur <= bkmax) {
                        bool stopped = false;
                        if (cur < nN)
                            cur = nN; // N entries never stop the scan
                        while (!stopped && cur <= bkmax && cur < nN + nZ) { // the zero entries
                            if (rng.coin())
                                stopped = true;
                            else
                                ++cur;
                        }
                        if (!stopped && cur <= bkmax) {
                            if (cur < r)
                                stopped = true; // first S entry
                            else { // R[r_size]: what an earlier, longer row left there
                                const Float sv = __hip_atomic_load(&staleg[r], __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_AGENT);
                                if (sv == 0)
                                    stopped = rng.coin();
                                else
                                    stopped = maximize ? sv < 0 : sv > 0;
                            }
                        }
                        if (stopped)
                            sel = cur - 1;
                    }
                }
                int qdummy;
                auto key_at = [&](int rank, bool& good) -> unsigned long long {
                    if (rank == 0)
                        return g1;
                    if (rank == 1)
                        return (anytie && run_of(g1, qdummy) >= 2) ? g1 : second_smallest();
                    if (rank == r - 1)
                        return gx;
                    if (nN > 0 && rank == nN - 1)
                        return gxN;
                    if (rank >= nN && rank < nN + nZ)
                        return K0;
                    if (rank == nN + nZ)
                        return gnS;
                    good = false;
                    return 0ull;
                };
                if (sel < 0) {
                    k_sel = key_at(0, ok);
                } else if (sel + 1 >= r) {
                    k_sel = key_at(r - 1, ok);
                } else {
                    k_sel = key_at(sel, ok);
                    k_sel1 = key_at(sel + 1, ok);
                    if (ok && anytie && k_sel1 == k_sel) {
                        // the selected position splits a run: where the run starts, how long it is
                        int q = 0;
                        const int len = run_of(k_sel, q);
                        int first = -1;
                        if (k_sel == g1)
                            first = 0;
                        else if (k_sel == K0)
                            first = nN;
                        else if (k_sel == gnS)
                            first = nN + nZ;
                        if (len < 2 || first < 0 || len > WIDE_TRUN || sel < first || sel + 1 >= first + len)
                            ok = false;
                        else {
                            f_split = true;
                            f_split_q = q;
                            f_split_off = sel - first;
                            f_split_len = len;
                        }
                    }
                }
            }
            if (!ok) {
                rng.s = saved_rng; // the sort path redoes the shuffles and the scan
                f_split = false;
            } else {
                const Float one = 1, two = 2;
                const Float middle = (two + one) / two;
                Float d = delta;
                const Float v_sel = key_value<Float>(k_sel, maximize);
                if (sel < 0) {
                    d += kfac * (v_sel / two);
                    f_mode = 0;
                } else if (sel + 1 >= r) {
                    d += kfac * (v_sel * middle);
                    f_mode = 1;
                } else {
                    const Float v_sel1 = key_value<Float>(k_sel1, maximize);
                    const Float old_pi = piw[k];
                    const Float new_pi = old_pi + ((v_sel + v_sel1) / two);
                    d += kfac * (v_sel1 - v_sel);
                    f_mode = 2;
                    f_thr = k_sel;
                    f_newpi = new_pi;
                    f_pich = signbit(old_pi) != signbit(new_pi);
                }
                f_d = d;
                f_nN = nN;
                f_nZ = nZ;
                fast = true;
            }
        }

#ifdef BXH_WPROF
        if (tid == 0 && grp == 0 && st.prof && row_tied) { // tied rows: fast, fast with a split run, sorted
            unsigned long long* pr = st.prof + (size_t)inst * 32;
            pr[24] += fast ? 1ull : 0ull;
            pr[25] += (fast && f_split) ? 1ull : 0ull;
            pr[26] += fast ? 0ull : 1ull;
        }
#endif
        if (fast && f_split) {
            // ---- the run the selected position splits: std::shuffle of its members --
            // (in the order calculator_sort leaves them in: ascending position in the row)
            int mine = 0;
            for (int l = tid; l < cnt; l += T)
                mine += order_key<Float>(val_s[l], maximize) == f_thr ? 1 : 0;
            mine = wave_sum(mine);
            if (tid == 0)
                misc_s[MS_WAVES + 33] = 0;
            __syncthreads();
            if (lane == 0 && mine)
                atomicAdd(&misc_s[MS_WAVES + 33], mine);
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(&tdl[8 + grp], misc_s[MS_WAVES + 33], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            if (!wide_sync(cnt_sl, (unsigned)G * (nslow + 1u), err_word, misc_s + WM_DEAD)) {
                dead = true;
                break;
            }
            ++nslow;
            int before = 0; // members in the workgroups in front of this one
            for (int g = 0; g < grp; ++g)
                before += __hip_atomic_load(&tdl[8 + g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ordinals of the own members, ascending in l: a block of T positions at a time
            for (int l0 = 0; l0 < cnt; l0 += T) {
                const int l = l0 + tid;
                const bool mem = l < cnt && order_key<Float>(val_s[l], maximize) == f_thr;
                const unsigned long long mask = __ballot(mem);
                if (lane == 0)
                    misc_s[MS_WAVES + wave] = __popcll(mask);
                __syncthreads();
                int off = before;
                for (int w = 0; w < wave; ++w)
                    off += misc_s[MS_WAVES + w];
                int blk = 0;
                for (int w = 0; w < T / 64; ++w)
                    blk += misc_s[MS_WAVES + w];
                if (l < cnt)
                    tk_ord[l] = mem ? off + __popcll(mask & ((1ull << lane) - 1ull)) : -1;
                before += blk;
                __syncthreads();
            }
            // the shuffle itself, on the ordinals: one lane, the engine as it stood in front of the run
            for (int i = tid; i < f_split_len; i += T)
                tk_list[i] = i;
            __syncthreads();
            if (tid == 0)
                lane_run_shuffle(tk_list, (unsigned)f_split_len, tk_state[f_split_q]);
            __syncthreads();
            for (int p_ = tid; p_ < f_split_len; p_ += T)
                tk_inv[tk_list[p_]] = p_;
            __syncthreads();
        }

        if (fast) {
            // ---- affect without the sorted order ------------------------------
            __syncthreads(); // every thread has read pi_k
            if (tid == 0) {
                if (grp == 0)
                    alg_bytes += pb.row_bytes[k];
                if (f_mode == 2) {
                    piw[k] = f_newpi;
                    if (f_pich)
                        misc_s[MS_PICH] = 1;
                }
            }
            for (int l = tid; l < cnt; l += T) {
                const int e = rs + start + l;
                const unsigned long long key = order_key<Float>(val_s[l], maximize);
                const int j = pb.row_col[e];
                const int pe = pb.row_pos[e];
                bool up = f_mode == 1 || (f_mode == 2 && key <= f_thr);
                if (f_split && key == f_thr) // in front of the selected position, or at it?
                    up = tk_inv[tk_ord[l]] <= f_split_off;
                if (PM1 && pb.fac_csr[e] < 0)
                    up = !up;
                const Float p = pown_s[l]; // P[pe] * theta
                if (up)
                    atomicOr(&xg[j >> 5], 1u << (j & 31));
                else
                    atomicAnd(&xg[j >> 5], ~(1u << (j & 31)));
                __hip_atomic_store(&P[pe], up ? p + f_d : p - f_d, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                // R[pos].value after the sort only matters through
                // stop_iterating: its class (N / zero / S)
                const int pos = start + l;
                const Float repN = maximize ? (Float)1 : (Float)-1;
                __hip_atomic_store(&staleg[pos],
                                   pos < f_nN ? repN : (pos < f_nN + f_nZ ? (Float)0 : -repN),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            // ---- calculator_sort: the whole row, in every workgroup alike ----
            for (int l = tid; l < cnt; l += T)
                __hip_atomic_store(&vbuf[start + l],
                                   (unsigned long long)__double_as_longlong((double)val_s[l]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!wide_sync(cnt_sl, (unsigned)G * (nslow + 1u), err_word, misc_s + WM_DEAD)) {
                dead = true;
                break;
            }
            ++nslow;
            if (tid == 0)
                misc_s[MS_TIE] = 0;
            if (r <= WIDE_SMALL) {
                for (int i = tid; i < r; i += T)
                    val_all[i] = (Float)__longlong_as_double((long long)__hip_atomic_load(
                      &vbuf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                __syncthreads();
                if (r > 1) {
                    Prof prof_;
                    run_merge_sort<Float>(val_all, r, maximize, key_run, id_run, key_s, id_s, &prof_);
                } else {
                    if (tid == 0) {
                        key_s[0] = order_key<Float>(val_all[0], maximize);
                        id_s[0] = 0;
                    }
                    __syncthreads();
                }
                for (int i = tid; i + 1 < r; i += T)
                    if (key_s[i] == key_s[i + 1])
                        misc_s[MS_TIE] = 1;
                __syncthreads();
                bool shuffled = false;
                if (misc_s[MS_TIE]) {
                    unsigned stt = rng.s;
                    shuffled = parallel_tie_shuffle(key_s, id_s, r, &stt, pb.lcg_pow, (int*)key_run,
                                                    (int*)key_run + n2s, id_run, misc_s + MS_WAVES);
                    if (shuffled)
                        rng.s = stt;
                }
                if (tid == 0) {
                    if (grp == 0)
                        alg_bytes += pb.row_bytes[k];
                    if (misc_s[MS_TIE] && !shuffled) {
                        int b0 = 0;
                        for (int i = 1; i <= r; ++i) {
                            if (i == r || key_s[i] != key_s[b0]) {
                                rng.BXH_WIDE_SHUFFLE(id_s + b0, (unsigned)(i - b0));
                                b0 = i;
                            }
                        }
                    }
                    auto Rv = [&](int i) -> Float { // R[i].value, R[r] is stale
                        return i < r ? val_all[id_s[i]]
                                     : __hip_atomic_load(&staleg[i], __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                    };
                    auto stop = [&](Float v) -> bool { // itm-common.hpp:150-163
                        if (v == 0)
                            return rng.coin();
                        return maximize ? v < 0 : v > 0;
                    };
                    const int c_size = bk_neg;
                    int bkmin = bk_min, bkmax = bk_max;
                    int sel;
                    if (kind == 0 || kind == 2 || bkmin == bkmax) {
                        sel = min(bkmin + c_size, r) - 1;
                    } else {
                        if (kind == 1) {
                            bkmin = min(bkmin, r);
                            bkmax = min(bkmax, r);
                        } else {
                            bkmin += c_size;
                            bkmax = min(bkmax + c_size, r);
                        }
                        sel = bkmax - 1;
                        int nN = 0, nNZ = 0;
                        {
                            int lo = 0, hi = r; // first key >= K0
                            while (lo < hi) {
                                const int mid = (lo + hi) >> 1;
                                if (key_s[mid] < K0)
                                    lo = mid + 1;
                                else
                                    hi = mid;
                            }
                            nN = lo;
                            hi = r; // first key > K0
                            while (lo < hi) {
                                const int mid = (lo + hi) >> 1;
                                if (key_s[mid] <= K0)
                                    lo = mid + 1;
                                else
                                    hi = mid;
                            }
                            nNZ = lo;
                        }
                        int cur = bkmin;
                        if (cur <= bkmax) {
                            bool stopped = false;
                            if (cur < nN)
                                cur = nN;
                            while (!stopped && cur <= bkmax && cur < nNZ) {
                                if (rng.coin())
                                    stopped = true;
                                else
                                    ++cur;
                            }
                            if (!stopped && cur <= bkmax) {
                                if (cur < r)
                                    stopped = true; // first S entry
                                else
                                    stopped = stop(Rv(r)); // R[r_size] (stale)
                            }
                            if (stopped)
                                sel = cur - 1;
                        }
                    }
                    const Float one = 1, two = 2;
                    const Float middle = (two + one) / two;
                    Float d = delta;
                    if (sel < 0) {
                        d += kfac * (Rv(0) / two);
                    } else if (sel + 1 >= r) {
                        d += kfac * (Rv(sel) * middle);
                    } else {
                        const Float old_pi = piw[k];
                        const Float new_pi = old_pi + ((Rv(sel) + Rv(sel + 1)) / two);
                        piw[k] = new_pi;
                        d += kfac * (Rv(sel + 1) - Rv(sel));
                        if (signbit(old_pi) != signbit(new_pi))
                            misc_s[MS_PICH] = 1;
                    }
                    misc_s[MS_SEL] = sel;
                    *d_s = d;
                    misc_s[MS_RNG] = (int)rng.s;
                }
                __syncthreads();
                rng.s = (unsigned)misc_s[MS_RNG];
                {
                    const int sel = misc_s[MS_SEL];
                    const Float d = *d_s;
                    for (int i = tid; i < r; i += T) {
                        const int idx = id_s[i];
                        if (i >= start && i < start + cnt) // this workgroup's positions of R
                            __hip_atomic_store(&staleg[i], val_all[idx], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        const int l = idx - start;
                        if (l < 0 || l >= cnt)
                            continue; // another workgroup's entry
                        const int e = rs + idx;
                        const int j = pb.row_col[e];
                        const int pe = pb.row_pos[e];
                        bool up = i <= sel;
                        if (PM1 && pb.fac_csr[e] < 0)
                            up = !up;
                        const Float p = pown_s[l];
                        if (up)
                            atomicOr(&xg[j >> 5], 1u << (j & 31));
                        else
                            atomicAnd(&xg[j >> 5], ~(1u << (j & 31)));
                        __hip_atomic_store(&P[pe], up ? p + d : p - d, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else {
                // ---- a long row: sorted in global scratch (every workgroup its
                // own copy), tied runs shuffled by one thread ------------------
#ifdef BXH_WPROF
                unsigned long long wprof_t = (unsigned long long)clock64();
#endif
                const int n2 = next_pow2(r);
                unsigned long long* const bkey0 = wp.bkey + (size_t)inst * (G + 2) * (size_t)wp.n2big;
                int* const bid0 = wp.bid + (size_t)inst * (G + 2) * (size_t)wp.n2big;
                unsigned long long* bkey = bkey0 + (size_t)grp * (size_t)wp.n2big;
                int* bid = bid0 + (size_t)grp * (size_t)wp.n2big;
                // LDS scratch of this path: the pi window (refilled at the start of every
                // tile set) and the staging slots (idle outside the column sums)
                const int ring_bytes = wp.ring * (int)sizeof(Float);
                unsigned long long* lk = (unsigned long long*)ring_s;
                int* li = (int*)uni;
                // the workgroups of the instance sort a share each (chunks of Cc pairs,
                // about one per workgroup) where LDS holds such a chunk ...
                int Cc = next_pow2((n2 + G - 1) / G);
                while (Cc > 64 && (Cc > 4 * T || Cc * 8 > ring_bytes || Cc * 4 > lp.union_bytes))
                    Cc >>= 1;
                const bool coop = G > 1 && !wp.own_sort && Cc >= 64 && Cc <= n2 && Cc <= 4 * T &&
                                  Cc * 8 <= ring_bytes && Cc * 4 <= lp.union_bytes;
                if (coop) {
                    if (!wide_sort_coop<Float>(vbuf, r, n2, maximize, grp, G, bkey0 + (size_t)G * wp.n2big,
                                               bid0 + (size_t)G * wp.n2big, bkey0 + (size_t)(G + 1) * wp.n2big,
                                               bid0 + (size_t)(G + 1) * wp.n2big, bid, lk, li, Cc, cnt_sl,
                                               nslow, err_word, misc_s + WM_DEAD
#ifdef BXH_WPROF
                                               ,
                                               (tid == 0 && grp == 0 && st.prof) ? st.prof + (size_t)inst * 32
                                                                                  : nullptr
#endif
                                               )) {
                        dead = true;
                        break;
                    }
                } else {
                    // ... or every workgroup the whole row, in its own copy
                    for (int i = tid; i < n2; i += T) {
                        unsigned long long key = ~0ull;
                        if (i < r)
                            key = order_key<Float>(
                              (Float)__longlong_as_double((long long)__hip_atomic_load(
                                &vbuf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)),
                              maximize);
                        bkey[i] = key;
                        bid[i] = i;
                    }
                    __syncthreads();
                    int C = 512;
                    while (C * 2 <= n2 && C * 2 * 8 <= ring_bytes && C * 2 * 4 <= lp.union_bytes)
                        C *= 2;
                    if (C * 8 <= ring_bytes && C * 4 <= lp.union_bytes && C <= n2)
                        wide_sort_big(bkey, bid, n2, lk, li, C);
                    else
                        bitonic_sort(bkey, bid, n2);
                }
                BXH_WSTAMP(0); // the sort
                // the sorted keys: this workgroup's own copy, or -- after the shared sort -- the one
                // copy of the instance, read through L2 (its lines may sit in this CU's L1 from an
                // earlier row)
                const bool shared_keys = coop;
                if (coop)
                    bkey = bkey0 + (size_t)(G + 1) * (size_t)wp.n2big;
                auto bk = [&](int i) -> unsigned long long {
                    return shared_keys
                             ? __hip_atomic_load(&bkey[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : bkey[i];
                };
                // ---- the tied runs.  Draw number g of the row's engine is a jump of the
                // linear congruential engine, and a run of L elements takes L / 2 draws
                // unless one is rejected: so every run's first draw is known from the run
                // lengths, the runs can be shuffled side by side (one lane each: runs of
                // up to 512; all threads: up to 2048), and one pass over all draws tells
                // whether -- and in which run first -- a rejection shifts the stream; from
                // that run on thread 0 shuffles serially, as for rows with too many runs.
                int serial_head = 0;     // first position left to thread 0's serial shuffle
                unsigned serial_state = rng.s;
                bool runs_listed = false;
                int* rl_head = (int*)uni;
                const int RCAP = lp.union_bytes / 12;
                int* rl_len = rl_head + RCAP;
                int* rl_base = rl_len + RCAP;
                // ordered lists of the runs' first and last positions.  Every wave takes one
                // contiguous stretch of the row (whole 64-position steps: coalesced reads), counts
                // its heads and tails, and after ONE barrier writes them behind those of the waves
                // in front of it: two barriers per row (rounds 2 - 4: two per 512 positions and
                // pass, 160 for a row of 20 000 elements -- half of what a tied row cost on top).
                // The keys of four steps are in flight together, a position's neighbours come from
                // the lanes next to it, and "is there a tie at all" is the count of the heads.
                int nh = 0, nt = 0;
                {
                    const int NWv = T / 64;
                    const int seg = (((r + NWv - 1) / NWv) + 63) & ~63;
                    const int s0 = min(r, wave * seg), s1 = min(r, s0 + seg);
                    // heads / tails among the positions c0 ... c0 + 255 below s1, as ballots
                    auto run_marks = [&](int c0, unsigned long long (&mh)[4], unsigned long long (&mt)[4]) {
                        unsigned long long kc[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int i = c0 + 64 * u + lane;
                            kc[u] = i < r ? bk(i) : 0ull;
                        }
                        const unsigned long long kfront = c0 > 0 ? bk(c0 - 1) : 0ull;
                        const unsigned long long kback = c0 + 256 < r ? bk(c0 + 256) : 0ull;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int i = c0 + 64 * u + lane;
                            unsigned long long kn = __shfl(kc[u], (lane + 1) & 63);
                            unsigned long long kp = __shfl(kc[u], (lane + 63) & 63);
                            const unsigned long long kn_edge = u < 3 ? __shfl(kc[u < 3 ? u + 1 : u], 0) : kback;
                            const unsigned long long kp_edge = u > 0 ? __shfl(kc[u > 0 ? u - 1 : u], 63) : kfront;
                            if (lane == 63)
                                kn = kn_edge;
                            if (lane == 0)
                                kp = kp_edge;
                            const bool in = i < s1;
                            const bool eq_next = in && i + 1 < r && kn == kc[u];
                            const bool eq_prev = in && i > 0 && kp == kc[u];
                            mh[u] = __ballot(eq_next && !eq_prev);
                            mt[u] = __ballot(eq_prev && !eq_next);
                        }
                    };
                    int ch = 0, ct = 0;
                    for (int c0 = s0; c0 < s1; c0 += 256) {
                        unsigned long long mh[4], mt[4];
                        run_marks(c0, mh, mt);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            ch += __popcll(mh[u]);
                            ct += __popcll(mt[u]);
                        }
                    }
                    if (lane == 0) {
                        misc_s[MS_WAVES + wave] = ch;
                        misc_s[MS_WAVES + 16 + wave] = ct;
                    }
                    __syncthreads();
                    int offh = 0, offt = 0;
                    for (int w = 0; w < NWv; ++w) {
                        const int cw = misc_s[MS_WAVES + w], tw = misc_s[MS_WAVES + 16 + w];
                        if (w < wave) {
                            offh += cw;
                            offt += tw;
                        }
                        nh += cw;
                        nt += tw;
                    }
                    BXH_WSTAMP(1); // tie detection = the runs counted
                    if (nh > 0 && nh <= RCAP) {
                        for (int c0 = s0; c0 < s1; c0 += 256) {
                            unsigned long long mh[4], mt[4];
                            run_marks(c0, mh, mt);
                            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int i = c0 + 64 * u + lane;
                                const int ph = offh + __popcll(mh[u] & below), pt = offt + __popcll(mt[u] & below);
                                if ((mh[u] >> lane) & 1ull)
                                    rl_head[ph] = i;
                                if ((mt[u] >> lane) & 1ull)
                                    rl_len[pt] = i; // (the last position, for now)
                                offh += __popcll(mh[u]);
                                offt += __popcll(mt[u]);
                            }
                        }
                    }
                    __syncthreads();
                }
                const bool tied = nh > 0;
                if (tied) {
                    BXH_WSTAMP(2); // the runs listed
                    const int nruns = nh;
                    (void)nt;
                    int run_kinds = 0; // bit 0: a run of more than 512 elements; bit 1: of more than 46 340
                    if (nruns <= RCAP) {
                        int mine = 0;
                        for (int q = tid; q < nruns; q += T) {
                            const int len = rl_len[q] - rl_head[q] + 1;
                            rl_len[q] = len;
                            mine |= (len > 512 ? 1 : 0) | (len > 46340 ? 2 : 0);
                        }
                        run_kinds = wide_or_bits(mine, misc_s + MS_WAVES + 36);
                    }
                    // (beyond 46 340 elements std::shuffle draws once per element, not once per pair:
                    // the draw numbers below do not hold for such a row -- thread 0 takes it whole)
                    if (nruns <= RCAP && !(run_kinds & 2)) {
                        runs_listed = true;
                        const bool has_long = (run_kinds & 1) != 0;
#ifdef BXH_WPROF
                        if (tid == 0 && grp == 0 && st.prof) { // the row's runs: how many, how long
                            unsigned long long tot = 0, mx = 0;
                            for (int q = 0; q < nruns; ++q) {
                                tot += (unsigned long long)rl_len[q];
                                mx = max(mx, (unsigned long long)rl_len[q]);
                            }
                            unsigned long long* pr = st.prof + (size_t)inst * 32;
                            pr[16] += (unsigned long long)nruns;
                            pr[17] += tot;
                            pr[18] = max(pr[18], mx);
                            pr[19] += 1ull;
                            pr[20] += (unsigned long long)r;
                            pr[21] = max(pr[21], (unsigned long long)(nruns + (r - (int)tot)));
                            pr[22] += (unsigned long long)(nruns + (r - (int)tot));
                        }
#endif
                        int qb = 0;               // runs in front of qb are done
                        unsigned st_row = rng.s;  // the engine in front of run qb
                        while (qb < nruns) {
                            // which run from qb on is the first with a rejected draw?
                            int total_draws = 0;
                            int first_rej = wide_first_rejection(rl_len, rl_base, qb, nruns, has_long, st_row,
                                                                 pb.lcg_pow, (int*)ring_s, misc_s, &total_draws);
                            if (first_rej < 0) // (more draws than the table of the engine's powers holds)
                                first_rej = qb;
                            BXH_WSTAMP(3); // every draw checked for a rejection
                            // the runs in front of it, side by side: one LANE per run of up to 512
                            // elements (round 5).  No draw of these runs is rejected (checked above),
                            // so run q starts at draw rl_base[q] of the row's engine, and its shuffle is
                            // the serial std::shuffle on a private engine, in 32-bit arithmetic (all
                            // ranges are below 2^31).  Rounds 2 - 4 gave every run a WAVE that traced
                            // each final position back through all L swaps -- O(L^2 / 64) dependent LDS
                            // reads per run: the 100 runs of ~200 equal costs of an integer-cost row of
                            // 20 000 elements took ~270 us of its ~330 us of tie handling.  A lane per
                            // run does L swaps in the workgroup's own copy of the ids (L2-resident),
                            // all runs of the row at once.
                            for (int q = qb + tid; q < first_rej; q += T) {
                                const int L = rl_len[q];
                                if (L > 512)
                                    continue;
                                lane_run_shuffle(bid + rl_head[q], (unsigned)L,
                                                 lcg_jump(pb.lcg_pow, st_row, rl_base[q]));
                            }
                            __syncthreads();
                            BXH_WSTAMP(4); // the short runs shuffled, a lane each
                            // the longer runs in front of it, one after the other
                            {
                                int* ids_l = (int*)ring_s;
                                const int cap_run = min(2048, ring_bytes / 12);
                                int* jt_l = ids_l + cap_run;
                                int* out_l = jt_l + cap_run;
                                for (int q = qb; has_long && q < first_rej; ++q) {
                                    const int L = rl_len[q];
                                    if (L <= 512)
                                        continue;
                                    const int head = rl_head[q];
                                    unsigned stq = lcg_jump(pb.lcg_pow, st_row, rl_base[q]);
                                    if (L <= cap_run) {
                                        for (int i = tid; i < L; i += T)
                                            ids_l[i] = bid[head + i];
                                        __syncthreads();
                                        // (no draw of this run is rejected: checked above)
                                        wide_run_shuffle(ids_l, L, &stq, pb.lcg_pow, jt_l, out_l, misc_s + WM_FLAG);
                                        for (int i = tid; i < L; i += T)
                                            bid[head + i] = ids_l[i];
                                        __syncthreads();
                                    } else {
                                        if (tid == 0) {
                                            MinStd e;
                                            e.s = stq;
                                            e.BXH_WIDE_SHUFFLE(bid + head, (unsigned)L);
                                        }
                                        __syncthreads();
                                    }
                                }
                            }
                            if (first_rej >= nruns) {
                                st_row = lcg_jump(pb.lcg_pow, st_row, total_draws);
                                qb = nruns;
                            } else {
                                // the run with the rejected draw: by one thread, in LDS when it fits
                                const int L = rl_len[first_rej], head = rl_head[first_rej];
                                const unsigned stq = lcg_jump(pb.lcg_pow, st_row, rl_base[first_rej]);
                                int* ids_l = (int*)ring_s;
                                const bool in_lds = L * 4 <= ring_bytes;
                                if (in_lds)
                                    for (int i = tid; i < L; i += T)
                                        ids_l[i] = bid[head + i];
                                __syncthreads();
                                if (tid == 0) {
                                    MinStd e;
                                    e.s = stq;
                                    if (in_lds)
                                        e.BXH_WIDE_SHUFFLE(ids_l, (unsigned)L);
                                    else
                                        e.BXH_WIDE_SHUFFLE(bid + head, (unsigned)L);
                                    misc_s[MS_RNG] = (int)e.s;
                                }
                                __syncthreads();
                                if (in_lds)
                                    for (int i = tid; i < L; i += T)
                                        bid[head + i] = ids_l[i];
                                st_row = (unsigned)misc_s[MS_RNG];
                                qb = first_rej + 1;
                                __syncthreads();
                            }
                        }
                        BXH_WSTAMP(5); // long runs, runs behind a rejected draw
                        serial_head = r;
                        serial_state = st_row;
                    }
                    __syncthreads();
                    if (tid == 0)
                        ring_s[wp.ring] = 0; // (the scratch took the padding entry of the window)
                    __syncthreads();
                }
                auto val_of = [&](int idx) -> Float {
                    return (Float)__longlong_as_double((long long)__hip_atomic_load(
                      &vbuf[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                };
                if (tid == 0) {
                    if (grp == 0)
                        alg_bytes += pb.row_bytes[k];
                    if (tied) { // what is left of the tied runs (all of them when the
                        rng.s = runs_listed ? serial_state : rng.s; // runs were not listed), serially
                        int b0 = runs_listed ? serial_head : 0;
                        unsigned long long kb0 = b0 < r ? bk(b0) : 0ull;
                        for (int i = b0 + 1; i <= r; ++i) {
                            const unsigned long long ki = i < r ? bk(i) : 0ull;
                            if (i == r || ki != kb0) {
                                rng.BXH_WIDE_SHUFFLE(bid + b0, (unsigned)(i - b0));
                                b0 = i;
                                kb0 = ki;
                            }
                        }
                    }
                    auto Rv = [&](int i) -> Float {
                        return i < r ? val_of(bid[i])
                                     : __hip_atomic_load(&staleg[i], __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                    };
                    auto stop = [&](Float v) -> bool {
                        if (v == 0)
                            return rng.coin();
                        return maximize ? v < 0 : v > 0;
                    };
                    const int c_size = bk_neg;
                    int bkmin = bk_min, bkmax = bk_max;
                    int sel;
                    if (kind == 0 || kind == 2 || bkmin == bkmax) {
                        sel = min(bkmin + c_size, r) - 1;
                    } else {
                        if (kind == 1) {
                            bkmin = min(bkmin, r);
                            bkmax = min(bkmax, r);
                        } else {
                            bkmin += c_size;
                            bkmax = min(bkmax + c_size, r);
                        }
                        sel = bkmax - 1;
                        int nN = 0, nNZ = 0;
                        {
                            int lo = 0, hi = r;
                            while (lo < hi) {
                                const int mid = (lo + hi) >> 1;
                                if (bk(mid) < K0)
                                    lo = mid + 1;
                                else
                                    hi = mid;
                            }
                            nN = lo;
                            hi = r;
                            while (lo < hi) {
                                const int mid = (lo + hi) >> 1;
                                if (bk(mid) <= K0)
                                    lo = mid + 1;
                                else
                                    hi = mid;
                            }
                            nNZ = lo;
                        }
                        int cur = bkmin;
                        if (cur <= bkmax) {
                            bool stopped = false;
                            if (cur < nN)
                                cur = nN;
                            while (!stopped && cur <= bkmax && cur < nNZ) {
                                if (rng.coin())
                                    stopped = true;
                                else
                                    ++cur;
                            }
                            if (!stopped && cur <= bkmax) {
                                if (cur < r)
                                    stopped = true;
                                else
                                    stopped = stop(Rv(r));
                            }
                            if (stopped)
                                sel = cur - 1;
                        }
                    }
                    const Float one = 1, two = 2;
                    const Float middle = (two + one) / two;
                    Float d = delta;
                    if (sel < 0) {
                        d += kfac * (Rv(0) / two);
                    } else if (sel + 1 >= r) {
                        d += kfac * (Rv(sel) * middle);
                    } else {
                        const Float old_pi = piw[k];
                        const Float new_pi = old_pi + ((Rv(sel) + Rv(sel + 1)) / two);
                        piw[k] = new_pi;
                        d += kfac * (Rv(sel + 1) - Rv(sel));
                        if (signbit(old_pi) != signbit(new_pi))
                            misc_s[MS_PICH] = 1;
                    }
                    misc_s[MS_SEL] = sel;
                    *d_s = d;
                    misc_s[MS_RNG] = (int)rng.s;
                }
                __syncthreads();
                BXH_WSTAMP(6); // thread 0: what is left of the runs, the selection, pi and d
                rng.s = (unsigned)misc_s[MS_RNG];
                {
                    const int sel = misc_s[MS_SEL];
                    const Float d = *d_s;
                    for (int i = tid; i < r; i += T) {
                        const int idx = bid[i];
                        if (i >= start && i < start + cnt)
                            __hip_atomic_store(&staleg[i], val_of(idx), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        const int l = idx - start;
                        if (l < 0 || l >= cnt)
                            continue;
                        const int e = rs + idx;
                        const int j = pb.row_col[e];
                        const int pe = pb.row_pos[e];
                        bool up = i <= sel;
                        if (PM1 && pb.fac_csr[e] < 0)
                            up = !up;
                        const Float p = pown_s[l];
                        if (up)
                            atomicOr(&xg[j >> 5], 1u << (j & 31));
                        else
                            atomicAnd(&xg[j >> 5], ~(1u << (j & 31)));
                        __hip_atomic_store(&P[pe], up ? p + d : p - d, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        BXH_RSTAMP(31); // selection and affect (either path)
        // the row's keys leave the tie table (with their counts and the list of the tied ones)
        for (int l = tid; l < cnt; l += T) {
            const int s = hslot_s[l];
            if (s >= 0) {
                __hip_atomic_exchange(&htab[s], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (row_tied && wp.tied_fast)
                    __hip_atomic_store(&hcount[s], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (row_tied && wp.tied_fast && grp == 0 && tid == 0) {
            __hip_atomic_store(&tdl[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&tdl[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        BXH_RSTAMP(23); // the keys leave the table
        // ---- second meeting of the row: affect's stores have landed --------
        if (!wide_sync(cnt_s2, (unsigned)G * (unsigned)(it + 1), err_word, misc_s + WM_DEAD)) {
            dead = true;
            break;
        }
        BXH_RSTAMP(7); // second meeting
    }
    if (dead)
        return; // the error word is set: the host reports BXH_ERR_DEVICE

    // ---- write back, then compute_violated_constraints (itm-common.hpp:102-115)
    if (grp == 0)
        for (int i = tid; i < m; i += T)
            pig[i] = piw[i];
    if (wp.skip_scan) { // the scan follows in its own launch, shared by the instances
        if (grp == 0 && tid == 0) {
            DevStatus s = st.status[inst];
            s.pi_changed = misc_s[MS_PICH];
            s.rng = rng.s;
            s.rows_in_sweep = (unsigned)nrows;
            s.row_updates += (unsigned long long)nrows;
            s.alg_bytes += alg_bytes;
            st.status[inst] = s;
        }
        return;
    }
    {
        const int per = (m + G - 1) / G;
        const int k0 = grp * per, k1 = min(m, k0 + per);
        int* vf = wp.vflag32 + (size_t)inst * m;
        for (int k = k0 + wave; k < k1; k += T / 64) {
            const int rs = pb.row_ptr[k], re = pb.row_ptr[k + 1];
            int v = 0;
            for (int e = rs + lane; e < re; e += 64) {
                const int j = pb.row_col[e];
                const int bit =
                  (__hip_atomic_load(&xg[j >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >>
                   (j & 31)) &
                  1u;
                v += PM1 ? (int)pb.fac_csr[e] * bit : bit;
            }
            v = wave_sum(v);
            if (lane == 0)
                __hip_atomic_store(&vf[k], (int)!(pb.bmin[k] <= v && v <= pb.bmax[k]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!wide_sync(cnt_end, (unsigned)G, err_word, misc_s + WM_DEAD))
            return;
        if (grp != 0)
            return;
        int base = 0;
        int* scratch_s = misc_s + MS_WAVES;
        for (int c = 0; c < m; c += T) {
            const int k = c + tid;
            const bool f = k < m && __hip_atomic_load(&vf[k], __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT) != 0;
            const unsigned long long mask = __ballot(f);
            const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
            if (lane == 0)
                scratch_s[wave] = __popcll(mask);
            __syncthreads();
            int off = 0, tot = 0;
            for (int w = 0; w < T / 64; ++w) {
                const int cw = scratch_s[w];
                if (w < wave)
                    off += cw;
                tot += cw;
            }
            if (f)
                vlist[base + off + prefix] = k;
            base += tot;
            __syncthreads();
        }
        if (tid == 0) {
            st.vcount[inst] = base;
            DevStatus s = st.status[inst];
            s.remaining = base;
            s.pi_changed = misc_s[MS_PICH];
            s.rng = rng.s;
            s.rows_in_sweep = (unsigned)nrows;
            s.row_updates += (unsigned long long)nrows;
            s.alg_bytes += alg_bytes;
            st.status[inst] = s;
        }
    }
}

// compute_order::init (itm-common.hpp:660-680) for the wide layout: the flags
// of the violated rows on `parts` workgroups per instance (x from global
// memory), then the ordered compaction.
__global__ __launch_bounds__(512) void
wide_flags_kernel(DevProblem pb, DevState st, WideParams wp, const unsigned char* active, int parts)
{
    const int inst = blockIdx.x / parts, part = blockIdx.x % parts;
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (active && !active[inst])
        return;
    const bool pm1 = pb.kind == 2 || pb.kind == 3;
    const unsigned* xg = st.x + (size_t)inst * (2 * pb.nwords);
    int* vf = wp.vflag32 + (size_t)inst * pb.m;
    const int per = (pb.m + parts - 1) / parts;
    const int k0 = part * per, k1 = min(pb.m, k0 + per);
    for (int k = k0 + wave; k < k1; k += T >> 6) {
        const int rs = pb.row_ptr[k], re = pb.row_ptr[k + 1];
        int v = 0;
        for (int e = rs + lane; e < re; e += 64) {
            const int j = pb.row_col[e];
            const int bit = (xg[j >> 5] >> (j & 31)) & 1u;
            v += pm1 ? (int)pb.fac_csr[e] * bit : bit;
        }
        v = wave_sum(v);
        if (lane == 0)
            vf[k] = !(pb.bmin[k] <= v && v <= pb.bmax[k]);
    }
}

// compute_violated_constraints of ALL instances in one pass over A: x is
// transposed first (one word per variable, bit b = x of instance b), then a wave
// per row adds up, per instance, the bits of the row's columns -- the matrix
// (8 GB at the XL shape) is read once instead of once per instance.
__global__ __launch_bounds__(256) void
wide_transpose_kernel(DevProblem pb, DevState st, unsigned* xt, int batch)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= pb.n)
        return;
    unsigned word = 0;
    for (int b = 0; b < batch; ++b)
        word |= ((st.x[(size_t)b * (2 * pb.nwords) + (j >> 5)] >> (j & 31)) & 1u) << b;
    xt[j] = word;
}

template<int NB> // instances handled (16 / 32)
__global__ __launch_bounds__(512) void
wide_flags_all_kernel(DevProblem pb, WideParams wp, const unsigned* xt, const unsigned char* active,
                      int batch)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool pm1 = pb.kind == 2 || pb.kind == 3;
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    for (int k = blockIdx.x * (blockDim.x >> 6) + wave; k < pb.m; k += nwaves) {
        const int rs = pb.row_ptr[k], re = pb.row_ptr[k + 1];
        int cnt[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
            cnt[b] = 0;
        for (int e = rs + lane; e < re; e += 64) {
            const unsigned w = xt[pb.row_col[e]];
            if (pm1) {
                const int sgn = (int)pb.fac_csr[e];
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    cnt[b] += (int)((w >> b) & 1u) * sgn;
            } else {
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    cnt[b] += (int)((w >> b) & 1u);
            }
        }
        const int lo = pb.bmin[k], hi = pb.bmax[k];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int v = wave_sum(cnt[b]);
            if (lane == 0 && b < batch && (!active || active[b]))
                wp.vflag32[(size_t)b * pb.m + k] = !(lo <= v && v <= hi);
        }
    }
}

// The same for up to 16 instances of a 01 matrix whose rows hold their columns in ascending order
// (round 5).  wide_flags_all_kernel gathers x's transposed word of every element out of L2: 2 of the
// 64 bytes of every sector it pulls are used, and the scan of the XL shape runs at the L2's pace
// (9 ms for 8 GB of column indices).  Here a workgroup holds a BAND of 61 440 columns of the
// transposed x in LDS (16 bits a column) and walks a chunk of the rows: a row's elements inside the
// band are one stretch of its column list (two binary searches, 64 rows at a time, a lane each),
// streamed 16 bytes a lane and four loads ahead; the gathers hit LDS.  Per instance the bits are
// counted in byte fields (the mask's four low bits times 0x00204081 puts them into four bytes), folded
// to 16-bit fields after every batch, summed over the wave with DPP, and lanes 0 - 15 add the row's
// band totals to vsum[row][instance]; wide_flags_finish_kernel compares them with the bounds and
// zeroes them again.  A band total cannot pass 61 440: the 16-bit fields hold it.
constexpr int WIDE_BAND = 61440;

__device__ inline unsigned
wide_row16_sum(unsigned v) // every lane of a row of 16: the row's sum
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);  // quad_perm(1,0,3,2)
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);  // quad_perm(2,3,0,1)
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); // row_half_mirror
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); // row_mirror
    return v;
}

__global__ __launch_bounds__(1024) void
wide_flags_band_kernel(DevProblem pb, const unsigned* xt, int* vsum, int nbands, int nchunks)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned short* tab = (unsigned short*)smem;
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwave = T >> 6;
    const int band = blockIdx.x % nbands, chunk = blockIdx.x / nbands;
    const int c0 = band * WIDE_BAND, c1 = min(pb.n, c0 + WIDE_BAND);
    for (int i = tid; i < c1 - c0; i += T)
        tab[i] = (unsigned short)xt[c0 + i];
    __syncthreads();
    const int per = (pb.m + nchunks - 1) / nchunks;
    const int r0 = chunk * per, r1 = min(pb.m, r0 + per);
    const long long nnz = (long long)pb.row_ptr[pb.m];
    for (int base = r0 + wave * 64; base < r1; base += nwave * 64) {
        const int row = base + lane;
        int lo = 0, hi = 0;
        if (row < r1) {
            const int rs = pb.row_ptr[row], re = pb.row_ptr[row + 1];
            int a = rs, b = re; // first element of the row with a column >= c0 ...
            while (a < b) {
                const int mid = a + ((b - a) >> 1); // (a + b passes 2^31 at the XL shape)
                if (pb.row_col[mid] < c0)
                    a = mid + 1;
                else
                    b = mid;
            }
            lo = a;
            b = re; // ... and the first with a column >= c1
            while (a < b) {
                const int mid = a + ((b - a) >> 1);
                if (pb.row_col[mid] < c1)
                    a = mid + 1;
                else
                    b = mid;
            }
            hi = a;
        }
        const int nrow = min(64, r1 - base);
        for (int j = 0; j < nrow; ++j) {
            const int lo_j = __builtin_amdgcn_readlane(lo, j), hi_j = __builtin_amdgcn_readlane(hi, j);
            if (lo_j >= hi_j)
                continue;
            unsigned s16[8];
#pragma unroll
            for (int w = 0; w < 8; ++w)
                s16[w] = 0u;
            for (int e0 = lo_j; e0 < hi_j; e0 += 1024) {
                int cv[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { // four 16-byte loads a lane, issued together
                    const long long e = (long long)e0 + u * 256 + lane * 4;
                    if (e + 3 < nnz && e < hi_j) {
                        int4 q;
                        __builtin_memcpy(&q, pb.row_col + e, 16);
                        cv[u][0] = q.x;
                        cv[u][1] = q.y;
                        cv[u][2] = q.z;
                        cv[u][3] = q.w;
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            cv[u][t] = (e + t < hi_j) ? pb.row_col[e + t] : c0;
                    }
                }
                unsigned a8[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int e = e0 + u * 256 + lane * 4 + t;
                        const unsigned mk = e < hi_j ? (unsigned)tab[cv[u][t] - c0] : 0u;
                        a8[0] += ((mk & 0xFu) * 0x00204081u) & 0x01010101u;
                        a8[1] += (((mk >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
                        a8[2] += (((mk >> 8) & 0xFu) * 0x00204081u) & 0x01010101u;
                        a8[3] += (((mk >> 12) & 0xFu) * 0x00204081u) & 0x01010101u;
                    }
#pragma unroll
                for (int w = 0; w < 4; ++w) { // byte f of a8[w] is instance 4 w + f
                    s16[2 * w] += a8[w] & 0x00FF00FFu;            // instances 4 w, 4 w + 2
                    s16[2 * w + 1] += (a8[w] >> 8) & 0x00FF00FFu; // instances 4 w + 1, 4 w + 3
                }
            }
            unsigned tot[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const unsigned rsum = wide_row16_sum(s16[w]);
                tot[w] = (unsigned)__builtin_amdgcn_readlane((int)rsum, 0) +
                         (unsigned)__builtin_amdgcn_readlane((int)rsum, 16) +
                         (unsigned)__builtin_amdgcn_readlane((int)rsum, 32) +
                         (unsigned)__builtin_amdgcn_readlane((int)rsum, 48);
            }
            if (lane < 16) { // instance `lane` = 4 w + f: word 2 w + (f & 1), half f >> 1
                const int w = lane >> 2, f = lane & 3;
                const int idx = 2 * w + (f & 1);
                unsigned word = tot[0];
#pragma unroll
                for (int i = 1; i < 8; ++i)
                    word = idx == i ? tot[i] : word;
                const int val = (int)((word >> (16 * (f >> 1))) & 0xFFFFu);
                if (val)
                    atomicAdd(&vsum[(size_t)(base + j) * 16 + lane], val);
            }
        }
    }
}

// vflag32[b][k] from the sums of wide_flags_band_kernel; the sums go back to zero for the next scan
__global__ __launch_bounds__(256) void
wide_flags_finish_kernel(DevProblem pb, WideParams wp, int* vsum, const unsigned char* active, int batch)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)pb.m * 16)
        return;
    const int k = (int)(i >> 4), b = (int)(i & 15);
    const int v = vsum[i];
    if (v)
        vsum[i] = 0;
    if (b < batch && (!active || active[b]))
        wp.vflag32[(size_t)b * pb.m + k] = !(pb.bmin[k] <= v && v <= pb.bmax[k]);
}

__global__ __launch_bounds__(512) void
wide_compact_kernel(DevProblem pb, DevState st, WideParams wp, const unsigned char* active)
{
    __shared__ int scratch_s[16];
    const int inst = blockIdx.x;
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (active && !active[inst])
        return;
    const int* vf = wp.vflag32 + (size_t)inst * pb.m;
    int* vlist = st.vlist + (size_t)inst * st.lcap;
    int base = 0;
    for (int c = 0; c < pb.m; c += T) {
        const int k = c + tid;
        const bool f = k < pb.m && vf[k] != 0;
        const unsigned long long mask = __ballot(f);
        const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0)
            scratch_s[wave] = __popcll(mask);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < (T >> 6); ++w) {
            const int cw = scratch_s[w];
            if (w < wave)
                off += cw;
            tot += cw;
        }
        if (f)
            vlist[base + off + prefix] = k;
        base += tot;
        __syncthreads();
    }
    if (tid == 0) {
        st.vcount[inst] = base;
        st.status[inst].remaining = base;
    }
}

} // namespace bxh
