// baryonyx_amd/csrc/bxh_kernels.hpp -- gfx950 kernels of the in-the-middle
// row update (device side of include/bxh.h).
//
// One workgroup == one Solver instance (the reference runs one Solver per
// worker thread, lib/src/itm-optimizer-common.hpp:649-653).  Rows of one
// instance are strictly sequential (row k+1 reads the pi_k and P_k. written
// by row k), so the parallelism is lanes-over-the-columns-of-a-row times
// instances-per-GPU.
//
// Floating-point results must be bit-identical to the reference's, which
// fixes the order of every sum: a lane walks ITS column in CSC order with two
// separate accumulators exactly like compute_reduced_costs
// (lib/src/itm-solver-inequalities-01.cpp:166-191).  Build with
// -ffp-contract=off.
//
// Data layout in HBM: P is stored in CSC order so a column
// walk reads one contiguous run; `row_pos[e]` maps CSR element e to its CSC
// slot for the two row-shaped accesses (decay and affect).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "bxh_rng.hpp"

namespace bxh {

struct DevProblem
{
    int m, n, nnz, rmax, nwords;
    int nnzp;     // CSC slots of one owner's slice (columns padded to 16 entries)
    int groups;   // workgroups (column owners, column j -> j % groups) per instance
    int romax;    // longest per-owner part of a row
    int spare;    // 1: a spare set of 16 staging slots exists (overflow chunk)
    int pipe;     // pipelined sweep: workgroups per instance (1 row chain + producers), 0 = off
    int lookahead; // pipelined sweep: a producer works on the row D positions past the last finished one
    int sw_chain, sw_prod; // pipelined sweep: waves with staging slots in the chain / a producer
    int gather_other; // pipelined sweep: entries of a row whose values the chain's non-staging waves fetch
    int static_late;  // pipelined sweep: 1 = late columns of consecutive rows from the static gaps (0: always the marks)
    int cmax;     // longest column (real entries)
    int kind;     // bxh_kind
    int maximize; // 0 / 1
    const int* row_ptr;        // [m+1]
    const int* row_col;        // [nnz]   column of CSR element e
    const int* row_pos;        // [nnz]   padded CSC slot of CSR element e
    const int2* row_cs;        // [nnz]   {first slot, padded length} of e's column
    const int* col_start;      // [n]     first padded CSC slot of column j
    const int* col_len;        // [n]     real entries of column j
    const int* col_row;        // [nnzp]  row of CSC slot q, | 0x40000000 when the
                               //         coefficient is -1; padding slots hold m
                               //         (pi_s[m] == 0, P == 0 there for ever)
    const unsigned short* col_row16; // [nnzp] the same in 16 bits (bit 15 = -1),
                               //         only when m < 32768 (v2 kernel)
    const int* sub_ptr;        // [groups][m+1] per-owner part of row k in sub_e
    const int* sub_e;          // CSR element ids, row order, owner after owner
    const int4* sub_desc;      // per sub_e slot: {first CSC slot, padded length,
                               //  CSC slot of the entry, column}: one load per entry
    const void* desc_cost;     // [nnz] Float: cost of the descriptor's column
    const int4* chain_rec;     // [nnz + 64] pipelined sweep, row chain: {first CSC slot, padded length |
                               //  position of the entry in its column << 16, column, w} of CSR element e,
                               //  w = rows up | rows down << 4 to the column's neighbouring entries (clipped
                               //  to 15, 0 = none) | the neighbour's index in its row << 8 (up) / << 18 (down)
                               //  where it is the adjacent row: a row's records are one run, moved to LDS
                               //  two rows ahead by LDS-DMA
    const unsigned* ownmask;   // [groups][2*nwords] x bits of each owner
    const signed char* fac_csr; // [nnz]  coefficient of CSR element (101 kinds)
    // inequalities-Z (general integer coefficients)
    const int* fac32;          // [nnz]   coefficient of CSR element
    const int* col_fac;        // [nnzp]  |coefficient| of CSC slot q (0 in padding)
    const signed char* zkind;  // [m]     sub-solver of row k (ZKind)
    const int* ex_start;       // [m]     first pattern of row k in ex_flat (exhaustive rows)
    const int* ex_count;       // [m]     number of patterns
    const int* ex_flat;        //         patterns: factor or 0 per element, row after row
    const unsigned long long* zconf; // [m] bit d - 1: row k + d shares a column with row k (d = 1 .. 64)
    int zwave;                 // 1: sweep_kernel_z<Float, true> (rows of up to 63 elements: a wave per row)
    const int4* zdesc;         // [2 nnz] wave form: {first slot, padded length, own slot, column},
                               //         {coefficient, cost bits (lo, hi), real entries of the column} of CSR element e
    const int2* zcolrf;        // [nnzp]  wave form: {row, |coefficient|} of CSC slot q
    const unsigned* zcolp;     // [nnzp]  Z kernel, wave form: row | |coefficient| << 16 (m and |a| below 65536)
    const int* bmin;           // [m] clamped as the Solver constructor does
    const int* bmax;           // [m]
    const int* bneg;           // [m] negative_factor (== c_size of the row)
    const void* cost;          // [n] Float (linear part)
    // quadratic_cost_type (itm-common.hpp:1150-1440): partners of variable j are
    // q_id[q_ptr[j] .. q_ptr[j+1]); null when the objective is linear
    const int* q_ptr;          // [n+1]
    const int* q_id;           // [n_quad]
    const void* q_factor;      // [n_quad] Float
    const double* row_bytes;   // [m] algorithmic bytes of one update of row k
    const unsigned* lcg_pow;   // [32770] 16807^k mod (2^31 - 1): jumps of minstd_rand0
};

struct DevStatus // == bxh_status
{
    int remaining;
    int pi_changed;
    unsigned rng;
    unsigned rows_in_sweep;
    unsigned long long row_updates;
    double alg_bytes; // cumulative algorithmic bytes of the rows updated
};

struct DevState
{
    void* P;             // [batch][nnzp] Float, padded CSC order
    void* pi;            // [batch][m]   Float
    unsigned* x;         // [batch][2*nwords] u32 halves of the bit_array blocks
    void* rstale;        // [batch][rmax+1] Float: R[].value as the reference's
                         //   persistent R array holds it between rows
    int* vlist;          // [batch][lcap] compute_order::R (the violated rows)
    int* vcount;         // [batch]
    signed char* vflag;  // [batch][m] scratch
    DevStatus* status;   // [batch]
    unsigned long long* prof; // [batch][16] phase ticks (profiling build only)
    int* vwork;          // [batch][groups][lcap] private copy of the row list
    int* vpart;          // [batch][groups][m] per-owner row activities
    unsigned long long* xbuf; // [batch][2][rmax][2] tagged halves of the row's reduced costs
    unsigned* cnt;       // [batch] row arrivals, [batch] sweep-end arrivals, [1] error
    unsigned long long* msg; // [batch][m][2] pipelined sweep: tagged pi of every finished row position
    // compute_order's second list and scratch (itm-common.hpp:627-915)
    int lcap;            // capacity of the row lists: max(m, n)
    int2* mlist;         // [batch][lcap] m_order = {row, infeasibility}
    int* mcount;         // [batch]
    int* mrun;           // [batch][lcap] row ids of the sorted m_order / list scratch
    int2* mtmp;          // [batch][lcap] scratch of the sorts
    int* slist;          // [batch][lcap] violated rows of the orders that keep R
    int* vamt;           // [batch][m] infeasibility of every row (0: satisfied)
    unsigned long long* skey; // [batch][next_pow2(lcap)] sort keys
    int* sid;            // [batch][next_pow2(lcap)] sort payload
    unsigned char* zscratch; // [batch][zscratch_bytes] node stack of the B&B rows
    size_t zscratch_bytes;
};

struct Prof
{
    unsigned long long acc[16];
    unsigned long long t;
};
#if defined(BXH_PROFILE) && !defined(BXH_NO_STAMPS)
#define BXH_PSTAMP(pp, slot)                                                   \
    do {                                                                       \
        if (threadIdx.x == 0) {                                                \
            const unsigned long long now_ = wall_clock64();                   \
            (pp)->acc[slot] += now_ - (pp)->t;                                 \
            (pp)->t = now_;                                                    \
        }                                                                      \
    } while (0)
#else
#define BXH_PSTAMP(pp, slot)                                                   \
    do {                                                                       \
    } while (0)
#endif
#if defined(BXH_PROFILE) && !defined(BXH_NO_STAMPS)
#define BXH_STAMP(slot)                                                        \
    do {                                                                       \
        if (tid == 0) {                                                        \
            const unsigned long long now_ = wall_clock64();                   \
            prof.acc[slot] += now_ - prof.t;                                   \
            prof.t = now_;                                                     \
        }                                                                      \
    } while (0)
#else
#define BXH_STAMP(slot)                                                        \
    do {                                                                       \
    } while (0)
#endif

// Profiling build: arrival skew of the waves at the row loop's barriers
// (slots 0..3 of a second accumulator set; part_s[16..47] hold the stamps).
#ifdef BXH_PROFILE
#define BXH_BARRIER(id)                                                        \
    do {                                                                       \
        if ((tid & 63) == 0)                                                   \
            part_s[16 + (id) * 8 + (tid >> 6)] = wall_clock64();               \
        __syncthreads();                                                       \
        if (tid == 0) {                                                        \
            unsigned long long mn_ = ~0ull, mx_ = 0ull;                        \
            for (int w_ = 0; w_ < (T >> 6); ++w_) {                            \
                const unsigned long long v_ = part_s[16 + (id) * 8 + w_];      \
                mn_ = v_ < mn_ ? v_ : mn_;                                     \
                mx_ = v_ > mx_ ? v_ : mx_;                                     \
            }                                                                  \
            skew[id] += mx_ - mn_;                                             \
            own[id] += mx_ - part_s[16 + (id) * 8];                            \
        }                                                                      \
    } while (0)
#else
#define BXH_BARRIER(id) __syncthreads()
#endif

// Timeline build (-DBXH_TIMELINE, tools/debug/chain_timeline.py): lane 0 of every wave of
// the row chain of instance 0 writes the shader clock at 16 points of rows 200..207 into
// the profile buffer ([row][wave][point]).  Not for any other use.
#ifdef BXH_TIMELINE
#ifdef BXH_TIMELINE_LDS
// Stamps without stores in between (-DBXH_TIMELINE_LDS, tools/debug/chain_timeline_light.py):
// six points per wave (-DBXH_TL_P0 .. P5; default 3, 6, 7, 8, 9, 12) go to LDS (part_s[16 ..])
// and are written out at the top of the next row.  A global store in front of a
// `s_waitcnt vmcnt(0)` (there is one on either side of barrier 1) costs its round trip, and
// the full-stamp build shows that round trip as time of the phase behind the stamp.
#ifndef BXH_TL_P0
#define BXH_TL_P0 3
#define BXH_TL_P1 6
#define BXH_TL_P2 7
#define BXH_TL_P3 8
#define BXH_TL_P4 9
#define BXH_TL_P5 12
#endif
#define BXH_TL_SLOT(p_) ((p_) == BXH_TL_P0 ? 0 : (p_) == BXH_TL_P1 ? 1 : (p_) == BXH_TL_P2 ? 2 : (p_) == BXH_TL_P3 ? 3 : (p_) == BXH_TL_P4 ? 4 : (p_) == BXH_TL_P5 ? 5 : -1)
#define BXH_TL(p_)                                                             \
    do {                                                                       \
        if (PIPE && grp == 0 && inst == 0 && (tid & 63) == 0) {                \
            if ((p_) == 0 && it >= 201 && it < 209) {                          \
                const int pts_[6] = { BXH_TL_P0, BXH_TL_P1, BXH_TL_P2, BXH_TL_P3, BXH_TL_P4, BXH_TL_P5 }; \
                for (int s_ = 0; s_ < 6; ++s_)                                 \
                    st.prof[((it - 201) * 8 + (tid >> 6)) * 16 + pts_[s_]] = part_s[16 + (tid >> 6) * 6 + s_]; \
            }                                                                  \
            if (BXH_TL_SLOT(p_) >= 0 && it >= 200 && it < 208)                 \
                part_s[16 + (tid >> 6) * 6 + BXH_TL_SLOT(p_)] = (unsigned long long)clock64(); \
        }                                                                      \
    } while (0)
#define BXH_TLP(p_)                                                            \
    do {                                                                       \
    } while (0)
#else
#define BXH_TL(p_)                                                             \
    do {                                                                       \
        if (PIPE && grp == 0 && inst == 0 && it >= 200 && it < 208 && (tid & 63) == 0) \
            st.prof[((it - 200) * 8 + (tid >> 6)) * 16 + (p_)] = (unsigned long long)clock64(); \
    } while (0)
// the same for the first producer of instance 0 (second half of the profile buffer)
#define BXH_TLP(p_)                                                            \
    do {                                                                       \
        if (PIPE && grp == 1 && inst == 0 && it >= 200 && it < 208 && (tid & 63) == 0) \
            st.prof[1024 + ((it - 200) * 8 + (tid >> 6)) * 16 + (p_)] = (unsigned long long)clock64(); \
    } while (0)
#endif
#else
#define BXH_TL(p_)                                                             \
    do {                                                                       \
    } while (0)
#define BXH_TLP(p_)                                                            \
    do {                                                                       \
    } while (0)
#endif

struct DevSweep
{
    const double* kappa; // [batch]
    double delta, theta, obj_amp;
    int is_push;
    int order;                   // row order of every instance ...
    const int* order_b;          // ... or per instance (compute_order's state), may be null
    const unsigned char* active; // [batch] or null
    unsigned epoch;              // launch number (tags of the row exchange)
};

// ---------------------------------------------------------------------------
// std::minstd_rand0 and the libstdc++ algorithms the reference calls on it
// (itm-common.hpp:45,117-163).  Restated from GCC 11's <bits/random.h>,
// <bits/uniform_int_dist.h> (the "2 divisions" downscaling branch -- the
// engine's range is not 2^32-1) and <bits/stl_algo.h> std::shuffle.
struct MinStd
{
    unsigned s;
    __device__ unsigned next()
    {
        s = (unsigned)(((unsigned long long)s * 16807ull) % 2147483647ull);
        return s;
    }
    // uniform_int_distribution<unsigned long>{0, urange}(g)
    __device__ unsigned long long uniform(unsigned long long urange)
    {
        const unsigned long long urngrange = 2147483645ull; // max() - min()
        // urange < urngrange always holds here (urange <= rmax^2 + rmax)
        const unsigned long long uerange = urange + 1;
        const unsigned long long scaling = urngrange / uerange;
        const unsigned long long past = uerange * scaling;
        unsigned long long ret;
        do
            ret = (unsigned long long)next() - 1ull;
        while (ret >= past);
        return ret / scaling;
    }
    // std::bernoulli_distribution(0.5)(g): generate_canonical<double,53> needs
    // two draws of this engine (log2 of its range is 30).
    __device__ bool coin()
    {
        const double r = 2147483646.0;
        double sum = 0.0;
        double tmp = 1.0;
        sum += (double)(next() - 1u) * tmp;
        tmp *= r;
        sum += (double)(next() - 1u) * tmp;
        tmp = 4611686009837453312.0; // double((long double)r * r)
        double ret = sum / tmp;
        if (ret >= 1.0)
            ret = 0.99999999999999988897769753748434595763683319091796875;
        return ret < 0.5;
    }
    // std::shuffle(first, first + len, g) on an int array.
    template<typename Ptr>
    __device__ void shuffle(Ptr a, unsigned len)
    {
        shuffle_inl(a, len);
    }
    // (the body, for callers that must not leave it to the compiler whether this becomes a call:
    // sweep_kernel_wide, bxh_wide.hpp)
    template<typename Ptr>
    __device__ __attribute__((always_inline)) void shuffle_inl(Ptr a, unsigned len)
    {
        if (len < 2)
            return;
        auto swp = [&](unsigned i, unsigned j) {
            int t = a[i];
            a[i] = a[j];
            a[j] = t;
        };
        const unsigned long long urngrange = 2147483645ull;
        if (urngrange / len >= len) {
            unsigned i = 1;
            if ((len % 2) == 0) {
                swp(i, (unsigned)uniform(1));
                ++i;
            }
            while (i != len) {
                const unsigned long long swap_range = (unsigned long long)i + 1;
                const unsigned long long b1 = swap_range + 1;
                const unsigned long long x = uniform(swap_range * b1 - 1);
                swp(i, (unsigned)(x / b1));
                ++i;
                swp(i, (unsigned)(x % b1));
                ++i;
            }
            return;
        }
        for (unsigned i = 1; i != len; ++i)
            swp(i, (unsigned)uniform(i));
    }
};

// ---------------------------------------------------------------------------
template<typename Float>
__device__ inline unsigned long long
order_key(Float v, bool maximize)
{
    // Monotone map of the value to an unsigned key; -0.0 and +0.0 must be a
    // tie, as `lhs.value < rhs.value` makes them (itm-common.hpp:139-144).
    double d = (double)v;
    if (d == 0.0)
        d = 0.0;
    unsigned long long b = (unsigned long long)__double_as_longlong(d);
    b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    return maximize ? ~b : b;
}

// c(j, x) of quadratic_cost_type (itm-common.hpp:1413-1425): the linear cost
// plus the factor of every quadratic partner whose x bit is set, in the
// order of the adjacency list.  `lin` is the linear cost of j.
template<typename Float>
__device__ inline Float
cost_x(const DevProblem& pb, const unsigned* x_s, int j, Float lin)
{
    if (pb.q_ptr) {
        const Float* qf = (const Float*)pb.q_factor;
        for (int f = pb.q_ptr[j], l = pb.q_ptr[j + 1]; f != l; ++f) {
            const int o = pb.q_id[f];
            if ((x_s[o >> 5] >> (o & 31)) & 1u)
                lin += qf[f];
        }
    }
    return lin;
}

// Inverse of order_key (the value's sign of zero is lost: +0.0 comes back,
// which no arithmetic of affect() can tell from -0.0 when the row has no
// equal reduced costs).
template<typename Float>
__device__ inline Float
key_value(unsigned long long k, bool maximize)
{
    if (maximize)
        k = ~k;
    k = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return (Float)__longlong_as_double((long long)k);
}

// Lane exchange inside a row of 16 lanes with DPP (no LDS round trip):
// 0xB1 quad_perm(1,0,3,2), 0x4E quad_perm(2,3,0,1), 0x141 row_half_mirror,
// 0x140 row_mirror.  After those four steps of an idempotent reduction every
// lane of a row holds the row's result; rows are combined with two
// ds_bpermute steps (xor 16, xor 32).
template<int CTRL>
__device__ inline unsigned long long
dpp_u64(unsigned long long v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// quad_perm broadcast inside lane pairs: 0xA0 = [0,0,2,2] (every lane takes the even
// lane's value), 0xF5 = [1,1,3,3] (the odd lane's)
template<int CTRL, typename Float>
__device__ inline Float
pair_bcast(Float v)
{
    if constexpr (sizeof(Float) == 8) {
        const unsigned long long r = dpp_u64<CTRL>((unsigned long long)__double_as_longlong((double)v));
        return (Float)__longlong_as_double((long long)r);
    } else {
        const int r = __builtin_amdgcn_update_dpp(0, __float_as_int((float)v), CTRL, 0xf, 0xf, false);
        return (Float)__int_as_float(r);
    }
}

#define BXH_WAVE_REDUCE(V, OP)                                                 \
    do {                                                                       \
        unsigned long long t_;                                                 \
        t_ = dpp_u64<0xB1>(V);                                                 \
        V = OP(t_, V);                                                         \
        t_ = dpp_u64<0x4E>(V);                                                 \
        V = OP(t_, V);                                                         \
        t_ = dpp_u64<0x141>(V);                                                \
        V = OP(t_, V);                                                         \
        t_ = dpp_u64<0x140>(V);                                                \
        V = OP(t_, V);                                                         \
        t_ = __shfl_xor(V, 16, 64);                                            \
        V = OP(t_, V);                                                         \
        t_ = __shfl_xor(V, 32, 64);                                            \
        V = OP(t_, V);                                                         \
    } while (0)

// Reduction inside each row of 16 lanes only (DPP, no LDS traffic): the four
// row results are then combined by LDS atomics of lanes 0, 16, 32, 48.
#define BXH_ROW_REDUCE(V, OP)                                                  \
    do {                                                                       \
        unsigned long long t_;                                                 \
        t_ = dpp_u64<0xB1>(V);                                                 \
        V = OP(t_, V);                                                         \
        t_ = dpp_u64<0x4E>(V);                                                 \
        V = OP(t_, V);                                                         \
        t_ = dpp_u64<0x141>(V);                                                \
        V = OP(t_, V);                                                         \
        t_ = dpp_u64<0x140>(V);                                                \
        V = OP(t_, V);                                                         \
    } while (0)

__device__ inline unsigned long long
umin64(unsigned long long a, unsigned long long b)
{
    return a < b ? a : b;
}
__device__ inline unsigned long long
umax64(unsigned long long a, unsigned long long b)
{
    return a > b ? a : b;
}

__device__ inline unsigned long long
wave_min_u64(unsigned long long v)
{
    BXH_WAVE_REDUCE(v, umin64);
    return v;
}

__device__ inline unsigned long long
wave_max_u64(unsigned long long v)
{
    BXH_WAVE_REDUCE(v, umax64);
    return v;
}

__device__ inline unsigned long long
bcast_first_u64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Wave-wide minima of 32-bit values on the DPP path only (no LDS): four
// butterfly steps inside each row of 16 lanes, then row_bcast:15 / row_bcast:31
// carry the row results up to row 3, and lane 63 is read back as a wave-uniform
// scalar.  v_min_u32_dpp folds the lane exchange into the min.  A DPP source
// written by the previous VALU instruction needs two wait states: the variant
// with four independent chains interleaves them (three instructions between
// two steps of one chain), the single chain pays an s_nop per step.  Every
// lane of the wave must be active.
#define BXH_DPP1(R, CTRL) "v_min_u32_dpp " R ", " R ", " R " " CTRL "\n"
#define BXH_DPP4(CTRL) BXH_DPP1("%0", CTRL) BXH_DPP1("%1", CTRL) BXH_DPP1("%2", CTRL) BXH_DPP1("%3", CTRL)
__device__ inline void
wave_umin32x4(unsigned& a, unsigned& b, unsigned& c, unsigned& d)
{
    asm volatile("s_nop 4\n" BXH_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                   BXH_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                     BXH_DPP4("row_half_mirror row_mask:0xf bank_mask:0xf")
                       BXH_DPP4("row_mirror row_mask:0xf bank_mask:0xf")
                         BXH_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                           BXH_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1\n"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    a = (unsigned)__builtin_amdgcn_readlane((int)a, 63);
    b = (unsigned)__builtin_amdgcn_readlane((int)b, 63);
    c = (unsigned)__builtin_amdgcn_readlane((int)c, 63);
    d = (unsigned)__builtin_amdgcn_readlane((int)d, 63);
}

__device__ inline unsigned
wave_umin32(unsigned a)
{
    asm volatile("s_nop 4\n" BXH_DPP1("%0", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") "s_nop 1\n"
                   BXH_DPP1("%0", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") "s_nop 1\n"
                     BXH_DPP1("%0", "row_half_mirror row_mask:0xf bank_mask:0xf") "s_nop 1\n"
                       BXH_DPP1("%0", "row_mirror row_mask:0xf bank_mask:0xf") "s_nop 1\n"
                         BXH_DPP1("%0", "row_bcast:15 row_mask:0xa bank_mask:0xf") "s_nop 1\n"
                           BXH_DPP1("%0", "row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1\n"
                 : "+v"(a));
    return (unsigned)__builtin_amdgcn_readlane((int)a, 63);
}
#undef BXH_DPP4
#undef BXH_DPP1

// 64-bit minima: the high words first, then the low words of the lanes whose
// high word is the minimum.  Results are wave-uniform.
__device__ inline void
wave_umin64x4(unsigned long long& a, unsigned long long& b, unsigned long long& c,
              unsigned long long& d)
{
    unsigned ah = (unsigned)(a >> 32), bh = (unsigned)(b >> 32), ch = (unsigned)(c >> 32),
             dh = (unsigned)(d >> 32);
    unsigned mah = ah, mbh = bh, mch = ch, mdh = dh;
    wave_umin32x4(mah, mbh, mch, mdh);
    unsigned al = ah == mah ? (unsigned)a : ~0u, bl = bh == mbh ? (unsigned)b : ~0u,
             cl = ch == mch ? (unsigned)c : ~0u, dl = dh == mdh ? (unsigned)d : ~0u;
    wave_umin32x4(al, bl, cl, dl);
    a = ((unsigned long long)mah << 32) | al;
    b = ((unsigned long long)mbh << 32) | bl;
    c = ((unsigned long long)mch << 32) | cl;
    d = ((unsigned long long)mdh << 32) | dl;
}

__device__ inline unsigned long long
wave_umin64(unsigned long long a)
{
    const unsigned ah = (unsigned)(a >> 32);
    const unsigned mah = wave_umin32(ah);
    const unsigned mal = wave_umin32(ah == mah ? (unsigned)a : ~0u);
    return ((unsigned long long)mah << 32) | mal;
}

__device__ inline unsigned long long
uniform_u64(unsigned long long v) // a value every lane holds -> scalar registers
{
    return bcast_first_u64(v);
}

// Tie filter: an LDS bitmap of 2^16 bits; a key sets three bits of ONE 32-bit
// word (word and positions from its hash) with a single atomic OR.  A key that
// finds them all already set MAY equal an earlier key of the row; of two equal
// keys the later one always does (one atomic per key: no ordering between
// bits).  At 500 keys about one row in 40 flags a key that is no duplicate.
__device__ inline bool
filter_insert(unsigned* bitmap, unsigned long long key)
{
    unsigned x = (unsigned)(key >> 32) * 0x9E3779B1u + (unsigned)key * 0x85EBCA77u;
    x ^= x >> 15;
    x *= 0xC2B2AE3Du;
    x ^= x >> 13;
    const unsigned word = x >> 21;                      // 2048 words
    const unsigned bits = (1u << (x & 31)) | (1u << ((x >> 5) & 31)) | (1u << ((x >> 10) & 31));
    return (atomicOr(&bitmap[word], bits) & bits) == bits;
}

// The same in two halves: the atomic is issued here, what it returned is looked at by
// filter_hit() -- later, so that its round trip runs under other work.
__device__ inline unsigned
filter_issue(unsigned* bitmap, unsigned long long key, unsigned& bits)
{
    unsigned x = (unsigned)(key >> 32) * 0x9E3779B1u + (unsigned)key * 0x85EBCA77u;
    x ^= x >> 15;
    x *= 0xC2B2AE3Du;
    x ^= x >> 13;
    const unsigned word = x >> 21;
    bits = (1u << (x & 31)) | (1u << ((x >> 5) & 31)) | (1u << ((x >> 10) & 31));
    return atomicOr(&bitmap[word], bits);
}
__device__ inline bool
filter_hit(unsigned old, unsigned bits)
{
    return (old & bits) == bits;
}

// Open-addressing insert of a non-zero key; returns true when the key was
// already there (two equal reduced costs in the row).
__device__ inline bool
hash_insert(unsigned long long* tab, unsigned mask, unsigned long long key)
{
    unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 40) & mask;
    for (;;) {
        const unsigned long long old = atomicCAS(&tab[h], 0ull, key);
        if (old == 0ull)
            return false;
        if (old == key)
            return true;
        h = (h + 1) & mask;
    }
}

// In-place bitonic sort of (key, id) pairs in LDS; n2 is a power of two.
// All pairs are distinct (ids are), so the result is the unique sorted order:
// "by value, ties by row-local id" -- the canonical tie rule of DESIGN.md.
__device__ inline void
bitonic_sort(unsigned long long* key, int* id, int n2)
{
    const int tid = threadIdx.x, T = blockDim.x;
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool asc = (i & k) == 0;
                const unsigned long long ka = key[i], kb = key[l];
                const int ia = id[i], ib = id[l];
                const bool gt = (ka > kb) || (ka == kb && ia > ib);
                if (gt == asc) {
                    key[i] = kb;
                    key[l] = ka;
                    id[i] = ib;
                    id[l] = ia;
                }
            }
            __syncthreads();
        }
    }
}

// Sort of n <= 2 * blockDim (and <= 1024) (key, id) pairs: every wave sorts
// runs of 64 pairs in registers (bitonic over lane shuffles, no barrier), the
// runs go to LDS, and each pair finds its final rank as the sum over the other
// runs of a 6-step binary search (pairs are distinct, so ranks are too).  Two
// barriers instead of the ~50 of an LDS bitonic sort.  `key_out/id_out`
// receive the sorted sequence; `key_run/id_run` are scratch of the same size.
struct KeyId
{
    unsigned long long key;
    int id;
};

__device__ inline bool
pair_less(unsigned long long ka, int ia, unsigned long long kb, int ib)
{
    return ka < kb || (ka == kb && ia < ib);
}

template<typename Float>
__device__ __attribute__((noinline)) void
run_merge_sort(const Float* val_by_id, // [n] reduced cost of row-local id
               int n,
               bool maximize,
               unsigned long long* key_run,
               int* id_run,
               unsigned long long* key_out,
               int* id_out,
               Prof* prof)
{
    (void)prof;
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63;
    const int nruns = (n + 63) >> 6;

    unsigned long long mk[2];
    int mi[2];
    int mrun[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int e = tid + s * T; // element index == id before the sort
        mrun[s] = e >> 6;
        if (e < n) {
            mk[s] = order_key<Float>(val_by_id[e], maximize);
            mi[s] = e;
        } else {
            mk[s] = ~0ull;
            mi[s] = 0x40000000 + e;
        }
    }
    // mrun[] is wave-uniform; the second slot only exists for n > blockDim
    const bool two = mrun[1] < nruns;
    for (int k = 2; k <= 64; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool lower = (lane & j) == 0;
            const bool asc = (lane & k) == 0;
            const unsigned long long ok0 = __shfl_xor(mk[0], j, 64);
            const int oi0 = __shfl_xor(mi[0], j, 64);
            unsigned long long ok1 = 0;
            int oi1 = 0;
            if (two) {
                ok1 = __shfl_xor(mk[1], j, 64);
                oi1 = __shfl_xor(mi[1], j, 64);
            }
            {
                const bool mine_less = pair_less(mk[0], mi[0], ok0, oi0);
                if (!((lower == asc) ? mine_less : !mine_less)) {
                    mk[0] = ok0;
                    mi[0] = oi0;
                }
            }
            if (two) {
                const bool mine_less = pair_less(mk[1], mi[1], ok1, oi1);
                if (!((lower == asc) ? mine_less : !mine_less)) {
                    mk[1] = ok1;
                    mi[1] = oi1;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (mrun[s] < nruns) {
            key_run[mrun[s] * 64 + lane] = mk[s];
            id_run[mrun[s] * 64 + lane] = mi[s];
        }
    }
    BXH_PSTAMP(prof, 8);
    __syncthreads();
    BXH_PSTAMP(prof, 9);

#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (mrun[s] < nruns && mi[s] < 0x40000000) {
            int rank = lane;
            // the searches in the other runs are independent: 8 at a time, so
            // each of the 7 probe rounds issues 8 LDS reads back to back
            for (int q0 = 0; q0 < nruns; q0 += 8) {
                int lo[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    lo[u] = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    unsigned long long kq[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int q = (q0 + u < nruns) ? q0 + u : q0;
                        kq[u] = key_run[q * 64 + lo[u] + step - 1];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int q = (q0 + u < nruns) ? q0 + u : q0;
                        const int iq = id_run[q * 64 + lo[u] + step - 1];
                        const bool less =
                          kq[u] < mk[s] || (kq[u] == mk[s] && iq < mi[s]);
                        lo[u] += less ? step : 0;
                    }
                }
                {
                    unsigned long long kq[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int q = (q0 + u < nruns) ? q0 + u : q0;
                        kq[u] = key_run[q * 64 + lo[u]];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int q = (q0 + u < nruns) ? q0 + u : q0;
                        const int iq = id_run[q * 64 + lo[u]];
                        const bool less =
                          kq[u] < mk[s] || (kq[u] == mk[s] && iq < mi[s]);
                        const int cnt = (lo[u] == 63 && less) ? 64 : lo[u];
                        rank += (q0 + u < nruns && q != mrun[s]) ? cnt : 0;
                    }
                }
            }
            key_out[rank] = mk[s];
            id_out[rank] = mi[s];
        }
    }
    BXH_PSTAMP(prof, 10);
    __syncthreads();
    BXH_PSTAMP(prof, 11);
}

__device__ inline int
wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

// compute_violated_constraints (itm-common.hpp:102-115): rebuild the ordered
// list of rows whose activity is outside [bmin, bmax].  x is read from LDS.
// First the flags of rows [k0, k1) (wave per row), then the ordered compaction.
__device__ inline int
flag_violated(const DevProblem& pb, const unsigned* x_s, signed char* vflag, int k0, int k1)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
    const bool pm1 = pb.kind == 2 || pb.kind == 3;
    if ((long long)pb.nnz <= 64ll * pb.m) {
        // Short rows: a THREAD per row, eight elements per round trip.  (A wave per row
        // is two dependent loads of latency per row and wave -- 0.3 ms for 1000 rows of
        // 40 elements, a twelfth of a sweep of the Z kernel -- with 60 idle lanes.)
        for (int k = k0 + tid; k < k1; k += T) {
            const int rs = pb.row_ptr[k], re = pb.row_ptr[k + 1];
            int v = 0;
            for (int e0 = rs; e0 < re; e0 += 8) {
                int j8[8], f8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u < re ? e0 + u : rs;
                    j8[u] = pb.row_col[e];
                    f8[u] = pb.kind == 4 ? pb.fac32[e] : (pm1 ? (int)pb.fac_csr[e] : 1);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int bit = (x_s[j8[u] >> 5] >> (j8[u] & 31)) & 1u;
                    v += e0 + u < re ? f8[u] * bit : 0;
                }
            }
            vflag[k] = !(pb.bmin[k] <= v && v <= pb.bmax[k]);
        }
        return 0;
    }
    for (int k = k0 + wave; k < k1; k += nwaves) {
        const int rs = pb.row_ptr[k], re = pb.row_ptr[k + 1];
        int v = 0;
        for (int e = rs + lane; e < re; e += 64) {
            const int j = pb.row_col[e];
            const int bit = (x_s[j >> 5] >> (j & 31)) & 1u;
            v += pb.kind == 4 ? pb.fac32[e] * bit : (pm1 ? (int)pb.fac_csr[e] * bit : bit);
        }
        v = wave_sum(v);
        if (lane == 0)
            vflag[k] = !(pb.bmin[k] <= v && v <= pb.bmax[k]);
    }
    return 0;
}

__device__ inline int
rebuild_violated(const DevProblem& pb,
                 const unsigned* x_s,
                 signed char* vflag,
                 int* vlist,
                 int* scratch_s /* >= 64 ints */,
                 bool flags_ready = false)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;

    if (!flags_ready)
        flag_violated(pb, x_s, vflag, 0, pb.m);
    __syncthreads(); // vflag is only read back by this workgroup (same CU)

    int base = 0;
    for (int c = 0; c < pb.m; c += T) {
        const int k = c + tid;
        const bool f = k < pb.m && vflag[k] != 0;
        const unsigned long long mask = __ballot(f);
        const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0)
            scratch_s[wave] = __popcll(mask);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < nwaves; ++w) {
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
    return base;
}

// Sort of a short row (n <= blockDim.x, a few dozen elements) by counting: every
// element reads all keys (LDS broadcast reads, independent of each other) and
// its rank is the number of pairs (key, id) below its own -- the same order as
// the bitonic / run-merge sorts (by value, ties by row-local id), without their
// dependent compare-exchange steps.  key_tmp: n u64 of scratch.
template<typename Float>
__device__ inline void
rank_sort_small(const Float* val_by_id,
                int n,
                bool maximize,
                unsigned long long* key_tmp,
                unsigned long long* key_out,
                int* id_out)
{
    const int tid = threadIdx.x;
    unsigned long long mine = ~0ull;
    const int n8 = (n + 7) & ~7; // (key_tmp holds whole 64-entry runs: the padding fits)
    if (tid < n8) {
        if (tid < n)
            mine = order_key<Float>(val_by_id[tid], maximize);
        key_tmp[tid] = mine; // padding: the largest key, behind every element
    }
    __syncthreads();
    // eight keys per round trip to LDS (uniform addresses: broadcast reads).
    // First by the keys alone (one compare per pair): without two equal keys the
    // counts are the ranks; equal keys land on one position, which the loser
    // of the two stores sees -- the row then counts again with the ids.
    int rank = 0;
    if (tid < n) {
        for (int j0 = 0; j0 < n8; j0 += 8) {
            ulonglong2 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                q[u] = reinterpret_cast<const ulonglong2*>(key_tmp + j0)[u];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rank += q[u].x < mine;
                rank += q[u].y < mine;
            }
        }
        key_out[rank] = mine;
        id_out[rank] = tid;
    }
    __syncthreads();
    const bool lost = tid < n && id_out[rank] != tid;
    if (__syncthreads_or(lost)) {
        if (tid < n) {
            rank = 0;
            for (int j0 = 0; j0 < n8; j0 += 8) {
                ulonglong2 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    q[u] = reinterpret_cast<const ulonglong2*>(key_tmp + j0)[u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    rank += (q[u].x < mine) || (q[u].x == mine && j0 + 2 * u < tid);
                    rank += (q[u].y < mine) || (q[u].y == mine && j0 + 2 * u + 1 < tid);
                }
            }
            key_out[rank] = mine;
            id_out[rank] = tid;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// random_shuffle_unique (itm-common.hpp:117-131) for a whole workgroup: after
// the sort every run of equal keys is shuffled with std::shuffle and the
// row's engine.  The engine is a linear congruential one, so draw number g is
// lcg_pow[g] * state (mod 2^31 - 1) and every position of every run can work out
// its own swap partner at once (libstdc++'s pairwise scheme: one draw decides
// two swaps, uniform_int_distribution's downscaling).  The Fisher-Yates swaps
// of a run are then applied by tracing, for every final position, where its
// element came from (backwards through the run's swaps).  A rejected draw
// (probability below 1e-4 per draw) shifts every later draw: the function then
// returns false, has changed nothing, and the caller takes the serial path.
// Needs r <= 2 * blockDim.x <= 1024.  ids are permuted in place; *state is
// advanced by the number of draws.  Scratch: three int arrays of r entries.
__device__ inline unsigned
lcg_jump(const unsigned* lcg_pow, unsigned state, int draws)
{
    unsigned long long x = (unsigned long long)lcg_pow[draws] * state; // < 2^62
    x = (x & 0x7fffffffull) + (x >> 31);
    x = (x & 0x7fffffffull) + (x >> 31);
    return (unsigned)(x >= 0x7fffffffull ? x - 0x7fffffffull : x);
}

// (not inlined: the sort path is cold in tie-free sweeps, and its registers
// would otherwise count against the row loop)
__device__ __attribute__((noinline)) bool
parallel_tie_shuffle(const unsigned long long* key_s,
                     int* id_s,
                     int r,
                     unsigned* state, // in / out (uniform)
                     const unsigned* lcg_pow,
                     int* start_s,
                     int* jt_s,
                     int* tmp_s,
                     int* scratch_s /* >= 40 ints */)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = T >> 6;
    const int p0 = 2 * tid, p1 = 2 * tid + 1;
    const bool in0 = p0 < r, in1 = p1 < r;

    // ---- run starts: inclusive max-scan of the head positions --------------
    const bool h0 = in0 && (p0 == 0 || key_s[p0] != key_s[p0 - 1]);
    const bool h1 = in1 && key_s[p1] != key_s[p0];
    int m0 = h0 ? p0 : -1;
    int m1 = h1 ? p1 : m0;
    int agg = m1; // of the pair
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(agg, off, 64);
        if (lane >= off)
            agg = max(agg, o);
    }
    if (lane == 63)
        scratch_s[wave] = agg;
    if (tid == 0)
        scratch_s[32] = 0; // rejection flag
    __syncthreads();
    int before = -1; // max over the earlier threads
    {
        const int up = __shfl_up(agg, 1, 64);
        if (lane > 0)
            before = up;
        for (int w = 0; w < wave; ++w)
            before = max(before, scratch_s[w]);
    }
    const int s0 = max(m0, before), s1 = max(m1, before);
    if (in0)
        start_s[p0] = s0;
    if (in1)
        start_s[p1] = s1;
    // the last element of a run publishes its length at the run's start
    const bool l0 = in0 && (p0 + 1 == r || (in1 ? h1 : true));
    if (l0)
        tmp_s[s0] = p0 - s0 + 1;
    __syncthreads();
    const bool l1 = in1 && (p1 + 1 == r || key_s[p1 + 1] != key_s[p1]);
    if (l1)
        tmp_s[s1] = p1 - s1 + 1;
    __syncthreads();
    const int L0 = in0 ? tmp_s[s0] : 1, L1 = in1 ? tmp_s[s1] : 1;

    // ---- draw numbers: exclusive prefix sum of floor(L / 2) over the runs ---
    const int d0 = (h0 ? (L0 >> 1) : 0), d1 = (h1 ? (L1 >> 1) : 0);
    int sum = d0 + d1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(sum, off, 64);
        if (lane >= off)
            sum += o;
    }
    __syncthreads(); // scratch_s[0 .. nw) is reused
    if (lane == 63)
        scratch_s[wave] = sum;
    __syncthreads();
    int excl = sum - (d0 + d1);
    int total = 0;
    for (int w = 0; w < nw; ++w) {
        if (w < wave)
            excl += scratch_s[w];
        total += scratch_s[w];
    }
    // draws before the run of p0 / p1: a head carries its own run's base
    const int base0 = excl, base1 = excl + d0;
    if (h0)
        jt_s[p0] = base0; // base of the run, read through start_s below
    if (h1)
        jt_s[p1] = base1;
    __syncthreads();
    const int rb0 = in0 ? jt_s[s0] : 0, rb1 = in1 ? jt_s[s1] : 0;
    __syncthreads();

    // ---- the swap partner of every position i >= 1 of a run -----------------
    const unsigned st0 = *state;
    bool reject = false;
    auto partner = [&](int p, int sp, int L, int rbase) -> int {
        const int i = p - sp;
        if (L < 2 || i == 0)
            return 0;
        const bool even = (L & 1) == 0;
        unsigned urange; // the draw is uniform over [0, urange]
        int d, half;
        unsigned b1 = 2;
        if (even && i == 1) {
            d = 0;
            half = 0;
            urange = 1u;
            b1 = 2u; // result / 1 ... handled below
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
        const unsigned raw = lcg_jump(lcg_pow, st0, rbase + d + 1) - 1u;
        if (raw >= past)
            reject = true;
        const unsigned x = raw / scaling;
        if (even && i == 1)
            return (int)x; // uniform(1)
        return half == 0 ? (int)(x / b1) : (int)(x % b1);
    };
    const int j0 = in0 ? partner(p0, s0, L0, rb0) : 0;
    const int j1 = in1 ? partner(p1, s1, L1, rb1) : 0;
    if (reject)
        scratch_s[32] = 1;
    if (in0)
        jt_s[p0] = j0;
    if (in1)
        jt_s[p1] = j1;
    __syncthreads();
    if (scratch_s[32])
        return false;

    // ---- where the element of every final position came from ----------------
    auto origin = [&](int p, int sp, int L) -> int {
        int pos = p - sp;
        for (int i = L - 1; i >= 1; --i) {
            const int j = jt_s[sp + i];
            pos = pos == i ? j : (pos == j ? i : pos);
        }
        return sp + pos;
    };
    const int o0 = in0 ? id_s[origin(p0, s0, L0)] : 0;
    const int o1 = in1 ? id_s[origin(p1, s1, L1)] : 0;
    __syncthreads();
    if (in0)
        id_s[p0] = o0;
    if (in1)
        id_s[p1] = o1;
    *state = lcg_jump(lcg_pow, st0, total);
    __syncthreads();
    return true;
}

// LDS carve shared by the kernels (host computes the same sizes).
struct LdsPlan
{
    int off_pi, off_x, off_val, off_key, off_id, off_stale, off_misc, total;
};

__host__ __device__ inline int
next_pow2(int v)
{
    int p = 1;
    while (p < v)
        p <<= 1;
    return p;
}

__host__ __device__ inline LdsPlan
lds_plan(int m, int nwords, int rmax, int fsize)
{
    auto al = [](int v) { return (v + 15) & ~15; };
    LdsPlan p;
    int o = 0;
    p.off_pi = o;
    o = al(o + m * fsize);
    p.off_x = o;
    o = al(o + nwords * 8);
    p.off_val = o;
    o = al(o + rmax * fsize);
    const int n2 = next_pow2(rmax < 2 ? 2 : rmax);
    p.off_key = o;
    o = al(o + n2 * 8);
    p.off_id = o;
    o = al(o + n2 * 4);
    p.off_stale = o;
    o = al(o + (rmax + 1) * fsize);
    p.off_misc = o;
    o = al(o + 96 * 4);
    p.total = o;
    return p;
}

// misc_s slots
enum
{
    MS_SEL = 0,
    MS_TIE = 1,
    MS_PICH = 2,
    MS_CNT = 3,
    MS_BMIN = 6,   // row constants fetched with the row descriptor
    MS_BMAX = 7,
    MS_BNEG = 12,
    MS_RBYTES = 14, // double (8 bytes)
    MS_RNG = 13,   // engine state after a sort-path row (all threads re-read it)
    MS_FAST = 4,   // 1: the row was resolved without sorting
    MS_SELMODE = 5, // 0: none up, 1: all up, 2: up iff key <= threshold
    MS_D = 8,      // Float d (8 bytes reserved)
    MS_THR = 10,   // u64 threshold key (8 bytes)
    MS_WAVES = 16, // 64 ints of scratch (also the per-wave partials below)
    MS_LCNT = 80,  // pipelined sweep: late entries of the row in either descriptor buffer (2 ints)
    MS_DEAD = 82   // pipelined sweep: a partner timed out, stop polling
};

// Row order of an instance this sweep: 0 none, 1 reversing, 2 random_sorting,
// 3 / 4 infeasibility decr / incr, 5 / 6 lagrangian decr / incr, 7
// pi_sign_change (solver_parameters::constraint_order, core:286-296; `cycle`
// is resolved to one of these by the host, itm-common.hpp:640-658).
__device__ inline int
order_of(const DevSweep& a, int inst)
{
    return a.order_b ? a.order_b[inst] : a.order;
}

// infeasibility_local_compute_violated_constraints (itm-common.hpp:893-914):
// m_order = {k, how far row k is from its bounds} for the violated rows, in
// increasing k.  Returns their number.
__device__ inline int
rebuild_infeasibility(const DevProblem& pb,
                      const unsigned* x_s,
                      int* vamt,
                      int2* mlist,
                      int* scratch_s /* >= 64 ints */)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
    const bool pm1 = pb.kind == 2 || pb.kind == 3;

    for (int k = wave; k < pb.m; k += nwaves) {
        const int rs = pb.row_ptr[k], re = pb.row_ptr[k + 1];
        int v = 0;
        for (int e = rs + lane; e < re; e += 64) {
            const int j = pb.row_col[e];
            const int bit = (x_s[j >> 5] >> (j & 31)) & 1u;
            v += pb.kind == 4 ? pb.fac32[e] * bit : (pm1 ? (int)pb.fac_csr[e] * bit : bit);
        }
        v = wave_sum(v);
        if (lane == 0) {
            int amt = 0;
            if (pb.bmin[k] > v)
                amt = pb.bmin[k] - v;
            else if (pb.bmax[k] < v)
                amt = v - pb.bmax[k];
            vamt[k] = amt;
        }
    }
    __syncthreads(); // vamt is only read back by this workgroup (same CU)

    int base = 0;
    for (int c = 0; c < pb.m; c += T) {
        const int k = c + tid;
        const int amt = k < pb.m ? vamt[k] : 0;
        const bool f = amt != 0;
        const unsigned long long mask = __ballot(f);
        const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0)
            scratch_s[wave] = __popcll(mask);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < nwaves; ++w) {
            const int cw = scratch_s[w];
            if (w < wave)
                off += cw;
            tot += cw;
        }
        if (f)
            mlist[base + off + prefix] = make_int2(k, amt);
        base += tot;
        __syncthreads();
    }
    return base;
}

// compute_order::init (itm-common.hpp:660-680) of the instances `active`
// selects (all when null), by each instance's current order.
// First half of compute_violated_constraints on `parts` workgroups per
// instance (blockIdx = inst * parts + part): which rows are violated.
__global__ __launch_bounds__(1024) void
order_flags_kernel(DevProblem pb, DevState st, const int* order_b, const unsigned char* active,
                   int parts)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned* x_s = (unsigned*)smem;
    const int inst = blockIdx.x / parts, part = blockIdx.x % parts;
    const int tid = threadIdx.x, T = blockDim.x;
    if (active && !active[inst])
        return;
    const int ord = order_b ? order_b[inst] : 0;
    if (ord == 3 || ord == 4)
        return; // the infeasibility orders compute amounts, in order_init_kernel
    const unsigned* xg = st.x + (size_t)inst * (2 * pb.nwords);
    for (int i = tid; i < 2 * pb.nwords; i += T)
        x_s[i] = xg[i];
    __syncthreads();
    const int per = (pb.m + parts - 1) / parts;
    const int k0 = part * per, k1 = min(pb.m, k0 + per);
    flag_violated(pb, x_s, st.vflag + (size_t)inst * pb.m, k0, k1);
}

template<typename Float>
__global__ __launch_bounds__(1024) void
order_init_kernel(DevProblem pb, DevState st, const int* order_b, const unsigned char* active,
                  int flags_ready)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const LdsPlan lp = lds_plan(pb.m, pb.nwords, pb.rmax, (int)sizeof(Float));
    unsigned* x_s = (unsigned*)(smem + lp.off_x);
    int* misc_s = (int*)(smem + lp.off_misc);
    const int inst = blockIdx.x;
    const int tid = threadIdx.x, T = blockDim.x;
    if (active && !active[inst])
        return;
    const int ord = order_b ? order_b[inst] : 0;

    const unsigned* xg = st.x + (size_t)inst * (2 * pb.nwords);
    for (int i = tid; i < 2 * pb.nwords; i += T)
        x_s[i] = xg[i];
    __syncthreads();

    int* vlist = st.vlist + (size_t)inst * st.lcap;
    if (ord == 3 || ord == 4) {
        const int cnt = rebuild_infeasibility(pb,
                                              x_s,
                                              st.vamt + (size_t)inst * pb.m,
                                              st.mlist + (size_t)inst * st.lcap,
                                              misc_s + MS_WAVES);
        if (tid == 0) {
            st.mcount[inst] = cnt;
            st.status[inst].remaining = cnt;
        }
        return;
    }
    if (ord == 7) { // std::iota over `variables` entries (itm-common.hpp:672-674)
        for (int i = tid; i < pb.n; i += T)
            vlist[i] = i;
        // remaining as solver_functor would first see it
        const int cnt = rebuild_violated(pb,
                                         x_s,
                                         st.vflag + (size_t)inst * pb.m,
                                         st.slist + (size_t)inst * st.lcap,
                                         misc_s + MS_WAVES,
                                         flags_ready != 0);
        if (tid == 0) {
            st.vcount[inst] = pb.n;
            st.status[inst].remaining = cnt;
        }
        return;
    }
    const int cnt = rebuild_violated(
      pb, x_s, st.vflag + (size_t)inst * pb.m, vlist, misc_s + MS_WAVES, flags_ready != 0);
    if (tid == 0) {
        st.vcount[inst] = cnt;
        st.status[inst].remaining = cnt;
    }
}

// The sorts of compute_order::run / push_and_run (itm-common.hpp:690-760,
// 808-850) ahead of the row loop: m_order by infeasibility (3 / 4), R by pi
// (5 / 6).  std::sort there is unstable; ties keep their list position here
// (the canonical tie rule of DESIGN.md, the oracle's SORT_CANONICAL).
template<typename Float>
__global__ __launch_bounds__(1024) void
order_prep_kernel(DevProblem pb, DevState st, DevSweep a)
{
    const int inst = blockIdx.x;
    const int tid = threadIdx.x, T = blockDim.x;
    if (a.active && !a.active[inst])
        return;
    const int ord = order_of(a, inst);
    if (ord < 3 || ord > 6)
        return;
    const bool by_m = ord <= 4;
    const int n = by_m ? st.mcount[inst] : st.vcount[inst];
    if (n < 1)
        return;
    const int n2 = next_pow2(n < 2 ? 2 : n);
    const int n2cap = next_pow2(st.lcap < 2 ? 2 : st.lcap);
    unsigned long long* key = st.skey + (size_t)inst * n2cap;
    int* id = st.sid + (size_t)inst * n2cap;
    int2* mlist = st.mlist + (size_t)inst * st.lcap;
    int2* mtmp = st.mtmp + (size_t)inst * st.lcap;
    int* mrun = st.mrun + (size_t)inst * st.lcap;
    int* vlist = st.vlist + (size_t)inst * st.lcap;
    const Float* pig = (const Float*)st.pi + (size_t)inst * pb.m;

    for (int i = tid; i < n2; i += T) {
        unsigned long long k = ~0ull;
        if (i < n) {
            if (by_m) {
                const int2 e = mlist[i];
                mtmp[i] = e;
                const unsigned amt = (unsigned)e.y;
                k = ord == 3 ? (unsigned long long)(0xffffffffu - amt) : (unsigned long long)amt;
            } else {
                const int row = vlist[i];
                mrun[i] = row;
                // `pi[r] < pi[l]` first (decr) == ascending on the reversed key
                k = order_key<Float>(pig[row], ord == 5);
                if (k == ~0ull)
                    k = ~0ull - 1; // padding stays last
            }
        }
        key[i] = k;
        id[i] = i;
    }
    __syncthreads();
    bitonic_sort(key, id, n2);
    for (int i = tid; i < n; i += T) {
        if (by_m) {
            const int2 e = mtmp[id[i]];
            mlist[i] = e;
            mrun[i] = e.x;
        } else {
            vlist[i] = mrun[id[i]];
        }
    }
}

// After a sweep in an infeasibility order: m_order of the new x
// (itm-common.hpp:720-722, 832-834).
template<typename Float>
__global__ __launch_bounds__(1024) void
order_post_kernel(DevProblem pb, DevState st, DevSweep a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const LdsPlan lp = lds_plan(pb.m, pb.nwords, pb.rmax, (int)sizeof(Float));
    unsigned* x_s = (unsigned*)(smem + lp.off_x);
    int* misc_s = (int*)(smem + lp.off_misc);
    const int inst = blockIdx.x;
    const int tid = threadIdx.x, T = blockDim.x;
    if (a.active && !a.active[inst])
        return;
    const int ord = order_of(a, inst);
    if (ord != 3 && ord != 4)
        return;
    const unsigned* xg = st.x + (size_t)inst * (2 * pb.nwords);
    for (int i = tid; i < 2 * pb.nwords; i += T)
        x_s[i] = xg[i];
    __syncthreads();
    const int cnt = rebuild_infeasibility(pb,
                                          x_s,
                                          st.vamt + (size_t)inst * pb.m,
                                          st.mlist + (size_t)inst * st.lcap,
                                          misc_s + MS_WAVES);
    if (tid == 0)
        st.mcount[inst] = cnt;
}

// One sweep of one instance: compute_order::run / push_and_run
// (itm-common.hpp:682-880) around compute_update_row /
// push_and_compute_update_row of the 01 and 101 solvers
// (itm-solver-inequalities-01.cpp:208-296, -equalities-01.cpp:184-270,
//  -equalities-101.cpp:208-307, -inequalities-101.cpp:250-347).
template<typename Float, bool PM1>
__global__ __launch_bounds__(1024) void
sweep_kernel(DevProblem pb, DevState st, DevSweep a)
{
#if defined(BXH_DEV_ONLY_NP) && !defined(BXH_DEV_KEEP_V1)
    return; // (development builds of another kernel: see sweep_kernel_v2)
#endif

    extern __shared__ __align__(16) unsigned char smem[];
    const int inst = blockIdx.x;
    if (a.active && !a.active[inst])
        return;

    const LdsPlan lp = lds_plan(pb.m, pb.nwords, pb.rmax, (int)sizeof(Float));
    Float* pi_s = (Float*)(smem + lp.off_pi);
    unsigned* x_s = (unsigned*)(smem + lp.off_x);
    Float* val_s = (Float*)(smem + lp.off_val);
    unsigned long long* key_s = (unsigned long long*)(smem + lp.off_key);
    int* id_s = (int*)(smem + lp.off_id);
    Float* stale_s = (Float*)(smem + lp.off_stale);
    int* misc_s = (int*)(smem + lp.off_misc);
    Float* d_s = (Float*)(misc_s + MS_D);

    const int tid = threadIdx.x, T = blockDim.x;
    const bool maximize = pb.maximize != 0;
    const int kind = pb.kind;

    Float* P = (Float*)st.P + (size_t)inst * pb.nnzp;
    Float* pig = (Float*)st.pi + (size_t)inst * pb.m;
    unsigned* xg = st.x + (size_t)inst * (2 * pb.nwords);
    Float* staleg = (Float*)st.rstale + (size_t)inst * (pb.rmax + 1);
    // compute_order::run / push_and_run by this instance's order
    const int ord = order_of(a, inst);
    const bool use_m = ord == 3 || ord == 4;          // rows of the sorted m_order
    const bool keep = use_m || ord == 7;              // R is not recomputed
    int* vlist = use_m ? st.mrun + (size_t)inst * st.lcap : st.vlist + (size_t)inst * st.lcap;
    int* out_list = keep ? st.slist + (size_t)inst * st.lcap : st.vlist + (size_t)inst * st.lcap;
    const Float* cost = (const Float*)pb.cost;

    const Float kappa = (Float)a.kappa[inst];
    const Float delta = (Float)a.delta;
    const Float theta = (Float)a.theta;
    const Float obj_amp = (Float)a.obj_amp;
    const bool push = a.is_push != 0;

    for (int i = tid; i < pb.m; i += T)
        pi_s[i] = pig[i];
    for (int i = tid; i < 2 * pb.nwords; i += T)
        x_s[i] = xg[i];
    for (int i = tid; i <= pb.rmax; i += T)
        stale_s[i] = staleg[i];

    MinStd rng;
    rng.s = st.status[inst].rng; // meaningful in thread 0 only
    const int nrows = use_m ? st.mcount[inst] : st.vcount[inst];
    double alg_bytes = 0.0; // thread 0 only

    if (tid == 0) {
        misc_s[MS_PICH] = 0;
        // compute_order: random_sorting shuffles the list before the sweep
        // (itm-common.hpp:708-714 / :817-823)
        if (ord == 2 || ord == 7)
            rng.shuffle(vlist, (unsigned)nrows);
    }
    __syncthreads();

    for (int it = 0; it < nrows; ++it) {
        const int k = (ord == 1) ? vlist[nrows - 1 - it] : vlist[it];
        const int rs = pb.row_ptr[k];
        const int r = pb.row_ptr[k + 1] - rs;

        // ---- decrease_preference + compute_reduced_costs (+ push) ----------
        for (int idx = tid; idx < r; idx += T) {
            const int e = rs + idx;
            const int j = pb.row_col[e];
            const int pe = pb.row_pos[e];
            Float sum_a_pi = 0;
            Float sum_a_p = 0;
            const int q0 = pb.col_start[j];
            const int qe = q0 + pb.col_len[j];
            for (int q = q0; q < qe; ++q) {
                Float p = P[q];
                if (q == pe)
                    p = p * theta; // P[e] *= theta, written back by affect
                const int ri = pb.col_row[q];
                const Float piv = pi_s[ri & 0x3fffffff];
                if (PM1) {
                    const Float fa = (ri & 0x40000000) ? (Float)-1 : (Float)1;
                    sum_a_pi += fa * piv;
                    sum_a_p += fa * p;
                } else {
                    sum_a_pi += piv;
                    sum_a_p += p;
                }
            }
            const Float cj = cost_x<Float>(pb, x_s, j, cost[j]);
            Float v = cj - sum_a_pi - sum_a_p;
            if (PM1 && pb.fac_csr[e] < 0)
                v = -v;
            if (push) // the 01 solvers index the cost with the LOCAL id (quirk)
                v += obj_amp * (PM1 ? cj : cost_x<Float>(pb, x_s, idx, cost[idx]));
            val_s[idx] = v;
        }
        const int n2 = next_pow2(r < 2 ? 2 : r);
        if (tid == 0)
            misc_s[MS_TIE] = 0;
        __syncthreads();

        // ---- calculator_sort: sort, then shuffle every run of equal values -
        for (int i = tid; i < n2; i += T) {
            key_s[i] = i < r ? order_key<Float>(val_s[i], maximize) : ~0ull;
            id_s[i] = i;
        }
        __syncthreads();
        if (r > 1)
            bitonic_sort(key_s, id_s, n2);
        for (int i = tid; i + 1 < r; i += T)
            if (key_s[i] == key_s[i + 1])
                misc_s[MS_TIE] = 1;
        __syncthreads();

        // ---- thread 0: tie shuffle, select_variables, pi and d of affect ---
        if (tid == 0) {
            alg_bytes += pb.row_bytes[k];
            if (misc_s[MS_TIE]) {
                int b0 = 0;
                for (int i = 1; i <= r; ++i) {
                    if (i == r || key_s[i] != key_s[b0]) {
                        rng.shuffle(id_s + b0, (unsigned)(i - b0));
                        b0 = i;
                    }
                }
            }

            auto Rv = [&](int i) -> Float { // R[i].value, R[r] is stale
                return i < r ? val_s[id_s[i]] : stale_s[i];
            };
            auto stop = [&](Float v) -> bool { // itm-common.hpp:150-163
                if (v == 0)
                    return rng.coin();
                return maximize ? v < 0 : v > 0;
            };

            const int c_size = PM1 ? pb.bneg[k] : 0;
            int bkmin = pb.bmin[k], bkmax = pb.bmax[k];
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
                for (int i = bkmin; i <= bkmax; ++i)
                    if (stop(Rv(i))) {
                        sel = i - 1;
                        break;
                    }
            }

            // affect, itm-common.hpp:393-436
            const Float one = 1, two = 2;
            const Float middle = (two + one) / two;
            Float d = delta;
            if (sel < 0) {
                d += (kappa / (one - kappa)) * (Rv(0) / two);
            } else if (sel + 1 >= r) {
                d += (kappa / (one - kappa)) * (Rv(sel) * middle);
            } else {
                const Float old_pi = pi_s[k];
                const Float new_pi = old_pi + ((Rv(sel) + Rv(sel + 1)) / two);
                pi_s[k] = new_pi;
                d += (kappa / (one - kappa)) * (Rv(sel + 1) - Rv(sel));
                if (signbit(old_pi) != signbit(new_pi))
                    misc_s[MS_PICH] = 1;
            }
            misc_s[MS_SEL] = sel;
            *d_s = d;
        }
        __syncthreads();

        // ---- affect: x bits and P of the row, by sorted position -----------
        {
            const int sel = misc_s[MS_SEL];
            const Float d = *d_s;
            for (int i = tid; i < r; i += T) {
                const int idx = id_s[i];
                const int e = rs + idx;
                const int j = pb.row_col[e];
                const int pe = pb.row_pos[e];
                bool up = i <= sel; // sel < 0: none; sel + 1 >= r: all
                if (PM1 && pb.fac_csr[e] < 0)
                    up = !up;
                const Float p = P[pe] * theta;
                if (up) {
                    atomicOr(&x_s[j >> 5], 1u << (j & 31));
                    P[pe] = p + d;
                } else {
                    atomicAnd(&x_s[j >> 5], ~(1u << (j & 31)));
                    P[pe] = p - d;
                }
                stale_s[i] = val_s[idx]; // R[i].value after the sort
            }
        }
        __syncthreads();
    }

    // ---- write back, then compute_violated_constraints ---------------------
    for (int i = tid; i < pb.m; i += T)
        pig[i] = pi_s[i];
    for (int i = tid; i < 2 * pb.nwords; i += T)
        xg[i] = x_s[i];
    for (int i = tid; i <= pb.rmax; i += T)
        staleg[i] = stale_s[i];

    const int cnt = rebuild_violated(
      pb, x_s, st.vflag + (size_t)inst * pb.m, out_list, misc_s + MS_WAVES);

    if (tid == 0) {
        if (!keep)
            st.vcount[inst] = cnt;
        DevStatus s = st.status[inst];
        s.remaining = cnt;
        s.pi_changed = misc_s[MS_PICH];
        s.rng = rng.s;
        s.rows_in_sweep = (unsigned)nrows;
        s.row_updates += (unsigned long long)nrows;
        s.alg_bytes += alg_bytes;
        st.status[inst] = s;
    }
}


// ---------------------------------------------------------------------------
// v2: staged column loads.
//
// The reduced costs of row k need, for each of its r columns, the sequential
// sums of P and pi over the whole column (about r * c = 25 000 elements for
// the scpnrh1 shape).  v1 gave each lane one column and a dependent
// load-add loop: one L2 / Infinity-Cache round trip per element.  Here the
// row's columns are taken G = blockDim/8 at a time ("pass"); an 8-lane group
// loads one column in 8-element (64-byte) pieces, all loads of a pass issued
// back to back (NB pieces per lane in flight), staged through LDS, and then
// ONE lane per column and per accumulator adds them in CSC order -- the
// summation order of compute_reduced_costs
// (itm-solver-inequalities-01.cpp:173-188) is untouched, so results stay
// bit-identical.  The loads of pass p+1 are in flight while pass p is summed.
// Wait until `*cnt >= target` (arrivals of the cooperating workgroups), one
// polling lane, bounded: a missing partner sets the error word instead of
// hanging the GPU.
__device__ inline void
wait_arrivals(unsigned* cnt, unsigned target, unsigned* err)
{
    long spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 4000000L) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

// compute_violated_constraints with several column owners: every workgroup
// adds up the row activities of its own columns, the partial sums are
// exchanged through L2 (one arrival per workgroup), and each workgroup builds
// the same ordered list from the totals.
__device__ inline int
rebuild_violated_owners(const DevProblem& pb,
                        int grp,
                        int ngroups,
                        const unsigned* x_s,
                        int* vpart_inst, // [groups][m] of this instance
                        unsigned* cnt_end,
                        unsigned* err_word,
                        signed char* vflag,
                        int* vout,
                        int* scratch_s)
{
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
    const bool pm1 = pb.kind >= 2;
    const int* sub_ptr = pb.sub_ptr + (size_t)grp * (pb.m + 1);

    for (int k = wave; k < pb.m; k += nwaves) {
        const int os = sub_ptr[k], oe = sub_ptr[k + 1];
        int v = 0;
        for (int t = os + lane; t < oe; t += 64) {
            const int e = pb.sub_e[t];
            const int j = pb.row_col[e];
            const int bit = (x_s[j >> 5] >> (j & 31)) & 1u;
            v += pm1 ? (int)pb.fac_csr[e] * bit : bit;
        }
        v = wave_sum(v);
        if (lane == 0)
            __hip_atomic_store(&vpart_inst[(size_t)grp * pb.m + k], v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(cnt_end, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wait_arrivals(cnt_end, (unsigned)ngroups, err_word);
    }
    __syncthreads();
    for (int k = tid; k < pb.m; k += T) {
        int v = 0;
        for (int g = 0; g < ngroups; ++g)
            v += __hip_atomic_load(&vpart_inst[(size_t)g * pb.m + k], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        vflag[k] = !(pb.bmin[k] <= v && v <= pb.bmax[k]);
    }
    __syncthreads();

    int base = 0;
    for (int c = 0; c < pb.m; c += T) {
        const int k = c + tid;
        const bool f = k < pb.m && vflag[k] != 0;
        const unsigned long long mask = __ballot(f);
        const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0)
            scratch_s[wave] = __popcll(mask);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < nwaves; ++w) {
            const int cw = scratch_s[w];
            if (w < wave)
                off += cw;
            tot += cw;
        }
        if (f)
            vout[base + off + prefix] = k;
        base += tot;
        __syncthreads();
    }
    return base;
}

// ---------------------------------------------------------------------------
// inequalities-Z: general integer coefficients (itm-solver-inequalities-Z.cpp).
enum ZKind : int // :73-80
{
    Z_BB = 0,
    Z_EXHAUSTIVE = 1,
    Z_UNCONSTRAINED_BB = 2,
    Z_UNCONSTRAINED_EXHAUSTIVE = 3,
    Z_LINEAR = 4
};

// Node stack of branch_and_bound_solver (branch-and-bound-solver.hpp:195-536)
// : per node a 0/1 vector (bit set), the sums and the level.  In LDS when it is small
// (Z_SCRATCH_LDS; rows of up to a few hundred elements), else in global memory: the search is one
// thread walking it word by word, a global round trip per word.
__host__ __device__ inline size_t
zscratch_size(int rmax, int fsize)
{
    const size_t words = static_cast<size_t>((rmax + 31) / 32);
    const size_t depth = static_cast<size_t>(rmax) + 2;
    // vars[depth][words] + solution[words] + sumr[depth] + sumf[depth] + level[depth]
    return (depth + 1) * words * 4 + depth * (static_cast<size_t>(fsize) + 8) + 64;
}

constexpr int Z_SCRATCH_LDS = 16 * 1024;
constexpr int Z_LDS_LIMIT = 160 * 1024; // the LDS of a gfx950 CU (bxh_device.hip: lds_limit)

// LDS of the Z kernel: the plan of the lane-per-column kernel, then the
// factor of every R position and a 0/1 result per position.
// The wave-per-row form (sweep_kernel_z<Float, true>): every wave has its own set
// of row buffers (64 entries each: values by element, values / keys / element ids /
// factors by position, positions by element, sub-solver results) and the workgroup
// a table of the next ZW_TABLE rows of the list (scalars + conflict masks).
constexpr int ZW_WAVES = 4;  // waves of the Z kernel's workgroup (256 threads)
constexpr int ZW_TABLE = 64; // rows of the list staged in LDS at a time (== a wave: the group scan uses ballots)
constexpr int ZW_CAP = 128;  // entries of a wave's row buffers: rows of up to ZW_RMAX elements go wave per row
constexpr int ZW_RMAX = 127; // (a lane takes two elements of a row of more than 63)
__host__ __device__ constexpr int
zw_wave_bytes(int fsize)
{
    return ZW_CAP * (2 * fsize + 8 + 4 * 4);
}
__host__ __device__ constexpr int
zw_table_bytes()
{
    return ZW_TABLE * (8 * 4 + 8 + 8 + 2 * 16);
}

__host__ __device__ inline int
lds_bytes_z_base(int m, int nwords, int rmax, int fsize, bool zw)
{
    const LdsPlan lp = lds_plan(m, nwords, rmax, fsize);
    const int n2 = next_pow2(rmax < 64 ? 64 : rmax); // whole 64-entry runs
    // + factor and 0/1 result per R position, + the sorted runs of
    // run_merge_sort, + the row's descriptors {first slot, padded length,
    // own slot, column} by row-local id
    return lp.total + 2 * (((rmax + 1) * 4 + 15) & ~15) + ((n2 * 8 + 15) & ~15) +
           ((n2 * 4 + 15) & ~15) + ((rmax * 16 + 15) & ~15) +
           (zw ? ZW_WAVES * zw_wave_bytes(fsize) + zw_table_bytes() : 0);
}

// The node stack sits in LDS only when it is small AND the plan still fits with
// it; otherwise the kernel uses the per-instance global scratch (host and kernel
// decide with this one function).
__host__ __device__ inline bool
z_stack_in_lds(int m, int nwords, int rmax, int fsize, bool zw)
{
    const int zs = (int)((zscratch_size(rmax, fsize) + 15) & ~(size_t)15);
    return zs <= Z_SCRATCH_LDS && lds_bytes_z_base(m, nwords, rmax, fsize, zw) + zs <= Z_LDS_LIMIT;
}

__host__ __device__ inline int
lds_bytes_z(int m, int nwords, int rmax, int fsize, bool zw)
{
    const int zs = (int)((zscratch_size(rmax, fsize) + 15) & ~(size_t)15);
    return lds_bytes_z_base(m, nwords, rmax, fsize, zw) +
           (z_stack_in_lds(m, nwords, rmax, fsize, zw) ? zs : 0);
}

// One thread: the depth-first search of branch_and_bound_solver::solve
// (:450-533) over the items in R order.  The reference first sorts the items
// with item::operator< (:212-220) -- R was just sorted by calculator_sort, on
// which that (stable) sort is the identity -- and afterwards by (result,
// value): a stable partition here.  Writes the 0/1 result of every position to
// res[]; returns false when the node budget is exhausted.
template<typename Float>
__device__ inline bool
bb_search(const Float* val_s,
          const int* id_s,
          const int* fz_s,
          int n,
          int b_min,
          int b_max,
          bool maximize,
          unsigned char* scratch,
          int* res)
{
    const int W = (n + 31) / 32;
    const int depth = n + 2;
    unsigned* vars = reinterpret_cast<unsigned*>(scratch);          // [depth][W]
    unsigned* sol = vars + (size_t)depth * W;                       // [W]
    Float* sumr = reinterpret_cast<Float*>(
      scratch + ((((size_t)depth + 1) * W * 4 + 15) & ~(size_t)15)); // [depth]
    int* sumf = reinterpret_cast<int*>(sumr + depth);               // [depth]
    int* level = sumf + depth;                                      // [depth]

    for (int i = 0; i < n; ++i) // negative coefficients: complemented variable
        if (fz_s[i] < 0) {
            b_min += -fz_s[i];
            b_max += -fz_s[i];
        }
    auto r_of = [&](int i) -> Float { return val_s[id_s[i]]; };
    auto f_of = [&](int i) -> int { return fz_s[i] < 0 ? -fz_s[i] : fz_s[i]; };

    const Float inf = __builtin_huge_val();
    Float lower_bound = maximize ? -inf : inf;
    auto is_best = [&](Float z) -> bool { return maximize ? z > lower_bound : z < lower_bound; };
    for (int w = 0; w < W; ++w)
        sol[w] = 0u;
    int top = -1;
    auto copy_sol = [&](int node) {
        for (int w = 0; w < W; ++w)
            sol[w] = vars[(size_t)node * W + w];
    };
    auto push0 = [&]() {
        ++top;
        for (int w = 0; w < W; ++w)
            vars[(size_t)top * W + w] = 0u;
        sumr[top] = 0;
        sumf[top] = 0;
        level[top] = 0;
    };
    auto push1 = [&]() {
        ++top;
        for (int w = 0; w < W; ++w)
            vars[(size_t)top * W + w] = 0u;
        vars[(size_t)top * W] = 1u;
        sumr[top] = r_of(0);
        sumf[top] = f_of(0);
        level[top] = 0;
    };
    const bool equality = b_max == b_min;
    if (equality) { // solve_equality, :386-409
        push0();
        if (sumf[top] == b_min) {
            copy_sol(top);
            top = -1; // returns at once
        } else {
            push1();
            if (sumf[top] == b_min) {
                copy_sol(top);
                --top;
            }
        }
    } else { // solve_inequality, :411-440
        push0();
        if (b_min <= sumf[top] && sumf[top] <= b_max)
            copy_sol(top);
        else if (sumf[top] > b_max)
            --top;
        push1();
        if (b_min <= sumf[top] && sumf[top] <= b_max)
            copy_sol(top);
        else if (sumf[top] > b_max)
            --top;
    }
    long budget = 50000000L; // every wave must reach the end of the kernel
    while (top >= 0) { // make_next_*_node, :307-384
        if (--budget < 0)
            return false;
        const int lv = level[top] + 1;
        if (lv >= n) {
            --top;
            continue;
        }
        level[top] = lv;
        const int child = top + 1;
        for (int w = 0; w < W; ++w)
            vars[(size_t)child * W + w] = vars[(size_t)top * W + w];
        vars[(size_t)child * W + (lv >> 5)] |= 1u << (lv & 31);
        sumr[child] = sumr[top] + r_of(lv);
        sumf[child] = sumf[top] + f_of(lv);
        level[child] = lv;
        top = child;
        if (equality) {
            if (sumf[top] == b_min) {
                if (is_best(sumr[top])) {
                    copy_sol(top);
                    lower_bound = sumr[top];
                }
                --top;
            }
        } else {
            if (b_min <= sumf[top] && sumf[top] <= b_max) {
                if (is_best(sumr[top])) {
                    copy_sol(top);
                    lower_bound = sumr[top];
                }
            } else if (sumf[top] > b_max) {
                --top;
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        const int bit = (sol[i >> 5] >> (i & 31)) & 1u;
        res[i] = fz_s[i] > 0 ? bit : 1 - bit;
    }
    return true;
}

// One sweep of one instance of the Z solver: compute_order::run /
// push_and_run around compute_update_row / push_and_compute_update_row of
// itm-solver-inequalities-Z.cpp:441-564 (one workgroup per instance).
//
// ZW = true (linear objective): rows of up to 63 elements are worked on by ONE WAVE
// each, wave-synchronously (no workgroup barrier inside a row), and up to four
// consecutive rows of the list that share no column -- they commute exactly: neither
// reads pi, P or x entries the other writes -- run side by side, a wave each, between
// two barriers.  What the reference does in sequence stays in sequence: the tie
// shuffles draw from the engine in row order (thread 0, behind the first barrier),
// and R's leftovers (`stale`) are written in row order.  Longer rows and the
// constrained sub-solver kinds take the general path below, alone.
template<typename Float, bool ZW>
__global__ __launch_bounds__(256) void // (256 threads at most: the cached column slots want registers)
sweep_kernel_z(DevProblem pb, DevState st, DevSweep a)
{
#if defined(BXH_DEV_ONLY_NP) && !defined(BXH_DEV_KEEP_Z)
    return; // (development builds of another kernel: see sweep_kernel_v2)
#endif

    extern __shared__ __align__(16) unsigned char smem[];
    const int inst = blockIdx.x;
    if (a.active && !a.active[inst])
        return;

    const LdsPlan lp = lds_plan(pb.m, pb.nwords, pb.rmax, (int)sizeof(Float));
    Float* pi_s = (Float*)(smem + lp.off_pi);
    unsigned* x_s = (unsigned*)(smem + lp.off_x);
    Float* val_s = (Float*)(smem + lp.off_val);
    unsigned long long* key_s = (unsigned long long*)(smem + lp.off_key);
    int* id_s = (int*)(smem + lp.off_id);
    Float* stale_s = (Float*)(smem + lp.off_stale);
    int* misc_s = (int*)(smem + lp.off_misc);
    Float* d_s = (Float*)(misc_s + MS_D);
    int* fz_s = (int*)(smem + lp.total);                                   // R[i].f
    int* res_s = (int*)(smem + lp.total + (((pb.rmax + 1) * 4 + 15) & ~15)); // 0/1 per position
    const int n2max = next_pow2(pb.rmax < 64 ? 64 : pb.rmax); // whole 64-entry runs
    unsigned long long* key_run =
      (unsigned long long*)(smem + lp.total + 2 * (((pb.rmax + 1) * 4 + 15) & ~15));
    int* id_run = (int*)((unsigned char*)key_run + ((n2max * 8 + 15) & ~15));
    int4* desc_s = (int4*)((unsigned char*)id_run + ((n2max * 4 + 15) & ~15)); // by row-local id

    const int tid = threadIdx.x, T = blockDim.x;
    const bool maximize = pb.maximize != 0;

    Float* P = (Float*)st.P + (size_t)inst * pb.nnzp;
    Float* pig = (Float*)st.pi + (size_t)inst * pb.m;
    unsigned* xg = st.x + (size_t)inst * (2 * pb.nwords);
    Float* staleg = (Float*)st.rstale + (size_t)inst * (pb.rmax + 1);
    const int ord = order_of(a, inst);
    const bool use_m = ord == 3 || ord == 4;
    const bool keep = use_m || ord == 7;
    int* vlist = use_m ? st.mrun + (size_t)inst * st.lcap : st.vlist + (size_t)inst * st.lcap;
    int* out_list = keep ? st.slist + (size_t)inst * st.lcap : st.vlist + (size_t)inst * st.lcap;
    const Float* cost = (const Float*)pb.cost;
    // (the node stack of the branch-and-bound rows: behind the descriptors in LDS when small)
    unsigned char* const zw_base = (unsigned char*)desc_s + ((pb.rmax * 16 + 15) & ~15);
    constexpr int ZWB = ZW ? ZW_WAVES * zw_wave_bytes((int)sizeof(Float)) + zw_table_bytes() : 0;
    unsigned char* scratch = z_stack_in_lds(pb.m, pb.nwords, pb.rmax, (int)sizeof(Float), ZW)
                               ? zw_base + ZWB
                               : st.zscratch + (size_t)inst * st.zscratch_bytes;
    // ZW: the row table (next ZW_TABLE rows of the list) behind the waves' buffers
    unsigned char* const zt_base = zw_base + ZW_WAVES * zw_wave_bytes((int)sizeof(Float));
    unsigned long long* const t_conf = (unsigned long long*)zt_base;
    double* const t_rb = (double*)(zt_base + ZW_TABLE * 8);
    int* const t_k = (int*)(zt_base + ZW_TABLE * 16);
    int* const t_rs = t_k + ZW_TABLE;
    int* const t_re = t_rs + ZW_TABLE;
    int* const t_z = t_re + ZW_TABLE;
    int* const t_bmin = t_z + ZW_TABLE;
    int* const t_bmax = t_bmin + ZW_TABLE;
    int* const t_g = t_bmax + ZW_TABLE;  // rows of the group that starts at this slot (0: general path)
    int* const t_cb = t_g + ZW_TABLE;    // (spare)
    // the scalars a group start reads, packed: {k, rs, re, kind} and {bmin, bmax, group size, r}
    int4* const t_a = (int4*)(t_cb + ZW_TABLE);
    int4* const t_b = t_a + ZW_TABLE;
    unsigned* const err_word = st.cnt + 2 * gridDim.x;

    const Float kappa = (Float)a.kappa[inst];
    const Float delta = (Float)a.delta;
    const Float theta = (Float)a.theta;
    const Float obj_amp = (Float)a.obj_amp;
    const bool push = a.is_push != 0;

    for (int i = tid; i < pb.m; i += T)
        pi_s[i] = pig[i];
    for (int i = tid; i < 2 * pb.nwords; i += T)
        x_s[i] = xg[i];
    for (int i = tid; i <= pb.rmax; i += T)
        stale_s[i] = staleg[i];

    MinStd rng;
    rng.s = st.status[inst].rng; // meaningful in thread 0 only
    const int nrows = use_m ? st.mcount[inst] : st.vcount[inst];
    double alg_bytes = 0.0; // thread 0 only
    Prof prof;
#ifdef BXH_PROFILE
    for (int i = 0; i < 16; ++i)
        prof.acc[i] = 0;
    prof.t = wall_clock64();
#endif

    if (tid == 0) {
        misc_s[MS_PICH] = 0;
        if (ord == 2 || ord == 7)
            rng.shuffle(vlist, (unsigned)nrows);
    }
    __syncthreads();

    // c_j - sum over column j of |a| * (pi + P), one accumulator, CSC order
    // (local_compute_reduced_cost, :295-306); `pe` / `pown` substitute the
    // row's own entry (not yet in HBM)
    // (columns are stored padded to 16 slots with P = 0 and |a| = 0: eight
    // entries are loaded at a time, all loads in flight together, and a
    // padding term 0 * (pi + 0) leaves the running sum bit-identical)
    auto column_rc = [&](int j, int q0, int padlen, int pe, Float pown) -> Float {
        Float acc = 0;
        const int qe = q0 + padlen;
        for (int qb = q0; qb < qe; qb += 8) {
            Float p8[8];
            int f8[8], r8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p8[u] = P[qb + u];
                f8[u] = pb.col_fac[qb + u];
                r8[u] = pb.col_row[qb + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const Float p = qb + u == pe ? pown : p8[u];
                const int ri = r8[u] < pb.m ? r8[u] : 0; // padding: any finite pi
                acc += (Float)f8[u] * (pi_s[ri] + p);
            }
        }
        return cost_x<Float>(pb, x_s, j, cost[j]) - acc;
    };

    // The two sub-solvers (all threads; barriers inside): submode 1 = branch and
    // bound, 2 = exhaustive, on a row whose values sit in val_s (by element), the
    // sorted order in id_s / fz_s (by position); leaves R's new order in id_s and the
    // split index in misc_s[MS_SEL].
    auto subsolve = [&](const int submode, const int k, const int r, const int bkmin, const int bkmax,
                        const Float* val_s, int* id_s, int* fz_s, int* res_s) {
            if (submode == 1) {
                if (tid == 0) {
                    if (!bb_search<Float>(val_s, id_s, fz_s, r, bkmin, bkmax, maximize, scratch, res_s))
                        __hip_atomic_store(err_word, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            } else if (submode == 2) {
                // exhaustive_solver::solve, exhaustive-solver.hpp:185-252: the
                // best of the row's precomputed feasible patterns, the first one
                // among equals (one thread per pattern, sums in element order)
                const int start = pb.ex_start[k], nsol = pb.ex_count[k];
                if (tid == 0) {
                    key_s[0] = ~0ull;
                    misc_s[MS_CNT] = 0x7fffffff;
                }
                __syncthreads();
                for (int t = tid; t < nsol; t += T) {
                    Float z = 0;
                    for (int jx = 0; jx < r; ++jx)
                        if (pb.ex_flat[start + t * r + jx])
                            z += val_s[id_s[jx]];
                    atomicMin(&key_s[0], order_key<Float>(z, maximize));
                }
                __syncthreads();
                for (int t = tid; t < nsol; t += T) {
                    Float z = 0;
                    for (int jx = 0; jx < r; ++jx)
                        if (pb.ex_flat[start + t * r + jx])
                            z += val_s[id_s[jx]];
                    if (order_key<Float>(z, maximize) == key_s[0])
                        atomicMin(&misc_s[MS_CNT], t);
                }
                __syncthreads();
                if (tid == 0) {
                    const int best = misc_s[MS_CNT];
                    for (int i = 0; i < r; ++i)
                        res_s[i] = pb.ex_flat[start + best * r + i] ? 1 : 0;
                }
                __syncthreads();
            }
            if (submode != 0) {
                // both sub-solvers end with a sort by (result first, then value)
                // and return the split index (:219-252, :505-533)
                if (tid == 0) {
                    int* tmp = (int*)key_s; // ids in the new order
                    int cnt1 = 0;
                    if (submode == 1) { // R is sorted by value: a stable partition
                        for (int i = 0; i < r; ++i)
                            if (res_s[i])
                                tmp[cnt1++] = id_s[i];
                        int z0 = cnt1;
                        for (int i = 0; i < r; ++i)
                            if (!res_s[i])
                                tmp[z0++] = id_s[i];
                    } else { // at most 9 items: libstdc++'s insertion sort
                        for (int i = 0; i < r; ++i) {
                            tmp[i] = id_s[i];
                            tmp[r + i] = res_s[i];
                            cnt1 += res_s[i];
                        }
                        auto less = [&](int ia, int ra, int ib, int rb) -> bool {
                            if (ra == rb)
                                return maximize ? val_s[ia] > val_s[ib] : val_s[ia] < val_s[ib];
                            return ra > rb;
                        };
                        for (int i = 1; i < r; ++i) {
                            const int vi = tmp[i], vr = tmp[r + i];
                            int jx = i;
                            while (jx > 0 && less(vi, vr, tmp[jx - 1], tmp[r + jx - 1])) {
                                tmp[jx] = tmp[jx - 1];
                                tmp[r + jx] = tmp[r + jx - 1];
                                --jx;
                            }
                            tmp[jx] = vi;
                            tmp[r + jx] = vr;
                        }
                    }
                    for (int i = 0; i < r; ++i)
                        id_s[i] = tmp[i];
                    misc_s[MS_SEL] = cnt1 == r ? r : cnt1 - 1;
                }
                __syncthreads();
            }
    };

    // the next row's scalars travel while this row is worked on
    struct RowZ
    {
        int k, rs, re, zk, bmin, bmax;
        double rbytes;
    };
    // (the row's number is fetched a row before its scalars: neither load is waited
    // for in the row that issues it)
    auto row_id = [&](int it_) -> int { return (ord == 1) ? vlist[nrows - 1 - it_] : vlist[it_]; };
    auto row_of = [&](int k_) -> RowZ {
        RowZ x;
        x.k = k_;
        x.rs = pb.row_ptr[x.k];
        x.re = pb.row_ptr[x.k + 1];
        x.zk = pb.zkind[x.k];
        x.bmin = pb.bmin[x.k];
        x.bmax = pb.bmax[x.k];
        x.rbytes = pb.row_bytes[x.k];
        return x;
    };
    RowZ nxt = { 0, 0, 0, 0, 0, 0, 0.0 };
    if (!ZW && nrows > 0)
        nxt = row_of(row_id(0));
    int k_ahead = (!ZW && nrows > 1) ? row_id(1) : 0; // the row after `nxt`
    bool pre_ok = false; // the lane's element descriptor of the next row is in registers
    int2 pre_cs = make_int2(0, 0);
    int pre_j = 0, pre_pe = 0, pre_f = 0;

    // ---- ZW: the row table, the groups, the wave-per-row path ---------------------
    const int wave = tid >> 6, lane = tid & 63, NW = T >> 6;
    int* const zflag = misc_s + MS_WAVES + 40; // [0..3] tie, [4..7] sub-solver wanted, [8..11] sel
    unsigned char* const zw_mine = zw_base + wave * zw_wave_bytes((int)sizeof(Float));
    auto w_val = [&](int w) { return (Float*)(zw_base + w * zw_wave_bytes((int)sizeof(Float))); };      // by element
    auto w_vs = [&](int w) { return w_val(w) + ZW_CAP; };                                              // by position
    auto w_key = [&](int w) { return (unsigned long long*)(w_val(w) + 2 * ZW_CAP); };                  // by position
    auto w_idp = [&](int w) { return (int*)(w_key(w) + ZW_CAP); };                                     // element by position
    auto w_fzp = [&](int w) { return w_idp(w) + ZW_CAP; };                                             // factor by position
    auto w_pos = [&](int w) { return w_idp(w) + 2 * ZW_CAP; };                                         // position by element
    auto w_res = [&](int w) { return w_idp(w) + 3 * ZW_CAP; };
    (void)zw_mine;
    int tbase = -0x40000000; // list position of table slot 0
    // inclusive prefix sum of an int over the wave (DPP: shifts inside a row of 16 lanes,
    // then the row totals handed on by row_bcast:15 / row_bcast:31)
    auto wave_scan = [&](int v) -> int {
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); // row_shr:1
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true); // row_shr:2
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); // row_shr:4
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true); // row_shr:8
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
        return v;
    };
    // the lane's element descriptor of its row in the NEXT group, fetched a group ahead
    double alg_w = 0.0; // algorithmic bytes of the rows this wave took
    RowZ nf = { 0, 0, 0, 0, 0, 0, 0.0 }; // a row of the NEXT table, fetched a table ahead
    unsigned long long nf_conf = 0ull;
    int nf_pos = 0x7fffffff;
    int zp_pos = -1;
    int4 zp_d0 = make_int4(0, 0, 0, 0), zp_d1 = make_int4(0, 0, 0, 0);

    for (int it = 0; it < nrows; ++it) {
        RowZ cur = nxt;
        if (ZW) {
            if (it + 2 * NW > tbase + ZW_TABLE && (tbase < 0 || tbase + ZW_TABLE < nrows)) {
                BXH_STAMP(7);
                __syncthreads(); // (nobody reads the old table any more)
                // (the rows of this table were fetched into registers when the last one was
                // filled -- positions old tbase + 48 ..: thread t holds position nf_pos)
                const bool have = tbase >= 0;
                tbase = it;
                if (have) {
                    const int sl_ = nf_pos - it;
                    if (sl_ >= 0 && sl_ < ZW_TABLE && nf_pos < nrows) {
                        t_k[sl_] = nf.k;
                        t_rs[sl_] = nf.rs;
                        t_re[sl_] = nf.re;
                        t_z[sl_] = nf.zk;
                        t_bmin[sl_] = nf.bmin;
                        t_bmax[sl_] = nf.bmax;
                        t_rb[sl_] = nf.rbytes;
                        t_conf[sl_] = nf_conf;
                        t_a[sl_] = make_int4(nf.k, nf.rs, nf.re, nf.zk);
                    }
                } else if (tid < ZW_TABLE && it + tid < nrows) {
                    const RowZ x = row_of(row_id(it + tid));
                    t_a[tid] = make_int4(x.k, x.rs, x.re, x.zk);
                    t_k[tid] = x.k;
                    t_rs[tid] = x.rs;
                    t_re[tid] = x.re;
                    t_z[tid] = x.zk;
                    t_bmin[tid] = x.bmin;
                    t_bmax[tid] = x.bmax;
                    t_rb[tid] = x.rbytes;
                    t_conf[tid] = pb.zconf[x.k];
                }
                __syncthreads();
                // which of the three rows in front does a row clash with (shares a column,
                // or too far apart in the matrix for the masks to tell), and is it a row a
                // wave can take: at most ZW_RMAX elements, a kind whose selection is the scan
                if (tid < ZW_TABLE) {
                    int cb = 0;
                    if (it + tid < nrows) {
                        const int rr = t_re[tid] - t_rs[tid];
                        if (rr >= 1 && rr <= ZW_RMAX && t_z[tid] >= Z_UNCONSTRAINED_BB)
                            cb = 0x100;
                        const int k2 = t_k[tid];
                        for (int u = 1; u <= 3 && u <= tid; ++u) {
                            const int k1 = t_k[tid - u];
                            const int lo = k1 < k2 ? tid - u : tid;
                            const int d = k1 < k2 ? k2 - k1 : k1 - k2;
                            if (d == 0 || d > 64 || ((t_conf[lo] >> (d - 1)) & 1ull))
                                cb |= 1 << (u - 1);
                        }
                    }
                    // the groups: consecutive simple rows, pairwise without a clash (wave 0
                    // has the 64 rows' bits in four ballots; lane 0 walks them)
                    const unsigned long long smask = __ballot((cb & 0x100) != 0);
                    const unsigned long long c1 = __ballot((cb & 1) != 0), c2 = __ballot((cb & 2) != 0),
                                             c3 = __ballot((cb & 4) != 0);
                    // (every lane works out the group that WOULD start at its slot; the loop only
                    // ever asks at slots where one does)
                    int gs = 0;
                    if ((smask >> tid) & 1ull) {
                        gs = 1;
                        while (gs < NW && tid + gs < ZW_TABLE && ((smask >> (tid + gs)) & 1ull)) {
                            const int at = tid + gs;
                            const int cbits = (int)((c1 >> at) & 1ull) | ((int)((c2 >> at) & 1ull) << 1) |
                                              ((int)((c3 >> at) & 1ull) << 2);
                            if (cbits & ((1 << gs) - 1))
                                break;
                            ++gs;
                        }
                    }
                    t_g[tid] = gs;
                    t_b[tid] = make_int4(t_bmin[tid], t_bmax[tid], gs, t_re[tid] - t_rs[tid]);
                }
                // the rows of the NEXT table: on their way from now on
                nf_pos = it + 48 + tid;
                if (tid < 80 && nf_pos < nrows) {
                    nf = row_of(row_id(nf_pos));
                    nf_conf = pb.zconf[nf.k];
                }
                __syncthreads();
            }
            int g = __builtin_amdgcn_readfirstlane(t_b[it - tbase].z);
            bool z_again = false; // the outer loop's top has to look at position `it` (end of the list, table refill)
            while (g > 0) { // group after group: the general path's state is not touched in here
                BXH_STAMP(8);
                const int s0 = it - tbase;
                const bool mine = wave < g;
                const int sl = s0 + (mine ? wave : 0);
                const int4 ta = t_a[sl], tb = t_b[sl];
                const int k = __builtin_amdgcn_readfirstlane(ta.x);
                const int rs = __builtin_amdgcn_readfirstlane(ta.y);
                const int r = mine ? __builtin_amdgcn_readfirstlane(tb.w) : 0;
                const int zk = __builtin_amdgcn_readfirstlane(ta.w);
                const int bkmin = __builtin_amdgcn_readfirstlane(tb.x);
                const int bkmax = __builtin_amdgcn_readfirstlane(tb.y);
                Float* const val_w = w_val(wave);
                Float* const vs_w = w_vs(wave);
                unsigned long long* const key_w = w_key(wave);
                int* const idp_w = w_idp(wave);
                int* const fzp_w = w_fzp(wave);
                int* const pos_w = w_pos(wave);

                // ---- phase A: the lane's column -> reduced cost -> rank ----------------
                // The first 16 slots of the lane's column stay in registers for the whole row:
                // P, |a| and -- the rows of a group share no column, so nothing a partner
                // writes is in this column -- the pi of the slots' rows, read from LDS ONCE;
                // the repair pass only swaps in the row's own slot (new pi_k, new P).
                Float c_p[16], c_f[16], c_pi[16];
                uint4 c_rp[4]; // row | |a| << 16 of four slots each, as loaded
                int4 c_desc = make_int4(0, 0, 0, 0);
                Float c_cost = 0, p_own_raw = 0;
                int fe = 0;
                if (lane < r) {
                    int4 d1;
                    if (zp_pos == it + wave) { // fetched while the previous group was worked on
                        c_desc = zp_d0;
                        d1 = zp_d1;
                    } else {
                        const int e = rs + lane;
                        c_desc = pb.zdesc[2 * e];
                        d1 = pb.zdesc[2 * e + 1];
                    }
                    fe = d1.x;
                    if (sizeof(Float) == 8)
                        c_cost = (Float)__hiloint2double(d1.z, d1.y);
                    else
                        c_cost = (Float)__int_as_float(d1.y);
                    const int2 cs = make_int2(c_desc.x, c_desc.y);
                    // (columns are padded to 16 slots: 16-byte loads, all in flight together)
                    if (sizeof(Float) == 8) {
                        const double2* pv = reinterpret_cast<const double2*>((const double*)P + cs.x);
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const double2 t2 = pv[u];
                            c_p[2 * u] = (Float)t2.x;
                            c_p[2 * u + 1] = (Float)t2.y;
                        }
                    } else {
                        const float4* pv = reinterpret_cast<const float4*>((const float*)P + cs.x);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float4 t4 = pv[u];
                            c_p[4 * u] = (Float)t4.x;
                            c_p[4 * u + 1] = (Float)t4.y;
                            c_p[4 * u + 2] = (Float)t4.z;
                            c_p[4 * u + 3] = (Float)t4.w;
                        }
                    }
                    const uint4* rfv = reinterpret_cast<const uint4*>(pb.zcolp + cs.x);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        c_rp[u] = rfv[u]; // (used -- and so waited for -- only behind the prefetch below)
                    p_own_raw = P[c_desc.z]; // (one more load instead of sixteen selects)
                }
                // the next group's descriptors: on their way while this one is worked on
                {
                    const int pos2 = it + g;
                    zp_pos = -1;
                    if (pos2 < nrows && pos2 + NW <= tbase + ZW_TABLE) {
                        const int g2 = __builtin_amdgcn_readfirstlane(t_b[pos2 - tbase].z);
                        if (wave < g2) {
                            const int sl2 = pos2 - tbase + wave;
                            const int4 ta2 = t_a[sl2];
                            const int e2 = ta2.y + lane;
                            zp_pos = pos2 + wave;
                            if (e2 < ta2.z) {
                                zp_d0 = pb.zdesc[2 * e2];
                                zp_d1 = pb.zdesc[2 * e2 + 1];
                            }
                        }
                    }
                }
                // sum over the lane's column as the cached arrays stand (set_own puts the row's own
                // entry -- pi_k and the not yet stored P -- into its slot first): the cached slots,
                // then (columns of more than 16 entries) the rest
                // (c_pi holds pi + P of every slot once phase A has formed the sums: the reference
                // adds `a * (pi + P)` slot after slot, so the inner sum can be kept instead of its
                // two terms -- half the registers, half the selects of the own slot)
                auto set_own = [&](Float pown, Float pik) {
                    const int off = c_desc.z - c_desc.x;
                    const Float so = pik + pown;
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        c_pi[u] = u == off ? so : c_pi[u];
                };
                auto z_rc = [&](Float pown, Float pik) -> Float {
                    Float acc = 0;
                    const int q0 = c_desc.x, pe = c_desc.z;
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        acc += c_f[u] * c_pi[u];
                    for (int qb = q0 + 16; qb < q0 + c_desc.y; qb += 8) {
                        Float p8[8], pi8[8];
                        int f8[8], r8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            p8[u] = P[qb + u];
                            const int2 rf = pb.zcolrf[qb + u];
                            r8[u] = rf.x;
                            f8[u] = rf.y;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            pi8[u] = pi_s[r8[u] < pb.m ? r8[u] : 0];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const bool own = qb + u == pe;
                            acc += (Float)f8[u] * ((own ? pik : pi8[u]) + (own ? pown : p8[u]));
                        }
                    }
                    return c_cost - acc;
                };
                auto z_own = [&]() -> Float { return p_own_raw; }; // P[pe] as it stands in memory
                Float val = 0;
                unsigned long long key = ~0ull;
                int rank = 0;
                bool tie = false;
                int sel = -2;
                // select_variables (:308-344) on ballots: lane t stands for i = t - 1.  (R[r],
                // the leftover of an earlier row, is read by the reference at i = r - 1, where it
                // cannot change `best` any more: left out.)
                auto wave_select = [&]() -> int {
                    const int i = lane - 1; // -1 .. 62
                    const bool in = i < r;
                    const int S = wave_scan((i >= 0 && i < r) ? fzp_w[i] : 0); // S_i (S_-1 = 0)
                    const Float nv = i + 1 < r ? vs_w[i + 1] : (Float)0;
                    const unsigned long long valid = __ballot(in && bkmin <= S && S <= bkmax);
                    const unsigned long long stops = __ballot(in && i + 1 < r && (maximize ? nv < 0 : nv > 0));
                    if (!valid)
                        return -2;
                    const int fv = __ffsll((long long)valid) - 1; // lane of the first valid i
                    const unsigned long long after = stops & ~((1ull << fv) - 1ull);
                    const int leave = after ? __ffsll((long long)after) - 1 : r; // lane; r == i = r - 1
                    const unsigned long long upto = valid & (leave >= 63 ? ~0ull : ((2ull << leave) - 1ull));
                    return (63 - __clzll((long long)upto)) - 1; // lane -> i
                };
                auto sub_wanted = [&](int sl_) -> int { // the unconstrained kinds, when the scan found nothing
                    return (sl_ == -2 && zk == Z_UNCONSTRAINED_BB)
                             ? 1
                             : ((sl_ == -2 && zk == Z_UNCONSTRAINED_EXHAUSTIVE) ? 2 : 0);
                };
                BXH_STAMP(9);
                Float pik_old = 0, p_decayed = 0;
                // ---- rows of 64 .. ZW_RMAX elements (round 5): the lane takes element `lane` and
                // element `lane + 64`, one after the other through the same cached column -- the
                // first was loaded above, the second is loaded here; phase C reloads the one that
                // is not in the registers.  Everything by position (vs, keys, ids, factors) is
                // 128 entries long; the scan of select_variables runs twice with the running sum
                // carried over.  Such rows are rare (21 of the 1000 of the telebus-shaped c5): the
                // point is that they no longer leave the groups' loop for the one-row path.
                const bool rlong = r > 63; // (wave-uniform)
                Float l_val[2] = { 0, 0 }, l_pdec[2] = { 0, 0 }, l_cost[2] = { 0, 0 };
                unsigned long long l_key[2] = { ~0ull, ~0ull };
                int l_rank[2] = { 0, 0 }, l_fe[2] = { 0, 0 };
                int4 l_desc[2] = { make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0) };
                int l_cached = 0; // which of the lane's two elements the cached column belongs to
                auto load_col = [&](int e) { // descriptor and the first 16 slots of element e's column
                    c_desc = pb.zdesc[2 * e];
                    const int4 d1 = pb.zdesc[2 * e + 1];
                    fe = d1.x;
                    if (sizeof(Float) == 8)
                        c_cost = (Float)__hiloint2double(d1.z, d1.y);
                    else
                        c_cost = (Float)__int_as_float(d1.y);
                    if (sizeof(Float) == 8) {
                        const double2* pv = reinterpret_cast<const double2*>((const double*)P + c_desc.x);
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const double2 t2 = pv[u];
                            c_p[2 * u] = (Float)t2.x;
                            c_p[2 * u + 1] = (Float)t2.y;
                        }
                    } else {
                        const float4* pv = reinterpret_cast<const float4*>((const float*)P + c_desc.x);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float4 t4 = pv[u];
                            c_p[4 * u] = (Float)t4.x;
                            c_p[4 * u + 1] = (Float)t4.y;
                            c_p[4 * u + 2] = (Float)t4.z;
                            c_p[4 * u + 3] = (Float)t4.w;
                        }
                    }
                    const uint4* rfv = reinterpret_cast<const uint4*>(pb.zcolp + c_desc.x);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        c_rp[u] = rfv[u];
                    p_own_raw = P[c_desc.z];
                };
                auto fill_pi = [&]() { // c_f, c_pi = pi + P of the cached slots (own slot: set_own)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned w4[4] = { c_rp[u].x, c_rp[u].y, c_rp[u].z, c_rp[u].w };
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int rw_ = (int)(w4[t] & 0xffffu);
                            c_pi[4 * u + t] = pi_s[rw_ < pb.m ? rw_ : 0];
                            c_f[4 * u + t] = (Float)(int)(w4[t] >> 16);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        c_pi[u] = c_pi[u] + c_p[u];
                };
                auto form_val = [&](int h) { // reduced cost and key of the cached element -> l_*[h]
                    fill_pi();
                    const Float pk = pi_s[k];
                    const Float pd = z_own() * theta; // decrease_preference (stored with affect's value)
                    set_own(pd, pk);
                    Float v = z_rc(pd, pk);
                    if (push)
                        v += obj_amp * c_cost;
                    l_val[h] = v;
                    l_key[h] = order_key<Float>(v, maximize);
                    l_pdec[h] = pd;
                    l_fe[h] = fe;
                    l_desc[h] = c_desc;
                    l_cost[h] = c_cost;
                };
                // select_variables for up to 127 positions: two passes of the ballot scan, lane t of
                // pass h standing for i = t - 1 + 64 h, the running sum of the factors carried over
                auto wave_select2 = [&]() -> int {
                    unsigned long long valid[2], stops[2];
                    int carry = 0;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int i = lane - 1 + 64 * h;
                        const bool in = i < r;
                        const int S = wave_scan((i >= 0 && i < r) ? fzp_w[i] : 0) + carry;
                        carry = __shfl(S, 63);
                        const Float nv = i + 1 < r ? vs_w[i + 1] : (Float)0;
                        valid[h] = __ballot(in && bkmin <= S && S <= bkmax);
                        stops[h] = __ballot(in && i + 1 < r && (maximize ? nv < 0 : nv > 0));
                    }
                    if (!valid[0] && !valid[1])
                        return -2;
                    // (positions p = lane + 64 h, i = p - 1)
                    const int fv = valid[0] ? __ffsll((long long)valid[0]) - 1 : 64 + __ffsll((long long)valid[1]) - 1;
                    const unsigned long long after0 = fv < 64 ? stops[0] & ~((1ull << fv) - 1ull) : 0ull;
                    const unsigned long long after1 = fv < 64 ? stops[1] : stops[1] & ~((1ull << (fv - 64)) - 1ull);
                    const int leave = after0 ? __ffsll((long long)after0) - 1
                                             : (after1 ? 64 + __ffsll((long long)after1) - 1 : r); // r: i = r - 1
                    const unsigned long long upto0 = valid[0] & (leave >= 63 ? ~0ull : ((2ull << leave) - 1ull));
                    const unsigned long long upto1 =
                      leave < 64 ? 0ull : valid[1] & (leave - 64 >= 63 ? ~0ull : ((2ull << (leave - 64)) - 1ull));
                    const int last = upto1 ? 64 + 63 - __clzll((long long)upto1) : 63 - __clzll((long long)upto0);
                    return last - 1; // position -> i
                };
                if (mine && rlong) {
                    const bool a1 = lane + 64 < r;
                    form_val(0); // (every lane has an element `lane`: r >= 64)
                    if (a1) {
                        load_col(rs + lane + 64);
                        form_val(1);
                        l_cached = 1;
                    }
                    BXH_STAMP(10);
                    key_w[lane] = l_key[0];
                    key_w[lane + 64] = l_key[1]; // (~0 behind the row's last element: counts nowhere)
                    __builtin_amdgcn_wave_barrier();
                    const int r8 = (r + 7) & ~7;
                    for (int j0 = 0; j0 < r8; j0 += 16) {
                        ulonglong2 q[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            q[u] = reinterpret_cast<const ulonglong2*>(key_w + j0)[u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            l_rank[0] += q[u].x < l_key[0];
                            l_rank[0] += q[u].y < l_key[0];
                            l_rank[1] += q[u].x < l_key[1];
                            l_rank[1] += q[u].y < l_key[1];
                        }
                    }
                    idp_w[l_rank[0]] = lane;
                    if (a1)
                        idp_w[l_rank[1]] = lane + 64;
                    __builtin_amdgcn_wave_barrier();
                    const bool lost = idp_w[l_rank[0]] != lane || (a1 && idp_w[l_rank[1]] != lane + 64);
                    tie = __ballot(lost) != 0ull;
                    if (tie) {
                        l_rank[0] = l_rank[1] = 0;
                        for (int j0 = 0; j0 < r8; j0 += 8) {
                            ulonglong2 q[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                q[u] = reinterpret_cast<const ulonglong2*>(key_w + j0)[u];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                l_rank[0] += (q[u].x < l_key[0]) || (q[u].x == l_key[0] && j0 + 2 * u < lane);
                                l_rank[0] += (q[u].y < l_key[0]) || (q[u].y == l_key[0] && j0 + 2 * u + 1 < lane);
                                l_rank[1] += (q[u].x < l_key[1]) || (q[u].x == l_key[1] && j0 + 2 * u < lane + 64);
                                l_rank[1] += (q[u].y < l_key[1]) || (q[u].y == l_key[1] && j0 + 2 * u + 1 < lane + 64);
                            }
                        }
                        idp_w[l_rank[0]] = lane;
                        if (a1)
                            idp_w[l_rank[1]] = lane + 64;
                    }
                    __builtin_amdgcn_wave_barrier(); // (every lane has read the keys by element)
                    val_w[lane] = l_val[0];
                    vs_w[l_rank[0]] = l_val[0];
                    fzp_w[l_rank[0]] = l_fe[0];
                    if (a1) {
                        val_w[lane + 64] = l_val[1];
                        vs_w[l_rank[1]] = l_val[1];
                        fzp_w[l_rank[1]] = l_fe[1];
                    }
                    if (tie) {
                        __builtin_amdgcn_wave_barrier();
                        key_w[l_rank[0]] = l_key[0]; // by position: the runs the shuffle goes over
                        if (a1)
                            key_w[l_rank[1]] = l_key[1];
                    } else {
                        __builtin_amdgcn_wave_barrier();
                        sel = wave_select2();
                    }
                    if (lane == 0) {
                        zflag[wave] = tie ? 1 : 0;
                        zflag[4 + wave] = tie ? 0 : sub_wanted(sel);
                        zflag[8 + wave] = sel;
                    }
                }
                if (mine && !rlong) {
                    if (lane < r) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) { // (padding: any finite pi; its factor is 0)
                            const unsigned w4[4] = { c_rp[u].x, c_rp[u].y, c_rp[u].z, c_rp[u].w };
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const int rw_ = (int)(w4[t] & 0xffffu);
                                c_pi[4 * u + t] = pi_s[rw_ < pb.m ? rw_ : 0];
                                c_f[4 * u + t] = (Float)(int)(w4[t] >> 16);
                            }
                        }
                        pik_old = pi_s[k];
                        p_decayed = z_own() * theta; // decrease_preference (stored with affect's value)
#pragma unroll
                        for (int u = 0; u < 16; ++u)
                            c_pi[u] = c_pi[u] + c_p[u]; // pi + P, as the reference adds them
                        set_own(p_decayed, pik_old);
                        val = z_rc(p_decayed, pik_old);
                        if (push) // :467-469: the true column
                            val += obj_amp * c_cost;
                        key = order_key<Float>(val, maximize);
                    }
                    BXH_STAMP(10);
                    // calculator_sort: the element's position = the number of keys below its own
                    // (keys by element in LDS, read by broadcast, eight per round trip; the
                    // padding up to a multiple of eight holds the largest key).  Two equal keys
                    // land on one position, which the loser of the two stores sees: the row then
                    // counts again with the ids (the canonical tie rule) and flags the tie.
                    key_w[lane] = key; // (by element for now; lanes >= r: ~0)
                    __builtin_amdgcn_wave_barrier();
                    const int r8 = (r + 7) & ~7;
                    // (sixteen keys per round trip: the lanes from r on hold the largest key, which counts nowhere)
                    for (int j0 = 0; j0 < r8; j0 += 16) {
                        ulonglong2 q[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            q[u] = reinterpret_cast<const ulonglong2*>(key_w + j0)[u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            rank += q[u].x < key;
                            rank += q[u].y < key;
                        }
                    }
                    if (lane < r)
                        idp_w[rank] = lane;
                    __builtin_amdgcn_wave_barrier();
                    const bool lost = lane < r && idp_w[rank] != lane;
                    tie = __ballot(lost) != 0ull;
                    if (tie) {
                        rank = 0;
                        for (int j0 = 0; j0 < r8; j0 += 8) {
                            ulonglong2 q[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                q[u] = reinterpret_cast<const ulonglong2*>(key_w + j0)[u];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                rank += (q[u].x < key) || (q[u].x == key && j0 + 2 * u < lane);
                                rank += (q[u].y < key) || (q[u].y == key && j0 + 2 * u + 1 < lane);
                            }
                        }
                        if (lane < r)
                            idp_w[rank] = lane;
                    }
                    __builtin_amdgcn_wave_barrier(); // (every lane has read the keys by element)
                    if (lane < r) {
                        val_w[lane] = val;
                        vs_w[rank] = val;
                        fzp_w[rank] = fe;
                    }
                    if (tie) {
                        __builtin_amdgcn_wave_barrier();
                        if (lane < r)
                            key_w[rank] = key; // by position: the runs the shuffle goes over
                    } else {
                        __builtin_amdgcn_wave_barrier();
                        sel = wave_select();
                    }
                    if (lane == 0) {
                        zflag[wave] = tie ? 1 : 0;
                        zflag[4 + wave] = tie ? 0 : sub_wanted(sel);
                        zflag[8 + wave] = sel;
                    }
                }
                BXH_STAMP(11);
                __syncthreads(); // B1: every row of the group is sorted
                BXH_STAMP(12);

                // ---- the tie shuffles, in row order, from the one engine -----------------
                bool any_tie = false;
                for (int w = 0; w < g; ++w)
                    any_tie = any_tie || zflag[w] != 0;
                if (any_tie) {
                    if (tid == 0)
                        for (int w = 0; w < g; ++w)
                            if (zflag[w]) {
                                const int rw = t_re[s0 + w] - t_rs[s0 + w];
                                const unsigned long long* kw = w_key(w);
                                int* iw = w_idp(w);
                                int b0 = 0;
                                for (int i = 1; i <= rw; ++i)
                                    if (i == rw || kw[i] != kw[b0]) {
                                        rng.shuffle(iw + b0, (unsigned)(i - b0));
                                        b0 = i;
                                    }
                            }
                    __syncthreads();
                    if (mine && tie && rlong) {
                        const bool a1 = lane + 64 < r;
                        pos_w[idp_w[lane]] = lane;
                        if (a1)
                            pos_w[idp_w[lane + 64]] = lane + 64;
                        __builtin_amdgcn_wave_barrier();
                        l_rank[0] = pos_w[lane];
                        fzp_w[l_rank[0]] = l_fe[0];
                        if (a1) {
                            l_rank[1] = pos_w[lane + 64];
                            fzp_w[l_rank[1]] = l_fe[1];
                        }
                        __builtin_amdgcn_wave_barrier();
                        sel = wave_select2();
                        if (lane == 0) {
                            zflag[4 + wave] = sub_wanted(sel);
                            zflag[8 + wave] = sel;
                        }
                    } else if (mine && tie) { // the new positions (values of a run are equal: vs stays)
                        if (lane < r)
                            pos_w[idp_w[lane]] = lane;
                        __builtin_amdgcn_wave_barrier();
                        if (lane < r) {
                            rank = pos_w[lane];
                            fzp_w[rank] = fe;
                        }
                        __builtin_amdgcn_wave_barrier();
                        sel = wave_select();
                        if (lane == 0) {
                            zflag[4 + wave] = sub_wanted(sel);
                            zflag[8 + wave] = sel;
                        }
                    }
                    __syncthreads();
                }
                bool any_sub = false;
                for (int w = 0; w < g; ++w)
                    any_sub = any_sub || zflag[4 + w] != 0;
                if (any_sub) {
                    for (int w = 0; w < g; ++w)
                        if (zflag[4 + w]) {
                            const int slw = s0 + w;
                            subsolve(zflag[4 + w], t_k[slw], t_re[slw] - t_rs[slw], t_bmin[slw], t_bmax[slw],
                                     w_val(w), w_idp(w), w_fzp(w), w_res(w));
                            if (tid == 0)
                                zflag[8 + w] = misc_s[MS_SEL];
                            __syncthreads();
                        }
                    if (mine && zflag[4 + wave] && rlong) {
                        const bool a1 = lane + 64 < r;
                        sel = zflag[8 + wave];
                        pos_w[idp_w[lane]] = lane;
                        if (a1)
                            pos_w[idp_w[lane + 64]] = lane + 64;
                        __builtin_amdgcn_wave_barrier();
                        l_rank[0] = pos_w[lane];
                        vs_w[l_rank[0]] = l_val[0];
                        fzp_w[l_rank[0]] = l_fe[0];
                        if (a1) {
                            l_rank[1] = pos_w[lane + 64];
                            vs_w[l_rank[1]] = l_val[1];
                            fzp_w[l_rank[1]] = l_fe[1];
                        }
                    } else if (mine && zflag[4 + wave]) { // R's new order: positions, values, factors
                        sel = zflag[8 + wave];
                        if (lane < r)
                            pos_w[idp_w[lane]] = lane;
                        __builtin_amdgcn_wave_barrier();
                        if (lane < r) {
                            rank = pos_w[lane];
                            vs_w[rank] = val;
                            fzp_w[rank] = fe;
                        }
                    }
                    __syncthreads();
                }

                // ---- local_affect (:346-439): pi and d, then every variable ---------------
                BXH_STAMP(13);
                if (mine) {
                    auto stale_at = [&](int pp) -> Float { // R[pp] as the rows in front left it
                        for (int w = wave - 1; w >= 0; --w)
                            if (t_re[s0 + w] - t_rs[s0 + w] > pp)
                                return w_vs(w)[pp];
                        return stale_s[pp <= pb.rmax ? pp : pb.rmax];
                    };
                    auto Rv = [&](int i) -> Float { return i < r ? vs_w[i] : stale_at(i); };
                    const Float one = 1, two = 2;
                    const Float middle = (two + one) / two;
                    const Float old_pi = pi_s[k];
                    Float new_pi;
                    if (sel < 0)
                        new_pi = old_pi + Rv(0) / two;
                    else if (sel + 1 >= r)
                        new_pi = old_pi + Rv(sel) * middle;
                    else
                        new_pi = old_pi + ((Rv(sel) + Rv(sel + 1)) / two);
                    if (lane == 0) {
                        pi_s[k] = new_pi;
                        if (signbit(old_pi) != signbit(new_pi))
                            atomicOr(&misc_s[MS_PICH], 1);
                    }
                    BXH_STAMP(1);
                    const Float d = (kappa / (one - kappa)) + delta;
                    if (rlong) {
                        // the lane's two elements: the one whose column is in the registers first,
                        // then the other (its column loaded again: nothing of it has changed but the
                        // row's own entry, which set_own puts in)
#pragma unroll
                        for (int step = 0; step < 2; ++step) {
                            const int hh = step == 0 ? l_cached : 1 - l_cached;
                            if (lane + 64 * hh < r) {
                                if (step == 1) {
                                    load_col(rs + lane + 64 * hh);
                                    fill_pi();
                                }
                                c_cost = l_cost[hh];
                                const int j = c_desc.w;
                                const int pe = c_desc.z;
                                const bool up = sel < 0 ? false : (sel + 1 >= r ? true : l_rank[hh] <= sel);
                                Float pp = up ? l_pdec[hh] + d : l_pdec[hh] - d;
                                if (up)
                                    atomicOr(&x_s[j >> 5], 1u << (j & 31));
                                else
                                    atomicAnd(&x_s[j >> 5], ~(1u << (j & 31)));
                                set_own(pp, new_pi);
                                const Float repair = z_rc(pp, new_pi);
                                if (up) {
                                    if (repair >= 0)
                                        pp = pp - repair + d;
                                } else {
                                    if (repair <= 0)
                                        pp = pp + repair - d;
                                }
                                P[pe] = pp;
                            }
                        }
                    } else if (lane < r) {
                        const int j = c_desc.w;
                        const int pe = c_desc.z;
                        const bool up = sel < 0 ? false : (sel + 1 >= r ? true : rank <= sel);
                        // (one walk of the column for both cases: the lanes of a wave take both)
                        Float pp = up ? p_decayed + d : p_decayed - d;
                        if (up)
                            atomicOr(&x_s[j >> 5], 1u << (j & 31));
                        else
                            atomicAnd(&x_s[j >> 5], ~(1u << (j & 31)));
                        set_own(pp, new_pi);
                        const Float repair = z_rc(pp, new_pi);
                        if (up) {
                            if (repair >= 0)
                                pp = pp - repair + d;
                        } else {
                            if (repair <= 0)
                                pp = pp + repair - d;
                        }
                        P[pe] = pp;
                    }
                    BXH_STAMP(2);
                }
                BXH_STAMP(14);
                __syncthreads(); // B2: the group is done (P in memory, pi / x in LDS)
                BXH_STAMP(15);
                // R as the reference leaves it: the rows of the group wrote it in turn, so a
                // position keeps the value of the LAST row long enough to reach it
                if (mine) {
                    int later = 0;
                    for (int w = wave + 1; w < g; ++w)
                        later = max(later, t_re[s0 + w] - t_rs[s0 + w]);
                    if (lane < r && lane >= later)
                        stale_s[lane] = vs_w[lane];
                    if (lane + 64 < r && lane + 64 >= later)
                        stale_s[lane + 64] = vs_w[lane + 64];
                }
                if (mine)
                    alg_w += t_rb[sl]; // (lane 0 of the wave counts; summed at the end)
                it += g;
                if (it >= nrows || (it + 2 * NW > tbase + ZW_TABLE && tbase + ZW_TABLE < nrows)) {
                    z_again = true;
                    break;
                }
                g = __builtin_amdgcn_readfirstlane(t_b[it - tbase].z);
            }
            if (z_again) {
                --it;
                continue;
            }
            // the general path for this row (the table holds its scalars)
            const int sl = it - tbase;
            cur.k = t_k[sl];
            cur.rs = t_rs[sl];
            cur.re = t_re[sl];
            cur.zk = t_z[sl];
            cur.bmin = t_bmin[sl];
            cur.bmax = t_bmax[sl];
            cur.rbytes = t_rb[sl];
            pre_ok = false;
            __syncthreads(); // (the stale values the last group wrote)
        } else {
            if (it + 1 < nrows)
                nxt = row_of(k_ahead);
            if (it + 2 < nrows)
                k_ahead = row_id(it + 2);
        }
        const int k = cur.k;
        const int rs = cur.rs;
        const int r = cur.re - cur.rs;
        const int zk = cur.zk;

        // ---- decrease_preference + compute_reduced_costs (+ push) ----------
        // A row that fits the workgroup, linear objective: lane = row element.
        // The first 16 slots of the lane's column (all of it for columns of up
        // to 16 entries) are fetched in ONE batch -- P, |a|, row -- and kept in
        // registers: the repair pass of local_affect walks the same column again
        // and nothing in it has changed by then but pi_k and the row's own P
        // entry, so it needs no second trip to memory.  (The descriptor of the
        // lane's element was fetched while the previous row was worked on.)
        const bool fastrow = pb.q_ptr == nullptr && r <= T;
        Float c_p[16];
        int c_f[16], c_r[16];
        int4 c_desc = make_int4(0, 0, 0, 0);
        Float c_cost = 0;
        if (fastrow) {
            if (tid < r) {
                const int e = rs + tid;
                int2 cs;
                int j, pe, fe;
                if (pre_ok) { // prefetched a row ago
                    cs = pre_cs;
                    j = pre_j;
                    pe = pre_pe;
                    fe = pre_f;
                } else {
                    cs = pb.row_cs[e];
                    j = pb.row_col[e];
                    pe = pb.row_pos[e];
                    fe = pb.fac32[e];
                }
                const int padlen = cs.y & 0xffff;
                c_desc = make_int4(cs.x, padlen, pe, j);
                desc_s[tid] = c_desc;
                res_s[tid] = fe;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    c_p[u] = P[cs.x + u];
                    c_f[u] = pb.col_fac[cs.x + u];
                    c_r[u] = pb.col_row[cs.x + u];
                }
                c_cost = cost[j];
            }
        }
        // the next row's descriptors: on their way while this row is worked on
        pre_ok = false;
        if (!ZW && it + 1 < nrows && pb.q_ptr == nullptr && nxt.re - nxt.rs <= T) {
            pre_ok = true;
            if (tid < nxt.re - nxt.rs) {
                const int e = nxt.rs + tid;
                pre_cs = pb.row_cs[e];
                pre_j = pb.row_col[e];
                pre_pe = pb.row_pos[e];
                pre_f = pb.fac32[e];
            }
        }
        // sum over the lane's column with `pown` for the row's own entry: the
        // cached slots first, then (columns of more than 16 entries) the rest
        auto cached_rc = [&](Float pown) -> Float {
            Float acc = 0;
            const int q0 = c_desc.x, pe = c_desc.z;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const Float pp = q0 + u == pe ? pown : c_p[u];
                const int ri = c_r[u] < pb.m ? c_r[u] : 0; // padding: any finite pi
                acc += (Float)c_f[u] * (pi_s[ri] + pp);
            }
            for (int qb = q0 + 16; qb < q0 + c_desc.y; qb += 8) {
                Float p8[8];
                int f8[8], r8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    p8[u] = P[qb + u];
                    f8[u] = pb.col_fac[qb + u];
                    r8[u] = pb.col_row[qb + u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const Float pp = qb + u == pe ? pown : p8[u];
                    const int ri = r8[u] < pb.m ? r8[u] : 0;
                    acc += (Float)f8[u] * (pi_s[ri] + pp);
                }
            }
            return c_cost - acc;
        };
        auto cached_own = [&]() -> Float { // P[pe] as it stands in memory
            const int off = c_desc.z - c_desc.x;
            Float v = 0;
            if (off < 16) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    v = u == off ? c_p[u] : v;
            } else
                v = P[c_desc.z];
            return v;
        };
        if (fastrow) {
            if (tid < r) {
                Float v = cached_rc(cached_own() * theta);
                if (push) // :467-469: the true column
                    v += obj_amp * c_cost;
                val_s[tid] = v;
            }
        } else
            for (int idx = tid; idx < r; idx += T) {
                // three independent loads from e (no chase through the column id)
                const int e = rs + idx;
                const int2 cs = pb.row_cs[e];
                const int j = pb.row_col[e];
                const int pe = pb.row_pos[e];
                const int padlen = cs.y & 0xffff;
                desc_s[idx] = make_int4(cs.x, padlen, pe, j);
                res_s[idx] = pb.fac32[e]; // factor by row-local id (res_s is free until the sub-solvers)
                Float v = column_rc(j, cs.x, padlen, pe, P[pe] * theta);
                if (push) // :467-469: the true column
                    v += obj_amp * cost_x<Float>(pb, x_s, j, cost[j]);
                val_s[idx] = v;
            }
        const int n2 = next_pow2(r < 2 ? 2 : r);
        if (tid == 0)
            misc_s[MS_TIE] = 0;
        BXH_STAMP(0);
        __syncthreads();
        BXH_STAMP(1);

        // ---- calculator_sort (every kind but the plain exhaustive one) -----
        if (zk != Z_EXHAUSTIVE) {
            if (r > 1 && r <= T && r <= 160) {
                rank_sort_small<Float>(val_s, r, maximize, key_run, key_s, id_s);
            } else if (r > 1 && r <= 2 * T && r <= 1024) {
                // wave-sorted runs + rank search: 2 barriers instead of ~25
                Prof noprof;
                noprof.t = 0;
                run_merge_sort<Float>(val_s, r, maximize, key_run, id_run, key_s, id_s, &noprof);
            } else {
                for (int i = tid; i < n2; i += T) {
                    key_s[i] = i < r ? order_key<Float>(val_s[i], maximize) : ~0ull;
                    id_s[i] = i;
                }
                __syncthreads();
                if (r > 1)
                    bitonic_sort(key_s, id_s, n2);
            }
            for (int i = tid; i + 1 < r; i += T)
                if (key_s[i] == key_s[i + 1])
                    misc_s[MS_TIE] = 1;
            __syncthreads();
            bool shuffled = false;
            if (misc_s[MS_TIE] && r > 64 && r <= 2 * T && r <= 1024) {
                if (tid == 0) // the engine lives in thread 0 in this kernel
                    misc_s[MS_RNG] = (int)rng.s;
                __syncthreads();
                unsigned stt = (unsigned)misc_s[MS_RNG];
                shuffled = parallel_tie_shuffle(key_s, id_s, r, &stt, pb.lcg_pow, (int*)key_run,
                                                (int*)key_run + n2max, id_run, misc_s + MS_WAVES);
                if (shuffled)
                    rng.s = stt;
            }
            if (tid == 0 && misc_s[MS_TIE] && !shuffled) {
                int b0 = 0;
                for (int i = 1; i <= r; ++i) {
                    if (i == r || key_s[i] != key_s[b0]) {
                        rng.shuffle(id_s + b0, (unsigned)(i - b0));
                        b0 = i;
                    }
                }
            }
        } else {
            for (int i = tid; i < r; i += T)
                id_s[i] = i;
        }
        __syncthreads();
        BXH_STAMP(2);
        for (int i = tid; i < r; i += T)
            fz_s[i] = res_s[id_s[i]];
        __syncthreads();
        BXH_STAMP(3);

        // ---- selection ------------------------------------------------------
        const int bkmin = cur.bmin, bkmax = cur.bmax;
        // select_variables (:308-344) on all lanes when the row fits the
        // workgroup: with S_i the running sum of the factors of R[0..i]
        // (S_-1 = 0), the scan keeps the last i whose S_i is within the bounds
        // and leaves at the first i, at or after the first such one, whose
        // successor R[i + 1] has a stopping value (R[r] as the reference's
        // array still holds it).
        const bool scan_kind = zk == Z_LINEAR || zk == Z_UNCONSTRAINED_BB ||
                               zk == Z_UNCONSTRAINED_EXHAUSTIVE;
        // a row of fewer than 64 elements: the same scan inside wave 0, on
        // ballots (no barrier): lane t stands for i = t - 1
        const bool wave_select = scan_kind && r < 64;
        if (wave_select && tid < 64) {
            const int i = tid - 1; // -1 .. 62
            const bool in = i < r;
            const int f = (i >= 0 && i < r) ? fz_s[i] : 0;
            int S = f; // inclusive scan over i = -1 .. (S_-1 = 0)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(S, off, 64);
                if (tid >= off)
                    S += o;
            }
            const Float nv = i + 1 < r ? val_s[id_s[i + 1]] : stale_s[i + 1 <= pb.rmax ? i + 1 : pb.rmax];
            const unsigned long long valid = __ballot(in && bkmin <= S && S <= bkmax);
            const unsigned long long stops = __ballot(in && (maximize ? nv < 0 : nv > 0));
            int sel = -2;
            if (valid) {
                const int fv = __ffsll((long long)valid) - 1;         // lane of the first valid i
                const unsigned long long after = stops & ~((1ull << fv) - 1ull);
                const int leave = after ? __ffsll((long long)after) - 1 : r; // lane; r == i = r - 1
                const unsigned long long upto = valid & (leave >= 63 ? ~0ull : ((2ull << leave) - 1ull));
                sel = (63 - __clzll((long long)upto)) - 1; // lane -> i
            }
            if (tid == 0)
                (misc_s + MS_WAVES)[34] = sel;
        }
        const bool par_select = !wave_select && r <= T && scan_kind;
        if (par_select) {
            const int lane = tid & 63, wave = tid >> 6, nw = T >> 6;
            int* wsum = misc_s + MS_WAVES; // per-wave totals, then the three results
            const int f = tid < r ? fz_s[tid] : 0;
            int incl = f; // inclusive scan inside the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off)
                    incl += o;
            }
            if (lane == 63)
                wsum[wave] = incl;
            if (tid == 0) {
                wsum[32] = 0x7fffffff; // first valid i
                wsum[33] = 0x7fffffff; // leaving i
                wsum[34] = -2;         // best
            }
            __syncthreads();
            int S = incl;
            for (int w = 0; w < wave; ++w)
                S += wsum[w];
            auto stops = [&](Float v) -> bool { return maximize ? v < 0 : v > 0; };
            const bool in_row = tid < r;
            const bool valid = in_row && bkmin <= S && S <= bkmax;          // i = tid
            const bool valid_m1 = bkmin <= 0 && 0 <= bkmax;                 // i = -1
            const Float next_v = tid + 1 < r ? val_s[id_s[tid + 1]] : stale_s[tid + 1 <= pb.rmax ? tid + 1 : pb.rmax];
            const bool stop_i = in_row && stops(next_v);                    // R[i + 1]
            const bool stop_m1 = stops(r > 0 ? val_s[id_s[0]] : stale_s[0]);
            if (valid)
                atomicMin(&wsum[32], tid);
            if (tid == 0 && valid_m1)
                atomicMin(&wsum[32], -1);
            __syncthreads();
            const int first_valid = wsum[32];
            if (first_valid != 0x7fffffff) {
                if (stop_i && tid >= first_valid)
                    atomicMin(&wsum[33], tid);
                if (tid == 0 && stop_m1 && -1 >= first_valid)
                    atomicMin(&wsum[33], -1);
            }
            __syncthreads();
            const int leave = wsum[33] == 0x7fffffff ? r - 1 : wsum[33];
            if (first_valid != 0x7fffffff) {
                if (valid && tid <= leave)
                    atomicMax(&wsum[34], tid);
                if (tid == 0 && valid_m1 && -1 <= leave)
                    atomicMax(&wsum[34], -1);
            }
            __syncthreads();
        }
        if (tid == 0) {
            alg_bytes += cur.rbytes;
            int sel = -2;
            if (r > T && scan_kind) {
                // select_variables, :308-344 (reads R[i + 1] up to R[r])
                auto Rv = [&](int i) -> Float { return i < r ? val_s[id_s[i]] : stale_s[i]; };
                int sum = 0;
                for (int i = -1; i < r; ++i) {
                    if (bkmin <= sum && sum <= bkmax)
                        sel = i;
                    if (sel != -2) {
                        const Float v = Rv(i + 1);
                        // stop_iterating without an engine, itm-common.hpp:182-190
                        if (maximize ? v < 0 : v > 0)
                            break;
                    }
                    if (i + 1 < r)
                        sum += fz_s[i + 1];
                    // (i + 1 == r: the reference adds the stale R[r].f and leaves)
                }
            }
            if (par_select || wave_select)
                sel = (misc_s + MS_WAVES)[34];
            misc_s[MS_SEL] = sel;
            misc_s[MS_SELMODE] = 0; // 1: B&B wanted, 2: exhaustive wanted
            if (zk == Z_BB || (zk == Z_UNCONSTRAINED_BB && sel == -2))
                misc_s[MS_SELMODE] = 1;
            if (zk == Z_EXHAUSTIVE || (zk == Z_UNCONSTRAINED_EXHAUSTIVE && sel == -2))
                misc_s[MS_SELMODE] = 2;
        }
        __syncthreads();
        const int submode = misc_s[MS_SELMODE];
        subsolve(submode, k, r, bkmin, bkmax, val_s, id_s, fz_s, res_s);

        BXH_STAMP(4);
        // ---- local_affect, :346-439: pi and d first ----------------------------
        if (tid == 0) {
            const int sel = misc_s[MS_SEL];
            auto Rv = [&](int i) -> Float { return i < r ? val_s[id_s[i]] : stale_s[i]; };
            const Float one = 1, two = 2;
            const Float middle = (two + one) / two;
            const Float old_pi = pi_s[k];
            Float new_pi;
            if (sel < 0)
                new_pi = old_pi + Rv(0) / two;
            else if (sel + 1 >= r)
                new_pi = old_pi + Rv(sel) * middle;
            else
                new_pi = old_pi + ((Rv(sel) + Rv(sel + 1)) / two);
            pi_s[k] = new_pi;
            if (signbit(old_pi) != signbit(new_pi))
                misc_s[MS_PICH] = 1;
            *d_s = (kappa / (one - kappa)) + delta;
        }
        __syncthreads();

        BXH_STAMP(5);
        // ---- every variable: x, P, then its repaired preference ------------
        if (fastrow) {
            // by row element (the lane that holds the column's slots): its
            // position in R comes from the inverse of the sorted order
            const int sel = misc_s[MS_SEL];
            const Float d = *d_s;
            int* pos_s = (int*)key_run; // (free since the sort)
            if (tid < r)
                pos_s[id_s[tid]] = tid;
            __syncthreads();
            if (tid < r) {
                const int i = pos_s[tid];
                const int j = c_desc.w;
                const int pe = c_desc.z;
                const bool up = sel < 0 ? false : (sel + 1 >= r ? true : i <= sel);
                Float p = cached_own() * theta;
                if (up) {
                    atomicOr(&x_s[j >> 5], 1u << (j & 31));
                    p = p + d;
                    const Float repair = cached_rc(p);
                    if (repair >= 0)
                        p = p - repair + d;
                } else {
                    atomicAnd(&x_s[j >> 5], ~(1u << (j & 31)));
                    p = p - d;
                    const Float repair = cached_rc(p);
                    if (repair <= 0)
                        p = p + repair - d;
                }
                P[pe] = p;
            }
        } else {
            const int sel = misc_s[MS_SEL];
            const Float d = *d_s;
            // with a quadratic objective c(j, x) of a later variable sees the
            // x bits set for the earlier ones: one thread, in R order
            const bool serial = pb.q_ptr != nullptr;
            for (int i = serial ? 0 : tid; i < (serial && tid != 0 ? 0 : r); i += serial ? 1 : T) {
                const int idx = id_s[i];
                const int4 dd = desc_s[idx];
                const int j = dd.w;
                const int pe = dd.z;
                const bool up = sel < 0 ? false : (sel + 1 >= r ? true : i <= sel);
                Float p = P[pe] * theta;
                if (up) {
                    atomicOr(&x_s[j >> 5], 1u << (j & 31));
                    p = p + d;
                    const Float repair = column_rc(j, dd.x, dd.y, pe, p);
                    if (repair >= 0)
                        p = p - repair + d;
                } else {
                    atomicAnd(&x_s[j >> 5], ~(1u << (j & 31)));
                    p = p - d;
                    const Float repair = column_rc(j, dd.x, dd.y, pe, p);
                    if (repair <= 0)
                        p = p + repair - d;
                }
                P[pe] = p;
            }
        }
        __syncthreads();
        BXH_STAMP(6);
        // R as the reference leaves it (values by final position)
        for (int i = tid; i < r; i += T)
            stale_s[i] = val_s[id_s[i]];
        __syncthreads();
    }

    for (int i = tid; i < pb.m; i += T)
        pig[i] = pi_s[i];
    for (int i = tid; i < 2 * pb.nwords; i += T)
        xg[i] = x_s[i];
    for (int i = tid; i <= pb.rmax; i += T)
        staleg[i] = stale_s[i];

    const int cnt = rebuild_violated(
      pb, x_s, st.vflag + (size_t)inst * pb.m, out_list, misc_s + MS_WAVES);
    BXH_STAMP(7);
#ifdef BXH_PROFILE
    if (tid == 0)
        for (int i = 0; i < 16; ++i)
            st.prof[(size_t)inst * 32 + i] += prof.acc[i];
#endif

    if (ZW) { // the rows the waves took by themselves
        if (lane == 0)
            t_rb[wave] = alg_w;
        __syncthreads();
        if (tid == 0)
            for (int w = 0; w < NW; ++w)
                alg_bytes += t_rb[w];
    }
    if (tid == 0) {
        if (!keep)
            st.vcount[inst] = cnt;
        DevStatus s = st.status[inst];
        s.remaining = cnt;
        s.pi_changed = misc_s[MS_PICH];
        s.rng = rng.s;
        s.rows_in_sweep = (unsigned)nrows;
        s.row_updates += (unsigned long long)nrows;
        s.alg_bytes += alg_bytes;
        st.status[inst] = s;
    }
}

// 16 / 4 bytes per lane from global memory straight into LDS, no registers: lane L's bytes
// land at lds_dst + 16 * L (4 * L); lds_dst is wave-uniform.  Counted by vmcnt like any load;
// the data may be read from LDS once the issuing wave has waited for it (and a barrier has
// passed, for the other waves).
__device__ inline void
lds_dma16(const void* src, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n"
                 "s_mov_b32 m0, %2\n"
                 "s_nop 0\n"
                 "global_load_lds_dwordx4 %1, off\n"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}
__device__ inline void
lds_dma4(const void* src, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n"
                 "s_mov_b32 m0, %2\n"
                 "s_nop 0\n"
                 "global_load_lds_dword %1, off\n"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}

struct LdsPlan2
{
    int off_pi, off_x, off_val, off_sp, off_cs, off_pe, off_col, off_cost, off_pown, off_spi,
      off_oidx, off_loc, off_stale, off_misc, off_part, off_union, stage_i_off, total;
    // pipelined sweep (role 1: row chain, role 2: producer)
    int off_mark, off_map, off_flag, off_lnp, off_markl, off_newp, off_patch, off_htab, off_land;
};

template<bool V>
struct PipeScope // compile-time scope of a P load
{
    static constexpr bool value = V;
};
constexpr int PIPE_RING = 4; // row slots of the producers' value ring (> lookahead)

// The row descriptors (cs, pe, col, cost, oidx, loc) are double buffered: the
// one of row it + 1 is written while the loads of row it are in flight.
// role 0: the lock-step kernel; role 1 / 2: the row chain / a producer of the
// pipelined sweep (n = number of variables: one 16-bit mark per column).
__host__ __device__ inline LdsPlan2
lds_plan2(int m, int nwords, int rmax, int romax, int fsize, int slots, int np, bool mw,
          int role = 0, int n = 0)
{
    auto al = [](int v) { return (v + 15) & ~15; };
    LdsPlan2 p;
    int o = 0;
    p.off_pi = o;
    o = al(o + (m + 1) * fsize); // + one zero entry read through column padding
    p.off_x = o;
    if (role != 2)
        o = al(o + nwords * 8);
    p.off_val = o;
    if (role != 2)
        o = al(o + rmax * fsize);
    p.off_sp = o;
    if (role == 0) // (the chain's late waves and the producers combine in registers)
        o = al(o + romax * fsize);
    p.off_cs = o;
    o = al(o + 2 * romax * 8);
    p.off_pe = o;
    o = al(o + 2 * romax * 4);
    p.off_col = o;
    o = al(o + 2 * romax * 4);
    p.off_cost = o;
    o = al(o + 2 * romax * fsize);
    p.off_pown = o; // the row's own P entries after the decay
    if (role != 2)
        o = al(o + romax * fsize);
    p.off_spi = p.off_val; // one owner: the pi sums sit in val_s until combined
    p.off_oidx = o;
    p.off_loc = o;
    if (mw) { // several owners: own arrays are indexed by the owner's position
        p.off_spi = o;
        o = al(o + romax * fsize);
        p.off_oidx = o;
        o = al(o + 2 * romax * 4);
        p.off_loc = o;
        o = al(o + 2 * rmax * 2);
    }
    p.off_stale = o;
    if (role != 2)
        o = al(o + (rmax + 1) * fsize);
    p.off_misc = o;
    o = al(o + 96 * 4);
    p.off_part = o; // partial reductions of the sort-free selection
    if (role != 2)
        o = al(o + (64 + 8) * 8);
    p.off_mark = p.off_map = p.off_flag = p.off_lnp = p.off_markl = p.off_newp = p.off_patch = o;
    p.off_land = o;
    if (role == 1) {
        // one 16-bit mark per column: code of the last row position that held it << 10 |
        // index of the column's entry in that row (the producers keep the code in a byte)
        p.off_markl = o;
        o = al(o + n * 2);
        p.off_land = o; // the next row's descriptor records, as LDS-DMA drops them (whole 1 KB pieces)
        o = al(o + ((rmax + 63) & ~63) * 16);
        p.off_newp = o; // what affect stored for every entry of the row
        o = al(o + rmax * fsize);
        p.off_patch = o; // late entry -> {entry of the previous row in the same column or -1,
                         //                its position in the column}
        o = al(o + 2 * rmax * 4);
    }
    if (role != 0) {
        p.off_mark = o; // producers: code (1..63, by position mod 63) of the last row that held column j
        if (role == 2)
            o = al(o + n);
        p.off_map = o; // chain: the row's late entries; producer: entry or -1
        o = al(o + 2 * rmax * 2);
        p.off_flag = o; // chain: 1 = the entry's column is summed here
        if (role == 1)
            o = al(o + 2 * rmax);
        p.off_lnp = o; // chain: largest piece count of every late chunk
        if (role == 1)
            o = al(o + 2 * ((rmax + 15) / 16) * 4);
    }
    p.off_htab = o;
    if (role == 1) // the tie filter's bitmap: not over the staging slots, which the late
        o = al(o + 512 * 16); // waves may still be reading when the other waves clear it
    p.off_union = o;
    if (role != 1)
        p.off_htab = o;
    // staging: per slot np*16 (+2) Float and np*16 (+4) 16-bit row indices
    const int stage_p = al(slots * (np * 16 + 2) * fsize);
    const int stage = stage_p + al(slots * (np * 16 + 4) * 2);
    p.stage_i_off = stage_p;
    const int n2 = next_pow2(rmax < 64 ? 64 : rmax); // whole 64-entry runs
    const int sortb = role == 2 ? 0 : 2 * (al(n2 * 8) + al(n2 * 4)); // runs + merged output
    o = al(o + (stage > sortb ? stage : sortb));
    p.total = o;
    return p;
}

// MW: several workgroups cooperate on one instance.  Column j belongs to
// workgroup j % groups, which alone reads and writes the P entries and the x
// bit of that column; every workgroup keeps its own copy of pi, of the row's
// reduced costs (exchanged once per row through L2) and of the engine, and
// takes the same selection decisions from them.
//
// PIPE: the pipelined sweep.  Row k + 1 reads what row k wrote only through
// the columns the two rows share: pi_k enters the sums of the columns of row
// k alone, and so do the P entries row k changed.  The reduced cost of a
// column of row p is therefore final as soon as the last row before p that
// holds the column is finished -- on average c_j^-1 * m rows earlier.  One
// workgroup per instance (the row chain, grp 0) runs select / affect row
// after row and only sums the columns that one of the previous `lookahead`
// rows holds (the late columns, ~5 % per row of look-ahead at the scpnrh
// shape); pb.pipe - 1 producer workgroups run `lookahead` rows ahead of it,
// sum the other columns exactly as above and hand the reduced costs over
// through a ring of tagged granules in L2 (never a round trip: the chain
// publishes {row finished, pi_k} messages and does not wait for anybody but
// the values of the row it is at).  One CSC copy / P array per instance
// (pb.groups == 1 layout).  Every sum keeps its order, so the results stay
// bit-identical to the lock-step kernels'.
template<typename Float, bool PM1, int NP, bool MW, bool PIPE = false>
__global__ __launch_bounds__(512) void
sweep_kernel_v2(DevProblem pb, DevState st, DevSweep a)
{
    static_assert(!(MW && PIPE), "the pipelined sweep has one P array per instance");
#ifdef BXH_DEV_ONLY_NP
    // development builds (A/B libraries loaded with BXH_LIB, never the shipped one): only the
    // fp64 01-solver forms with this piece count keep their bodies -- a build in a fraction of
    // the time; every other form of this kernel returns at once
    if (!(NP == BXH_DEV_ONLY_NP && sizeof(Float) == 8 && !PM1))
        return;
#endif
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int LR = 2;
    const int NG = MW ? pb.groups : (PIPE ? pb.pipe : 1);
    int inst = blockIdx.x, grp = 0;
    if (MW || PIPE) {
        const int b = blockIdx.x, ninst = gridDim.x / NG;
        if ((ninst & 7) == 0) {
            // blocks 8 apart share an XCD under round-robin dispatch: keep the
            // partners of an instance on one L2 (speed only)
            inst = (b & 7) + 8 * ((b >> 3) / NG);
            grp = (b >> 3) % NG;
        } else {
            inst = b / NG;
            grp = b % NG;
        }
    }
    if (a.active && !a.active[inst])
        return;

    const int tid = threadIdx.x, T = blockDim.x;
    const int role = PIPE ? (grp == 0 ? 1 : 2) : 0;
    const int SW = PIPE ? (grp == 0 ? pb.sw_chain : pb.sw_prod) : (T >> 6); // waves with staging slots
    const int G = PIPE ? SW * 8 * LR
                       : ((T >> 6) + pb.spare) * 8 * LR; // staging slots: 8 * LR columns per wave (+ a spare set)
    constexpr int SP = NP * 16 + 2; // Float entries of a P slot (16-byte rows)
    constexpr int SI = NP * 16 + 4; // 16-bit entries of an index slot (8-byte rows)
    typedef Float Float2 __attribute__((ext_vector_type(2)));
    const LdsPlan2 lp =
      lds_plan2(pb.m, pb.nwords, pb.rmax, MW ? pb.romax : pb.rmax, (int)sizeof(Float), G, NP, MW,
                role, pb.n);
    Float* pi_s = (Float*)(smem + lp.off_pi);
    unsigned* x_s = (unsigned*)(smem + lp.off_x);
    Float* val_s = (Float*)(smem + lp.off_val);
    Float* sp_s = (Float*)(smem + lp.off_sp);
    Float* spi_s = (Float*)(smem + lp.off_spi); // == val_s with one owner
    // double-buffered row descriptors (buffer it & 1 holds row it)
    const int RO = MW ? pb.romax : pb.rmax;
    int2* const cs_b = (int2*)(smem + lp.off_cs);
    int* const pe_b = (int*)(smem + lp.off_pe);
    int* const col_b = (int*)(smem + lp.off_col);
    Float* const cost_b = (Float*)(smem + lp.off_cost);
    int* const oidx_b = (int*)(smem + lp.off_oidx);
    short* const loc_b = (short*)(smem + lp.off_loc);
    Float* pown_s = (Float*)(smem + lp.off_pown);
    Float* stale_s = (Float*)(smem + lp.off_stale);
    int* misc_s = (int*)(smem + lp.off_misc);
    Float* d_s = (Float*)(misc_s + MS_D);
    // union: staging during the reduced-cost passes, sort buffers afterwards
    Float* stg_p = (Float*)(smem + lp.off_union);
    unsigned short* stg_i = (unsigned short*)(smem + lp.off_union + lp.stage_i_off);
    const int n2max = next_pow2(pb.rmax < 64 ? 64 : pb.rmax); // whole 64-entry runs
    unsigned long long* key_s = (unsigned long long*)(smem + lp.off_union);
    int* id_s = (int*)(smem + lp.off_union + ((n2max * 8 + 15) & ~15));
    // tie-detection table of the sort-free path: 2 * n2max 64-bit slots, over
    // the same bytes (the two paths never run for the same row)
    unsigned long long* htab = (unsigned long long*)(smem + lp.off_htab);
    unsigned long long* part_s = (unsigned long long*)(smem + lp.off_part);
    const int sort_half = ((n2max * 8 + 15) & ~15) + ((n2max * 4 + 15) & ~15);
    unsigned long long* key_run = (unsigned long long*)(smem + lp.off_union + sort_half);
    int* id_run = (int*)(smem + lp.off_union + sort_half + ((n2max * 8 + 15) & ~15));

    const bool maximize = pb.maximize != 0;
    const int kind = pb.kind;

    Float* P = (Float*)st.P + (MW ? ((size_t)inst * NG + grp) : (size_t)inst) * pb.nnzp;
    const unsigned short* crow = pb.col_row16 + (MW ? (size_t)grp * pb.nnzp : (size_t)0);
    // pipelined sweep: column marks, late list / early map, flags
    unsigned char* const mark_s = (unsigned char*)(smem + lp.off_mark);
    short* const map_b = (short*)(smem + lp.off_map);
    unsigned char* const flag_b = (unsigned char*)(smem + lp.off_flag);
    int* const lnp_b = (int*)(smem + lp.off_lnp);
    unsigned short* const markl_s = (unsigned short*)(smem + lp.off_markl);
    Float* const newp_s = (Float*)(smem + lp.off_newp);
    short2* const patch_b = (short2*)(smem + lp.off_patch);
    // row chain: the landing buffer of the next row's descriptor records, and the cost of the
    // columns of late entries (low / high words apart: that is how LDS-DMA drops them)
    const int4* const land_s = (const int4*)(smem + lp.off_land);
    unsigned* const lc_lo = (unsigned*)(smem + lp.off_cost);
    unsigned* const lc_hi = lc_lo + 2 * RO;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto chain_cost = [&](int buf, int l) -> Float {
        if (sizeof(Float) == 8) {
            const unsigned long long b =
              (unsigned long long)lc_lo[buf * RO + l] | ((unsigned long long)lc_hi[buf * RO + l] << 32);
            return (Float)__longlong_as_double((long long)b);
        }
        return (Float)__uint_as_float(lc_lo[buf * RO + l]);
    };
    auto chain_cost_store = [&](int buf, int l, Float c) {
        if (sizeof(Float) == 8) {
            const unsigned long long b = (unsigned long long)__double_as_longlong((double)c);
            lc_lo[buf * RO + l] = (unsigned)b;
            lc_hi[buf * RO + l] = (unsigned)(b >> 32);
        } else
            lc_lo[buf * RO + l] = __float_as_uint((float)c);
    };
    unsigned* const cnt_row = st.cnt + inst;
    unsigned* const cnt_end = st.cnt + gridDim.x / NG + inst;
    unsigned* const err_word = st.cnt + 2 * (gridDim.x / NG);
    Float* pig = (Float*)st.pi + (size_t)inst * pb.m;
    unsigned* xg = st.x + (size_t)inst * (2 * pb.nwords);
    Float* staleg = (Float*)st.rstale + (size_t)inst * (pb.rmax + 1);
    // compute_order::run / push_and_run by this instance's order
    const int ord = order_of(a, inst);
    const bool use_m = ord == 3 || ord == 4;          // rows of the sorted m_order
    const bool keep = use_m || ord == 7;              // R is not recomputed
    int* vlist = use_m ? st.mrun + (size_t)inst * st.lcap : st.vlist + (size_t)inst * st.lcap;
    int* out_list = keep ? st.slist + (size_t)inst * st.lcap : st.vlist + (size_t)inst * st.lcap;
    // several owners: each works on a private copy of the row list
    int* vrows = (MW || PIPE) ? st.vwork + ((size_t)inst * NG + grp) * st.lcap : vlist;
    const Float* cost = (const Float*)pb.cost;

    const Float kappa = (Float)a.kappa[inst];
    // affect() evaluates kappa / (one - kappa) for every row; same operands,
    // same quotient: computed once per sweep
    const Float kfac = kappa / ((Float)1 - kappa);
    const Float delta = (Float)a.delta;
    const Float theta = (Float)a.theta;
    const Float obj_amp = (Float)a.obj_amp;
    const bool push = a.is_push != 0;

    for (int i = tid; i < pb.m; i += T)
        pi_s[i] = pig[i];
    if (tid == 0)
        pi_s[pb.m] = 0; // read through the padding of the staged columns
    if (role != 2) {
        for (int i = tid; i < 2 * pb.nwords; i += T)
            x_s[i] = xg[i];
        for (int i = tid; i <= pb.rmax; i += T)
            stale_s[i] = staleg[i];
    }
    if (PIPE) {
        if (role == 1)
            for (int i = tid; i < pb.n; i += T)
                markl_s[i] = 0;
        else
            for (int i = tid; i < pb.n; i += T)
                mark_s[i] = 0;
    }

    MinStd rng;
    rng.s = st.status[inst].rng; // meaningful in thread 0 only
    const int nrows = use_m ? st.mcount[inst] : st.vcount[inst];
    double alg_bytes = 0.0; // thread t0 only

    if (MW || PIPE) {
        for (int i = tid; i < nrows; i += T)
            vrows[i] = vlist[i];
        __syncthreads();
    }
    if (tid == 0) {
        if (PIPE) {
            misc_s[MS_LCNT] = misc_s[MS_LCNT + 1] = 0;
            misc_s[MS_DEAD] = 0;
        }
        misc_s[MS_PICH] = 0;
        if (ord == 2 || ord == 7)
            rng.shuffle(vrows, (unsigned)nrows);
        misc_s[MS_RNG] = (int)rng.s;
    }
    __syncthreads();
    // (pi_sign_change keeps the shuffled R: written back behind the row loop,
    // when every partner has copied the list)
    rng.s = (unsigned)misc_s[MS_RNG];
    // Pipelined sweep over CONSECUTIVE rows (every full sweep in the orders none /
    // reversing): the rows before a row are k - 1, k - 2, .. (k + 1, .. when reversing), so
    // "which of the D previous rows hold this column" is the static gap to the column's
    // neighbouring entry that bxh_create put into the descriptors -- no column marks, no
    // look-up of the previous row's entry.  smode 1: ascending, 2: descending, 0: any other
    // list (the marks).  Chain and producers decide alike: they walk the same list.
    int smode = 0;
    if (PIPE && pb.static_late && (ord == 0 || ord == 1)) {
        int bad = 0;
        for (int i = tid; i + 1 < nrows; i += T)
            bad |= (vrows[i + 1] - vrows[i] != 1);
        if (!__syncthreads_or(bad))
            smode = ord == 0 ? 1 : 2;
    }

    const int l8 = tid & 7;
    // The duties of a single thread (row constants, pi_k and d of a row, the
    // serial part of a tied row, the status block): in the pipelined chain on
    // the non-staging waves -- the staging waves are the critical path.
    const int t0 = (PIPE && grp == 0) ? SW * 64 : 0;
    const int t1 = (PIPE && grp == 0) ? (SW * 64 + 64) % T : 64 % T;
    Prof prof;
#ifdef BXH_PROFILE
    unsigned long long skew[4] = { 0, 0, 0, 0 }, own[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < 16; ++i)
        prof.acc[i] = 0;
    prof.t = wall_clock64();
#endif

    // Rows ahead.  k -> row_ptr / sub_ptr -> descriptor records are dependent
    // global loads: the row id travels four rows ahead, the extents three, the
    // per-entry descriptor (registers) two and its LDS copy one, so that none
    // of them is waited for at the head of a row.
    auto row_at = [&](int it_) -> int {
        return (ord == 1) ? vrows[nrows - 1 - it_] : vrows[it_];
    };
    const int* sub_ptr = MW ? pb.sub_ptr + (size_t)grp * (pb.m + 1) : nullptr;
    struct RowX
    {
        int k, rs, re, os, oe;
    };
    auto row_extent = [&](int kid) -> RowX {
        RowX x;
        x.k = kid;
        x.rs = pb.row_ptr[kid];
        x.re = pb.row_ptr[kid + 1];
        x.os = MW ? sub_ptr[kid] : 0;
        x.oe = MW ? sub_ptr[kid + 1] : 0;
        return x;
    };
    auto own_len = [&](const RowX& x) -> int { return MW ? x.oe - x.os : x.re - x.rs; };
    RowX x0 = { 0, 0, 0, 0, 0 }, x1 = x0, x2 = x0;
    if (nrows > 0)
        x0 = row_extent(row_at(0));
    if (nrows > 1)
        x1 = row_extent(row_at(1));
    if (nrows > 2)
        x2 = row_extent(row_at(2));
    RowX x3 = x2;
    int k3 = nrows > 3 ? row_at(3) : 0, k4 = 0;
    // The extents of the row three ahead and the id of the row four ahead are asked for in
    // the middle of a row, for the rows' shift at its end.  Their addresses are uniform; as
    // vector loads the compiler moved each value to a scalar register right behind its load --
    // a `s_waitcnt vmcnt(0)` per load in the middle of the row: two global round trips one
    // behind the other on every wave, and on the chain's staging waves a wait for the late
    // loads they had just issued (2 400 of a row's 18 000 cycles, found with the LDS-stamped
    // timeline).  row_ptr / sub_ptr never change, and neither does the row list while the
    // rows are walked in its own order (orders none / reversing: the private copy is the
    // list): read through the constant address space they are scalar loads, which no vector
    // memory counter waits for.
    typedef const __attribute__((address_space(4))) int* cint_p;
    const cint_p row_ptr_c = (cint_p)pb.row_ptr;
    const cint_p sub_ptr_c = (cint_p)sub_ptr;
    const cint_p vlist_c = (cint_p)vlist;
    auto prefetch_x3k4 = [&](int it_) {
        if (it_ + 3 < nrows) {
            x3.k = k3;
            x3.rs = row_ptr_c[k3];
            x3.re = row_ptr_c[k3 + 1];
            x3.os = MW ? sub_ptr_c[k3] : 0;
            x3.oe = MW ? sub_ptr_c[k3 + 1] : 0;
            if (it_ + 4 < nrows)
                k4 = ord == 0 ? vlist_c[it_ + 4] : (ord == 1 ? vlist_c[nrows - 1 - (it_ + 4)] : row_at(it_ + 4));
        }
    };
    // descriptor of a row's own entries (column start / padded length | chunk
    // pieces, CSC slot, column, cost, position in the whole row) in registers
    int2 d_cs[2];
    int d_pe[2], d_col[2], d_oidx[2];
    Float d_cost[2];
    auto fetch_desc = [&](const RowX& x) { // own entries l < ro_
        const int ro_ = own_len(x);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int l = tid + s2 * T;
            if (l < ro_) {
                if (MW) { // independent loads: no pointer chase per entry
                    const int4 dd = pb.sub_desc[x.os + l];
                    d_cs[s2] = make_int2(dd.x, dd.y);
                    d_pe[s2] = dd.z;
                    d_col[s2] = dd.w;
                    d_oidx[s2] = pb.sub_e[x.os + l] - x.rs;
                    d_cost[s2] = ((const Float*)pb.desc_cost)[x.os + l];
                } else {
                    const int e = x.rs + l;
                    d_cs[s2] = pb.row_cs[e];
                    d_pe[s2] = pb.row_pos[e];
                    d_col[s2] = pb.row_col[e];
                    d_oidx[s2] = l;
                    d_cost[s2] = ((const Float*)pb.desc_cost)[e];
                }
            }
        }
    };
    auto store_desc = [&](const RowX& x, int buf) { // registers -> LDS buffer
        const int ro_ = own_len(x);
        int2* cs_w = cs_b + buf * RO;
        int* pe_w = pe_b + buf * RO;
        int* col_w = col_b + buf * RO;
        Float* cost_w = cost_b + buf * RO;
        int* oidx_w = oidx_b + buf * RO;
        if (ro_ <= 2 * T) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int l = tid + s2 * T;
                if (l < ro_) {
                    cs_w[l] = d_cs[s2];
                    pe_w[l] = d_pe[s2];
                    col_w[l] = d_col[s2];
                    if (PIPE && role == 1)
                        chain_cost_store(buf, l, d_cost[s2]);
                    else
                        cost_w[l] = d_cost[s2];
                    if (MW)
                        oidx_w[l] = d_oidx[s2];
                }
            }
        } else {
            for (int l = tid; l < ro_; l += T) {
                const int e = MW ? pb.sub_e[x.os + l] : x.rs + l;
                cs_w[l] = pb.row_cs[e];
                pe_w[l] = pb.row_pos[e];
                col_w[l] = pb.row_col[e];
                if (PIPE && role == 1)
                    chain_cost_store(buf, l, cost[pb.row_col[e]]);
                else
                    cost_w[l] = cost[pb.row_col[e]];
                if (MW)
                    oidx_w[l] = e - x.rs;
            }
        }
        if (MW) { // own entry of a row position, filled in after the next barrier
            short* loc_w = loc_b + buf * pb.rmax;
            for (int idx = tid; idx < x.re - x.rs; idx += T)
                loc_w[idx] = -1;
        }
    };
    auto fill_loc = [&](const RowX& x, int buf) {
        if (MW) {
            const int ro_ = own_len(x);
            short* loc_w = loc_b + buf * pb.rmax;
            const int* oidx_w = oidx_b + buf * RO;
            for (int l = tid; l < ro_; l += T)
                loc_w[oidx_w[l]] = (short)l;
        }
    };
    // row constants of select_variables, fetched one row ahead by one thread
    int c_bmin = 0, c_bmax = 0, c_bneg = 0;
    double c_rbytes = 0.0;
    auto fetch_consts = [&](int k_) {
        if (tid == t1) {
            c_bmin = pb.bmin[k_];
            c_bmax = pb.bmax[k_];
            c_bneg = pb.bneg[k_];
            c_rbytes = pb.row_bytes[k_];
        }
    };
    if (nrows > 0) {
        fetch_desc(x0);
        store_desc(x0, 0);
        fetch_consts(x0.k);
        if (nrows > 1)
            fetch_desc(x1);
        __syncthreads();
        fill_loc(x0, 0);
    }

    // ---- reduced-cost pipeline: loads of a wave's chunk of 16 columns --------
    // Columns are stored padded to 16 entries (128 bytes of P), so a column
    // is a whole number of 16-entry pieces: an 8-lane group loads one piece
    // per instruction (16 B of P + 4 B of row indices per lane = exactly one
    // cache line of P).  Every column of a chunk loads the chunk's largest
    // piece count (wave-uniform, from the descriptor); pieces past a column's
    // own end re-read its last piece (a cache hit): no branch per lane and no
    // use of the values at issue, so all loads stay in flight together.
    constexpr int CW = 8 * LR; // columns of a wave's chunk
    const int wave = tid >> 6, lane = tid & 63, nwaves = T >> 6;
    const int g8 = lane >> 3;
    Float* const wst_p = stg_p + wave * (CW * SP);
    unsigned short* const wst_i = stg_i + wave * (CW * SI);
    // the spare set of slots behind those of the last wave (overflow chunk)
    Float* const xst_p = stg_p + nwaves * (CW * SP);
    unsigned short* const xst_i = stg_i + nwaves * (CW * SI);
    Float2 regP[LR][NP];
    unsigned regI[LR][NP]; // two 16-bit row indices per lane and piece
    const char* const Pb = reinterpret_cast<const char*>(P);
    const char* const Ib = reinterpret_cast<const char*>(crow);
    auto chunk_pieces = [&](const int2* cs_r, int chunk) -> int {
        return __builtin_amdgcn_readfirstlane((cs_r[chunk * CW].y >> 16) & 0xff);
    };
    auto issue = [&](const int2* cs_r, int ro_, int chunk, int npm) {
#pragma unroll
        for (int h = 0; h < LR; ++h) {
            const int t = chunk * CW + h * 8 + g8;
            if (t < ro_) {
                const int2 cs = cs_r[t];
                const unsigned last = (unsigned)((cs.y & 0xffff) >> 4) - 1u;
                const unsigned q0 = (unsigned)cs.x + 2u * (unsigned)l8;
#pragma unroll
                for (int b = 0; b < NP; ++b) {
                    if (b < npm) { // wave-uniform
                        // 32-bit offsets from a uniform base
                        const unsigned q = q0 + (min((unsigned)b, last) << 4);
                        regP[h][b] =
                          *reinterpret_cast<const Float2*>(Pb + q * (unsigned)sizeof(Float));
                        regI[h][b] = *reinterpret_cast<const unsigned*>(Ib + q * 2u);
                    }
                }
            }
        }
    };

    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(st.xbuf + (size_t)inst * (PIPE ? PIPE_RING : 2) * pb.rmax * 2), 0,
      MW ? 2 * pb.rmax * 16 : (PIPE ? PIPE_RING * pb.rmax * 16 : 0), 0x00020000);

    // ---- pipelined sweep ---------------------------------------------------
    // messages of the row chain: position q (1-based) -> {pi of its row, tag}
    const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(st.msg + (PIPE ? (size_t)inst * pb.m * 2 : (size_t)0)), 0, PIPE ? pb.m * 16 : 0, 0x00020000);
    const int D = pb.lookahead;
    // Marks: per column the code of the last row position that held it (0: none
    // yet; a byte in the producers, the high six bits of a 16-bit word in the
    // chain, whose low ten bits are the column's entry in that row).  A column of
    // the row at position q is late when one
    // of the D previous positions marked it.  Codes repeat every 63
    // positions: a stale mark can only make an early column late, which both
    // sides decide alike and which is always safe (the chain then sums a
    // column whose rows are all finished).
    auto code_of = [](int q_) -> int { return q_ % 63 + 1; };
    auto is_late = [&](int mk, int q_) -> bool {
        bool late = false;
        for (int d_ = 1; d_ <= D; ++d_)
            late = late || (q_ - d_ >= 1 && mk == code_of(q_ - d_));
        return late;
    };
    const int LNPW = (pb.rmax + 15) / 16;
    auto tag_of = [&](int q_) -> unsigned { return ((a.epoch & 0x1ffffu) << 15) | (unsigned)q_; };
    // a granule pair {low half, tag, high half, tag}: polled until both tags
    // are the expected one (bounded; one time-out stops all polling)
    auto poll_pair = [&](const __amdgpu_buffer_rsrc_t& rs_, unsigned off, unsigned tag, v4u g) -> Float {
        long spins = 0;
        while (!(g.y == tag && g.w == tag)) {
            if (__hip_atomic_load(&misc_s[MS_DEAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000L) { // a partner is missing: give up loudly
                __hip_atomic_store(err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&misc_s[MS_DEAD], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                break;
            }
            g = __builtin_amdgcn_raw_buffer_load_b128(rs_, off, 0, 16);
        }
        const unsigned long long bits = (unsigned long long)g.x | ((unsigned long long)g.z << 32);
        return (Float)__longlong_as_double((long long)bits);
    };
    auto pair_of = [&](Float v, unsigned tag) -> v4u {
        const unsigned long long bits = (unsigned long long)__double_as_longlong((double)v);
        const v4u g = { (unsigned)bits, tag, (unsigned)(bits >> 32), tag };
        return g;
    };
    // chunk position t -> row entry through a map (-1 / past cnt: no column)
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)P, 0, PIPE ? (int)((unsigned)pb.nnzp * (unsigned)sizeof(Float)) : 0, 0x00020000);
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    // DEV: device-scope (sc1) loads -- what the producers read was stored by
    // another CU.  A compile-time choice: a run-time one makes the compiler
    // wait for every load before it issues the next.
    auto load_p2 = [&](unsigned byte_off, auto dev) -> Float2 { // 2 entries of P
        constexpr int AUX = decltype(dev)::value ? 16 : 0;
        Float2 out;
        if (sizeof(Float) == 8) {
            const v4u w = __builtin_amdgcn_raw_buffer_load_b128(prsrc, byte_off, 0, AUX);
            __builtin_memcpy(&out, &w, sizeof out);
        } else {
            const v2u w = __builtin_amdgcn_raw_buffer_load_b64(prsrc, byte_off, 0, AUX);
            __builtin_memcpy(&out, &w, sizeof out);
        }
        return out;
    };
    typedef PipeScope<true> DevScope;
    typedef PipeScope<false> CuScope;
    auto store_p = [&](unsigned slot, Float v) { // device-scope (write-through) store of one entry
        if (sizeof(Float) == 8) {
            v2u w;
            __builtin_memcpy(&w, &v, 8);
            __builtin_amdgcn_raw_buffer_store_b64(w, prsrc, slot * 8u, 0, 16);
        } else {
            unsigned w;
            __builtin_memcpy(&w, &v, 4);
            __builtin_amdgcn_raw_buffer_store_b32(w, prsrc, slot * 4u, 0, 16);
        }
    };
    // (direct: map[t] is t or -1 -- a producer's map -- so the descriptor is
    // read next to the map entry, not behind it: one LDS round trip, which is
    // slow while the other waves are summing)
    // No branch per lane: a lane without a column loads from past the end of
    // the buffers (nothing is fetched, zeros come back) and stages those zeros
    // in a slot nobody reads.  With the loads and the LDS writes under a
    // per-lane branch the compiler assumes loads whose staging was "skipped"
    // still pending and waits for every load before it issues the next one.
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)crow, 0, PIPE ? (int)((unsigned)pb.nnzp * 2u) : 0, 0x00020000);
    auto issue_m = [&](const int2* cs_r, const short* map, int cnt_, int chunk_, int npm_,
                       auto dev, bool direct = false) {
        // all addresses first, then all loads: where the compiler reuses a
        // register of the previous chunk for an address it waits for that
        // chunk's (long finished) load -- once, in front of the loads, not
        // between them
        unsigned qa[LR][NP], qi[LR][NP]; // byte offsets into P / into the row indices
#pragma unroll
        for (int h = 0; h < LR; ++h) {
            const int t = chunk_ * CW + h * 8 + g8;
            const int tt = min(t, cnt_ - 1);
            const int l = t < cnt_ ? (int)map[tt] : -1;
            const int2 csd = cs_r[tt];
            const int2 cs = direct ? csd : cs_r[max(l, 0)];
            const unsigned last = (unsigned)((cs.y & 0xffff) >> 4) - 1u;
            const unsigned q0 = (unsigned)cs.x + 2u * (unsigned)l8;
#pragma unroll
            for (int b = 0; b < NP; ++b) {
                const unsigned q = l >= 0 ? q0 + (min((unsigned)b, last) << 4) : 0x0ffffff0u;
                qa[h][b] = q * (unsigned)sizeof(Float);
                qi[h][b] = q * 2u;
            }
        }
        // (opaque to the compiler, or it computes every address again next to its load)
#pragma unroll
        for (int h = 0; h < LR; ++h)
#pragma unroll
            for (int b = 0; b < NP; ++b)
                asm volatile("" : "+v"(qa[h][b]), "+v"(qi[h][b]));
#pragma unroll
        for (int h = 0; h < LR; ++h) {
#pragma unroll
            for (int b = 0; b < NP; ++b) {
                if (b < npm_) { // wave-uniform
                    regP[h][b] = load_p2(qa[h][b], dev);
                    regI[h][b] = __builtin_amdgcn_raw_buffer_load_b32(irsrc, qi[h][b], 0, 0);
                }
            }
        }
    };
    auto stage_m = [&](const short* map, int cnt_, int chunk_, int npm_) {
#pragma unroll
        for (int h = 0; h < LR; ++h) {
            Float* dp = wst_p + (h * 8 + g8) * SP + 2 * l8;
            unsigned short* di = wst_i + (h * 8 + g8) * SI + 2 * l8;
#pragma unroll
            for (int b = 0; b < NP; ++b) {
                if (b < npm_) {
                    *reinterpret_cast<Float2*>(dp + b * 16) = regP[h][b];
                    *reinterpret_cast<unsigned*>(di + b * 16) = regI[h][b];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // Two lanes per staged column and kind: lanes 0-31 add the P entries, lanes
    // 32-63 the pi entries, lane pair (2u, 2u + 1) of each half column u's --
    // strictly in CSC order: of every 16-entry piece the even lane holds
    // entries 0-7 and the odd lane entries 8-15 (all read from LDS at once: half
    // as many LDS instructions per wave as with one lane per column and kind);
    // the running sum then goes even lane -> odd lane -> even lane of the next
    // piece (a DPP move inside the pair), so that every addition happens in the
    // order and with the operands of the scalar loop.  Returns the lane's sum
    // (both lanes of a pair hold it) and leaves the decayed own entry in dec_out
    // (even P lanes).  (row chain: a column loaded before the previous row's affect
    // first takes that row's new entry, patch[t] = the entry's index in the
    // previous row)
    auto sum_m = [&](const int2* cs_r, const int* pe_r, const short* map, int cnt_, int chunk_,
                     int& l_out, Float& dec_out, const short2* patch = nullptr,
                     bool direct = false) -> Float {
        const int u = (lane & 31) >> 1;
        const int sub = lane & 1;
        const bool is_pi = lane >= 32;
        const int t = chunk_ * CW + u;
        const int tt = min(t, cnt_ - 1);
        const int l = t < cnt_ ? (int)map[tt] : -1;
        const int2 csd = cs_r[tt];
        const int ped = pe_r[tt];
        l_out = l;
        Float acc = 0;
        if (l >= 0) {
            const int2 cs = direct ? csd : cs_r[l];
            const int cpad = cs.y & 0xffff;
            const Float2* sp2 = reinterpret_cast<const Float2*>(wst_p + u * SP) + 4 * sub;
            const uint2* si2 = reinterpret_cast<const uint2*>(wst_i + u * SI) + 2 * sub;
            // the row indices travel one piece ahead of the entries: the pi lanes
            // then wait for one LDS round trip per piece, not two (the first piece's
            // are asked for in front of the decay: nothing changes them)
            unsigned sgn[4]; // 8 packed 16-bit indices of the next piece (this lane's half)
            auto read_idx = [&](int i0_) {
                if (is_pi || PM1) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const uint2 w = si2[(i0_ >> 2) + b];
                        sgn[2 * b] = w.x;
                        sgn[2 * b + 1] = w.y;
                    }
                }
            };
            read_idx(0);
            if (!is_pi && sub == 0) { // P[e] *= theta for this row's entry
                if (patch) {
                    const short2 pp = patch[t];
                    if (pp.x >= 0)
                        wst_p[u * SP + pp.y] = newp_s[pp.x];
                }
                Float* pk = wst_p + u * SP + ((direct ? ped : pe_r[l]) - cs.x);
                const Float dec = *pk * theta; // (stored to HBM by affect)
                *pk = dec;
                dec_out = dec;
            }
            // (LDS operations of one wave execute in order: the reads below see the
            // patched and decayed entry; the compiler must not move them up)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int i0 = 0; i0 < cpad; i0 += 16) {
                Float v[8];
                unsigned sg[4];
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    sg[b] = sgn[b];
                if (is_pi) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        v[2 * b] = pi_s[sg[b] & 0x7fff];
                        v[2 * b + 1] = pi_s[(sg[b] >> 16) & 0x7fff];
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const Float2 w = sp2[(i0 >> 1) + b];
                        v[2 * b] = w.x;
                        v[2 * b + 1] = w.y;
                    }
                }
                if (i0 + 16 < cpad)
                    read_idx(i0 + 16);
                if (PM1) {
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const unsigned neg = (sg[b >> 1] >> ((b & 1) ? 31 : 15)) & 1u;
                        v[b] = neg ? -v[b] : v[b];
                    }
                }
                // entries 0-7 (the even lane's), then 8-15 (the odd lane's)
                Float a = acc;
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    a += v[b];
                acc = pair_bcast<0xA0>(a); // both lanes: the even lane's sum
                a = acc;
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    a += v[b];
                acc = pair_bcast<0xF5>(a); // both lanes: the odd lane's sum
            }
        }
        return acc;
    };

    // The row chain's form of sum_m: the late columns' sums are the head of the critical
    // path of a row (nothing else of the row can start before them), and a column sum is
    // one chain of c_j dependent additions however the lanes share it.  So (1) what the
    // prologue needs from LDS comes in two batches (entry + patch record, then extent /
    // own slot / the previous row's new value) instead of six dependent round trips, (2)
    // both halves of the wave run ONE instruction stream (the P lanes read their eight
    // entries of a piece with the same eight 8-byte reads the pi lanes gather with), and
    // (3) the values of piece i + 1 are read while piece i is added up (two register sets
    // taking turns; the chain has the registers for it: its load registers are idle while
    // it sums, a second late chunk of the same wave -- 1.5 % of the rows -- is only asked
    // for behind the sums).  Same order of additions, same operands.
    auto sum_late = [&](const int2* cs_r, const int* pe_r, const short* map, int cnt_, int chunk_,
                        int npm_, int& l_out, Float& dec_out, const short2* patch) -> Float {
        const int u = (lane & 31) >> 1;
        const int sub = lane & 1;
        const bool is_pi = lane >= 32;
        const int t = chunk_ * CW + u;
        const int tt = min(t, cnt_ - 1);
        const int lraw = (int)map[tt];
        const short2 pp = patch[tt];
        const int l = t < cnt_ ? lraw : -1;
        l_out = l;
        const int lc = max(l, 0);
        const int2 cs = cs_r[lc];
        const int pe = pe_r[lc];
        const Float npv = newp_s[max((int)pp.x, 0)];
        const int cpad = l >= 0 ? (cs.y & 0xffff) : 0; // (no column: every piece reads zeros)
        Float* const colp = wst_p + u * SP;
        const uint2* si2 = reinterpret_cast<const uint2*>(wst_i + u * SI) + 2 * sub;
        if (l >= 0 && !is_pi && sub == 0) {
            if (pp.x >= 0) // what the previous row's affect stored for its entry of this column
                colp[pp.y] = npv;
            Float* pk = colp + (pe - cs.x);
            const Float dec = *pk * theta; // P[e] *= theta (stored to HBM by affect)
            *pk = dec;
            dec_out = dec;
        }
        // (LDS operations of one wave execute in order: the reads below see the patched and
        // decayed entry; the compiler must not move them up)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // Element offsets from the start of LDS: P lanes walk the staged column, pi lanes
        // gather pi by the staged row index; a piece past the lane's own column reads
        // pi_s[m] == +0.0 eight times, and adding +0.0 leaves a sum that began as +0.0 as it
        // is.  With the piece count of the chunk (wave-uniform) instead of the column's own,
        // the pieces are straight-line code -- one copy per count: behind a branch that
        // merges, the compiler's wait counts drain the reads issued ahead.
        const Float* const fbase = reinterpret_cast<const Float*>(smem);
        const int p_off = (int)(colp - fbase) + 8 * sub;
        const int pi_off = (int)(pi_s - fbase);
        const int z_off = pi_off + pb.m;
        Float acc = 0;
        auto run = [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            Float v[2][8];
            unsigned sg[3][4]; // row indices: of the piece being added, the one being read, the next
            auto read_idx = [&](int k_, unsigned (&o)[4]) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const uint2 w = si2[4 * k_ + b];
                    o[2 * b] = w.x;
                    o[2 * b + 1] = w.y;
                }
            };
            auto load_piece = [&](int k_, Float (&o)[8], const unsigned (&ix)[4]) {
                const bool in = 16 * k_ < cpad;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int idx = (int)((b & 1) ? (ix[b >> 1] >> 16) & 0x7fffu : ix[b >> 1] & 0x7fffu);
                    const int off = is_pi ? pi_off + idx : p_off + 16 * k_ + b;
                    o[b] = fbase[in ? off : z_off];
                }
            };
            auto add_piece = [&](Float (&w)[8], const unsigned (&ix)[4]) {
                if (PM1) {
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const unsigned neg = (ix[b >> 1] >> ((b & 1) ? 31 : 15)) & 1u;
                        w[b] = neg ? -w[b] : w[b];
                    }
                }
                Float a = acc; // entries 0-7 (the even lane's), then 8-15 (the odd lane's)
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    a += w[b];
                acc = pair_bcast<0xA0>(a);
                a = acc;
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    a += w[b];
                acc = pair_bcast<0xF5>(a);
            };
            read_idx(0, sg[0]);
            if (K > 1)
                read_idx(1, sg[1]);
            load_piece(0, v[0], sg[0]);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                // indices two pieces ahead, values one piece ahead, then this piece's additions
                if (k + 2 < K)
                    read_idx(k + 2, sg[(k + 2) % 3]);
                if (k + 1 < K)
                    load_piece(k + 1, v[(k + 1) & 1], sg[(k + 1) % 3]);
                add_piece(v[k & 1], sg[k % 3]);
            }
        };
        static_assert(NP <= 8, "one straight-line copy per piece count");
        switch (__builtin_amdgcn_readfirstlane(npm_)) { // wave-uniform (said so: a scalar branch per case, no exec masks)
        case 1:
            run(std::integral_constant<int, 1>{});
            break;
        case 2:
            run(std::integral_constant<int, 2>{});
            break;
        case 3:
            run(std::integral_constant<int, (NP >= 3 ? 3 : 1)>{});
            break;
        case 4:
            run(std::integral_constant<int, (NP >= 4 ? 4 : 1)>{});
            break;
        case 5:
            run(std::integral_constant<int, (NP >= 5 ? 5 : 1)>{});
            break;
        case 6:
            run(std::integral_constant<int, (NP >= 6 ? 6 : 1)>{});
            break;
        case 7:
            run(std::integral_constant<int, (NP >= 7 ? 7 : 1)>{});
            break;
        case 8:
            run(std::integral_constant<int, (NP >= 8 ? 8 : 1)>{});
            break;
        default:
            break;
        }
        return acc;
    };

    if (PIPE && grp > 0) {
        // ---- a producer: the early columns of every row, `D` rows ahead ------
        // Software pipeline over the rows: the loads of a wave's first chunk of
        // row it + 1 are issued before the sums of its last chunk of row it, the
        // chain's message a row before it is needed.
        const int np_ = NG - 1, me = grp - 1;
        // a column is early when none of the D previous rows holds it (thread l
        // reads the descriptor entry it wrote itself)
        auto classify_p = [&](int r_, int buf, int q_) {
            const int* col_w = col_b + buf * RO;
            short* em = map_b + buf * pb.rmax;
            if (smode) { // consecutive rows: late = the column's neighbour is at most D rows away
                // (not in the first row of the walk: nothing was loaded ahead for it.  A neighbour in
                // a row in front of the walk's first makes a column of the second row late, which is
                // safe: the chain sums it, and patches only from a row it has walked.)
                const int2* cs_w = cs_b + buf * RO;
                const int reach = q_ >= 2 ? D : 0;
                for (int l = tid; l < r_; l += T) {
                    const int g = (cs_w[l].y >> (smode == 1 ? 24 : 28)) & 15;
                    em[l] = (g >= 1 && g <= reach) ? (short)-1 : (short)l;
                }
                return;
            }
            for (int l = tid; l < r_; l += T) {
                const int j = col_w[l];
                const int mk = mark_s[j];
                em[l] = is_late(mk, q_) ? (short)-1 : (short)l;
                mark_s[j] = (unsigned char)code_of(q_);
            }
        };
        auto first_chunk = [&](const RowX& x, int buf) { // loads of this wave's first chunk of a row
            const int r_ = x.re - x.rs;
            const int c = me + np_ * wave;
            if (wave < SW && c * CW < r_)
                issue_m(cs_b + buf * RO, map_b + buf * pb.rmax, r_, c, chunk_pieces(cs_b + buf * RO, c), DevScope{}, true);
        };
        v4u mpre = { 0u, 0u, 0u, 0u }; // the message asked for a row ahead (every wave for itself)
        // Position q + 1 reads P once the rows up to position q - D are finished.  Only the
        // loads of the NEXT row's first chunk ask for that, and each wave issues them at
        // its own time, behind the staging of its last chunk of this row -- so every wave
        // checks the chain's message there, for itself (asked for a row ahead: it is there
        // unless the chain is late), writes the finished row's pi (all waves the same
        // value; no column still being summed holds that row) and goes on.  Checked by one
        // thread at the top of the row, with a workgroup barrier behind it, the producers
        // stood still until the chain's message had come, a third of a row (found with the
        // timeline build), and the loop chain -> message -> producers -> values -> chain
        // went round once per row instead of once per D rows.
        auto await_rows = [&](int need_, int it_) {
            if (need_ >= 1)
                pi_s[row_at(need_ - 1)] =
                  poll_pair(mrsrc, (unsigned)(need_ - 1) * 16u, tag_of(need_), mpre);
            if (need_ >= 0 && it_ + 2 < nrows)
                mpre = __builtin_amdgcn_raw_buffer_load_b128(mrsrc, (unsigned)need_ * 16u, 0, 16);
            asm volatile("" ::: "memory");
        };
#ifdef BXH_PROFILE
        unsigned long long pacc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, pt = wall_clock64();
#define BXH_PROD_STAMP(slot)                                                   \
    do {                                                                       \
        if (tid == 0) {                                                        \
            const unsigned long long now_ = wall_clock64();                   \
            pacc[slot] += now_ - pt;                                           \
            pt = now_;                                                         \
        }                                                                      \
    } while (0)
#else
#define BXH_PROD_STAMP(slot)                                                   \
    do {                                                                       \
    } while (0)
#endif
        if (nrows > 0) {
            classify_p(x0.re - x0.rs, 0, 1);
            __syncthreads();
            __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): see below
            first_chunk(x0, 0); // (position 1 waits for no row)
        }
        for (int it = 0; it < nrows; ++it, x0 = x1, x1 = x2, x2 = x3, k3 = k4) {
            const int cur = it & 1, q = it + 1;
            const int rs = x0.rs;
            const int r = x0.re - x0.rs;
            const int2* cs_s = cs_b + cur * RO;
            const int* pe_s = pe_b + cur * RO;
            const Float* cost_s = cost_b + cur * RO;
            const short* emap = map_b + cur * pb.rmax;
            BXH_TLP(0);
            __syncthreads(); // every wave is through with row it - 1 (the other descriptor buffer)
            BXH_PROD_STAMP(0);
            BXH_TLP(1);
            if (it + 1 < nrows) {
                store_desc(x1, cur ^ 1);
                classify_p(x1.re - x1.rs, cur ^ 1, q + 1);
                BXH_PROD_STAMP(1);
                BXH_TLP(2);
            }
            const int need = q - D; // (see await_rows)
            BXH_PROD_STAMP(2);
            BXH_TLP(3);
            __syncthreads(); // the next row's map and descriptor
            BXH_PROD_STAMP(3);
            BXH_TLP(4);
            const int nchunks = (r + CW - 1) / CW;
            const unsigned roff = (unsigned)(q % PIPE_RING) * (unsigned)pb.rmax * 16u;
            const unsigned tagq = tag_of(q);
            if (wave < SW) {
                int i = wave;
                int chunk = me + np_ * i;
                if (chunk >= nchunks && it + 1 < nrows) {
                    // No chunk of this row: whatever this wave loaded last was
                    // never staged, and is long here.  Saying so (vmcnt(0)) on
                    // this path keeps the compiler from waiting, on every
                    // path, for "pending" loads between the loads it issues
                    // next, whose registers it reuses for the addresses.
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                    await_rows(need, it);
                    first_chunk(x1, cur ^ 1);
                }
                int npm = chunk < nchunks ? chunk_pieces(cs_s, chunk) : 0;
                int tlj = 0; // (timeline: the wave's first / second chunk of the row)
                (void)tlj;
                while (chunk < nchunks) {
                    stage_m(emap, r, chunk, npm);
                    BXH_PROD_STAMP(6);
                    BXH_TLP(5 + 4 * tlj);
                    i += SW;
                    const int cn = me + np_ * i;
                    int npn = 0;
                    if (cn < nchunks) {
                        npn = chunk_pieces(cs_s, cn);
                        issue_m(cs_s, emap, r, cn, npn, DevScope{}, true);
                    } else if (it + 1 < nrows) {
                        await_rows(need, it);
                        first_chunk(x1, cur ^ 1);
                    }
                    BXH_PROD_STAMP(7);
                    BXH_TLP(6 + 4 * tlj);
                    int l;
                    Float dec;
                    const Float cst = cost_s[min(chunk * CW + ((lane & 31) >> 1), r - 1)];
                    const Float acc = sum_m(cs_s, pe_s, emap, r, chunk, l, dec, nullptr, true);
                    BXH_TLP(7 + 4 * tlj);
                    const Float spi = __shfl_down(acc, 32); // the pi lanes of the same column
                    if (l >= 0 && lane < 32 && (lane & 1) == 0) {
                        Float v = cst - spi - acc;
                        if (PM1 && pb.fac_csr[rs + l] < 0)
                            v = -v;
                        __builtin_amdgcn_raw_buffer_store_b128(pair_of(v, tagq), xrsrc,
                                                               roff + (unsigned)l * 16u, 0, 16);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    BXH_TLP(8 + 4 * tlj);
                    tlj = tlj < 1 ? tlj + 1 : 1;
                    chunk = cn;
                    npm = npn;
                }
            }
            BXH_PROD_STAMP(4);
            BXH_TLP(13);
            if (it + 2 < nrows) {
                fetch_desc(x2);
                prefetch_x3k4(it);
            }
            BXH_PROD_STAMP(5);
            BXH_TLP(14);
        }
#ifdef BXH_PROFILE
        if (tid == 0 && grp == 1)
            for (int i = 0; i < 8; ++i)
                st.prof[(size_t)inst * 32 + 24 + i] += pacc[i];
#endif
        // compute_violated_constraints of the new x: a share of the rows (the
        // chain publishes x, every workgroup flags its rows, the chain compacts)
        if (tid == 0)
            wait_arrivals(cnt_row, 1u, err_word);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        unsigned* xp = (unsigned*)(smem + lp.off_union); // (the staging slots are free now)
        for (int i = tid; i < 2 * pb.nwords; i += T)
            xp[i] = xg[i];
        __syncthreads();
        flag_violated(pb, xp, st.vflag + (size_t)inst * pb.m, (int)((long long)pb.m * grp / NG),
                      (int)((long long)pb.m * (grp + 1) / NG));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0)
            __hip_atomic_fetch_add(cnt_end, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // the row chain: which entries of a row are late, and the compact list of
    // them (its order is free: every column is summed on its own)
    // `base`, `stride`: the threads that take part and the entries they take
    // (whole waves).  `rs_` >= 0: the descriptor of the row comes straight from
    // global memory (the chain's non-staging waves write the next row's
    // descriptor and classify it while the staging waves sum the late columns).
    auto classify_c = [&](int r_, int buf, int q_, int pr_len, int base, int stride, int rs_) {
        int* col_w = col_b + buf * RO;
        int2* cs_w = cs_b + buf * RO;
        int* pe_w = pe_b + buf * RO;
        short* lm = map_b + buf * pb.rmax;
        short2* lpt = patch_b + buf * pb.rmax;
        const int* pe_prev = pe_b + (buf ^ 1) * RO; // the row in front of it: the one being updated
        const int* col_prev = col_b + (buf ^ 1) * RO;
        unsigned char* lf = flag_b + buf * pb.rmax;
        int* ln = lnp_b + buf * LNPW;
        const bool all_costs = push && PM1; // (a push sweep of the 101 solvers amplifies by the column's cost)
        const int trips = (r_ + stride - 1) / stride;
        for (int s2 = 0; s2 < trips; ++s2) { // (wave-uniform)
            const int l = base + s2 * stride;
            bool late = false;
            int pcs = 0, ml = -1, ppos = 0;
            if (l < r_) {
                int j, recw = 0, ownpos;
                int2 csw;
                if (rs_ >= 0) { // the record LDS-DMA left here a row ago
                    const int4 rec = land_s[l];
                    csw = make_int2(rec.x, rec.y & 0xffff);
                    j = rec.z;
                    recw = rec.w;
                    ownpos = (int)((unsigned)rec.y >> 16);
                    cs_w[l] = csw;
                    pe_w[l] = rec.x + ownpos;
                    col_w[l] = j;
                } else {
                    j = col_w[l];
                    csw = cs_w[l];
                    ownpos = pe_w[l] - csw.x;
                    recw = (csw.y >> 24) & 0xff; // (the two gaps; the first row of a walk has no row in front of it)
                    csw.y &= 0xffff;
                }
                if (smode) {
                    // consecutive rows: the gap to the column's neighbouring entry says it all
                    // (the neighbour in the very next row is this column's entry of the
                    // previous row of the walk, one slot up / down the column)
                    const unsigned w = (unsigned)recw;
                    const int g = (int)(smode == 1 ? w & 15u : (w >> 4) & 15u);
                    late = g >= 1 && g <= D && q_ >= 2; // (the first row of a walk has nothing loaded ahead)
                    if (g == 1 && q_ >= 2) {
                        ml = (int)((smode == 1 ? w >> 8 : w >> 18) & 1023u);
                        ppos = ownpos + (smode == 1 ? -1 : 1);
                    }
                    lf[l] = late ? 1 : 0;
                    pcs = (csw.y & 0xffff) >> 4;
                } else {
                const unsigned mk16 = markl_s[j];
                const int mk = (int)(mk16 >> 10);
                late = is_late(mk, q_);
                if (q_ >= 2 && mk == code_of(q_ - 1)) { // the previous row holds the column too
                    ml = (int)(mk16 & 1023u);
                    if (ml >= pr_len || col_prev[ml] != j) // (unless the mark is 63 rows old)
                        ml = -1;
                }
                markl_s[j] = (unsigned short)((code_of(q_) << 10) | l);
                lf[l] = late ? 1 : 0;
                pcs = (csw.y & 0xffff) >> 4;
                if (ml >= 0)
                    ppos = pe_prev[ml] - csw.x;
                }
            }
            if (rs_ >= 0) {
                // the cost of a late entry's column: asked for here, a row before the late
                // sums read it, and dropped into lc_lo / lc_hi[buf][l] by LDS-DMA (no
                // register is held, nothing waits; the wait for the loads in front of
                // barrier 1 covers it)
                const bool want = l < r_ && (late || all_costs);
                if (__ballot(want)) { // wave-uniform
                    const unsigned dst = lds_base + (unsigned)lp.off_cost +
                                         (unsigned)(buf * RO + (l - lane)) * 4u; // (l - lane: the wave's first entry)
                    if (want) {
                        const char* src = (const char*)pb.desc_cost + (size_t)(rs_ + l) * sizeof(Float);
                        lds_dma4(src, __builtin_amdgcn_readfirstlane(dst));
                        if (sizeof(Float) == 8)
                            lds_dma4(src + 4, __builtin_amdgcn_readfirstlane(dst + (unsigned)(2 * RO) * 4u));
                    }
                }
            }
            const unsigned long long bal = __ballot(late);
            if (bal) { // wave-uniform
                int pbase = 0;
                if (lane == 0)
                    pbase = atomicAdd(&misc_s[MS_LCNT + buf], __popcll(bal));
                pbase = __builtin_amdgcn_readfirstlane(pbase);
                if (late) {
                    const int pos = pbase + __popcll(bal & ((1ull << lane) - 1ull));
                    lm[pos] = (short)l;
                    lpt[pos] = make_short2((short)ml, (short)ppos);
                    atomicMax(&ln[pos >> 4], pcs);
                }
            }
        }
    };
    // the descriptor records of the row at `x`, 1 KB (64 records) per wave and instruction,
    // by the waves w0, w0 + 1, .. of `nw` (whole pieces: the array is padded)
    auto dma_records = [&](const RowX& x, int w0, int nw) {
        const int rr = x.re - x.rs;
        for (int ch = wave - w0; ch * 64 < rr; ch += nw)
            lds_dma16(pb.chain_rec + x.rs + ch * 64 + lane,
                      __builtin_amdgcn_readfirstlane(lds_base + (unsigned)lp.off_land + (unsigned)ch * 1024u));
    };
    if (PIPE) {
        for (int i = tid; i < 2 * LNPW; i += T)
            lnp_b[i] = 0;
        if (nrows > 0)
            classify_c(x0.re - x0.rs, 0, 1, 0, tid, T, -1);
        if (nrows > 1) { // row 1 is classified at the top of row 0
            dma_records(x1, 0, nwaves);
            __builtin_amdgcn_s_waitcnt(0x0f70);
        }
        __syncthreads();
    }

    int kprev = 0; // (pipelined sweep: the row whose message is published a row later)
    for (int it = 0; it < nrows; ++it, x0 = x1, x1 = x2, x2 = x3, k3 = k4) {
        const int cur = it & 1;
        const int k = x0.k;
        const int rs = x0.rs;
        const int r = x0.re - x0.rs;  // the whole row
        const int ro = own_len(x0);   // this owner's entries
        const int2* cs_s = cs_b + cur * RO;
        const int* pe_s = pe_b + cur * RO;
        const int* col_s = col_b + cur * RO;
        const Float* cost_s = cost_b + cur * RO;
        const int* oidx_s = oidx_b + cur * RO;
        const short* loc_s = loc_b + cur * pb.rmax;
        BXH_TL(0);

        // ---- pipelined sweep, row chain: only the late columns are summed
        // here; the producers' values and the row's own P entries are loaded
        // meanwhile -----------------------------------------------------------
        const unsigned char* lflag = flag_b + cur * pb.rmax;
        // whose entries a thread fetches (the producers' value, the own P entry)
        // and turns into val_s: the non-staging waves take the first `go`
        // entries of the row, two each at most, the staging waves -- busy with
        // the late columns -- the rest
        int gl[2] = { tid, tid + T };
        if (PIPE) {
            const int STG = SW * 64, OTH = T - STG;
            const int go = min(pb.gather_other, 2 * OTH);
            if (wave >= SW) {
                gl[0] = tid - STG;
                gl[1] = OTH + tid - STG < go ? OTH + tid - STG : 0x7fffffff;
            } else {
                gl[0] = go + tid;
                gl[1] = go + STG + tid;
            }
        }
        Float gP[2] = { 0, 0 };
        v4u gG[2] = { { 0u, 0u, 0u, 0u }, { 0u, 0u, 0u, 0u } };
        const unsigned roff = (unsigned)((it + 1) % PIPE_RING) * (unsigned)pb.rmax * 16u;
        if (PIPE) {
            const short* lmap = map_b + cur * pb.rmax;
            const int nl = misc_s[MS_LCNT + cur];
            const int nlc = (nl + CW - 1) / CW;
            const int* lnp = lnp_b + cur * LNPW;
            // the producers' values and the own P entries of this thread's row entries.
            // A staging wave asks for them BEHIND the staging of its late chunk: staging
            // waits for the chunk's loads (issued a row ago), and the wait counter of a
            // wave is in order -- asked for in front of it, these loads (a round trip to
            // the producers' stores) would be waited for there, before the sums, instead
            // of travelling under them (3 000 of a row's 21 000 cycles, found with the
            // timeline build).
            auto ask_values = [&]() {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int l = gl[s2];
                    if (l < r && !lflag[l]) {
                        gP[s2] = P[pe_s[l]];
                        gG[s2] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, roff + (unsigned)l * 16u, 0, 16);
                    }
                }
            };
            if (wave >= SW || wave >= nlc)
                ask_values();
            BXH_STAMP(13);
            BXH_TL(1);
            if (wave >= SW && it + 1 < nrows)
                classify_c(x1.re - x1.rs, cur ^ 1, it + 2, r, tid - SW * 64, T - SW * 64, x1.rs);
            BXH_TL(2);
            if (wave < SW) {
                // (the wave's first late chunk was loaded during the previous
                // row, behind its barrier 1; what that row's affect changed
                // since is patched in the staged copy)
                const short2* lpatch = patch_b + cur * pb.rmax;
                int chunk = wave, npm = 0;
                if (chunk < nlc)
                    npm = __builtin_amdgcn_readfirstlane(lnp[chunk]);
                while (chunk < nlc) {
                    stage_m(lmap, nl, chunk, npm);
                    BXH_TL(13);
#ifdef BXH_TIMELINE
                    if (inst == 0 && it >= 200 && it < 208 && lane == 0) // (pieces, late columns of the row)
                        st.prof[((it - 200) * 8 + wave) * 16 + 14] = (unsigned long long)(npm * 1000 + nl);
#endif
                    if (chunk == wave)
                        ask_values();
                    const int cn = chunk + SW;
                    int npn = 0;
                    int l;
                    Float dec = 0;
#ifdef BXH_CHAIN_SUM_OLD
                    if (cn < nlc) {
                        npn = __builtin_amdgcn_readfirstlane(lnp[cn]);
                        issue_m(cs_s, lmap, nl, cn, npn, CuScope{});
                    }
                    const Float acc = sum_m(cs_s, pe_s, lmap, nl, chunk, l, dec, lpatch);
#else
                    const Float acc = sum_late(cs_s, pe_s, lmap, nl, chunk, npm, l, dec, lpatch);
                    if (cn < nlc) {
                        npn = __builtin_amdgcn_readfirstlane(lnp[cn]);
                        issue_m(cs_s, lmap, nl, cn, npn, CuScope{});
                    }
#endif
                    BXH_TL(15);
                    const Float spi = __shfl_down(acc, 32); // the pi lanes of the same column
                    if (l >= 0 && lane < 32 && (lane & 1) == 0) {
                        const Float cj = chain_cost(cur, l); // (linear objective: the host checks)
                        Float v = cj - spi - acc;
                        if (PM1 && pb.fac_csr[rs + l] < 0)
                            v = -v;
                        if (push) // the 01 solvers index the cost with the LOCAL id (quirk)
                            v += obj_amp * (PM1 ? cj : cost[l]);
                        val_s[l] = v;
                        pown_s[l] = dec;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    chunk = cn;
                    npm = npn;
                }
            }
        }
        BXH_TL(3);
        // ---- this row's first loads, then the next row's descriptor ---------
        const int nchunks = (ro + CW - 1) / CW;
        int chunk = wave;
        int npm = 0;
        if (!PIPE && chunk < nchunks) {
            npm = chunk_pieces(cs_s, chunk);
            issue(cs_s, ro, chunk, npm);
        }
        BXH_STAMP(0);

        // ---- decrease_preference + compute_reduced_costs ------------------
        // Wave-private pipeline, no workgroup barrier inside.  The pieces of
        // a 16-column chunk are staged in the wave's LDS slots, those of the
        // wave's next chunk issued before the sums of this one; lanes 0-15
        // then add the P entries and lanes 16-31 the pi entries of one column
        // each, strictly in CSC order
        // (itm-solver-inequalities-01.cpp:173-188); padding entries are +0.0
        // / pi_s[m] == +0.0 and leave the sums bit-identical.
        // An owner's part of a row may exceed the 16 columns per wave (8
        // waves: 128) by a few columns.  That overflow chunk is loaded by wave
        // 0 behind its own chunk, staged in a spare set of slots, and summed
        // by the wave's lanes 32-63 in the same pass as its own chunk -- not
        // in a second pass of its own, which the whole workgroup (and the
        // partners of the instance) would wait for.
        if (!PIPE) {
            const bool overflow = pb.spare && nchunks > nwaves; // chunk `nwaves` exists
            auto stage = [&](Float* base_p, unsigned short* base_i, int chunk_, int npm_) {
#pragma unroll
                for (int h = 0; h < LR; ++h) {
                    const int t = chunk_ * CW + h * 8 + g8;
                    if (t < ro) {
                        // every loaded piece is stored; the adders only read
                        // the column's own (padded) length
                        Float* dp = base_p + (h * 8 + g8) * SP + 2 * l8;
                        unsigned short* di = base_i + (h * 8 + g8) * SI + 2 * l8;
#pragma unroll
                        for (int b = 0; b < NP; ++b) {
                            if (b < npm_) {
                                *reinterpret_cast<Float2*>(dp + b * 16) = regP[h][b];
                                *reinterpret_cast<unsigned*>(di + b * 16) = regI[h][b];
                            }
                        }
                    }
                }
                // LDS operations of one wave execute in order: only the
                // compiler has to be kept from reordering across this point
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            };
            for (; chunk < nchunks; chunk += nwaves) {
                if (overflow && chunk == nwaves)
                    continue; // summed together with chunk 0 (wave 0)
                stage(wst_p, wst_i, chunk, npm);
                BXH_STAMP(12);
                const bool dual = overflow && chunk == 0;
                if (dual) {
                    const int npx = chunk_pieces(cs_s, nwaves);
                    issue(cs_s, ro, nwaves, npx);
                    stage(xst_p, xst_i, nwaves, npx);
                    if (2 * nwaves < nchunks) {
                        npm = chunk_pieces(cs_s, 2 * nwaves);
                        issue(cs_s, ro, 2 * nwaves, npm);
                    }
                } else if (chunk + nwaves < nchunks) {
                    npm = chunk_pieces(cs_s, chunk + nwaves);
                    issue(cs_s, ro, chunk + nwaves, npm);
                }
                BXH_STAMP(13);

                if (!dual) {
                    // two lanes per column and kind (see sum_m): lanes 0-31 the P entries,
                    // lanes 32-63 the pi entries, the even lane of a pair entries 0-7 of
                    // every piece, the odd lane entries 8-15; same order of additions
                    const int u = (lane & 31) >> 1;
                    const int sub = lane & 1;
                    const bool is_pi = lane >= 32;
                    const int t = chunk * CW + u;
                    if (t < ro) {
                        const int2 cs = cs_s[t];
                        const int cpad = cs.y & 0xffff;
                        const Float2* sp2 = reinterpret_cast<const Float2*>(wst_p + u * SP) + 4 * sub;
                        const uint2* si2 = reinterpret_cast<const uint2*>(wst_i + u * SI) + 2 * sub;
                        unsigned sgn[4];
                        auto read_idx = [&](int i0_) {
                            if (is_pi || PM1) {
#pragma unroll
                                for (int b = 0; b < 2; ++b) {
                                    const uint2 w = si2[(i0_ >> 2) + b];
                                    sgn[2 * b] = w.x;
                                    sgn[2 * b + 1] = w.y;
                                }
                            }
                        };
                        read_idx(0);
                        if (!is_pi && sub == 0) { // P[e] *= theta for this row's entry
                            Float* pk = wst_p + u * SP + (pe_s[t] - cs.x);
                            const Float dec = *pk * theta; // (stored to HBM by affect)
                            *pk = dec;
                            pown_s[t] = dec;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        Float acc = 0;
                        for (int i0 = 0; i0 < cpad; i0 += 16) {
                            Float v[8];
                            unsigned sg[4];
#pragma unroll
                            for (int b = 0; b < 4; ++b)
                                sg[b] = sgn[b];
                            if (is_pi) {
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    v[2 * b] = pi_s[sg[b] & 0x7fff];
                                    v[2 * b + 1] = pi_s[(sg[b] >> 16) & 0x7fff];
                                }
                            } else {
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    const Float2 w = sp2[(i0 >> 1) + b];
                                    v[2 * b] = w.x;
                                    v[2 * b + 1] = w.y;
                                }
                            }
                            if (i0 + 16 < cpad)
                                read_idx(i0 + 16);
                            if (PM1) {
#pragma unroll
                                for (int b = 0; b < 8; ++b) {
                                    const unsigned neg = (sg[b >> 1] >> ((b & 1) ? 31 : 15)) & 1u;
                                    v[b] = neg ? -v[b] : v[b];
                                }
                            }
                            Float a = acc;
#pragma unroll
                            for (int b = 0; b < 8; ++b)
                                a += v[b];
                            acc = pair_bcast<0xA0>(a);
                            a = acc;
#pragma unroll
                            for (int b = 0; b < 8; ++b)
                                a += v[b];
                            acc = pair_bcast<0xF5>(a);
                        }
                        if (sub == 0) {
                            if (is_pi)
                                spi_s[t] = acc;
                            else
                                sp_s[t] = acc;
                        }
                    }
                } else {
                    const int half = lane >> 5; // 1: the overflow chunk
                    const int lane5 = lane & 31;
                    const int u = lane5 & (CW - 1);
                    const bool is_pi = lane5 >= CW;
                    const int t = (half ? nwaves : chunk) * CW + u;
                    if ((half == 0 || dual) && t < ro) {
                        Float* const my_p = half ? xst_p : wst_p;
                        unsigned short* const my_i = half ? xst_i : wst_i;
                        const int2 cs = cs_s[t];
                        const int cpad = cs.y & 0xffff;
                        if (!is_pi) { // P[e] *= theta for this row's entry
                            Float* pk = my_p + u * SP + (pe_s[t] - cs.x);
                            const Float dec = *pk * theta; // (stored to HBM by affect)
                            *pk = dec;
                            pown_s[t] = dec;
                        }
                        const Float2* sp2 =
                          reinterpret_cast<const Float2*>(my_p + u * SP);
                        const uint2* si2 =
                          reinterpret_cast<const uint2*>(my_i + u * SI);
                        Float acc = 0;
                        for (int i0 = 0; i0 < cpad; i0 += 16) {
                            Float v[16];
                            unsigned sg[8]; // 16 packed 16-bit indices
                            if (is_pi || PM1) {
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    const uint2 w = si2[(i0 >> 2) + b];
                                    sg[2 * b] = w.x;
                                    sg[2 * b + 1] = w.y;
                                }
                            }
                            if (is_pi) {
#pragma unroll
                                for (int b = 0; b < 8; ++b) {
                                    v[2 * b] = pi_s[sg[b] & 0x7fff];
                                    v[2 * b + 1] = pi_s[(sg[b] >> 16) & 0x7fff];
                                }
                            } else {
#pragma unroll
                                for (int b = 0; b < 8; ++b) {
                                    const Float2 w = sp2[(i0 >> 1) + b];
                                    v[2 * b] = w.x;
                                    v[2 * b + 1] = w.y;
                                }
                            }
#pragma unroll
                            for (int b = 0; b < 16; ++b) {
                                Float term = v[b];
                                if (PM1) {
                                    const unsigned neg =
                                      (sg[b >> 1] >> ((b & 1) ? 31 : 15)) & 1u;
                                    term = neg ? -term : term;
                                }
                                acc += term;
                            }
                        }
                        if (is_pi)
                            spi_s[t] = acc;
                        else
                            sp_s[t] = acc;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                BXH_STAMP(14);
            }
        }
        BXH_STAMP(15);
        if (!PIPE) // (the chain's late waves write their reduced costs themselves)
            BXH_BARRIER(0);
        BXH_STAMP(2);
        // Row exchange among the owners: every reduced cost travels as two
        // naturally aligned 8-byte granules {32 value bits, 32-bit tag}, both
        // written by ONE write-through 16-byte store and read by one
        // L1-bypassing 16-byte load until each granule's tag is the current
        // (launch, row) number -- each granule validates itself, so no
        // counter, no barrier and no ordering is needed.  Two slots alternate:
        // a workgroup can only be one row ahead of its partners (it needs
        // their values).
        const unsigned xoff = (unsigned)(it & 1) * (unsigned)pb.rmax * 16u;
        const unsigned tag32 = ((a.epoch & 0x1ffffu) << 15) | (unsigned)(it + 1);

        // ---- the next row's descriptor -> LDS and the row constants, BEFORE
        // this row's stores: their registers were loaded a row ago, so the
        // wait for them returns at once here, while behind the write-through
        // stores it would also wait for the stores' acknowledgements -----------
        if (!PIPE && it + 1 < nrows)
            store_desc(x1, cur ^ 1);
        if (tid == t1) { // row constants of select_variables, off thread 0
            misc_s[MS_BMIN] = c_bmin;
            misc_s[MS_BMAX] = c_bmax;
            misc_s[MS_BNEG] = c_bneg;
            *(double*)(misc_s + MS_RBYTES) = c_rbytes;
            // slots of the selection's LDS-atomic reductions (part_s[64..])
            part_s[64] = 0ull;  // largest key
            part_s[65] = 0ull;  // largest N key
            part_s[66] = ~0ull; // smallest S key
            part_s[67] = 0ull;  // counts: N in the high half, Z in the low half
            part_s[68] = 0ull;  // tie flag
            part_s[69] = ~0ull; // smallest key
        }
        BXH_STAMP(11);

        if (PIPE) {
            const unsigned tagq = tag_of(it + 1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int l = gl[s2];
                if (l < r) {
                    if (!lflag[l]) { // (a late entry: written by the wave that summed it)
                        Float v = poll_pair(xrsrc, roff + (unsigned)l * 16u, tagq, gG[s2]);
                        pown_s[l] = gP[s2] * theta;
                        if (push) // the 01 solvers index the cost with the LOCAL id (quirk)
                            v += obj_amp * (PM1 ? chain_cost(cur, l) : cost[l]);
                        val_s[l] = v;
                    }
                }
            }
        }
        for (int l = tid; !PIPE && l < ro; l += T) {
            const int idx = MW ? oidx_s[l] : l; // position in the whole row
            const int j = col_s[l];
            // (a quadratic objective runs with one owner: x_s is complete)
            const Float cj = cost_x<Float>(pb, x_s, j, cost_s[l]);
            Float v = cj - spi_s[l] - sp_s[l];
            if (PM1 && pb.fac_csr[rs + idx] < 0)
                v = -v;
            if (push) // the 01 solvers index the cost with the LOCAL id (quirk)
                v += obj_amp * (PM1 ? cj : cost_x<Float>(pb, x_s, idx, cost[idx]));
            if (MW) {
                const unsigned long long bits =
                  (unsigned long long)__double_as_longlong((double)v);
                const v4u g = { (unsigned)bits, tag32, (unsigned)(bits >> 32), tag32 };
                // both granules by one write-through (sc1) 16-byte store
                __builtin_amdgcn_raw_buffer_store_b128(g, xrsrc, xoff + (unsigned)idx * 16u, 0, 16);
            }
            val_s[idx] = v; // own values do not travel
        }
        // the descriptor after the next -> registers, the next row constants
        // (loads only: nothing here waits; the chain issues them behind
        // barrier 1, after the loads of the next row's late columns)
        auto prefetch_rows = [&]() { // (the chain's descriptors do not travel through registers)
            if (it + 2 < nrows)
                prefetch_x3k4(it);
            if (it + 1 < nrows)
                fetch_consts(x1.k);
        };
        if (PIPE) {
        } else if (it + 2 < nrows) {
            fetch_desc(x2);
            prefetch_x3k4(it);
        }
        if (!PIPE && it + 1 < nrows)
            fetch_consts(x1.k);
        BXH_STAMP(1);
        BXH_TL(4);
        if (MW) {
            // The waves that stored do not poll: a wave's loads would also
            // wait for the acknowledgement of its own write-through stores
            // (one counter).  The other waves fetch the partners' values.
            const int nstore = min(T, ((ro + 63) >> 6) << 6);
            const int npoll = T - nstore;
            const int first = npoll >= 64 ? tid - nstore : tid;
            const int step = npoll >= 64 ? npoll : T;
            if (first >= 0) {
                for (int idx = first; idx < r; idx += step) {
                    if (loc_s[idx] >= 0)
                        continue; // own
                    v4u g;
                    long spins = 0;
                    for (;;) { // L1-bypassing (sc1) 16-byte load of both granules
                        g = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xoff + (unsigned)idx * 16u, 0, 16);
                        if (g.y == tag32 && g.w == tag32)
                            break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > 2000000L) { // a partner is missing: give up loudly
                            __hip_atomic_store(err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    const unsigned long long bits = (unsigned long long)g.x | ((unsigned long long)g.z << 32);
                    val_s[idx] = (Float)__longlong_as_double((long long)bits);
                }
            }
        }
        const int n2 = next_pow2(r < 2 ? 2 : r);
        if (tid == t0) {
            misc_s[MS_TIE] = 0;
            misc_s[MS_FAST] = 0;
        }
        // the tie filter's bitmap (2^16 bits) shares LDS with the staging
        // slots: clear
        for (int i = tid; i < 512; i += T)
            reinterpret_cast<uint4*>(htab)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (PIPE) // what this thread stored to P in the previous row has arrived (a thread
            __builtin_amdgcn_s_waitcnt(0x0f70); // without a load of its own would not wait)
        BXH_TL(5);
        BXH_BARRIER(1);
        BXH_TL(6);
        if (PIPE) {
            // this buffer's late list is used up: empty it for the row after the next
            if (tid < LNPW)
                lnp_b[cur * LNPW + tid] = 0;
            if (tid == t0)
                misc_s[MS_LCNT + cur] = 0;
            // the next row's late columns: first chunk of every staging wave.
            // (Nothing is on its way to this wave here: combine has waited for
            // its loads.  vmcnt(0) tells the compiler, which otherwise waits
            // between the loads below for loads "pending" from the last row.)
            __builtin_amdgcn_s_waitcnt(0x0f70);
            if (it + 1 < nrows && wave < SW) {
                // (an opaque copy of the wave index: what the loads below derive from it is then
                // worked out here, in three instructions -- kept across the row as loop invariants,
                // two of those values were spilled and came back through `s_waitcnt vmcnt(0)`, one
                // behind the other, in front of the very loads this row's top waits for)
                int wv = wave;
                asm volatile("" : "+v"(wv));
                const int nl1 = misc_s[MS_LCNT + (cur ^ 1)];
                if (wv * CW < nl1)
                    issue_m(cs_b + (cur ^ 1) * RO, map_b + (cur ^ 1) * pb.rmax, nl1, wv,
                            __builtin_amdgcn_readfirstlane(lnp_b[(cur ^ 1) * LNPW + wv]), CuScope{});
            }
            // every thread has waited for its loads, and with them for the P
            // entries it stored in the previous row: that row is finished for
            // the producers
            if (tid == SW * 64 && it > 0) // (a wave without loads to issue here)
                __builtin_amdgcn_raw_buffer_store_b128(pair_of(pi_s[kprev], tag_of(it)), mrsrc,
                                                       (unsigned)(it - 1) * 16u, 0, 16);
            kprev = k;
            // the records of the row after the next: the landing buffer is free (the next
            // row was classified in front of this barrier), they are read at the top of
            // the next row
            if (wave >= SW && it + 2 < nrows)
                dma_records(x2, SW, nwaves - SW);
            prefetch_rows();
        }
        if (it + 1 < nrows)
            fill_loc(x1, cur ^ 1);
        BXH_STAMP(3);
        BXH_TL(7);

        // ---- selection without sorting (rows with no two equal reduced
        // costs whose select_variables / affect only need the extreme order
        // statistics: the set-covering / set-partitioning shapes) ------------
        // Sorted by calculator_sort a row reads  N.. N Z S.. S  where N never
        // stops the scan of select_variables, Z (== 0) flips a coin and S
        // stops it (itm-common.hpp:150-163).  The scan result, R[selected],
        // R[selected + 1] and "is this entry at or before selected" then follow
        // from the class counts, the two smallest / the largest key and the
        // keys next to the class borders.  Anything else (a tie, which needs
        // the shuffle and its RNG draws, or an interior rank) takes the sort.
        bool fast = false;
        int f_mode = 0, f_nN = 0, f_nZ = 0;
        unsigned long long f_thr = 0ull;
        Float f_d = 0, f_newpi = 0;
        bool f_pich = false;
        unsigned long long mykey[2] = { 0ull, 0ull };
        if (r <= 2 * T && r >= 2) {
            const unsigned long long K0 = order_key<Float>((Float)0, maximize);
            // per lane: A smallest key, A2 second smallest, B ~largest, C
            // ~largest of class N, D smallest of class S (maxima as minima of
            // the complement: one kind of reduction)
            unsigned long long A = ~0ull, A2 = ~0ull, B = ~0ull, C = ~0ull, D = ~0ull;
            int cN = 0, cZ = 0;
            bool flagged[2] = { false, false };
            unsigned f_old[2] = { 0u, 0u }, f_bits[2] = { 1u, 1u }; // (the filter's answers, read behind the reductions)
            // Row chain: one entry per thread over all eight waves (two where the row has more
            // than 512 elements).  Behind barrier 1 the staging waves first issue the next row's
            // late loads (twenty load instructions each, ~2 400 cycles) and the other waves the
            // message, the record DMA and the row prefetches (~1 700); with the keys of the whole
            // row on the other waves alone, two per thread (round 3), those reached barrier 2
            // ~2 600 cycles behind the staging waves (profiles/r05/chain_timeline_r05.txt); split
            // evenly the sweep is 0.8 % shorter (7.31 -> 7.25 ms).  -DBXH_KEYS_OTHER: the round-3
            // placement.
            int ke[2] = { tid, tid + T };
#ifdef BXH_KEYS_OTHER
            if (PIPE) {
                const int STG = SW * 64, OTH = T - STG;
                if (wave >= SW) {
                    ke[0] = tid - STG;
                    ke[1] = OTH + tid - STG;
                } else {
                    ke[0] = 2 * OTH + tid;
                    ke[1] = 2 * OTH + STG + tid;
                }
            }
            const bool kskip = PIPE && (ke[0] & ~63) >= r; // wave-uniform: no entry of the row here
#else
            const bool kskip = false;
#endif
            if (!kskip) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int e = ke[s2];
                const bool have = e < r;
                unsigned long long k = ~0ull;
                if (have) {
                    k = order_key<Float>(val_s[e], maximize);
                    f_old[s2] = filter_issue(reinterpret_cast<unsigned*>(htab), k, f_bits[s2]);
                }
                mykey[s2] = have ? k : 0ull;
                const bool isN = have && k < K0, isZ = have && k == K0;
                const bool isS = have && k > K0;
                cN += __popcll(__ballot(isN));
                cZ += __popcll(__ballot(isZ));
                if (s2 == 0) {
                    A = k;
                    B = have ? ~k : ~0ull;
                } else {
                    A2 = umax64(A, k);
                    A = umin64(A, k);
                    B = umin64(B, have ? ~k : ~0ull);
                }
                C = umin64(C, isN ? ~k : ~0ull);
                D = umin64(D, isS ? k : ~0ull);
            }
            // wave reductions (uniform results); the second smallest is the
            // minimum of what is left once the wave's smallest key is known
            // (keys are distinct unless the row has a tie, which takes the
            // sort path anyway)
#ifdef BXH_TL_KEYS
            if (wave >= SW) BXH_TL(13);
#endif
            const unsigned long long ownA = A;
            wave_umin64x4(A, B, C, D);
            unsigned long long E = ownA == A ? A2 : ownA;
            E = wave_umin64(E);
            // a flagged key is looked up in the whole row by its wave (rare:
            // the bitmap is 128 times larger than the row).  Behind the reductions:
            // what the filter's atomics returned is first read here, so their round
            // trip runs under the DPP chains above.
#ifdef BXH_TL_KEYS
            if (wave >= SW) BXH_TL(15);
#endif
            bool wtie = false;
            asm volatile("" : "+v"(f_old[0]), "+v"(f_old[1])); // (not before the reductions above)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                flagged[s2] = filter_hit(f_old[s2], f_bits[s2]);
                unsigned long long pending = __ballot(flagged[s2]);
                while (pending && !wtie) { // one confirmed tie settles the row
                    const int src = __ffsll((long long)pending) - 1;
                    pending &= pending - 1;
                    const unsigned long long fk = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(
                                                     (int)(unsigned)(mykey[s2] >> 32), src)
                                                   << 32) |
                                                  (unsigned)__builtin_amdgcn_readlane(
                                                    (int)(unsigned)mykey[s2], src);
                    int same = 0; // uniform: ballots over 64 entries at a time
                    for (int e0 = 0; e0 < r; e0 += 64) {
                        const int e = e0 + (tid & 63);
                        const bool eq = e < r && order_key<Float>(val_s[e], maximize) == fk;
                        same += __popcll(__ballot(eq));
                    }
                    if (same >= 2)
                        wtie = true;
                }
            }
            {
                const int wave = tid >> 6, lane = tid & 63;
                if (lane == 0) {
                    atomicMax(&part_s[64], ~B); // largest key
                    atomicMax(&part_s[65], ~C); // largest N key
                    atomicMin(&part_s[66], D);  // smallest S key
                    atomicMin(&part_s[69], A);  // smallest key
                    part_s[wave * 2] = A;
                    part_s[wave * 2 + 1] = E;
                    atomicAdd(&part_s[67], ((unsigned long long)(unsigned)cN << 32) | (unsigned)cZ);
                    if (wtie)
                        atomicOr(&part_s[68], 1ull);
                }
            }
            } else if ((tid & 63) == 0) { // no keys in this wave: neutral partial minima
                part_s[(tid >> 6) * 2] = ~0ull;
                part_s[(tid >> 6) * 2 + 1] = ~0ull;
            }
            BXH_STAMP(8);
            BXH_TL(8);
            BXH_BARRIER(2);
            BXH_STAMP(9);
            BXH_TL(9);
            // Every wave then runs the same, wave-uniform selection logic on
            // scalar registers: no second barrier, no broadcast through LDS.
            // All threads keep identical copies of the engine state (re-read
            // after a sort-path row).
            {
                const int nw = T >> 6;
                // everything the selection may read from LDS, asked for in one batch behind the
                // barrier (one round trip instead of one per use: this stretch runs on every wave
                // of the workgroup at once and is nothing but latency)
                const int lane_u = tid & 63;
                const int pre_bmin = misc_s[MS_BMIN], pre_bmax = misc_s[MS_BMAX], pre_bneg = misc_s[MS_BNEG];
                const Float pre_pi = pi_s[k];
                const Float pre_stale = stale_s[r];
                const unsigned long long pre_a1 = part_s[(lane_u < nw ? lane_u : 0) * 2],
                                         pre_a2 = part_s[(lane_u < nw ? lane_u : 0) * 2 + 1];
                const unsigned long long g1 = uniform_u64(part_s[69]);
                const unsigned long long gx = uniform_u64(part_s[64]), gxN = uniform_u64(part_s[65]),
                                         gnS = uniform_u64(part_s[66]);
                // second smallest key, only where it is asked for: the minimum
                // over the waves of "second smallest of the wave that holds
                // g1, smallest of the others"
                auto second_smallest = [&]() -> unsigned long long {
                    unsigned long long cnd = ~0ull;
                    if (lane_u < nw)
                        cnd = pre_a1 == g1 ? pre_a2 : pre_a1;
                    return wave_umin64(cnd);
                };
                const unsigned long long cnt = uniform_u64(part_s[67]);
                const unsigned long long tflag = uniform_u64(part_s[68]);
                const int nN = (int)(cnt >> 32), nZ = (int)(cnt & 0xffffffffu);
                const bool anytie = tflag != 0ull;
                BXH_STAMP(4);
                bool ok = !anytie;
                const unsigned saved_rng = rng.s;
                int sel = 0;
                unsigned long long k_sel = 0ull, k_sel1 = 0ull;
                if (ok) {
                    auto stop = [&](Float v) -> bool { // itm-common.hpp:150-163
                        if (v == 0)
                            return rng.coin();
                        return maximize ? v < 0 : v > 0;
                    };
                    const int c_size = PM1 ? pre_bneg : 0;
                    int bkmin = pre_bmin, bkmax = pre_bmax;
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
                            if (cur <= bkmax && cur < nN + nZ) { // the zero entry
                                if (rng.coin())
                                    stopped = true;
                                else
                                    ++cur;
                            }
                            if (!stopped && cur <= bkmax) {
                                if (cur < r)
                                    stopped = true; // first S entry
                                else
                                    stopped = stop(pre_stale); // R[r_size] (stale)
                            }
                            if (stopped)
                                sel = cur - 1;
                        }
                    }
                    auto key_at = [&](int rank, bool& good) -> unsigned long long {
                        if (rank == 0)
                            return g1;
                        if (rank == 1)
                            return second_smallest();
                        if (rank == r - 1)
                            return gx;
                        if (nN > 0 && rank == nN - 1)
                            return gxN;
                        if (rank == nN)
                            return nZ ? K0 : gnS;
                        if (nZ && rank == nN + nZ)
                            return gnS;
                        good = false;
                        return 0ull;
                    };
                    if (sel < 0) {
                        k_sel = key_at(0, ok); // R[0]
                    } else if (sel + 1 >= r) {
                        k_sel = key_at(r - 1, ok); // R[selected] == R[r - 1]
                    } else {
                        k_sel = key_at(sel, ok);
                        k_sel1 = key_at(sel + 1, ok);
                    }
                }
                BXH_STAMP(5);
                if (!ok) {
                    rng.s = saved_rng; // the sort path redoes the scan
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
                        const Float old_pi = pre_pi;
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
            BXH_STAMP(10);
            BXH_TL(10);
        }
        if (PIPE && wave >= SW) // the records asked for behind barrier 1 are in LDS (long since;
            __builtin_amdgcn_s_waitcnt(0x0f70); // in front of the stores below, which would be waited for too)
        if (fast) {
            BXH_STAMP(1);
            // ---- affect without the sorted order ------------------------------
            const int mode = f_mode;
            const unsigned long long thr = f_thr;
            const Float d = f_d;
            const int nN = f_nN, nZ = f_nZ;
            if (tid == t0) {
                alg_bytes += *(const double*)(misc_s + MS_RBYTES);
                if (mode == 2) {
                    pi_s[k] = f_newpi;
                    if (f_pich)
                        misc_s[MS_PICH] = 1;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int l = tid + s2 * T; // own entries only
                if (l < ro) {
                    const int idx = MW ? oidx_s[l] : l;
                    // (row chain: the keys were formed by other threads, see above)
                    const unsigned long long key =
                      (MW || PIPE) ? order_key<Float>(val_s[idx], maximize) : mykey[s2];
                    const int j = col_s[l];
                    const int pe = pe_s[l];
                    bool up = mode == 1 || (mode == 2 && key <= thr);
                    if (PM1 && pb.fac_csr[rs + idx] < 0)
                        up = !up;
                    const Float p = pown_s[l]; // P[pe] * theta
                    if (up)
                        atomicOr(&x_s[j >> 5], 1u << (j & 31));
                    else
                        atomicAnd(&x_s[j >> 5], ~(1u << (j & 31)));
                    if (PIPE) {
                        const Float np_ = up ? p + d : p - d;
                        newp_s[l] = np_;
                        store_p((unsigned)pe, np_);
                    } else
                        P[pe] = up ? p + d : p - d;
                }
                // R[pos].value after the sort only matters through
                // stop_iterating: keep its class (N / zero / S)
                const int pos = tid + s2 * T;
                if (pos < r) {
                    const Float repN = maximize ? (Float)1 : (Float)-1;
                    stale_s[pos] = pos < nN ? repN : (pos < nN + nZ ? (Float)0 : -repN);
                }
            }
            BXH_STAMP(12);
            BXH_TL(11);
            BXH_BARRIER(3);
            BXH_STAMP(6);
            BXH_TL(12);
            continue;
        }

        // ---- calculator_sort: sort, then shuffle every run of equal values -
        if (r <= 2 * T && r <= 1024) {
            if (r > 1) {
                BXH_STAMP(4);
                run_merge_sort<Float>(val_s, r, maximize, key_run, id_run, key_s, id_s, &prof);
            } else {
                if (tid == 0) {
                    key_s[0] = order_key<Float>(val_s[0], maximize);
                    id_s[0] = 0;
                }
                __syncthreads();
            }
        } else {
            for (int i = tid; i < n2; i += T) {
                key_s[i] = i < r ? order_key<Float>(val_s[i], maximize) : ~0ull;
                id_s[i] = i;
            }
            __syncthreads();
            bitonic_sort(key_s, id_s, n2);
        }
        for (int i = tid; i + 1 < r; i += T)
            if (key_s[i] == key_s[i + 1])
                misc_s[MS_TIE] = 1;
        __syncthreads();
        BXH_STAMP(4);
        // the shuffle of the tied runs on all lanes (the buffers of the sorted
        // runs are free again); a rejected draw leaves it to thread 0 below
        bool shuffled = false;
        if (misc_s[MS_TIE] && r <= 2 * T && r <= 1024) {
            unsigned stt = rng.s;
            shuffled = parallel_tie_shuffle(key_s, id_s, r, &stt, pb.lcg_pow, (int*)key_run,
                                            (int*)key_run + n2max, id_run, misc_s + MS_WAVES);
            if (shuffled)
                rng.s = stt;
        }

        // ---- thread 0: tie shuffle, select_variables, pi and d of affect ---
        if (tid == t0) {
            alg_bytes += *(const double*)(misc_s + MS_RBYTES);
            if (misc_s[MS_TIE] && !shuffled) {
                int b0 = 0;
                for (int i = 1; i <= r; ++i) {
                    if (i == r || key_s[i] != key_s[b0]) {
                        rng.shuffle(id_s + b0, (unsigned)(i - b0));
                        b0 = i;
                    }
                }
            }

            auto Rv = [&](int i) -> Float { // R[i].value, R[r] is stale
                return i < r ? val_s[id_s[i]] : stale_s[i];
            };
            auto stop = [&](Float v) -> bool { // itm-common.hpp:150-163
                if (v == 0)
                    return rng.coin();
                return maximize ? v < 0 : v > 0;
            };

            const int c_size = PM1 ? misc_s[MS_BNEG] : 0;
            int bkmin = misc_s[MS_BMIN], bkmax = misc_s[MS_BMAX];
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
                // The scan `for i = bkmin .. bkmax: stop at the first R[i] that
                // stops` over the sorted row N.. N Z.. Z S.. S: entries of class
                // N never stop, every zero flips the coin, the first S stops --
                // the class borders come from two binary searches in the keys.
                const unsigned long long K0 = order_key<Float>((Float)0, maximize);
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
                            stopped = stop(stale_s[r]); // R[r_size] (stale)
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
                const Float old_pi = pi_s[k];
                const Float new_pi = old_pi + ((Rv(sel) + Rv(sel + 1)) / two);
                pi_s[k] = new_pi;
                d += kfac * (Rv(sel + 1) - Rv(sel));
                if (signbit(old_pi) != signbit(new_pi))
                    misc_s[MS_PICH] = 1;
            }
            misc_s[MS_SEL] = sel;
            *d_s = d;
            misc_s[MS_RNG] = (int)rng.s;
        }
        __syncthreads();
        rng.s = (unsigned)misc_s[MS_RNG]; // every thread tracks the engine
        BXH_STAMP(5);

        // ---- affect: x bits and P of the row, by sorted position -----------
        {
            const int sel = misc_s[MS_SEL];
            const Float d = *d_s;
            for (int i = tid; i < r; i += T) {
                const int idx = id_s[i];
                stale_s[i] = val_s[idx];
                const int l = MW ? (int)loc_s[idx] : idx; // own entry?
                if (l < 0)
                    continue;
                const int j = col_s[l];
                const int pe = pe_s[l];
                bool up = i <= sel;
                if (PM1 && pb.fac_csr[rs + idx] < 0)
                    up = !up;
                const Float p = pown_s[l]; // P[pe] * theta
                if (up)
                    atomicOr(&x_s[j >> 5], 1u << (j & 31));
                else
                    atomicAnd(&x_s[j >> 5], ~(1u << (j & 31)));
                if (PIPE) {
                    const Float np_ = up ? p + d : p - d;
                    newp_s[l] = np_;
                    store_p((unsigned)pe, np_);
                } else
                    P[pe] = up ? p + d : p - d;
            }
        }
        __syncthreads();
        BXH_STAMP(6);
    }

    if (grp == 0) {
        for (int i = tid; i < pb.m; i += T)
            pig[i] = pi_s[i];
        for (int i = tid; i <= pb.rmax; i += T)
            staleg[i] = stale_s[i];
        if ((MW || PIPE) && ord == 7) // pi_sign_change keeps the shuffled R
            for (int i = tid; i < nrows; i += T)
                vlist[i] = vrows[i];
    }
    int cnt;
    if (MW) {
        // every owner merges the x bits of its columns into the shared words
        const unsigned* own = pb.ownmask + (size_t)grp * (2 * pb.nwords);
        for (int i = tid; i < 2 * pb.nwords; i += T) {
            const unsigned mk = own[i];
            if (mk) {
                atomicAnd(&xg[i], ~mk);
                atomicOr(&xg[i], x_s[i] & mk);
            }
        }
        cnt = rebuild_violated_owners(pb,
                                      grp,
                                      NG,
                                      x_s,
                                      st.vpart + (size_t)inst * NG * pb.m,
                                      cnt_end,
                                      err_word,
                                      st.vflag + ((size_t)inst * NG + grp) * pb.m,
                                      grp == 0 ? out_list : vrows,
                                      misc_s + MS_WAVES);
    } else if (PIPE) {
        for (int i = tid; i < 2 * pb.nwords; i += T)
            xg[i] = x_s[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) // x is there: the producers flag their shares of the rows
            __hip_atomic_store(cnt_row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        signed char* vf = st.vflag + (size_t)inst * pb.m;
        flag_violated(pb, x_s, vf, 0, (int)((long long)pb.m / NG));
        if (tid == 0)
            wait_arrivals(cnt_end, (unsigned)(NG - 1), err_word);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        cnt = rebuild_violated(pb, x_s, vf, out_list, misc_s + MS_WAVES, true);
    } else {
        for (int i = tid; i < 2 * pb.nwords; i += T)
            xg[i] = x_s[i];
        cnt = rebuild_violated(
          pb, x_s, st.vflag + (size_t)inst * pb.m, out_list, misc_s + MS_WAVES);
    }
    BXH_STAMP(7);
#ifdef BXH_PROFILE
    if (tid == 0 && grp == 0) {
        for (int i = 0; i < 16; ++i)
            st.prof[(size_t)inst * 32 + i] += prof.acc[i];
        for (int i = 0; i < 4; ++i) {
            st.prof[(size_t)inst * 32 + 16 + i] += skew[i];
            st.prof[(size_t)inst * 32 + 20 + i] += own[i];
        }
    }
#endif

    if (tid == t0 && grp == 0) {
        if (!keep)
            st.vcount[inst] = cnt;
        DevStatus s = st.status[inst];
        s.remaining = cnt;
        s.pi_changed = misc_s[MS_PICH];
        s.rng = rng.s;
        s.rows_in_sweep = (unsigned)nrows;
        s.row_updates += (unsigned long long)nrows;
        s.alg_bytes += alg_bytes;
        st.status[inst] = s;
    }
}

// ---------------------------------------------------------------------------
// Restart of an instance with the population on the device (bxh_reinit,
// include/bxh.h): crossover (lib/src/itm-optimizer-common.hpp:359-371) and
// mutation (:494-526) of `best_solution_recorder::reinit` (:528-554).  One
// workgroup per job.  The engine is sequential, so thread 0 draws; the mutation
// loop spends two draws per variable and two more where the first Bernoulli
// hits (one variable in 10^4 by default): the lanes look for the next hit with
// jumps of the engine, thread 0 handles it.
struct ReinitJob // == bxh_reinit_job
{
    int inst, cross, first, second;
    unsigned rng_cross;
    int mutate;
    unsigned rng_mut, pad_;
    double var_p, val_p;
};

__global__ __launch_bounds__(64) void
reinit_kernel(DevProblem pb, DevState st, const ReinitJob* jobs, const unsigned* pop)
{
    __shared__ int next_hit;
    const ReinitJob jb = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const int nw32 = 2 * pb.nwords;
    unsigned* xg = st.x + (size_t)jb.inst * nw32;
    unsigned state = jb.rng_mut;
    if (jb.cross) {
        const unsigned* x1 = pop + (size_t)jb.first * nw32;
        const unsigned* x2 = pop + (size_t)jb.second * nw32;
        // the masks: sequential draws (three per block, more when one is rejected)
        if (lane == 0) {
            Lcg g{ jb.rng_cross };
            for (int i = 0; i < pb.nwords; ++i) {
                const unsigned long long mask = lcg_uniform_u64(g, ~0ull);
                const unsigned long long a = ((unsigned long long)x1[2 * i + 1] << 32) | x1[2 * i];
                const unsigned long long b = ((unsigned long long)x2[2 * i + 1] << 32) | x2[2 * i];
                const unsigned long long v = ((a ^ b) & mask) ^ a;
                xg[2 * i] = (unsigned)v;
                xg[2 * i + 1] = (unsigned)(v >> 32);
            }
        }
    }
    __syncthreads();
    if (jb.mutate) {
        // variables [from, n): every lane tests a stretch of them at the draw the
        // variable would get if none before it in [from, n) hit
        int from = 0;
        const int CH = 1024; // jumps of at most 2 * CH draws (lcg_pow holds 32770 powers)
        while (from < pb.n) {
            const int upto = min(pb.n, from + CH);
            if (lane == 0)
                next_hit = upto;
            __syncthreads();
            for (int i = from + lane; i < upto; i += 64) {
                Lcg g{ lcg_jump(pb.lcg_pow, state, 2 * (i - from)) };
                if (lcg_bernoulli(g, jb.var_p))
                    atomicMin(&next_hit, i);
            }
            __syncthreads();
            const int hit = next_hit;
            state = lcg_jump(pb.lcg_pow, state, 2 * (hit - from));
            if (hit < upto) {
                Lcg g{ state };
                (void)lcg_bernoulli(g, jb.var_p); // the hit itself
                const bool v = lcg_bernoulli(g, jb.val_p);
                state = g.s;
                if (lane == 0) {
                    // bit_array_impl::set(int, bool) as the reference wrote it
                    // (bit-array.hpp:144-149): the value only reaches offset 0 of a
                    // 64-bit block, every other offset is cleared
                    const unsigned bit = 1u << (hit & 31);
                    const unsigned w = xg[hit >> 5];
                    xg[hit >> 5] = (v && (hit & 63) == 0) ? (w | bit) : (w & ~bit);
                }
                from = hit + 1;
            } else
                from = upto;
            __syncthreads();
        }
    }
    if (lane == 0)
        st.status[jb.inst].rng = state;
}

} // namespace bxh
