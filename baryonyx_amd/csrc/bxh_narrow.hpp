// baryonyx_amd/csrc/bxh_narrow.hpp -- short rows: one wavefront per Solver
// instance, several instances per CU.
//
// The staged kernel (bxh_kernels.hpp, sweep_kernel_v2) spends a 512-thread
// workgroup and most of a CU's LDS on one instance; on rows of 20..250
// elements over columns of a dozen (OR-Library scp4x, the aa0x set-partitioning
// shapes) its row time is barriers and memory round trips, not data.  Here a
// row is ONE wave's job:
//
//   * lane l owns elements l, l + 64, ... of the row (NE per lane) and walks
//     the whole column of each -- 16 padded CSC slots, one 128-byte line of P
//     plus 32 bytes of row ids -- with the two sums of compute_reduced_costs in
//     CSC order (itm-solver-inequalities-01.cpp:166-191; the order fixes the bits);
//   * no workgroup barrier anywhere: the block IS the wave (s_barrier of a
//     one-wave block is a no-op), reductions are DPP, the small sorts are the
//     wave-level ones of bxh_kernels.hpp;
//   * rows are software-pipelined through LDS: while row `it` is being decided,
//     the column lines of row it+1 and the element descriptors of row it+2 are
//     on their way, moved by LDS-DMA loads (global_load_lds_dwordx4: no
//     registers, eight lanes per 128-byte line so that a load instruction
//     touches 8 lines, not 64), into swizzled slots the owning lane then reads
//     without bank conflicts;
//   * the columns of row it+1 are read BEFORE row it stores its P values; the
//     one slot of a column that row it changes (the column's entry in row it)
//     is recomputed in the staged copy -- P*theta + d when x_j was set, - d
//     when cleared (affect, itm-common.hpp:393-436), the same two operations
//     on the same operands -- so the sums see exactly what a reload would
//     return.  Which slot that is comes from a byte per column ("slot of the
//     column's entry in the row that touched it last"), checked against the
//     staged row ids;
//   * compute_violated_constraints (itm-common.hpp:102-115) at the end of the
//     sweep goes over the SET bits of x and their columns, not over the rows.
//
// LDS per instance (aa04 shape, fp64): pi 3.4 KB, x 1.8, row list and row
// constants 8.5, slot bytes 7, descriptors 3 x 4, column lines 2 x 20, the
// buffers of one row 10: about 85 KB.
#pragma once

#include "bxh_kernels.hpp"

namespace bxh {

struct NarrowParams
{
    const int4* desc; // [nnz][2] per CSR element: {first CSC slot of its column, real entries of
                      //  the column, CSC slot of the element, column (bit 31: coefficient -1)},
                      //  {cost of the column as Float bits, 0, 0}
    const int4* rowc; // [m] per row: {first element, length | c_size << 16, bmin, bmax} (the bounds as
                      //  the Solver constructors clamp them)
};

constexpr int NARROW_RING = 8; // rows of the list whose constants wait in LDS (fetched four rows ahead)

// Several sweeps of solver_functor's loop (itm-solver-common.hpp:135-213) in one
// launch: after every sweep the instance's kappa moves as the host would move
// it, the launch ends when no row is violated, when kappa passes kappa_max or
// after max_sweeps; pi, x and the row list stay in LDS in between.  max_sweeps
// == 0: one sweep, nothing of this (bxh_sweep).
struct NarrowRun
{
    int max_sweeps;
    int first_i;             // loop index of the launch's first sweep
    int w_limit;             // kappa moves behind a sweep with index > w_limit
    double kappa_max;
    const double* kappa_inc; // [m + 1] what kappa gains behind a sweep that left that many rows
    int* out;                // [batch][4] sweeps done, fewest rows left, its sweep index, why it ended
    double* kappa_out;       // [batch]
    unsigned* xbest;         // [batch][2 * nwords] x behind the sweep that left the fewest rows
};

struct NarrowLds
{
    int off_pi, off_x, off_xbest, off_rows, off_rowc, off_slot, off_desc, off_cols, off_val, off_key, off_id,
      off_tab, off_rank, off_stale, off_misc, off_part, total, n2;
    int rp;        // row positions staged (32 * RPH)
    int desc_size; // bytes of one descriptor buffer
    int cols_size; // bytes of one column buffer (P lines, then row ids)
    int ids_off;   // offset of the row ids inside a column buffer
};

// rph: staged row positions / 32 (rmax <= 32 * rph): whole load instructions' worth of the longest row
__host__ __device__ inline NarrowLds
narrow_lds(int m, int n, int nwords, int rmax, int fsize, int rph)
{
    auto al = [](int v) { return (v + 15) & ~15; };
    NarrowLds p;
    const int n2 = next_pow2(rmax < 2 ? 2 : rmax);
    p.n2 = n2;
    p.rp = 32 * rph;
    p.desc_size = p.rp * 32;
    p.ids_off = p.rp * 16 * fsize;
    p.cols_size = p.ids_off + p.rp * 32;
    int o = 0;
    p.off_pi = o;
    o = al(o + (m + 1) * fsize);
    p.off_x = o;
    o = al(o + nwords * 8);
    p.off_xbest = o; // (unused: x behind the best sweep so far goes to NarrowRun::xbest directly)
    p.off_rows = o;
    o = al(o + m * 4);
    p.off_rowc = o; // {first element, length | c_size << 16, bmin, bmax} of the next rows of the list
    o = al(o + NARROW_RING * 16);
    p.off_slot = o; // per column: slot (inside the column) of its entry in the row that wrote it last
    o = al(o + n);
    p.off_desc = o; // 3 descriptor buffers (rows it, it+1, it+2)
    o = al(o + 3 * p.desc_size);
    p.off_cols = o; // 2 column buffers (rows it, it+1); the row activities after the sweep
    // The buffers of the sort path (and the tie filter) use the column buffer of
    // the NEXT row: its lines are still in registers while a row is decided, and
    // are stored there afterwards.
    int rowbuf = al(rmax * fsize) + al(n2 * 8) + al(n2 * 4) + al(2 * n2 * 8 > 1024 ? 2 * n2 * 8 : 1024) + al(n2 * 4);
    int cs2 = p.cols_size > rowbuf ? p.cols_size : rowbuf;
    p.cols_size = cs2;
    o = al(o + (2 * cs2 > (m + 1) * 4 ? 2 * cs2 : (m + 1) * 4));
    // offsets inside a column buffer
    {
        int q = 0;
        p.off_val = q;
        q = al(q + rmax * fsize);
        p.off_key = q;
        q = al(q + n2 * 8);
        p.off_id = q;
        q = al(q + n2 * 4);
        p.off_tab = q; // tie filter of the sort-free path (1 KB); scratch of the sorts (2 * n2 keys)
        q = al(q + (2 * n2 * 8 > 1024 ? 2 * n2 * 8 : 1024));
        p.off_rank = q;
    }
    p.off_stale = o;
    o = al(o + (rmax + 1) * fsize);
    p.off_misc = o;
    o = al(o + 96 * 4);
    p.off_part = o; // 2 x 16 u64: partial reductions of the waves, one set per row parity
    o = al(o + 2 * 16 * 8);
    p.total = o;
    return p;
}

// One padded piece (16 CSC slots) of a column: the P values and the packed row
// ids (16 bits each: row, bit 15 = coefficient -1; padding slots hold row m).
template<typename Float>
struct NarrowPiece
{
    Float p[16];
    unsigned id[8];
};

template<typename Float>
__device__ inline void
narrow_load_piece(const Float* P, const unsigned short* ids, int q, NarrowPiece<Float>& out)
{
    if constexpr (sizeof(Float) == 8) {
        const double2* src = reinterpret_cast<const double2*>(P + q);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double2 t = src[u];
            out.p[2 * u] = t.x;
            out.p[2 * u + 1] = t.y;
        }
    } else {
        const float4* src = reinterpret_cast<const float4*>(P + q);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 t = src[u];
            out.p[4 * u] = t.x;
            out.p[4 * u + 1] = t.y;
            out.p[4 * u + 2] = t.z;
            out.p[4 * u + 3] = t.w;
        }
    }
    const uint4* is = reinterpret_cast<const uint4*>(ids + q);
    const uint4 a = is[0], b = is[1];
    out.id[0] = a.x;
    out.id[1] = a.y;
    out.id[2] = a.z;
    out.id[3] = a.w;
    out.id[4] = b.x;
    out.id[5] = b.y;
    out.id[6] = b.z;
    out.id[7] = b.w;
}

// 16 bytes per lane from global memory straight into LDS (no registers): lane
// L's bytes land at lds_dst + 16 * L.  lds_dst is wave-uniform.  Counted by
// vmcnt like any load; the data may be read from LDS after s_waitcnt vmcnt(0).
__device__ inline void
narrow_dma16(const void* src, unsigned lds_dst)
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

// Tie filter of a row of at most 256 keys: 256 words of LDS, three bits of one
// word per key (see filter_insert).
__device__ inline bool
narrow_filter_insert(unsigned* bitmap, unsigned long long key)
{
    unsigned x = (unsigned)(key >> 32) * 0x9E3779B1u + (unsigned)key * 0x85EBCA77u;
    x ^= x >> 15;
    x *= 0xC2B2AE3Du;
    x ^= x >> 13;
    const unsigned word = x >> 24;
    const unsigned bits = (1u << (x & 31)) | (1u << ((x >> 5) & 31)) | (1u << ((x >> 10) & 31));
    return (atomicOr(&bitmap[word], bits) & bits) == bits;
}

// compute_order::run / push_and_run (itm-common.hpp:682-880) around
// compute_update_row / push_and_compute_update_row of the 01 and 101 solvers
// (itm-solver-inequalities-01.cpp:208-296, -equalities-01.cpp:184-270,
//  -equalities-101.cpp:208-307, -inequalities-101.cpp:250-347), one wave per
// instance.  NE: row elements per lane (rmax <= 64 * NE), NPC: pieces of a
// column that are staged ahead (longer columns fetch the rest on demand).
typedef unsigned narrow_v4u __attribute__((ext_vector_type(4)));

// compute_order::run / push_and_run (itm-common.hpp:682-880) around
// compute_update_row / push_and_compute_update_row of the 01 and 101 solvers
// (itm-solver-inequalities-01.cpp:208-296, -equalities-01.cpp:184-270,
//  -equalities-101.cpp:208-307, -inequalities-101.cpp:250-347).  W waves per
// instance (1: no barrier does anything; 4: one wave per SIMD of the CU, two
// barriers per row), RPH * 32 staged row positions (rmax <= 32 * RPH).  The
// first 16-entry piece of every column is staged ahead; longer columns fetch
// the rest on demand.
// OCC = 2 (W = 4): at most 256 registers, so that two workgroups -- one wave each per SIMD -- share
// a CU where their LDS plans allow it; the launch takes that build when the batch exceeds the CUs
// (the registers it gives up cost the lone workgroup of a CU an eighth of its speed).
template<typename Float, bool PM1, int W, int RPH, int OCC = 1>
__global__ __launch_bounds__(64 * W, OCC) void
sweep_kernel_narrow(DevProblem pb, DevState st, DevSweep a, NarrowParams np, NarrowRun run)
{
#if defined(BXH_DEV_ONLY_NP) && !defined(BXH_DEV_KEEP_NARROW)
    return; // (development builds of another kernel: see sweep_kernel_v2)
#endif

    extern __shared__ __align__(16) unsigned char smem[];
    const int inst = blockIdx.x;
    if (a.active && !a.active[inst])
        return;

    constexpr int T = 64 * W;
    constexpr int FS = (int)sizeof(Float);
    constexpr int PB = 16 * FS;      // bytes of one piece of P
    constexpr int CH = PB / 16;      // its 16-byte chunks (8 / 4)
    constexpr int EPC = 16 / FS;     // values per chunk (2 / 4)
    constexpr int CPI = 64 / CH;     // columns one load instruction covers (8 / 16)
    constexpr int RP = 32 * RPH;     // row positions
    constexpr int NE = (RP + T - 1) / T; // row elements per thread
    constexpr int NIP = RP / CPI, NII = RP / 32, NID = RP / 32; // load instructions of a row
    constexpr int NIPW = (NIP + W - 1) / W, NIIW = (NII + W - 1) / W; // ... of one wave
    const NarrowLds L = narrow_lds(pb.m, pb.n, pb.nwords, pb.rmax, FS, RPH);
    Float* pi_s = (Float*)(smem + L.off_pi);
    unsigned* x_s = (unsigned*)(smem + L.off_x);
    int* rows_s = (int*)(smem + L.off_rows);
    int4* rowc_s = (int4*)(smem + L.off_rowc);
    unsigned char* slot_s = smem + L.off_slot;
    int* act_s = (int*)(smem + L.off_cols);
    Float* stale_s = (Float*)(smem + L.off_stale);
    int* misc_s = (int*)(smem + L.off_misc);
    Float* d_s = (Float*)(misc_s + MS_D);
    unsigned long long* part_all = (unsigned long long*)(smem + L.off_part);
    // LDS addresses (the DMA destination is an address, not a pointer)
    // (the dynamic segment starts where the static LDS of the kernel, none here, ends)
    const unsigned lds0 = (unsigned)__builtin_amdgcn_groupstaticsize();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool maximize = pb.maximize != 0;
    const int kind = pb.kind;

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
    const unsigned short* ids16 = pb.col_row16;

    Float kappa = (Float)a.kappa[inst];
    const Float delta = (Float)a.delta;
    const Float theta = (Float)a.theta;
    const Float obj_amp = (Float)a.obj_amp;
    const bool push = a.is_push != 0;
    Float kfac = kappa / ((Float)1 - kappa);

    int nrows = use_m ? st.mcount[inst] : st.vcount[inst];
    for (int i = tid; i < pb.m; i += T)
        pi_s[i] = pig[i];
    if (tid == 0)
        pi_s[pb.m] = (Float)0; // the row of the padding slots
    for (int i = tid; i < 2 * pb.nwords; i += T)
        x_s[i] = xg[i];
    for (int i = tid; i <= pb.rmax; i += T)
        stale_s[i] = staleg[i];
    for (int i = tid; i < nrows; i += T)
        rows_s[i] = vlist[i];
    for (int i = tid; i < (pb.n + 3) / 4; i += T)
        ((unsigned*)slot_s)[i] = 0u;
    // push sweeps of the 01 solvers add the cost at the LOCAL id of the element
    // (a quirk of the reference): the same per thread for every row
    Float cost_local[NE];
#pragma unroll
    for (int s = 0; s < NE; ++s)
        cost_local[s] = (!PM1 && push && tid + T * s < pb.n) ? cost[tid + T * s] : (Float)0;

    MinStd rng;
    rng.s = st.status[inst].rng; // every thread tracks the engine
    Prof prof;
    (void)prof;
#ifdef BXH_PROFILE
    for (int i = 0; i < 16; ++i)
        prof.acc[i] = 0;
    prof.t = wall_clock64();
#endif

    auto part_reset = [&](unsigned long long* ps) { // thread 0
        ps[8] = 0ull;   // largest key
        ps[9] = 0ull;   // largest N key
        ps[10] = ~0ull; // smallest S key
        ps[11] = 0ull;  // counts: N in the high half, Z in the low half
        ps[12] = 0ull;  // tie flag
        ps[13] = ~0ull; // smallest key
    };
    int cnt = 0;                  // rows violated behind the last sweep
    int sweeps_done = 0, stop_why = 0;
    int best_rem = 0x7fffffff, best_i = 0;
    unsigned long long rows_total = 0;
    double alg_total = 0.0;
    unsigned rows_last = 0;
    for (int sw = 0;; ++sw) { // (one pass unless run.max_sweeps says otherwise)
    if (tid == 0) {
        misc_s[MS_PICH] = 0;
        part_reset(part_all);
        part_reset(part_all + 16);
    }
    __syncthreads();
    kfac = kappa / ((Float)1 - kappa);
    // compute_order: random_sorting shuffles the list before the sweep
    // (itm-common.hpp:708-714 / :817-823)
    if (ord == 2 || ord == 7) {
        if (tid == 0) {
            rng.shuffle(rows_s, (unsigned)nrows);
            misc_s[MS_RNG] = (int)rng.s;
        }
        __syncthreads();
        rng.s = (unsigned)misc_s[MS_RNG];
    }

    // The constants of the first rows of the list, in sweep order (a ring in LDS: the row
    // loop fetches those of row it + 4 while it works on row it -- no memory round trip per
    // row), and the algorithmic bytes of the sweep (every row of the list is updated
    // exactly once; the figures are whole numbers, any order adds up the same).
    double alg_bytes = 0.0; // thread 0
    {
        double bytes = 0.0;
        for (int i = tid; i < nrows; i += T) {
            const int k = (ord == 1) ? rows_s[nrows - 1 - i] : rows_s[i];
            if (i < NARROW_RING / 2) // (the row loop fetches the others, four rows ahead)
                rowc_s[i] = np.rowc[k];
            bytes += pb.row_bytes[k];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            bytes += __shfl_xor(bytes, o, 64);
        if (lane == 0)
            ((double*)(misc_s + MS_WAVES))[wave] = bytes;
        __syncthreads();
        for (int w = 0; w < W; ++w)
            alg_bytes += ((double*)(misc_s + MS_WAVES))[w];
    }
    __syncthreads();

    // ---- the two stages of loads ahead of the row loop ------------------------
    // descriptors of row `it` -> buffer it % 3: lane L brings half (L & 1) of the
    // 32-byte descriptor of element 32 * k + (L >> 1): 1 KB, contiguous, per
    // instruction (LDS-DMA: no registers); instruction k is wave k % W's
    auto issue_desc = [&](int it) {
        if (it >= nrows)
            return;
        const int4 c = rowc_s[it & (NARROW_RING - 1)];
        const int rs = __builtin_amdgcn_readfirstlane(c.x);
        const int r = __builtin_amdgcn_readfirstlane(c.y & 0xffff);
        const unsigned dst = lds0 + (unsigned)L.off_desc + (unsigned)(it % 3) * (unsigned)L.desc_size;
        const char* src = (const char*)np.desc + ((size_t)rs * 32 + (size_t)lane * 16);
#pragma unroll
        for (int kk = 0; kk < (NID + W - 1) / W; ++kk) {
            const int k = wave + W * kk;
            if (32 * k < r) { // (wave-uniform)
                if (32 * k + (lane >> 1) < r)
                    narrow_dma16(src + k * 1024, __builtin_amdgcn_readfirstlane(dst + k * 1024));
            }
        }
    };
    // Column lines of row `it`, by the descriptors in buffer it % 3, into
    // registers; land_cols() stores them to buffer it & 1 once they have arrived.
    // P: lane L brings a 16-byte chunk of the line of column CPI * i + L / CH; which
    // chunk is swizzled by the column so that the owner's reads (thread == column)
    // of one chunk number spread over all banks.  Row ids: two lanes per column.
    // Instruction i is wave i % W's.  (The LDS-DMA form of these loads issued
    // four times slower.)  Lanes beyond the row read the row's first column again.
    narrow_v4u creg[NIPW + NIIW];
    auto issue_cols = [&](int it) {
        if (it >= nrows)
            return;
        const int r = __builtin_amdgcn_readfirstlane(rowc_s[it & (NARROW_RING - 1)].y & 0xffff);
        const int4* dsc = (const int4*)(smem + L.off_desc + (it % 3) * L.desc_size);
        int qp[NIPW], qi[NIIW];
#pragma unroll
        for (int t = 0; t < NIPW; ++t) {
            const int c = CPI * (wave + W * t) + lane / CH;
            qp[t] = dsc[(c < r ? c : 0) * 2].x;
        }
#pragma unroll
        for (int t = 0; t < NIIW; ++t) {
            const int c = 32 * (wave + W * t) + (lane >> 1);
            qi[t] = dsc[(c < r ? c : 0) * 2].x;
        }
#pragma unroll
        for (int t = 0; t < NIPW; ++t) {
            const int c = CPI * (wave + W * t) + lane / CH;
            const int u = (lane % CH) ^ (c & (CH - 1));
            creg[t] = *(const narrow_v4u*)((const char*)(P + qp[t]) + u * 16);
        }
#pragma unroll
        for (int t = 0; t < NIIW; ++t) {
            const int c = 32 * (wave + W * t) + (lane >> 1);
            const int h = (lane & 1) ^ ((c >> 2) & 1);
            creg[NIPW + t] = *(const narrow_v4u*)((const char*)(ids16 + qi[t]) + h * 16);
        }
    };
    auto land_cols = [&](int it) {
        if (it >= nrows)
            return;
        unsigned char* cb = smem + L.off_cols + (it & 1) * L.cols_size;
#pragma unroll
        for (int t = 0; t < NIPW; ++t)
            if (wave + W * t < NIP)
                *(narrow_v4u*)(cb + (wave + W * t) * 1024 + lane * 16) = creg[t];
#pragma unroll
        for (int t = 0; t < NIIW; ++t)
            if (wave + W * t < NII)
                *(narrow_v4u*)(cb + L.ids_off + (wave + W * t) * 1024 + lane * 16) = creg[NIPW + t];
    };
    auto wait_loads = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    issue_desc(0);
    issue_desc(1);
    wait_loads();
    __syncthreads();
    issue_cols(0);
    wait_loads();
    land_cols(0);
    // the tie filter of row 0 (in the column buffer of row 1)
    if (tid < 64)
        reinterpret_cast<uint4*>(smem + L.off_cols + L.cols_size + L.off_tab)[tid] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    BXH_PSTAMP(&prof, 0);

    int kprev = -1;       // the row updated just before (its P stores are not in the staged lines)
    Float dprev = (Float)0;

    for (int it = 0; it < nrows; ++it) {
        // ---- the loads of the rows ahead (complete at this row's wait_loads) -----
        // the constants of row it + 4 first (one lane of the last wave; into the ring behind this
        // row's wait for its loads).  In FRONT of the column loads: the compiler waits for a
        // register's older load before it writes the register again, and its wait counter is
        // in order -- behind them that wait would be for the columns just asked for
        int4 rc_ahead = make_int4(0, 0, 0, 0);
        const bool rc_mine = tid == T - 64 && it + NARROW_RING / 2 < nrows;
        if (rc_mine) {
            const int ia = it + NARROW_RING / 2;
            rc_ahead = np.rowc[(ord == 1) ? rows_s[nrows - 1 - ia] : rows_s[ia]];
        }
        issue_desc(it + 2);
        issue_cols(it + 1);
        BXH_PSTAMP(&prof, 1);

        const int4 rowc = rowc_s[it & (NARROW_RING - 1)];
        const int k = __builtin_amdgcn_readfirstlane(
          (ord == 1) ? rows_s[nrows - 1 - it] : rows_s[it]);
        const int r = __builtin_amdgcn_readfirstlane(rowc.y & 0xffff);
        const int row_bmin = __builtin_amdgcn_readfirstlane(rowc.z);
        const int row_bmax = __builtin_amdgcn_readfirstlane(rowc.w);
        const int row_csize = PM1 ? __builtin_amdgcn_readfirstlane(rowc.y >> 16) : 0;
        const int4* dsc = (const int4*)(smem + L.off_desc + (it % 3) * L.desc_size);
        unsigned char* cbuf = smem + L.off_cols + (it & 1) * L.cols_size;
        // the sort path's buffers and the tie filter: in the other column buffer
        unsigned char* rbuf = smem + L.off_cols + ((it + 1) & 1) * L.cols_size;
        Float* val_s = (Float*)(rbuf + L.off_val);
        unsigned long long* key_s = (unsigned long long*)(rbuf + L.off_key);
        int* id_s = (int*)(rbuf + L.off_id);
        unsigned* filt_s = (unsigned*)(rbuf + L.off_tab);
        unsigned long long* key_run = (unsigned long long*)(rbuf + L.off_tab);
        int* id_run = (int*)(key_run + L.n2);
        int* rank_s = (int*)(rbuf + L.off_rank);
        unsigned long long* part_s = part_all + (it & 1) * 16;

        // ---- decrease_preference + compute_reduced_costs (+ push) ----------
        struct Elem
        {
            int pe, jw, own; // CSC slot of the element, column | sign bit, slot inside the column
        };
        Elem el[NE];
        Float v[NE], pown[NE];
        unsigned long long key[NE];
#pragma unroll
        for (int s = 0; s < NE; ++s) {
            const int idx = tid + T * s;
            v[s] = (Float)0;
            pown[s] = (Float)0;
            key[s] = ~0ull;
            el[s].pe = 0;
            el[s].jw = 0;
            el[s].own = 0;
            if (idx < r) {
                const int4 d0 = dsc[idx * 2];
                const int4 d1 = dsc[idx * 2 + 1];
                const int q0 = d0.x, len = d0.y, pe = d0.z;
                el[s].pe = pe;
                el[s].jw = d0.w;
                const int j = d0.w & 0x7fffffff;
                const bool eneg = PM1 && d0.w < 0;
                const int own = pe - q0;
                el[s].own = own;
                Float cj;
                if constexpr (FS == 8)
                    cj = (Float)__longlong_as_double(((long long)(unsigned)d1.y << 32) | (unsigned)d1.x);
                else
                    cj = (Float)__int_as_float(d1.x);
                // where element t (< 16) of the column sits in the staged copy
                const int cb_off = L.off_cols + (it & 1) * L.cols_size; // (offsets from smem: the
                auto p_at = [&](int t) -> int {                       //  accesses stay LDS accesses)
                    const int u = (t / EPC) ^ (idx & (CH - 1));
                    return cb_off + idx * PB + u * 16 + (t % EPC) * FS;
                };
                auto id_at = [&](int t) -> unsigned {
                    const int h = (t >> 3) ^ ((idx >> 2) & 1);
                    return *((const unsigned short*)(cbuf + L.ids_off + idx * 32 + h * 16) + (t & 7));
                };
                // the slot the previous row has just stored: recomputed in the staged copy
                {
                    const int sp = (int)slot_s[j];
                    if (sp < 16 && sp < len && (int)(id_at(sp) & 0x7fffu) == kprev) {
                        const bool xbit = (x_s[j >> 5] >> (j & 31)) & 1u;
                        const int po_ = p_at(sp);
                        Float pv = *(Float*)(smem + po_) * theta;
                        pv = pv + (xbit ? dprev : -dprev);
                        *(Float*)(smem + po_) = pv;
                    }
                }
                // P[e] *= theta (written back by affect)
                Float po = 0;
                if (own < 16) {
                    const int po_ = p_at(own);
                    po = *(Float*)(smem + po_) * theta;
                    *(Float*)(smem + po_) = po;
                }
                Float spi = 0, sp = 0;
                auto accumulate = [&](const NarrowPiece<Float>& pcs, int base, bool check_own) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const unsigned h = (pcs.id[t >> 1] >> ((t & 1) * 16)) & 0xffffu;
                        const int row = (int)(h & 0x7fffu);
                        Float pv = pcs.p[t];
                        if (check_own && base + t == own) {
                            pv = pv * theta;
                            po = pv;
                        }
                        const Float piv = pi_s[row];
                        if (PM1) {
                            const bool neg = (h & 0x8000u) != 0;
                            spi += neg ? -piv : piv;
                            sp += neg ? -pv : pv;
                        } else {
                            spi += piv;
                            sp += pv;
                        }
                    }
                };
                {
                    NarrowPiece<Float> pcs;
                    const unsigned char* pl = cbuf + idx * PB;
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        const uint4 w = *(const uint4*)(pl + ((u ^ (idx & (CH - 1))) * 16));
                        if constexpr (FS == 8) {
                            pcs.p[2 * u] = __longlong_as_double(((long long)w.y << 32) | w.x);
                            pcs.p[2 * u + 1] = __longlong_as_double(((long long)w.w << 32) | w.z);
                        } else {
                            pcs.p[4 * u] = __uint_as_float(w.x);
                            pcs.p[4 * u + 1] = __uint_as_float(w.y);
                            pcs.p[4 * u + 2] = __uint_as_float(w.z);
                            pcs.p[4 * u + 3] = __uint_as_float(w.w);
                        }
                    }
                    const unsigned char* il = cbuf + L.ids_off + idx * 32;
                    const uint4 ia = *(const uint4*)(il + (((idx >> 2) & 1) * 16));
                    const uint4 ib = *(const uint4*)(il + ((((idx >> 2) & 1) ^ 1) * 16));
                    pcs.id[0] = ia.x;
                    pcs.id[1] = ia.y;
                    pcs.id[2] = ia.z;
                    pcs.id[3] = ia.w;
                    pcs.id[4] = ib.x;
                    pcs.id[5] = ib.y;
                    pcs.id[6] = ib.z;
                    pcs.id[7] = ib.w;
                    accumulate(pcs, 0, false);
                }
                for (int b = 16; b < len; b += 16) { // the rest of a long column: fetched now
                    NarrowPiece<Float> extra;
                    narrow_load_piece<Float>(P, ids16, q0 + b, extra);
                    accumulate(extra, b, true);
                }
                Float vv = cj - spi - sp;
                if (eneg)
                    vv = -vv;
                if (push) // the 01 solvers index the cost with the LOCAL id (quirk)
                    vv += obj_amp * (PM1 ? cj : cost_local[s]);
                v[s] = vv;
                pown[s] = po;
                key[s] = order_key<Float>(vv, maximize);
            }
        }

        BXH_PSTAMP(&prof, 2);
        // ---- selection without sorting (see sweep_kernel_v2): class counts,
        // the extreme keys and the keys next to the class borders; the waves'
        // partial results meet in LDS ----------------------------------------
        bool fast = false;
        int f_mode = 0, f_nN = 0, f_nZ = 0;
        unsigned long long f_thr = 0ull;
        Float f_d = 0, f_newpi = 0;
        bool f_pich = false;
        const unsigned long long K0 = order_key<Float>((Float)0, maximize);
        // an equality row (or bmin == bmax) that selects exactly one element only asks for
        // the two smallest keys: the other extreme keys are not reduced then
        const bool only_two = (kind == 0 || kind == 2 || row_bmin == row_bmax) &&
                              min(row_bmin + row_csize, r) - 1 == 0;
        if (r >= 2) {
            unsigned long long A = ~0ull, A2 = ~0ull, B = ~0ull, C = ~0ull, D = ~0ull;
            int cN = 0, cZ = 0;
            bool flagged = false;
#pragma unroll
            for (int s = 0; s < NE; ++s) {
                const bool have = tid + T * s < r;
                const unsigned long long kk = key[s];
                // a flagged key MAY equal an earlier one: the row then takes the sort
                // path, which is always right
                if (have)
                    flagged |= narrow_filter_insert(filt_s, kk);
                const bool isN = have && kk < K0, isZ = have && kk == K0, isS = have && kk > K0;
                cN += __popcll(__ballot(isN));
                cZ += __popcll(__ballot(isZ));
                if (kk < A) {
                    A2 = A;
                    A = kk;
                } else
                    A2 = umin64(A2, kk);
                B = umin64(B, have ? ~kk : ~0ull);
                C = umin64(C, isN ? ~kk : ~0ull);
                D = umin64(D, isS ? kk : ~0ull);
            }
            const bool wtie = __ballot(flagged) != 0ull;
            const unsigned long long ownA = A;
            if (only_two)
                A = wave_umin64(A);
            else
                wave_umin64x4(A, B, C, D);
            unsigned long long E = ownA == A ? A2 : ownA;
            E = wave_umin64(E);
            if (lane == 0) {
                const unsigned long long cnt = ((unsigned long long)(unsigned)cN << 32) | (unsigned)cZ;
                if (W > 1) {
                    if (!only_two) {
                        atomicMax(&part_s[8], ~B);
                        atomicMax(&part_s[9], ~C);
                        atomicMin(&part_s[10], D);
                    }
                    atomicMin(&part_s[13], A);
                    atomicAdd(&part_s[11], cnt);
                    if (wtie)
                        atomicOr(&part_s[12], 1ull);
                } else {
                    part_s[8] = ~B;
                    part_s[9] = ~C;
                    part_s[10] = D;
                    part_s[13] = A;
                    part_s[11] = cnt;
                    part_s[12] = wtie ? 1ull : 0ull;
                }
                part_s[wave * 2] = A;
                part_s[wave * 2 + 1] = E;
            }
        }
        __syncthreads();
        if (W > 1 && tid == 0) // the other set, for the next row
            part_reset(part_all + ((it + 1) & 1) * 16);
        if (r >= 2) {
            const unsigned long long g1 = uniform_u64(part_s[13]);
            const unsigned long long gx = uniform_u64(part_s[8]), gxN = uniform_u64(part_s[9]),
                                     gnS = uniform_u64(part_s[10]);
            const unsigned long long cnt = uniform_u64(part_s[11]);
            const int nN = (int)(cnt >> 32), nZ = (int)(cnt & 0xffffffffu);
            const bool anytie = uniform_u64(part_s[12]) != 0ull;
            // second smallest key: the minimum over the waves of "second smallest of
            // the wave that holds g1, smallest of the others"
            unsigned long long cnd = ~0ull;
            if (lane < W) {
                const unsigned long long a1 = part_s[lane * 2], a2 = part_s[lane * 2 + 1];
                cnd = a1 == g1 ? a2 : a1;
            }
            const unsigned long long E2 = wave_umin64(cnd);
            bool ok = !anytie;
            const unsigned saved_rng = rng.s;
            int sel = 0;
            unsigned long long k_sel = 0ull, k_sel1 = 0ull;
            if (ok) {
                auto stop = [&](Float sv) -> bool { // itm-common.hpp:150-163
                    if (sv == 0)
                        return rng.coin();
                    return maximize ? sv < 0 : sv > 0;
                };
                const int c_size = row_csize;
                int bkmin = row_bmin, bkmax = row_bmax;
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
                                stopped = stop(stale_s[r]); // R[r_size] (stale)
                        }
                        if (stopped)
                            sel = cur - 1;
                    }
                }
                auto key_at = [&](int rank, bool& good) -> unsigned long long {
                    if (rank == 0)
                        return g1;
                    if (rank == 1)
                        return E2;
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
                    k_sel = key_at(0, ok);
                } else if (sel + 1 >= r) {
                    k_sel = key_at(r - 1, ok);
                } else {
                    k_sel = key_at(sel, ok);
                    k_sel1 = key_at(sel + 1, ok);
                }
            }
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
                    const Float old_pi = pi_s[k];
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

        BXH_PSTAMP(&prof, 3);
        // the columns read ahead must have arrived before this row's stores to
        // the same lines are issued
        wait_loads();
        if (rc_mine)
            rowc_s[(it + NARROW_RING / 2) & (NARROW_RING - 1)] = rc_ahead;
        BXH_PSTAMP(&prof, 4);
        // the tie filter of the next row lives where this row's lines were read from
        if (tid < 64)
            reinterpret_cast<uint4*>(cbuf + L.off_tab)[tid] = make_uint4(0u, 0u, 0u, 0u);

        if (fast) {
            const int mode = f_mode;
            const Float d = f_d;
            if (tid == 0 && mode == 2) {
                pi_s[k] = f_newpi;
                if (f_pich)
                    misc_s[MS_PICH] = 1;
            }
#pragma unroll
            for (int s = 0; s < NE; ++s) {
                const int idx = tid + T * s;
                if (idx < r) {
                    const int j = el[s].jw & 0x7fffffff;
                    const int pe = el[s].pe;
                    bool up = mode == 1 || (mode == 2 && key[s] <= f_thr);
                    if (PM1 && el[s].jw < 0)
                        up = !up;
                    const Float p = pown[s]; // P[pe] * theta
                    if (up)
                        atomicOr(&x_s[j >> 5], 1u << (j & 31));
                    else
                        atomicAnd(&x_s[j >> 5], ~(1u << (j & 31)));
                    P[pe] = up ? p + d : p - d;
                    slot_s[j] = (unsigned char)el[s].own;
                    // R[pos].value after the sort only matters through
                    // stop_iterating: keep its class (N / zero / S)
                    const Float repN = maximize ? (Float)1 : (Float)-1;
                    stale_s[idx] = idx < f_nN ? repN : (idx < f_nN + f_nZ ? (Float)0 : -repN);
                }
            }
            dprev = d;
            BXH_PSTAMP(&prof, 5);
        } else {
            // ---- calculator_sort: sort, then shuffle every run of equal values
#pragma unroll
            for (int s = 0; s < NE; ++s)
                if (tid + T * s < r)
                    val_s[tid + T * s] = v[s];
            if (tid == 0)
                misc_s[MS_TIE] = 0;
            __syncthreads();
            const int n2 = next_pow2(r < 2 ? 2 : r);
            if (r <= 2 * T) {
                if (r > 1) {
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
            bool shuffled = false;
            if (misc_s[MS_TIE] && r <= 2 * T) {
                unsigned stt = rng.s;
                shuffled = parallel_tie_shuffle(key_s, id_s, r, &stt, pb.lcg_pow, (int*)key_run,
                                                (int*)key_run + L.n2, id_run, misc_s + MS_WAVES);
                if (shuffled)
                    rng.s = stt;
            }
            if (tid == 0) {
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
                auto stop = [&](Float sv) -> bool {
                    if (sv == 0)
                        return rng.coin();
                    return maximize ? sv < 0 : sv > 0;
                };
                const int c_size = row_csize;
                int bkmin = row_bmin, bkmax = row_bmax;
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
            rng.s = (unsigned)misc_s[MS_RNG];
            const int sel = misc_s[MS_SEL];
            const Float d = *d_s;
            for (int i = tid; i < r; i += T)
                rank_s[id_s[i]] = i;
            __syncthreads();
            for (int i = tid; i < r; i += T)
                stale_s[i] = val_s[id_s[i]]; // R[i].value after the sort
#pragma unroll
            for (int s = 0; s < NE; ++s) {
                const int idx = tid + T * s;
                if (idx < r) {
                    const int j = el[s].jw & 0x7fffffff;
                    const int pe = el[s].pe;
                    bool up = rank_s[idx] <= sel; // sel < 0: none; sel + 1 >= r: all
                    if (PM1 && el[s].jw < 0)
                        up = !up;
                    const Float p = pown[s];
                    if (up)
                        atomicOr(&x_s[j >> 5], 1u << (j & 31));
                    else
                        atomicAnd(&x_s[j >> 5], ~(1u << (j & 31)));
                    P[pe] = up ? p + d : p - d;
                    slot_s[j] = (unsigned char)el[s].own;
                }
            }
            __syncthreads(); // (the sort path's buffers are done with: the lines of the next row go there)
            dprev = d;
            BXH_PSTAMP(&prof, 6);
        }
        kprev = k;
        land_cols(it + 1);
        __syncthreads(); // x, pi, the slot bytes and the staged lines of the next row are in place
        BXH_PSTAMP(&prof, 7);
    }

    // ---- compute_violated_constraints ----------------------------------------
    __syncthreads();
    if (ord == 7) // pi_sign_change keeps the shuffled R
        for (int i = tid; i < nrows; i += T)
            vlist[i] = rows_s[i];

    // row activities from the set bits of x: every thread takes words of x, four
    // columns at a time (their descriptor loads, then their row ids, together)
    for (int i = tid; i <= pb.m; i += T)
        act_s[i] = 0;
    __syncthreads();
    for (int w = tid; w < 2 * pb.nwords; w += T) {
        unsigned word = x_s[w];
        while (word) {
            int jj[4], q0[4], ln[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (word) {
                    jj[u] = w * 32 + (__ffs((int)word) - 1);
                    word &= word - 1;
                    if (jj[u] >= pb.n)
                        jj[u] = -1;
                } else
                    jj[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                q0[u] = jj[u] >= 0 ? pb.col_start[jj[u]] : 0;
                ln[u] = jj[u] >= 0 ? pb.col_len[jj[u]] : 0;
            }
            uint4 ia[4], ib[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ln[u] > 0) {
                    const uint4* is = reinterpret_cast<const uint4*>(ids16 + q0[u]);
                    ia[u] = is[0];
                    ib[u] = is[1];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ln[u] > 0) {
                    const unsigned wd[8] = { ia[u].x, ia[u].y, ia[u].z, ia[u].w,
                                             ib[u].x, ib[u].y, ib[u].z, ib[u].w };
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const unsigned h = (wd[t >> 1] >> ((t & 1) * 16)) & 0xffffu;
                        if (t < ln[u])
                            atomicAdd(&act_s[h & 0x7fffu], (h & 0x8000u) ? -1 : 1);
                    }
                    for (int t = 16; t < ln[u]; ++t) {
                        const unsigned h = ids16[q0[u] + t];
                        atomicAdd(&act_s[h & 0x7fffu], (h & 0x8000u) ? -1 : 1);
                    }
                }
        }
    }
    __syncthreads();
    cnt = 0;
    {
        int* scratch_s = misc_s + MS_WAVES;
        for (int c = 0; c < pb.m; c += T) {
            const int kq = c + tid;
            bool f = false;
            if (kq < pb.m) {
                const int av = act_s[kq];
                f = !(pb.bmin[kq] <= av && av <= pb.bmax[kq]);
            }
            const unsigned long long mask = __ballot(f);
            const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
            int off = 0, tot = __popcll(mask);
            if (W > 1) {
                if (lane == 0)
                    scratch_s[wave] = __popcll(mask);
                __syncthreads();
                tot = 0;
                for (int w = 0; w < W; ++w) {
                    const int cw = scratch_s[w];
                    if (w < wave)
                        off += cw;
                    tot += cw;
                }
            }
            if (f) {
                out_list[cnt + off + prefix] = kq;
                if (!keep)
                    rows_s[cnt + off + prefix] = kq; // (the list of the next sweep of this launch)
            }
            cnt += tot;
            if (W > 1)
                __syncthreads();
        }
    }
    // ---- solver_functor's bookkeeping behind a sweep (every thread alike) ------
    rows_total += (unsigned long long)nrows;
    alg_total += alg_bytes;
    rows_last = (unsigned)nrows;
    sweeps_done = sw + 1;
    if (run.max_sweeps <= 0)
        break;
    {
        const int i = run.first_i + sw;
        if (cnt == 0)
            stop_why = 1;
        else {
            if (cnt < best_rem) {
                best_rem = cnt;
                best_i = i;
                for (int w = tid; w < 2 * pb.nwords; w += T) // (global memory: read by the host only)
                    run.xbest[(size_t)inst * (2 * pb.nwords) + w] = x_s[w];
            }
            if (i > run.w_limit)
                kappa = kappa + (Float)run.kappa_inc[cnt];
            if (kappa > (Float)run.kappa_max)
                stop_why = 2;
        }
        if (stop_why || sw + 1 >= run.max_sweeps)
            break;
    }
    nrows = cnt;
    __syncthreads();
    } // sweeps of this launch

    // ---- write back ---------------------------------------------------------------
    __syncthreads();
    for (int i = tid; i < pb.m; i += T)
        pig[i] = pi_s[i];
    for (int i = tid; i < 2 * pb.nwords; i += T)
        xg[i] = x_s[i];
    for (int i = tid; i <= pb.rmax; i += T)
        staleg[i] = stale_s[i];
    if (run.max_sweeps > 0) {
        if (tid == 0) {
            run.out[inst * 4 + 0] = sweeps_done;
            run.out[inst * 4 + 1] = best_rem;
            run.out[inst * 4 + 2] = best_i;
            run.out[inst * 4 + 3] = stop_why;
            run.kappa_out[inst] = (double)kappa;
        }
    }

    BXH_PSTAMP(&prof, 8);
#ifdef BXH_PROFILE
    if (tid == 0)
        for (int i = 0; i < 16; ++i)
            st.prof[(size_t)inst * 32 + i] += prof.acc[i];
#endif
    if (tid == 0) {
        if (!keep)
            st.vcount[inst] = cnt;
        DevStatus s = st.status[inst];
        s.remaining = cnt;
        s.pi_changed = misc_s[MS_PICH];
        s.rng = rng.s;
        s.rows_in_sweep = rows_last;
        s.row_updates += rows_total;
        s.alg_bytes += alg_total;
        st.status[inst] = s;
    }
}

} // namespace bxh
