// baryonyx_amd/csrc/bxh_swave.hpp -- short rows, a wave per row: the 01 / 101 solvers
// (equalities / inequalities with coefficients +1 or -1 / +1; the reference's
// lib/src/itm-solver-{equalities,inequalities}-{01,101}.cpp, compute_update_row /
// push_and_compute_update_row, with affect() of lib/src/itm-common.hpp:382-467) on instances
// whose rows all have at most 63 elements, in the form sweep_kernel_z<Float, true> introduced
// for the Z solver (bxh_kernels.hpp): ONE WAVE works a row, wave-synchronously, and up to four
// consecutive rows of the list that share no column -- they commute exactly: neither reads a pi,
// P or x entry the other writes -- run side by side, between two barriers.  What the reference
// does in sequence stays in sequence: the tie shuffles of calculator_sort and the coin flips of
// stop_iterating (a reduced cost of exactly zero, lib/src/itm-common.hpp:150-163) draw from the
// one engine in row order (thread 0, behind the first barrier), and R's leftovers -- the
// `R[r_size]` read of the inequality solvers' select_variables -- go by row order.
//
// Every sum keeps the reference's order (CSC order, two accumulators, a * pi and a * P for the
// 101 solvers), so pi / P / x are bit-identical to the oracle, as in every other kernel form.
#ifndef BXH_SWAVE_HPP
#define BXH_SWAVE_HPP

#include "bxh_kernels.hpp"
#include "bxh_narrow.hpp" // NarrowRun: several sweeps of solver_functor's loop per launch (bxh_run)

namespace bxh {

struct SwLds
{
    int off_pi, off_x, off_stale, off_misc, off_waves, off_table, total;
};
__host__ __device__ constexpr int
sw_table_bytes()
{
    return ZW_TABLE * (2 * 16 + 8 + 8); // {k, rs, re, c_size}, {bmin, bmax, group, r}, mask, bytes
}
__host__ __device__ inline SwLds
sw_lds(int m, int nwords, int rmax, int fsize)
{
    auto al = [](int v) { return (v + 15) & ~15; };
    SwLds p{};
    int o = 0;
    p.off_pi = o;
    o = al(o + (m + 1) * fsize); // (pi[m] = +0.0: the row of a column's padding slots)
    p.off_x = o;
    o = al(o + nwords * 8);
    p.off_stale = o;
    o = al(o + (rmax + 1) * fsize);
    p.off_misc = o;
    o = al(o + 96 * 4);
    p.off_waves = o;
    o = al(o + ZW_WAVES * zw_wave_bytes(fsize));
    p.off_table = o;
    o = al(o + sw_table_bytes());
    p.total = o;
    return p;
}

template<typename Float, bool PM1>
__global__ __launch_bounds__(256) void
sweep_kernel_sw(DevProblem pb, DevState st, DevSweep a, NarrowRun run)
{
#if defined(BXH_DEV_ONLY_NP) && !defined(BXH_DEV_KEEP_SW)
    return; // (development builds of another kernel: see sweep_kernel_v2)
#endif

    extern __shared__ __align__(16) unsigned char smem[];
    const int inst = blockIdx.x;
    if (a.active && !a.active[inst])
        return;
    const SwLds lp = sw_lds(pb.m, pb.nwords, pb.rmax, (int)sizeof(Float));
    Float* const pi_s = (Float*)(smem + lp.off_pi);
    unsigned* const x_s = (unsigned*)(smem + lp.off_x);
    Float* const stale_s = (Float*)(smem + lp.off_stale);
    int* const misc_s = (int*)(smem + lp.off_misc);
    unsigned char* const zw_base = smem + lp.off_waves;
    unsigned char* const zt_base = smem + lp.off_table;
    int4* const t_a = (int4*)zt_base;                       // {k, rs, re, c_size}
    int4* const t_b = t_a + ZW_TABLE;                       // {bmin, bmax, rows of the group starting here, r}
    unsigned long long* const t_conf = (unsigned long long*)(t_b + ZW_TABLE);
    double* const t_rb = (double*)(t_conf + ZW_TABLE);

    const int tid = threadIdx.x, T = blockDim.x;
    const int wave = tid >> 6, lane = tid & 63, NW = T >> 6;
    const bool maximize = pb.maximize != 0;
    const int kind = pb.kind;

    Float* const P = (Float*)st.P + (size_t)inst * pb.nnzp;
    Float* const pig = (Float*)st.pi + (size_t)inst * pb.m;
    unsigned* const xg = st.x + (size_t)inst * (2 * pb.nwords);
    Float* const staleg = (Float*)st.rstale + (size_t)inst * (pb.rmax + 1);
    const int ord = order_of(a, inst);
    const bool use_m = ord == 3 || ord == 4;
    const bool keep = use_m || ord == 7;
    int* const vlist = use_m ? st.mrun + (size_t)inst * st.lcap : st.vlist + (size_t)inst * st.lcap;
    int* const out_list = keep ? st.slist + (size_t)inst * st.lcap : st.vlist + (size_t)inst * st.lcap;
    const Float* const cost = (const Float*)pb.cost;

    Float kappa = (Float)a.kappa[inst]; // (moves behind a sweep of a multi-sweep launch, bxh_run)
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
    Prof prof;
    (void)prof;
#ifdef BXH_PROFILE
    for (int i = 0; i < 16; ++i)
        prof.acc[i] = 0;
    prof.t = wall_clock64();
#endif
    int nrows = use_m ? st.mcount[inst] : st.vcount[inst];
    double alg_w = 0.0; // algorithmic bytes of the rows this wave took (lane 0 counts)
    // Several sweeps of solver_functor's loop (itm-solver-common.hpp:135-213) in one launch, as in
    // the short-row kernel (bxh_narrow.hpp: NarrowRun): behind every sweep kappa moves as the host
    // would move it, the launch ends when no row is violated, when kappa passes kappa_max or after
    // max_sweeps; pi / x / R's leftovers stay in LDS in between, the row list is the one the sweep
    // left in memory.  run.max_sweeps == 0: one sweep, nothing of this (bxh_sweep).
    int cnt = 0, sweeps_done = 0, stop_why = 0, best_rem = 0x7fffffff, best_i = 0;
    unsigned long long rows_total = 0;
    unsigned rows_last = 0;
    for (int sw = 0;; ++sw) {

    if (tid == 0) {
        pi_s[pb.m] = 0;
        misc_s[MS_PICH] = 0;
        if (ord == 2 || ord == 7) // compute_order: random_sorting shuffles the list first
            rng.shuffle(vlist, (unsigned)nrows);
    }
    __syncthreads();

    auto row_id = [&](int it_) -> int { return (ord == 1) ? vlist[nrows - 1 - it_] : vlist[it_]; };
    int* const zflag = misc_s + MS_WAVES; // [0..3] tie, [4..7] selection left to thread 0, [8..11] sel
    auto w_val = [&](int w) { return (Float*)(zw_base + w * zw_wave_bytes((int)sizeof(Float))); }; // by element
    auto w_vs = [&](int w) { return w_val(w) + 64; };                                             // by position
    auto w_key = [&](int w) { return (unsigned long long*)(w_val(w) + 128); };                    // by position
    auto w_idp = [&](int w) { return (int*)(w_key(w) + 64); };                                    // element by position
    auto w_pos = [&](int w) { return w_idp(w) + 128; };                                           // position by element
    auto stops = [&](Float v) -> bool { return maximize ? v < 0 : v > 0; }; // (a zero flips a coin)

    // a row of the NEXT table, fetched a table ahead; the lane's element of its row in the NEXT group
    int4 nf_a = make_int4(0, 0, 0, 0);
    int nf_bmin = 0, nf_bmax = 0, nf_pos = 0x7fffffff;
    unsigned long long nf_conf = 0ull;
    double nf_rb = 0.0;
    int zp_pos = -1;
    int4 zp_d0 = make_int4(0, 0, 0, 0), zp_d1 = make_int4(0, 0, 0, 0);
    auto fetch_row = [&](int pos, int4& ra, int& bmin_, int& bmax_, unsigned long long& conf_, double& rb_) {
        const int k_ = row_id(pos);
        ra = make_int4(k_, pb.row_ptr[k_], pb.row_ptr[k_ + 1], PM1 ? pb.bneg[k_] : 0);
        bmin_ = pb.bmin[k_];
        bmax_ = pb.bmax[k_];
        conf_ = pb.zconf[k_];
        rb_ = pb.row_bytes[k_];
    };

    int tbase = -0x40000000; // list position of table slot 0
    BXH_PSTAMP(&prof, 0);
    for (int it = 0; it < nrows;) {
        // ---- the table of the next 64 rows of the list, and their groups -------------
        if (it + 2 * NW > tbase + ZW_TABLE && (tbase < 0 || tbase + ZW_TABLE < nrows)) {
            __syncthreads(); // (nobody reads the old table any more)
            const bool have = tbase >= 0;
            tbase = it;
            if (have) {
                const int sl_ = nf_pos - it;
                if (sl_ >= 0 && sl_ < ZW_TABLE && nf_pos < nrows) {
                    t_a[sl_] = nf_a;
                    t_b[sl_] = make_int4(nf_bmin, nf_bmax, 0, nf_a.z - nf_a.y);
                    t_conf[sl_] = nf_conf;
                    t_rb[sl_] = nf_rb;
                }
            } else if (tid < ZW_TABLE && it + tid < nrows) {
                int4 ra;
                int b0, b1;
                unsigned long long cf;
                double rb;
                fetch_row(it + tid, ra, b0, b1, cf, rb);
                t_a[tid] = ra;
                t_b[tid] = make_int4(b0, b1, 0, ra.z - ra.y);
                t_conf[tid] = cf;
                t_rb[tid] = rb;
            }
            __syncthreads();
            if (tid < ZW_TABLE) { // (wave 0: the 64 rows' bits in four ballots)
                int cb = 0;
                if (it + tid < nrows) {
                    const int rr = t_b[tid].w;
                    if (rr >= 1 && rr <= 63)
                        cb = 0x100;
                    const int k2 = t_a[tid].x;
                    for (int u = 1; u <= 3 && u <= tid; ++u) {
                        const int k1 = t_a[tid - u].x;
                        const int lo = k1 < k2 ? tid - u : tid;
                        const int d = k1 < k2 ? k2 - k1 : k1 - k2;
                        if (d == 0 || d > 64 || ((t_conf[lo] >> (d - 1)) & 1ull))
                            cb |= 1 << (u - 1);
                    }
                }
                const unsigned long long smask = __ballot((cb & 0x100) != 0);
                const unsigned long long c1 = __ballot((cb & 1) != 0), c2 = __ballot((cb & 2) != 0),
                                         c3 = __ballot((cb & 4) != 0);
                int gs = 0; // the group that WOULD start at this slot
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
                t_b[tid].z = gs;
            }
            nf_pos = it + 48 + tid; // the rows of the NEXT table: on their way from now on
            if (tid < 80 && nf_pos < nrows)
                fetch_row(nf_pos, nf_a, nf_bmin, nf_bmax, nf_conf, nf_rb);
            __syncthreads();
        }
        const int s0 = it - tbase;
        const int g = max(1, __builtin_amdgcn_readfirstlane(t_b[s0].z)); // (bxh_create: every row has 1 .. 63 elements)
        const bool mine = wave < g;
        const int sl = s0 + (mine ? wave : 0);
        const int4 ta = t_a[sl], tb = t_b[sl];
        const int k = __builtin_amdgcn_readfirstlane(ta.x);
        const int rs = __builtin_amdgcn_readfirstlane(ta.y);
        const int r = mine ? __builtin_amdgcn_readfirstlane(tb.w) : 0;
        const int c_size = __builtin_amdgcn_readfirstlane(ta.w);
        const int row_bmin = __builtin_amdgcn_readfirstlane(tb.x);
        const int row_bmax = __builtin_amdgcn_readfirstlane(tb.y);
        int rg[ZW_WAVES]; // the lengths of the group's rows (R's leftovers go by them)
#pragma unroll
        for (int u = 0; u < ZW_WAVES; ++u) {
            const int ru = t_b[s0 + (u < g ? u : 0)].w;
            rg[u] = u < g ? __builtin_amdgcn_readfirstlane(ru) : 0;
        }
        Float* const val_w = w_val(wave);
        Float* const vs_w = w_vs(wave);
        unsigned long long* const key_w = w_key(wave);
        int* const idp_w = w_idp(wave);
        int* const pos_w = w_pos(wave);

        // ---- phase A: the lane's column -> reduced cost -> position -------------------
        BXH_PSTAMP(&prof, 1);
        // Only as much of the columns as the row's longest one holds is fetched: `nq` quads of
        // four slots, wave-uniform (the slots behind a column's last entry are padding, P = +0.0
        // and the row m whose pi is +0.0: leaving their additions out changes no bit of a sum
        // that started at +0.0).  The rows of the slots come as 16-bit ids (bit 15: coefficient -1).
        Float c_p[16];
        uint4 c_id[2] = { make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u) };
        int4 c_desc = make_int4(0, 0, 0, 0);
        Float c_cost = 0, p_own_raw = 0, c_push = 0;
        int fe = 1, nq = 1;
        {
            int4 d1 = make_int4(1, 0, 0, 0);
            if (lane < r) {
                if (zp_pos == it + wave) { // fetched while the previous group was worked on
                    c_desc = zp_d0;
                    d1 = zp_d1;
                } else {
                    const int e = rs + lane;
                    c_desc = pb.zdesc[2 * e];
                    d1 = pb.zdesc[2 * e + 1];
                }
            }
            const int clen = lane < r ? d1.w : 0; // real entries of the lane's column
            nq = __ballot(clen > 12) ? 4 : (__ballot(clen > 8) ? 3 : (__ballot(clen > 4) ? 2 : 1));
            const bool far_own = __ballot(lane < r && c_desc.z - c_desc.x >= 16) != 0ull;
            if (lane < r) {
                fe = d1.x;
                if (sizeof(Float) == 8)
                    c_cost = (Float)__hiloint2double(d1.z, d1.y);
                else
                    c_cost = (Float)__int_as_float(d1.y);
                const int q0 = c_desc.x;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    if (qd < nq) {
                        if (sizeof(Float) == 8) {
                            const double2* pv = reinterpret_cast<const double2*>((const double*)P + q0);
                            const double2 t0 = pv[2 * qd], t1 = pv[2 * qd + 1];
                            c_p[4 * qd] = (Float)t0.x;
                            c_p[4 * qd + 1] = (Float)t0.y;
                            c_p[4 * qd + 2] = (Float)t1.x;
                            c_p[4 * qd + 3] = (Float)t1.y;
                        } else {
                            const float4 t4 = reinterpret_cast<const float4*>((const float*)P + q0)[qd];
                            c_p[4 * qd] = (Float)t4.x;
                            c_p[4 * qd + 1] = (Float)t4.y;
                            c_p[4 * qd + 2] = (Float)t4.z;
                            c_p[4 * qd + 3] = (Float)t4.w;
                        }
                    }
                const uint4* idv = reinterpret_cast<const uint4*>(pb.col_row16 + q0);
                c_id[0] = idv[0];
                if (nq > 2)
                    c_id[1] = idv[1];
                if (far_own) // (an own entry among the first 16 is taken out of the line below)
                    p_own_raw = P[c_desc.z];
                if (push && !PM1) // the 01 solvers amplify by the cost of variable number <local id> (quirk)
                    c_push = cost[lane];
            }
        }
        { // the next group's descriptors: on their way while this one is worked on
            const int pos2 = it + g;
            zp_pos = -1;
            if (pos2 < nrows && pos2 + NW <= tbase + ZW_TABLE) {
                const int g2 = __builtin_amdgcn_readfirstlane(t_b[pos2 - tbase].z);
                if (wave < g2) {
                    const int4 ta2 = t_a[pos2 - tbase + wave];
                    const int e2 = ta2.y + lane;
                    zp_pos = pos2 + wave;
                    if (e2 < ta2.z) {
                        zp_d0 = pb.zdesc[2 * e2];
                        zp_d1 = pb.zdesc[2 * e2 + 1];
                    }
                }
            }
        }
        BXH_PSTAMP(&prof, 2);
        Float val = 0, p_decayed = 0;
        unsigned long long key = ~0ull;
        int rank = 0, sel = -1;
        bool tie = false, serial = false;
        if (mine) {
            if (lane < r) {
                // decrease_preference + compute_reduced_costs: two sums in CSC order, the row's own
                // entry decayed (it reaches memory with affect's value)
                const int off = c_desc.z - c_desc.x;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    if (qd < nq) {
#pragma unroll
                        for (int u = 4 * qd; u < 4 * qd + 4; ++u)
                            p_own_raw = u == off ? c_p[u] : p_own_raw;
                    }
                p_decayed = p_own_raw * theta;
                Float sum_a_pi = 0, sum_a_p = 0;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    if (qd < nq) { // (padding slots: row m, whose pi is +0.0, and P = +0.0)
                        const uint4 iw = c_id[qd >> 1];
                        const unsigned w0 = (qd & 1) ? iw.z : iw.x, w1 = (qd & 1) ? iw.w : iw.y;
                        const unsigned h4[4] = { w0 & 0xffffu, w0 >> 16, w1 & 0xffffu, w1 >> 16 };
                        Float pv4[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            pv4[t] = pi_s[h4[t] & 0x7fffu];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int u = 4 * qd + t;
                            const Float pp = u == off ? p_decayed : c_p[u];
                            if (PM1) {
                                const Float fa = (h4[t] & 0x8000u) ? (Float)-1 : (Float)1;
                                sum_a_pi += fa * pv4[t];
                                sum_a_p += fa * pp;
                            } else {
                                sum_a_pi += pv4[t];
                                sum_a_p += pp;
                            }
                        }
                    }
                for (int qb = c_desc.x + 16; qb < c_desc.x + c_desc.y; qb += 8) { // longer columns
                    Float p8[8], pi8[8];
                    int2 rf8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        p8[u] = P[qb + u];
                        rf8[u] = pb.zcolrf[qb + u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        pi8[u] = pi_s[rf8[u].x];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const Float pp = qb + u == c_desc.z ? p_decayed : p8[u];
                        if (PM1) {
                            sum_a_pi += (Float)rf8[u].y * pi8[u];
                            sum_a_p += (Float)rf8[u].y * pp;
                        } else {
                            sum_a_pi += pi8[u];
                            sum_a_p += pp;
                        }
                    }
                }
                val = c_cost - sum_a_pi - sum_a_p;
                if (PM1 && fe < 0)
                    val = -val;
                if (push)
                    val += obj_amp * (PM1 ? c_cost : c_push);
                key = order_key<Float>(val, maximize);
            }
            BXH_PSTAMP(&prof, 3);
            // calculator_sort: the element's position = the number of keys below its own; two equal
            // keys collide on a position: the row counts again with the ids and flags the tie
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
            }
            __builtin_amdgcn_wave_barrier();
            if (tie && lane < r)
                key_w[rank] = key; // by position: the runs the shuffle goes over

            // select_variables.  Its answer depends on the VALUES by position only (a shuffle
            // moves elements inside a run of equal values), so it is taken here -- unless the scan
            // of an inequality row meets a zero (a coin from the engine) or runs on to R[r] (what
            // an earlier row left there) before a value stops it: thread 0 then takes it, in turn.
            if (kind == 0 || kind == 2 || row_bmin == row_bmax) {
                sel = min(row_bmin + c_size, r) - 1;
            } else {
                int lo, hi;
                if (kind == 1) {
                    lo = min(row_bmin, r);
                    hi = min(row_bmax, r);
                } else {
                    lo = row_bmin + c_size;
                    hi = min(row_bmax + c_size, r);
                }
                sel = hi - 1;
                const Float vi = lane < r ? vs_w[lane] : (Float)0; // position i = lane
                const bool in = lane >= lo && lane <= hi && lane < r;
                const unsigned long long zeros = __ballot(in && vi == 0);
                const unsigned long long halt = __ballot(in && vi != 0 && stops(vi));
                const unsigned long long first = zeros | halt;
                if (first) {
                    const int i0 = __ffsll((long long)first) - 1;
                    if ((zeros >> i0) & 1ull)
                        serial = true; // a coin decides
                    else
                        sel = i0 - 1;
                } else if (hi >= r && lo <= hi)
                    serial = true; // the scan reaches R[r]
            }
            if (lane == 0) {
                zflag[wave] = tie ? 1 : 0;
                zflag[4 + wave] = serial ? 1 : 0;
                zflag[8 + wave] = sel;
            }
        }
        BXH_PSTAMP(&prof, 4);
        __syncthreads(); // B1: every row of the group is sorted
        BXH_PSTAMP(&prof, 5);

        // ---- what the reference does in sequence: shuffles and coins, row after row -----
        bool any_seq = false;
        for (int w = 0; w < g; ++w)
            any_seq = any_seq || zflag[w] != 0 || zflag[4 + w] != 0;
        if (any_seq) {
            if (tid == 0)
                for (int w = 0; w < g; ++w) {
                    const int rw = rg[w];
                    if (zflag[w]) {
                        const unsigned long long* kw = w_key(w);
                        int* iw = w_idp(w);
                        int b0 = 0;
                        for (int i = 1; i <= rw; ++i)
                            if (i == rw || kw[i] != kw[b0]) {
                                rng.shuffle(iw + b0, (unsigned)(i - b0));
                                b0 = i;
                            }
                    }
                    if (zflag[4 + w]) {
                        const int4 wa = t_a[s0 + w], wb = t_b[s0 + w];
                        const Float* vw = w_vs(w);
                        auto Rv = [&](int i) -> Float { // R[i].value; R[r] is what the rows in front left
                            if (i < rw)
                                return vw[i];
                            for (int w2 = w - 1; w2 >= 0; --w2)
                                if (rg[w2] > i)
                                    return w_vs(w2)[i];
                            return stale_s[i <= pb.rmax ? i : pb.rmax];
                        };
                        int lo, hi;
                        if (kind == 1) {
                            lo = min(wb.x, rw);
                            hi = min(wb.y, rw);
                        } else {
                            lo = wb.x + wa.w;
                            hi = min(wb.y + wa.w, rw);
                        }
                        int s = hi - 1;
                        for (int i = lo; i <= hi; ++i) {
                            const Float v = Rv(i);
                            if (v == 0 ? rng.coin() : stops(v)) {
                                s = i - 1;
                                break;
                            }
                        }
                        zflag[8 + w] = s;
                    }
                }
            __syncthreads();
            if (mine) {
                if (tie) { // the new positions (values of a run are equal: vs stays)
                    if (lane < r)
                        pos_w[idp_w[lane]] = lane;
                    __builtin_amdgcn_wave_barrier();
                    if (lane < r)
                        rank = pos_w[lane];
                }
                if (serial)
                    sel = zflag[8 + wave];
            }
        }

        BXH_PSTAMP(&prof, 6);
        // ---- affect (itm-common.hpp:382-467): pi_k and d, then every variable -------------
        if (mine) {
            const Float one = 1, two = 2;
            const Float middle = (two + one) / two;
            Float d = delta;
            if (sel < 0) {
                d += (kappa / (one - kappa)) * (vs_w[0] / two);
            } else if (sel + 1 >= r) {
                d += (kappa / (one - kappa)) * (vs_w[sel] * middle);
            } else {
                const Float lo_v = vs_w[sel], hi_v = vs_w[sel + 1];
                const Float old_pi = pi_s[k];
                const Float new_pi = old_pi + ((lo_v + hi_v) / two);
                d += (kappa / (one - kappa)) * (hi_v - lo_v);
                if (lane == 0) {
                    pi_s[k] = new_pi;
                    if (signbit(old_pi) != signbit(new_pi))
                        atomicOr(&misc_s[MS_PICH], 1);
                }
            }
            if (lane < r) {
                const int j = c_desc.w;
                bool up = rank <= sel; // sel < 0: none; sel + 1 >= r: all
                if (PM1 && fe < 0)
                    up = !up;
                if (up)
                    atomicOr(&x_s[j >> 5], 1u << (j & 31));
                else
                    atomicAnd(&x_s[j >> 5], ~(1u << (j & 31)));
                P[c_desc.z] = up ? p_decayed + d : p_decayed - d;
            }
        }
        BXH_PSTAMP(&prof, 7);
        __syncthreads(); // B2: the group is done (P in memory, pi / x in LDS)
        BXH_PSTAMP(&prof, 8);
        // R as the reference leaves it: the rows of the group wrote it in turn, so a position
        // keeps the value of the LAST row long enough to reach it
        if (mine) {
            int later = 0;
#pragma unroll
            for (int w = 1; w < ZW_WAVES; ++w)
                later = max(later, w > wave ? rg[w] : 0);
            if (lane < r && lane >= later)
                stale_s[lane] = vs_w[lane];
            alg_w += t_rb[sl];
        }
        it += g;
    }
    BXH_PSTAMP(&prof, 9);

    __syncthreads();
    cnt = rebuild_violated(pb, x_s, st.vflag + (size_t)inst * pb.m, out_list, misc_s + MS_WAVES + 16);
    // ---- solver_functor's bookkeeping behind a sweep (every thread alike) ------------
    rows_total += (unsigned long long)nrows;
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
    nrows = cnt; // (out_list == vlist in the orders a multi-sweep launch takes: the next sweep's list)
    __syncthreads();
    } // sweeps of this launch

    __syncthreads();
    for (int i = tid; i < pb.m; i += T)
        pig[i] = pi_s[i];
    for (int i = tid; i < 2 * pb.nwords; i += T)
        xg[i] = x_s[i];
    for (int i = tid; i <= pb.rmax; i += T)
        staleg[i] = stale_s[i];
    if (run.max_sweeps > 0 && tid == 0) {
        run.out[inst * 4 + 0] = sweeps_done;
        run.out[inst * 4 + 1] = best_rem;
        run.out[inst * 4 + 2] = best_i;
        run.out[inst * 4 + 3] = stop_why;
        run.kappa_out[inst] = (double)kappa;
    }
    if (lane == 0)
        t_rb[wave] = alg_w;
    __syncthreads();
    if (tid == 0) {
        double alg_bytes = 0.0;
        for (int w = 0; w < NW; ++w)
            alg_bytes += t_rb[w];
        if (!keep)
            st.vcount[inst] = cnt;
        DevStatus s = st.status[inst];
        s.remaining = cnt;
        s.pi_changed = misc_s[MS_PICH];
        s.rng = rng.s;
        s.rows_in_sweep = rows_last;
        s.row_updates += rows_total;
        s.alg_bytes += alg_bytes;
        st.status[inst] = s;
    }
#ifdef BXH_PROFILE
    if (tid == 0)
        for (int i = 0; i < 16; ++i)
            st.prof[(size_t)inst * 32 + i] += prof.acc[i];
#endif
}

} // namespace bxh

#endif
