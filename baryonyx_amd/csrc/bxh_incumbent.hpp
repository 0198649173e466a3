// baryonyx_amd/csrc/bxh_incumbent.hpp -- the incumbent of a batch, found on the
// device (bxh_incumbent, include/bxh.h): which instance holds the point
// best_solution_recorder would keep (fewest violated rows, then the better
// objective: `storage::sort`'s order, lib/src/itm-optimizer-common.hpp:424-442,
// with raw_result_compare, lib/src/result.hpp:96-107), as one 32-byte record and the winner's
// x, both left in device memory for the exchange step over RCCL
// (baryonyx_amd/dist.py) -- no per-instance read-back, no host loop.
#ifndef BXH_INCUMBENT_HPP
#define BXH_INCUMBENT_HPP

#include "bxh_kernels.hpp"

namespace bxh {

// The exchange key.  Two non-negative 64-bit words compared lexicographically:
//   hi = violated rows (19 bits, clipped) << 44 | top 44 bits of the objective's ordered image
//   lo = low 20 bits of that image << 6 | rank (6 bits)
// (the ordered image of a double: smaller value <-> smaller unsigned word;
// complemented when maximising, so that MIN always names the best).  One
// all-reduce(MIN) of `hi` settles every comparison that differs in the count or
// in the first 44 bits of the objective; a second one of `lo`, among the ranks
// that tie there, makes the comparison exact in all 64 bits and picks the lowest
// rank among equals.
struct IncumbentRecord
{
    long long key_hi;
    long long key_lo;
    double value;  // the objective as summed here (see incumbent_value_kernel)
    int remaining; // violated rows of that instance
    int instance;
};
static_assert(sizeof(IncumbentRecord) == 32, "ABI (include/bxh.h: bxh_incumbent_record)");

constexpr int kIncRemainingBits = 19;
constexpr int kIncRankBits = 6;

__host__ __device__ inline unsigned long long
incumbent_ordered_bits(double value, bool maximize)
{
    unsigned long long b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = static_cast<unsigned long long>(__double_as_longlong(value));
#else
    static_assert(sizeof(double) == 8, "");
    __builtin_memcpy(&b, &value, 8);
#endif
    b = (b >> 63) ? ~b : (b | (1ull << 63));
    return maximize ? ~b : b;
}

__host__ __device__ inline void
incumbent_pack(int remaining, double value, int rank, bool maximize, long long& hi, long long& lo)
{
    const long long rmax = (1ll << kIncRemainingBits) - 1;
    const long long r = remaining < 0 ? 0 : (remaining > rmax ? rmax : remaining);
    const unsigned long long o = incumbent_ordered_bits(value, maximize);
    hi = (r << 44) | static_cast<long long>(o >> 20);
    lo = static_cast<long long>((o & 0xFFFFFull) << kIncRankBits) | (rank & ((1 << kIncRankBits) - 1));
}

// value[b] = constant + sum_j cost[j] x_bj: Cost::results (itm-common.hpp:1140-1147,
// the linear part) on the ORIGINAL costs.  One workgroup of 256 threads per
// instance; thread t adds the set bits of the 32-bit words t, t + 256, ... in
// rising j, the 256 partial sums are added in a fixed tree: deterministic, and
// equal to the reference's sequential sum whenever every partial sum is exact
// (integer costs).  With real costs the last bits may differ from the
// sequential sum: the value orders the instances, the caller recomputes the
// winner's objective from its x where it needs the exact figure.
__global__ void
__launch_bounds__(256) incumbent_value_kernel(const unsigned* __restrict__ x,
                                              int nwords,
                                              int n,
                                              const double* __restrict__ cost,
                                              double constant,
                                              double* __restrict__ value)
{
    __shared__ double part[256];
    const int b = blockIdx.x, t = threadIdx.x;
    const unsigned* xb = x + static_cast<size_t>(b) * 2 * nwords;
    const int words32 = (n + 31) / 32;
    double s = 0.0;
    for (int w = t; w < words32; w += 256) {
        unsigned bits = xb[w];
        while (bits) {
            const int k = __ffs(static_cast<int>(bits)) - 1;
            bits &= bits - 1;
            const int j = w * 32 + k;
            if (j < n)
                s += cost[j];
        }
    }
    part[t] = s;
    __syncthreads();
    for (int half = 128; half > 0; half >>= 1) {
        if (t < half)
            part[t] = part[t] + part[t + half];
        __syncthreads();
    }
    if (t == 0)
        value[b] = constant + part[0];
}

// The best instance of the batch by (violated rows, objective, instance id), its
// record and its x.  One workgroup of 256 threads.  An instance with violated
// rows has no objective to compare: it stands with the worst value (+inf, -inf
// when maximising), as a worker without a solution does in the recorder.
__global__ void
__launch_bounds__(256) incumbent_pick_kernel(const DevStatus* __restrict__ status,
                                             const double* __restrict__ value,
                                             const unsigned* __restrict__ x,
                                             int batch,
                                             int nwords,
                                             int maximize,
                                             int rank,
                                             IncumbentRecord* __restrict__ out,
                                             unsigned* __restrict__ out_x)
{
    __shared__ unsigned long long s_hi[256], s_lo[256];
    __shared__ int s_b[256];
    const int t = threadIdx.x;
    unsigned long long hi = ~0ull, lo = ~0ull;
    int best = -1;
    for (int b = t; b < batch; b += 256) {
        const int rem = status[b].remaining;
        const double worst = maximize ? -__builtin_huge_val() : __builtin_huge_val();
        const double v = rem == 0 ? value[b] : worst;
        const unsigned long long h =
          (static_cast<unsigned long long>(rem < 0 ? 0 : rem) << 32); // (exact count here: no clipping)
        const unsigned long long l = incumbent_ordered_bits(v, maximize != 0);
        // lexicographic (count, objective image, instance); b rises within a thread
        if (best < 0 || h < hi || (h == hi && l < lo)) {
            hi = h;
            lo = l;
            best = b;
        }
    }
    s_hi[t] = hi;
    s_lo[t] = lo;
    s_b[t] = best;
    __syncthreads();
    for (int half = 128; half > 0; half >>= 1) {
        if (t < half) {
            const int o = t + half;
            const bool take =
              s_b[o] >= 0 && (s_b[t] < 0 || s_hi[o] < s_hi[t] ||
                              (s_hi[o] == s_hi[t] && (s_lo[o] < s_lo[t] || (s_lo[o] == s_lo[t] && s_b[o] < s_b[t]))));
            if (take) {
                s_hi[t] = s_hi[o];
                s_lo[t] = s_lo[o];
                s_b[t] = s_b[o];
            }
        }
        __syncthreads();
    }
    const int w = s_b[0];
    if (t == 0) {
        IncumbentRecord r;
        r.instance = w;
        r.remaining = w >= 0 ? status[w].remaining : 0x7fffffff;
        const double worst = maximize ? -__builtin_huge_val() : __builtin_huge_val();
        r.value = (w >= 0 && r.remaining == 0) ? value[w] : worst;
        incumbent_pack(r.remaining, r.value, rank, maximize != 0, r.key_hi, r.key_lo);
        *out = r;
    }
    if (w >= 0 && out_x)
        for (int i = t; i < 2 * nwords; i += 256)
            out_x[i] = x[static_cast<size_t>(w) * 2 * nwords + i];
}

} // namespace bxh

#endif
