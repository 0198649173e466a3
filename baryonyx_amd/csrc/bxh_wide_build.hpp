// baryonyx_amd/csrc/bxh_wide_build.hpp -- the constraint matrix of the wide
// kernel built ON THE DEVICE from a device-resident CSR (bxh_create_device):
// what `sparse_matrix<int>`'s constructor does on the host
// (lib/src/sparse-matrix.hpp:86-186: element ids in constraint order, the
// column view by a stable sort on the column) and what the Solver constructors
// do with the bounds (itm-solver-inequalities-01.cpp:88-106, ...), for matrices
// of 2e9 elements that no host loop should touch.
//
// Stable order inside a column (rows ascending) without a sort: the rows are cut
// into NB blocks; cnt[b][j] = entries of column j in block b (atomics), a scan
// over the blocks of every column turns the counts into first slots, and one
// workgroup per block then walks ITS rows in order and hands out the slots.
#pragma once

#include <hip/hip_runtime.h>

namespace bxh {
namespace build {

// wave per row: length (-> rmax), coefficient / column validation
__global__ __launch_bounds__(256) void
row_stats_kernel(const int* row_ptr, const int* col, const int* factor, int m, int n, int pm1,
                 int* rmax, int* err)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    int mx = 0;
    bool bad = false, unsorted = false;
    for (int k = wave; k < m; k += nwaves) {
        const int rs = row_ptr[k], re = row_ptr[k + 1];
        if (re <= rs)
            bad = true;
        mx = max(mx, re - rs);
        for (int e = rs + lane; e < re; e += 64) {
            const int j = col[e];
            if (j < 0 || j >= n)
                bad = true;
            if (e + 1 < re && col[e + 1] <= j)
                unsorted = true; // (not an error: the band scan of the violated rows needs ascending columns)
            const int f = factor ? factor[e] : 1;
            if (pm1 ? (f != 1 && f != -1) : (f != 1))
                bad = true;
        }
    }
    if (lane == 0)
        atomicMax(rmax, mx);
    if (bad)
        atomicOr(err, 1);
    if (unsorted)
        atomicOr(err, 16);
}

// cnt[b][j] += 1 for every element of block b's rows
__global__ __launch_bounds__(1024) void
count_kernel(const int* row_ptr, const int* col, int m, int n, int rpb, unsigned* cnt)
{
    const int b = blockIdx.x;
    const int k0 = b * rpb, k1 = min(m, k0 + rpb);
    if (k0 >= k1)
        return;
    const int e0 = row_ptr[k0], e1 = row_ptr[k1];
    unsigned* c = cnt + (size_t)b * n;
    for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x)
        atomicAdd(&c[col[e]], 1u);
}

// thread per column: its length, over all blocks
__global__ void
col_len_kernel(const unsigned* cnt, int nb, int n, int* col_len)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    unsigned s = 0;
    for (int b = 0; b < nb; ++b)
        s += cnt[(size_t)b * n + j];
    col_len[j] = (int)s;
}

// thread per column: counts -> first slot of the column's entries of block b
__global__ void
bases_kernel(unsigned* cnt, int nb, int n, const int2* col_desc)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    unsigned run = (unsigned)col_desc[j].x;
    for (int b = 0; b < nb; ++b) {
        const unsigned t = cnt[(size_t)b * n + j];
        cnt[(size_t)b * n + j] = run;
        run += t;
    }
}

__global__ void
fill_int_kernel(int* p, size_t count, int v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        p[i] = v;
}

// one workgroup per row block, its rows in order: slot of every element
__global__ __launch_bounds__(1024) void
fill_kernel(const int* row_ptr, const int* col, const int* factor, int m, int n, int rpb,
            unsigned* base, int* col_row, int* row_pos, signed char* fac_csr)
{
    const int b = blockIdx.x;
    const int k0 = b * rpb, k1 = min(m, k0 + rpb);
    unsigned* bs = base + (size_t)b * n;
    for (int k = k0; k < k1; ++k) {
        const int rs = row_ptr[k], re = row_ptr[k + 1];
        for (int e = rs + threadIdx.x; e < re; e += blockDim.x) {
            const int j = col[e];
            const unsigned q = atomicAdd(&bs[j], 1u); // (memory-side: seen by the next row)
            const int f = factor ? factor[e] : 1;
            col_row[q] = f < 0 ? (k | 0x40000000) : k;
            row_pos[e] = (int)q;
            if (fac_csr)
                fac_csr[e] = (signed char)f;
        }
        __syncthreads();
    }
}

// wave per row: clamped bounds as each Solver constructor computes them, and
// the algorithmic bytes of one update of the row
__global__ __launch_bounds__(256) void
bounds_kernel(const int* row_ptr, const int* col, const int* factor, const int* cmin,
              const int* cmax, const int* col_len, int m, int kind, double fsize, int* bmin,
              int* bmax, int* bneg, double* row_bytes, int* err)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int k = wave; k < m; k += nwaves) {
        const int rs = row_ptr[k], re = row_ptr[k + 1];
        int lower = 0;
        double cb = 0.0;
        for (int e = rs + lane; e < re; e += 64) {
            if (factor && factor[e] < 0)
                ++lower;
            cb += (double)col_len[col[e]];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lower += __shfl_xor(lower, o, 64);
            cb += __shfl_xor(cb, o, 64);
        }
        if (lane == 0) {
            const int len = re - rs, upper = len - lower;
            const int lo = cmin[k], hi = cmax[k];
            int a = lo, c = hi, ng = 0;
            switch (kind) {
            case 0: // equalities-01
                if (lo != hi)
                    atomicOr(err, 2);
                break;
            case 1: // inequalities-01
                if (lo != hi) {
                    a = max(0, lo);
                    c = min(len, hi);
                }
                break;
            case 2: // equalities-101
                if (lo != hi)
                    atomicOr(err, 2);
                ng = lower;
                break;
            default: // inequalities-101
                if (lo != hi) {
                    a = max(-lower, lo);
                    c = min(upper, hi);
                }
                ng = lower;
                break;
            }
            if (a > c)
                atomicOr(err, 4);
            bmin[k] = a;
            bmax[k] = c;
            bneg[k] = ng;
            const double r = (double)len;
            row_bytes[k] = cb * (4.0 + 2.0 * fsize) + r * (4.0 + 5.0 * fsize) + 2.0 * fsize +
                           (double)((len + 7) / 8) * 2.0;
        }
    }
}

template<typename Float>
__global__ void
cost_kernel(const double* in, Float* out, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n)
        out[j] = (Float)in[j];
}

// P between the caller's CSR element order and the device's padded CSC order,
// a chunk of elements [e0, e0 + count) at a time
template<typename Float>
__global__ void
gather_P_kernel(const Float* P, const int* row_pos, size_t e0, size_t count, double* out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        out[i] = (double)P[row_pos[e0 + i]];
}
template<typename Float>
__global__ void
scatter_P_kernel(Float* P, const int* row_pos, size_t e0, size_t count, const double* in)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        P[row_pos[e0 + i]] = (Float)in[i];
}

} // namespace build
} // namespace bxh
