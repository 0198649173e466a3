// baryonyx_amd/csrc/bxh_device.hip -- host half of the C ABI in include/bxh.h:
// builds the device-resident constraint matrix, owns the per-instance solver
// state, launches the gfx950 kernels of bxh_kernels.hpp.
//
// There is deliberately NO CPU implementation behind these entry points: if
// HIP cannot run them they return BXH_ERR_DEVICE.
#include "../../include/bxh.h"
#include "bxh_kernels.hpp"
#include "bxh_wide.hpp"
#include "bxh_narrow.hpp"
#include "bxh_swave.hpp"
#include "bxh_wide_build.hpp"
#include "bxh_incumbent.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_error;

int
fail(int code, const std::string& msg)
{
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                          \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess)                                                  \
            return fail(e_ == hipErrorOutOfMemory ? BXH_ERR_OOM                \
                                                  : BXH_ERR_DEVICE,            \
                        std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

template<typename T>
struct DevBuf
{
    T* p = nullptr;
    size_t count = 0;
    ~DevBuf() { release(); }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        count = 0;
    }
    // Every buffer starts as zeros (hipMalloc hands out whatever the last owner
    // left there; a scratch array read before its first write must not depend on
    // that).  BXH_POISON=1 fills with 0xA5 instead: a diagnostic for such reads.
    hipError_t alloc(size_t n)
    {
        release();
        count = n;
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), bytes);
        if (e != hipSuccess)
            return e;
        static const bool poison = std::getenv("BXH_POISON") != nullptr;
        static const bool raw = std::getenv("BXH_NO_ZERO") != nullptr; // (diagnostic)
        if (raw)
            return hipSuccess;
        // (hipMemset on device memory is asynchronous, on the null stream; the
        // handle's work runs on a non-blocking stream of its own, which does not
        // wait for it: the fill must be over before anything else touches the buffer)
        e = hipMemset(p, poison ? 0xA5 : 0, bytes);
        if (e != hipSuccess)
            return e;
        return hipStreamSynchronize(nullptr);
    }
    hipError_t upload(const std::vector<T>& v)
    {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty())
            return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel on a
// device, not of a handle: handles of different shapes live side by side
// (device_pool, the tests), so every kernel that may ask for more than the
// 64 KiB default is raised to the LDS of a gfx950 CU once per device instead of
// to each handle's own plan.
hipError_t
set_lds_limits(int device)
{
    static std::mutex mu;
    static std::vector<char> done;
    std::lock_guard<std::mutex> lock(mu);
    if (device >= 0 && static_cast<size_t>(device) < done.size() && done[device])
        return hipSuccess;
    const int limit = 160 * 1024;
    // (a kernel's static LDS counts against the same 160 KiB.)  A refused call is
    // not fatal here: the kernel keeps the 64 KiB default, which is all a small
    // handle asks for; a handle whose plan needs more then fails at its launch,
    // loudly (BXH_ERR_DEVICE), instead of every create on the device failing.
    auto raise = [&](const void* fn) {
        hipFuncAttributes fa;
        hipError_t e = hipFuncGetAttributes(&fa, fn);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    limit - static_cast<int>(fa.sharedSizeBytes));
        if (e != hipSuccess)
            (void)hipGetLastError();
    };
#define BXH_SET_LDS(...) raise(reinterpret_cast<const void*>(&__VA_ARGS__))
    BXH_SET_LDS(bxh::sweep_kernel<float, false>);
    BXH_SET_LDS(bxh::sweep_kernel<float, true>);
    BXH_SET_LDS(bxh::sweep_kernel<double, false>);
    BXH_SET_LDS(bxh::sweep_kernel<double, true>);
    BXH_SET_LDS(bxh::sweep_kernel_z<float, false>);
    BXH_SET_LDS(bxh::sweep_kernel_z<double, false>);
    BXH_SET_LDS(bxh::sweep_kernel_sw<float, false>);
    BXH_SET_LDS(bxh::sweep_kernel_sw<float, true>);
    BXH_SET_LDS(bxh::sweep_kernel_sw<double, false>);
    BXH_SET_LDS(bxh::sweep_kernel_sw<double, true>);
    BXH_SET_LDS(bxh::sweep_kernel_z<float, true>);
    BXH_SET_LDS(bxh::sweep_kernel_z<double, true>);
    BXH_SET_LDS(bxh::wide_flags_band_kernel);
    BXH_SET_LDS(bxh::order_post_kernel<float>);
    BXH_SET_LDS(bxh::order_post_kernel<double>);
    BXH_SET_LDS(bxh::order_init_kernel<float>);
    BXH_SET_LDS(bxh::order_init_kernel<double>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 4, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 4, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 5, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 5, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 6, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 6, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 8, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 8, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 4, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 4, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 5, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 5, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 6, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 6, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 8, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 8, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 4, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 4, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 5, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 5, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 6, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 6, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 8, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 8, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 4, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 4, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 5, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 5, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 6, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 6, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 8, false>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 8, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 4, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 5, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 6, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, false, 8, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 4, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 5, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 6, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<float, true, 8, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 4, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 5, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 6, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, false, 8, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 4, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 5, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 6, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_v2<double, true, 8, false, true>);
    BXH_SET_LDS(bxh::sweep_kernel_wide<float, false>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 1, 1>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 1, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 1, 3>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 1, 4>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 1>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 3>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 4>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 5>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 6>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 7>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 8>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 3, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 4, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 5, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 6, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 7, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, false, 4, 8, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 1, 1>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 1, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 1, 3>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 1, 4>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 1>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 3>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 4>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 5>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 6>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 7>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 8>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 3, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 4, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 5, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 6, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 7, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<float, true, 4, 8, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 1, 1>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 1, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 1, 3>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 1, 4>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 1>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 3>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 4>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 5>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 6>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 7>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 8>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 3, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 4, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 5, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 6, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 7, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, false, 4, 8, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 1, 1>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 1, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 1, 3>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 1, 4>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 1>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 3>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 4>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 5>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 6>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 7>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 8>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 3, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 4, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 5, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 6, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 7, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_narrow<double, true, 4, 8, 2>);
    BXH_SET_LDS(bxh::sweep_kernel_wide<float, true>);
    BXH_SET_LDS(bxh::sweep_kernel_wide<double, false>);
    BXH_SET_LDS(bxh::sweep_kernel_wide<double, true>);
#undef BXH_SET_LDS
    if (device >= 0) {
        if (static_cast<size_t>(device) >= done.size())
            done.resize(static_cast<size_t>(device) + 1, 0);
        done[device] = 1;
    }
    return hipSuccess;
}

} // namespace

struct bxh_handle_s
{
    int device = 0;
    int kind = 0, float_type = 1, maximize = 0, batch = 0;
    int m = 0, n = 0, nnz = 0, rmax = 0, nwords = 0;
    int nnzp = 0; // CSC slots per instance (columns padded to 16 entries)
    size_t fsize = 8;
    int threads = 256;
    size_t lds_bytes = 0;      // sweep kernel
    size_t lds_bytes_init = 0; // order_init kernel
    int version = 1;           // 1: lane-per-column walk, 2: staged column loads
    int nb = 0;                // v2: 16-entry pieces per staged column
    int groups = 1;            // v2: workgroups (column owners) per instance
    int spare = 0;             // v2: spare staging slot set
    int pipe = 0;              // v2: pipelined sweep, workgroups per instance (row chain + producers)
    int lookahead = 2;         // v2: rows between a producer and the last finished row
    int sw_chain = 0, sw_prod = 0; // v2: staging waves of the chain / a producer
    int romax = 0;             // longest per-owner part of a row
    int wide_g = 0;            // wide kernel (version 4): workgroups per instance
    int narrow_w = 0, narrow_rph = 0; // narrow kernel (version 5): waves per instance, staged row positions / 32
    bool narrow_occ2 = false;         // ... its 256-register build (two workgroups per CU)
    DevBuf<int4> ndesc;        // narrow kernel: per CSR element descriptors
    DevBuf<int4> nrowc;        // narrow kernel: per row {first element, length | c_size << 16, bmin, bmax}
    bxh::NarrowParams npar{};
    bxh::NarrowRun nrun{};          // (max_sweeps == 0 outside bxh_run)
    DevBuf<double> run_out;         // [batch][2] the four ints of a launch's outcome, then [batch] kappa: one transfer
    DevBuf<double> run_inc;         // [m + 1]
    std::vector<double> run_inc_host; // ... as uploaded last (the table of a solve does not change)
    DevBuf<unsigned> run_xbest;     // [batch][2 * nwords]
    bool device_built = false; // bxh_create_device: no host copy of the matrix exists
    // 1: the instance's list R is known to be empty (a sweep over it is a no-op);
    // 0: unknown or not empty.  Maintained by the calls that read `remaining` back.
    std::vector<unsigned char> known_empty;
    // device-resident population (bxh_pop_*, bxh_reinit)
    DevBuf<unsigned> xt;      // wide kernel: x transposed over the instances (batch <= 32)
    DevBuf<unsigned> pop;     // [slots + 1][2 * nwords]
    int pop_slots = -1;
    DevBuf<bxh::ReinitJob> jobs_d;
    std::vector<unsigned char> skipped; // the instance's last sweep was such a no-op: nothing was launched
    DevBuf<double> stage_d;    // bxh_set_P / bxh_get_P of a device-built handle: one chunk
    unsigned epoch = 0;        // sweep launches so far (tags of the row exchange)

    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    // problem
    DevBuf<int> row_ptr, row_col, row_pos, col_start, col_len, col_row, bmin, bmax, bneg;
    DevBuf<int2> row_cs;
    DevBuf<signed char> fac_csr;
    DevBuf<unsigned short> col_row16;
    DevBuf<int> sub_ptr, sub_e;      // per-owner row parts (groups > 1)
    DevBuf<int4> sub_desc;
    DevBuf<int4> chain_rec;          // pipelined sweep: the row chain's descriptor records (DevProblem::chain_rec)
    DevBuf<unsigned char> desc_cost; // cost of each descriptor's column (Float)
    DevBuf<unsigned> ownmask;        // [groups][2*nwords] x bits of each owner
    DevBuf<int> vwork, vpart;        // [batch][groups][m]
    DevBuf<unsigned long long> xbuf; // [batch][2][rmax] reduced-cost exchange (pipelined: [batch][ring][rmax])
    DevBuf<unsigned long long> msg;  // [batch][m][2] pipelined sweep: finished rows
    DevBuf<unsigned> cnt;            // [2*batch] arrival counters + [1] error flag
    DevBuf<unsigned char> cost; // Float[n]
    DevBuf<double> row_bytes;   // [m]
    // host copies needed for state import/export and accounting
    std::vector<int> h_row_pos, h_row_ptr, h_col_ptr, h_row_col;
    std::vector<unsigned char> h_owner; // owner workgroup of column j

    // state
    DevBuf<unsigned char> P, pi, rstale;
    DevBuf<unsigned> x;
    DevBuf<int> vlist, vcount;
    // compute_order (itm-common.hpp:627-915): one state per instance
    int lcap = 0;              // capacity of the row lists: max(m, n)
    int policy = -1;           // constraint_order given to the constructor, -1: none yet
    bool use_cycle = false;
    std::vector<int> ostate;   // current order of every instance (0..7)
    DevBuf<int> order_b;       // the same, on the device
    int* h_order = nullptr;    // pinned staging
    DevBuf<int2> mlist, mtmp;
    // inequalities-Z
    DevBuf<int> fac32, col_fac, ex_start, ex_count, ex_flat;
    DevBuf<int> q_ptr, q_id;            // quadratic objective
    DevBuf<unsigned> lcg_pow;           // 16807^k mod 2^31-1
    DevBuf<unsigned char> q_factor;
    DevBuf<signed char> zkind;
    DevBuf<unsigned long long> zconf; // Z kernel: rows k + 1 .. k + 64 that share a column with row k
    DevBuf<int4> zdesc;               // Z kernel, wave form: two 16-byte records per CSR element
    DevBuf<int2> zcolrf;              // Z kernel, wave form: {row, |coefficient|} of every padded CSC slot
    DevBuf<unsigned> zcolp;           // ... packed: row | |coefficient| << 16
    int zwave = 0;                    // Z kernel: rows of up to 63 elements go wave per row (groups)
    int swave = 0;                    // 01 / 101 solvers, every row of 1 .. 63 elements: a wave per row (version 7)
    DevBuf<unsigned char> zscratch;
    DevBuf<int> mcount, mrun, slist, vamt, sid;
    DevBuf<unsigned long long> skey;
    DevBuf<signed char> vflag;
    DevBuf<bxh::DevStatus> status;
    DevBuf<unsigned long long> prof;
    DevBuf<double> kappa;
    DevBuf<unsigned char> active;
    // wide kernel (bxh_wide.hpp)
    DevBuf<int2> col_desc;
    DevBuf<unsigned char> piw;
    DevBuf<unsigned long long> whash, wrec, vbuf, bkey;
    DevBuf<unsigned> wcnt, wcount;
    DevBuf<int> wdl, vsum; // vsum: [m][16] sums of the band scan (wide_flags_band_kernel), when it applies
    DevBuf<int> bid, vflag32;
    bxh::WideParams wp{};

    // HIP events around every sweep kernel launch (bxh_kernel_time)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    double kernel_ms = 0.0;
    unsigned long long kernel_launches = 0;

    bxh::DevStatus* h_status = nullptr; // pinned
    double* h_kappa = nullptr;          // pinned
    unsigned char* h_active = nullptr;  // pinned

    // bxh_sweep_async's uploads (kappa, and the active mask behind it) leave from a
    // ring of pinned slots in ONE copy per call: a call waits for the copy that used
    // its slot kStageSlots calls ago, not for the previous launch
    static constexpr int kStageSlots = 8;
    unsigned char* h_stage[kStageSlots] = {};
    hipEvent_t stage_done[kStageSlots] = {};
    bool stage_busy[kStageSlots] = {};
    size_t stage_seq = 0;
    DevBuf<unsigned char> sargs;  // [batch] double kappa, then [batch] active
    std::vector<int> ostate_dev;  // order_b as uploaded last

    // bxh_incumbent
    DevBuf<double> obj_cost, inc_value;
    DevBuf<bxh::IncumbentRecord> inc_rec;
    DevBuf<unsigned> inc_x;
    double obj_constant = 0.0;
    int obj_maximize = 0;
    bool obj_set = false;

    bxh::DevProblem dp{};
    bxh::DevState ds{};

    ~bxh_handle_s()
    {
        if (h_status)
            (void)hipHostFree(h_status);
        if (h_kappa)
            (void)hipHostFree(h_kappa);
        if (h_active)
            (void)hipHostFree(h_active);
        if (h_order)
            (void)hipHostFree(h_order);
        for (int i = 0; i != kStageSlots; ++i) {
            if (h_stage[i])
                (void)hipHostFree(h_stage[i]);
            if (stage_done[i])
                (void)hipEventDestroy(stage_done[i]);
        }
        for (auto& ev : events) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        if (own_stream)
            (void)hipStreamDestroy(own_stream);
    }
};

namespace {

int
check_inst(bxh_handle h, int inst)
{
    if (!h)
        return fail(BXH_ERR_INVALID, "null handle");
    if (inst < 0 || inst >= h->batch)
        return fail(BXH_ERR_INVALID, "instance index out of range");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess)
        return fail(BXH_ERR_DEVICE, hipGetErrorString(e));
    return BXH_OK;
}

// Compute units of a device, asked once per device (hipGetDeviceProperties costs
// milliseconds; bxh_create needs the number in several places and optimize() may
// create one handle per worker).
int
device_cus(int device)
{
    static std::mutex mu;
    static std::vector<int> known;
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0)
        return 256;
    if (static_cast<size_t>(device) >= known.size())
        known.resize(static_cast<size_t>(device) + 1, 0);
    int& cus = known[static_cast<size_t>(device)];
    if (cus == 0) {
        hipDeviceProp_t prop;
        cus = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
                ? prop.multiProcessorCount
                : 256;
    }
    return cus;
}

template<typename Float>
void
to_float(const double* in, size_t n, std::vector<unsigned char>& out)
{
    out.resize(n * sizeof(Float));
    Float* o = reinterpret_cast<Float*>(out.data());
    for (size_t i = 0; i != n; ++i)
        o[i] = static_cast<Float>(in[i]);
}

template<typename Float>
void
from_float(const std::vector<unsigned char>& in, size_t n, double* out)
{
    const Float* p = reinterpret_cast<const Float*>(in.data());
    for (size_t i = 0; i != n; ++i)
        out[i] = static_cast<double>(p[i]);
}

int
set_vec(bxh_handle h, unsigned char* dst, const double* src, size_t n)
{
    std::vector<unsigned char> tmp;
    if (h->float_type == BXH_FLOAT)
        to_float<float>(src, n, tmp);
    else
        to_float<double>(src, n, tmp);
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(dst, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    return BXH_OK;
}

int
get_vec(bxh_handle h, const unsigned char* src, double* dst, size_t n)
{
    std::vector<unsigned char> tmp(n * h->fsize);
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(tmp.data(), src, tmp.size(), hipMemcpyDeviceToHost));
    if (h->float_type == BXH_FLOAT)
        from_float<float>(tmp, n, dst);
    else
        from_float<double>(tmp, n, dst);
    return BXH_OK;
}

// Fold the finished event pairs into the running kernel time; the stream
// must be idle.
int
drain_events(bxh_handle h)
{
    for (size_t i = 0; i != h->events_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->events[i].first, h->events[i].second));
        h->kernel_ms += static_cast<double>(ms);
        ++h->kernel_launches;
    }
    h->events_used = 0;
    return BXH_OK;
}

// The device copy of the per-instance row orders follows the host's (`ostate`):
// uploaded only when it differs from what was uploaded last.
int
sync_order_b(bxh_handle h)
{
    if (h->ostate_dev == h->ostate)
        return BXH_OK;
    HIP_TRY(hipStreamSynchronize(h->stream)); // the pinned buffer is reused
    std::memcpy(h->h_order, h->ostate.data(), static_cast<size_t>(h->batch) * sizeof(int));
    HIP_TRY(hipMemcpyAsync(h->order_b.p, h->h_order, static_cast<size_t>(h->batch) * sizeof(int),
                           hipMemcpyHostToDevice, h->stream));
    h->ostate_dev = h->ostate;
    return BXH_OK;
}

// compute_violated_constraints of the wide layout for all (active) instances:
// x transposed, one pass over A, the ordered compaction per instance
int
launch_wide_scan(bxh_handle h, const unsigned char* dev_active)
{
    hipLaunchKernelGGL(bxh::wide_transpose_kernel, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, h->dp,
                       h->ds, h->xt.p, h->batch);
    HIP_TRY(hipGetLastError());
    const int blocks = std::max(1, std::min(2048, (h->m + 7) / 8));
    if (h->vsum.p) { // x out of LDS, a band of columns per workgroup (bxh_wide.hpp)
        const int nbands = (h->n + bxh::WIDE_BAND - 1) / bxh::WIDE_BAND;
        const int nchunks = std::max(1, std::min(device_cus(h->device) / nbands, (h->m + 63) / 64));
        const size_t lds = (static_cast<size_t>(std::min(h->n, bxh::WIDE_BAND)) * 2 + 15) & ~size_t(15);
        hipLaunchKernelGGL(bxh::wide_flags_band_kernel, dim3(nbands * nchunks), dim3(1024), lds, h->stream, h->dp,
                           h->xt.p, h->vsum.p, nbands, nchunks);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(bxh::wide_flags_finish_kernel,
                           dim3(static_cast<unsigned>((static_cast<size_t>(h->m) * 16 + 255) / 256)), dim3(256), 0,
                           h->stream, h->dp, h->wp, h->vsum.p, dev_active, h->batch);
    } else if (h->batch <= 16)
        hipLaunchKernelGGL((bxh::wide_flags_all_kernel<16>), dim3(blocks), dim3(512), 0, h->stream, h->dp,
                           h->wp, h->xt.p, dev_active, h->batch);
    else
        hipLaunchKernelGGL((bxh::wide_flags_all_kernel<32>), dim3(blocks), dim3(512), 0, h->stream, h->dp,
                           h->wp, h->xt.p, dev_active, h->batch);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(bxh::wide_compact_kernel, dim3(h->batch), dim3(512), 0, h->stream, h->dp, h->ds,
                       h->wp, dev_active);
    HIP_TRY(hipGetLastError());
    return BXH_OK;
}

template<typename Float>
int
launch_sweep(bxh_handle h, const bxh::DevSweep& a)
{
    const bool pm1 = h->kind == BXH_EQ_101 || h->kind == BXH_INEQ_101;
    if (h->events_used == h->events.size()) {
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        h->events.emplace_back(e0, e1);
    }
    HIP_TRY(hipEventRecord(h->events[h->events_used].first, h->stream));
#define BXH_LAUNCH(K)                                                          \
    hipLaunchKernelGGL(K, dim3(h->batch * (h->pipe ? h->pipe : h->groups)), dim3(h->threads), \
                       h->lds_bytes, h->stream, h->dp, h->ds, a)
    if (h->version == 2) {
#define BXH_V2(PM, NPV)                                                        \
    do {                                                                       \
        if (h->pipe)                                                           \
            BXH_LAUNCH((bxh::sweep_kernel_v2<Float, PM, NPV, false, true>));   \
        else if (h->groups > 1)                                                \
            BXH_LAUNCH((bxh::sweep_kernel_v2<Float, PM, NPV, true>));          \
        else                                                                   \
            BXH_LAUNCH((bxh::sweep_kernel_v2<Float, PM, NPV, false>));         \
    } while (0)
        if (pm1) {
            if (h->nb == 4)
                BXH_V2(true, 4);
            else if (h->nb == 5)
                BXH_V2(true, 5);
            else if (h->nb == 6)
                BXH_V2(true, 6);
            else
                BXH_V2(true, 8);
        } else {
            if (h->nb == 4)
                BXH_V2(false, 4);
            else if (h->nb == 5)
                BXH_V2(false, 5);
            else if (h->nb == 6)
                BXH_V2(false, 6);
            else
                BXH_V2(false, 8);
        }
#undef BXH_V2
    } else if (h->version == 3) {
        if (h->zwave)
            BXH_LAUNCH((bxh::sweep_kernel_z<Float, true>));
        else
            BXH_LAUNCH((bxh::sweep_kernel_z<Float, false>));
    } else if (h->version == 7) {
        if (pm1)
            hipLaunchKernelGGL((bxh::sweep_kernel_sw<Float, true>), dim3(h->batch), dim3(h->threads), h->lds_bytes,
                               h->stream, h->dp, h->ds, a, h->nrun);
        else
            hipLaunchKernelGGL((bxh::sweep_kernel_sw<Float, false>), dim3(h->batch), dim3(h->threads), h->lds_bytes,
                               h->stream, h->dp, h->ds, a, h->nrun);
    } else if (h->version == 5) {
#define BXH_NARROW_O(PM, WW, RQ, OC)                                           \
    hipLaunchKernelGGL((bxh::sweep_kernel_narrow<Float, PM, WW, RQ, OC>), dim3(h->batch), dim3(64 * WW), \
                       h->lds_bytes, h->stream, h->dp, h->ds, a, h->npar, h->nrun)
#define BXH_NARROW4(PM, RQ)                                                    \
    do {                                                                       \
        if (RQ >= 3 && h->narrow_occ2)                                         \
            BXH_NARROW_O(PM, 4, (RQ >= 3 ? RQ : 3), 2);                        \
        else                                                                   \
            BXH_NARROW_O(PM, 4, RQ, 1);                                        \
    } while (0)
#define BXH_NARROW_PM(PM)                                                      \
    do {                                                                       \
        if (h->narrow_w == 4) {                                                \
            switch (h->narrow_rph) {                                           \
            case 1: BXH_NARROW4(PM, 1); break;                                 \
            case 2: BXH_NARROW4(PM, 2); break;                                 \
            case 3: BXH_NARROW4(PM, 3); break;                                 \
            case 4: BXH_NARROW4(PM, 4); break;                                 \
            case 5: BXH_NARROW4(PM, 5); break;                                 \
            case 6: BXH_NARROW4(PM, 6); break;                                 \
            case 7: BXH_NARROW4(PM, 7); break;                                 \
            default: BXH_NARROW4(PM, 8); break;                                \
            }                                                                  \
        } else {                                                               \
            switch (h->narrow_rph) {                                           \
            case 1: BXH_NARROW_O(PM, 1, 1, 1); break;                          \
            case 2: BXH_NARROW_O(PM, 1, 2, 1); break;                          \
            case 3: BXH_NARROW_O(PM, 1, 3, 1); break;                          \
            default: BXH_NARROW_O(PM, 1, 4, 1); break;                         \
            }                                                                  \
        }                                                                      \
    } while (0)
        if (pm1)
            BXH_NARROW_PM(true);
        else
            BXH_NARROW_PM(false);
#undef BXH_NARROW4
#undef BXH_NARROW_PM
#undef BXH_NARROW_O
    } else if (h->version == 4) {
        if (pm1)
            hipLaunchKernelGGL((bxh::sweep_kernel_wide<Float, true>), dim3(h->batch * h->wide_g),
                               dim3(bxh::WIDE_T), h->lds_bytes, h->stream, h->dp, h->ds, a, h->wp);
        else
            hipLaunchKernelGGL((bxh::sweep_kernel_wide<Float, false>), dim3(h->batch * h->wide_g),
                               dim3(bxh::WIDE_T), h->lds_bytes, h->stream, h->dp, h->ds, a, h->wp);
    } else {
        if (pm1)
            BXH_LAUNCH((bxh::sweep_kernel<Float, true>));
        else
            BXH_LAUNCH((bxh::sweep_kernel<Float, false>));
    }
#undef BXH_LAUNCH
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->events[h->events_used].second, h->stream));
    ++h->events_used;
    return BXH_OK;
}

} // namespace

extern "C" {

const char*
bxh_last_error(void)
{
    return g_error.c_str();
}

int
bxh_create(const bxh_problem* pb,
           int kind,
           int float_type,
           int maximize,
           int batch,
           int device,
           bxh_handle* out)
{
    if (!pb || !out)
        return fail(BXH_ERR_INVALID, "null argument");
    *out = nullptr;
    if (pb->m <= 0 || pb->n <= 0 || pb->nnz <= 0 || batch <= 0)
        return fail(BXH_ERR_INVALID, "empty problem or batch");
    if (!pb->row_ptr || !pb->col || !pb->cmin || !pb->cmax || !pb->cost)
        return fail(BXH_ERR_INVALID, "null problem array");
    if (kind < BXH_EQ_01 || kind > BXH_INEQ_Z)
        return fail(BXH_ERR_INVALID, "unknown solver kind");
    if (float_type != BXH_FLOAT && float_type != BXH_DOUBLE)
        return fail(BXH_ERR_UNSUPPORTED,
                    "only float and double exist on the device (long double "
                    "does not)");
    if (pb->n_quad < 0 || (pb->n_quad > 0 && (!pb->q_ptr || !pb->q_id || !pb->q_factor)))
        return fail(BXH_ERR_INVALID, "quadratic objective arrays");
    if (pb->n_quad > 0) {
        if (pb->q_ptr[0] != 0 || pb->q_ptr[pb->n] != pb->n_quad)
            return fail(BXH_ERR_INVALID, "q_ptr does not span the quadratic terms");
        for (int i = 0; i != pb->n_quad; ++i)
            if (pb->q_id[i] < 0 || pb->q_id[i] >= pb->n)
                return fail(BXH_ERR_INVALID, "quadratic partner out of range");
    }
    if (pb->nnz > 0x7fffffffLL)
        return fail(BXH_ERR_UNSUPPORTED, "more than 2^31-1 elements");
    if (pb->row_ptr[0] != 0 || pb->row_ptr[pb->m] != pb->nnz)
        return fail(BXH_ERR_INVALID, "row_ptr does not span [0, nnz]");

    hipError_t e0 = hipSetDevice(device);
    if (e0 != hipSuccess)
        return fail(BXH_ERR_DEVICE,
                    std::string("hipSetDevice: ") + hipGetErrorString(e0));

    bxh_handle h = new (std::nothrow) bxh_handle_s;
    if (!h)
        return fail(BXH_ERR_OOM, "host allocation failed");
    std::unique_ptr<bxh_handle_s> guard(h);

    h->device = device;
    h->kind = kind;
    h->float_type = float_type;
    h->maximize = maximize ? 1 : 0;
    h->batch = batch;
    h->m = pb->m;
    h->n = pb->n;
    h->nnz = static_cast<int>(pb->nnz);
    h->nwords = pb->n / 64 + 1;
    h->fsize = float_type == BXH_FLOAT ? 4 : 8;

    const int m = h->m, n = h->n, nnz = h->nnz;
    const bool pm1 = kind == BXH_EQ_101 || kind == BXH_INEQ_101;

    // ---- validate + CSC (stable by column == counting sort,
    //      sparse-matrix.hpp:155-174) ---------------------------------------
    std::vector<int> col_ptr(static_cast<size_t>(n) + 1, 0);
    int rmax = 0;
    for (int k = 0; k != m; ++k) {
        const int len = pb->row_ptr[k + 1] - pb->row_ptr[k];
        if (len <= 0)
            return fail(BXH_ERR_INVALID, "empty or negative-length row");
        rmax = std::max(rmax, len);
    }
    for (int e = 0; e != nnz; ++e) {
        if (pb->col[e] < 0 || pb->col[e] >= n)
            return fail(BXH_ERR_INVALID, "column index out of range");
        const int f = pb->factor ? pb->factor[e] : 1;
        if (kind == BXH_INEQ_Z ? f == 0 : (pm1 ? (f != 1 && f != -1) : (f != 1)))
            return fail(BXH_ERR_INVALID,
                        "coefficient not allowed by this solver kind");
        ++col_ptr[pb->col[e] + 1];
    }
    for (int j = 0; j != n; ++j)
        col_ptr[j + 1] += col_ptr[j];
    h->rmax = rmax;

    int cmax = 0;
    for (int j = 0; j != n; ++j)
        cmax = std::max(cmax, col_ptr[j + 1] - col_ptr[j]);

    // ---- kernel choice: staged pipeline (v2) where the instance fits it, and
    // then how many workgroups (column owners) cooperate on one instance: all
    // of them must be resident together, one per CU ----------------------
    const char* force_v1 = std::getenv("BXH_FORCE_V1");
    const bool zsolver = kind == BXH_INEQ_Z;
    // The wide kernel (bxh_wide.hpp) takes what the staged kernel cannot:
    // columns of more than 128 entries, m >= 32768 (pi beyond LDS); linear
    // objective, 01 / 101 solvers.  BXH_GROUPS=w<G> forces it (tests).
    int wide_g = 0;
    {
        const char* eg = std::getenv("BXH_GROUPS");
        const bool asked = eg && eg[0] == 'w';
        const bool fits_v2 = cmax <= 128 && m < 32768;
        if (!zsolver && pb->n_quad == 0 && (asked || (!fits_v2 && !(force_v1 && force_v1[0] == '1')))) {
            const int cus = device_cus(device);
            // every workgroup should get a tile (32 columns) per wave of the longest row
            wide_g = std::max(1, std::min({ cus / batch, rmax / 256, 64 }));
            if (asked && std::atoi(eg + 1) > 0)
                wide_g = std::atoi(eg + 1);
            if (batch * wide_g > cus || wide_g > 64)
                return fail(BXH_ERR_UNSUPPORTED,
                            "the cooperating workgroups of all instances must be resident "
                            "together (batch * workgroups per instance <= CUs)");
        }
    }
    // Short rows (bxh_narrow.hpp): one wave per instance, several instances per
    // CU.  Rows of at most 256 elements, columns of a few 16-entry pieces, linear
    // objective, 01 / 101 solvers; BXH_NARROW=0 keeps the staged kernel, =1 takes
    // the narrow one for any column length the staged kernel would take.
    // ... and where every row has 1 .. 63 elements, a wave per row with independent rows of the list
    // side by side (bxh_swave.hpp; up to two workgroups per CU, linear objective).  BXH_SWAVE=0 / 1.
    bool swave = false;
    {
        int rmin = m > 0 ? rmax : 0;
        for (int k = 0; k != m; ++k)
            rmin = std::min(rmin, pb->row_ptr[k + 1] - pb->row_ptr[k]);
        const char* es = std::getenv("BXH_SWAVE");
        const bool fits = !zsolver && !wide_g && pb->n_quad == 0 && rmin >= 1 && rmax <= 63 && m >= 1 && m < 32768 &&
                          bxh::sw_lds(m, h->nwords, rmax, float_type == BXH_FLOAT ? 4 : 8).total <= 160 * 1024 &&
                          !std::getenv("BXH_GROUPS") && !(force_v1 && force_v1[0] == '1');
        // (by itself where the list is long enough for the groups to count and the one launch per
        // sweep not to: the short-row kernel runs many sweeps of a small instance in one launch,
        // bxh_run.  164 registers: three workgroups share a CU, and the form stays ahead of the
        // one-wave kernel at every batch size measured -- c1 at 512 / 1024 / 2048 restarts
        // 249 / 272 / 341 against 135 / 264 / 275 M row-updates/s)
        swave = fits && (es ? es[0] == '1' : m >= 64);
    }
    bool narrow = false;
    {
        const char* en = std::getenv("BXH_NARROW");
        const bool asked = en && en[0] == '1';
        const bool off = (en && en[0] == '0') || std::getenv("BXH_GROUPS") || (force_v1 && force_v1[0] == '1');
        const int rph = (rmax + 31) / 32;
        const int nl = bxh::narrow_lds(m, n, h->nwords, rmax, float_type == BXH_FLOAT ? 4 : 8, rph).total;
        narrow = !swave && !wide_g && !zsolver && pb->n_quad == 0 && rmax <= 256 && cmax <= 128 && m < 32767 &&
                 nl <= 160 * 1024 && !off && (asked || cmax <= 32);
    }
    const bool v2_ok = !swave && !narrow && !wide_g && !zsolver && cmax <= 128 && m < 32768 && !(force_v1 && force_v1[0] == '1');
    int groups = 1;
    if (v2_ok) {
        const int cus = device_cus(device);
        // ... and only while every owner still gets a full round of 128 columns of
        // the longest row: below that the exchange costs more than the split saves
        // (measured: rows of 20 and of 120 elements run 10-20 % faster with one owner)
        while (groups < 4 && batch * groups * 2 <= cus && rmax / (groups * 2) >= 128)
            groups *= 2;
        if (const char* eg = std::getenv("BXH_GROUPS")) {
            const int g = std::atoi(eg);
            if ((g == 1 || g == 2 || g == 4) && batch * g <= cus)
                groups = g;
            else if (g == 8 && batch * g <= 2 * cus) // two workgroups per CU
                groups = g;
        }
    }
    if (pb->n_quad > 0) // c(j, x) reads every x bit: one owner
        groups = 1;
    // Pipelined sweep (one row chain + producers per instance, one CSC copy):
    // where the lock-step kernel would use 4 column owners, or on request
    // (BXH_GROUPS=p2 / p3 / p4; BXH_PIPE=0 keeps the lock-step kernel).
    {
        int want = 0;
        const char* eg = std::getenv("BXH_GROUPS");
        const char* ep = std::getenv("BXH_PIPE");
        const int cus = device_cus(device);
        if (eg && eg[0] == 'p')
            want = std::atoi(eg + 1);
        else if (!eg && groups == 4 && !(ep && ep[0] == '0'))
            want = 4;
        if (v2_ok && pb->n_quad == 0 && want >= 2 && want <= 4 && batch * want <= cus &&
            rmax <= 1024 && n < 65536) {
            const int np = cmax <= 64 ? 4 : (cmax <= 80 ? 5 : (cmax <= 96 ? 6 : 8));
            const int fs = float_type == BXH_FLOAT ? 4 : 8;
            int swc = 0, swp = 0;
            for (int w = 8; w >= 1 && !swp; --w)
                if (bxh::lds_plan2(m, h->nwords, rmax, rmax, fs, w * 16, np, false, 2, n).total <=
                    160 * 1024)
                    swp = w;
            for (int w = 4; w >= 1 && !swc; --w)
                if (bxh::lds_plan2(m, h->nwords, rmax, rmax, fs, w * 16, np, false, 1, n).total <=
                    160 * 1024)
                    swc = w;
            if (swp >= 4 && swc >= 2) {
                h->pipe = want;
                h->sw_chain = swc;
                h->sw_prod = swp;
                if (const char* ew = std::getenv("BXH_PIPE_SW")) // experiments: fewer staging waves
                    h->sw_prod = std::max(1, std::min(swp, std::atoi(ew)));
                h->lookahead = 2;
                if (const char* el = std::getenv("BXH_LOOKAHEAD"))
                    h->lookahead = std::max(2, std::min(bxh::PIPE_RING - 1, std::atoi(el))); // (1 would dead-lock: the producers need the message of the row D + 1 back to load a row ahead)
                groups = 1;
            }
        }
        if (eg && eg[0] == 'p' && !h->pipe)
            groups = 1;
    }
    h->groups = groups;

    // Column owners.  Every owner should hold about r_k / groups entries of
    // EVERY row (the partners of an instance wait for the slowest one at each
    // row): greedy assignment, each column goes to the owner that is least
    // loaded over the rows the column touches.
    std::vector<unsigned char> owner(static_cast<size_t>(n), 0);
    if (groups > 1) {
        std::vector<int> csc_rows(static_cast<size_t>(nnz));
        {
            std::vector<int> fill(col_ptr.begin(), col_ptr.end() - 1);
            for (int k = 0; k != m; ++k)
                for (int e = pb->row_ptr[k]; e != pb->row_ptr[k + 1]; ++e)
                    csc_rows[fill[pb->col[e]]++] = k;
        }
        std::vector<int> load(static_cast<size_t>(m) * groups, 0);
        std::vector<long long> total(static_cast<size_t>(groups), 0);
        for (int j = 0; j != n; ++j) {
            int best = 0;
            long long best_score = -1;
            for (int g = 0; g != groups; ++g) {
                long long score = 0;
                for (int q = col_ptr[j]; q != col_ptr[j + 1]; ++q) {
                    const long long l = load[static_cast<size_t>(csc_rows[q]) * groups + g];
                    score += l * l; // penalise the rows where g is already ahead
                }
                score = score * 4096 + total[g] % 4096;
                if (best_score < 0 || score < best_score) {
                    best_score = score;
                    best = g;
                }
            }
            owner[j] = static_cast<unsigned char>(best);
            for (int q = col_ptr[j]; q != col_ptr[j + 1]; ++q)
                ++load[static_cast<size_t>(csc_rows[q]) * groups + best];
            total[best] += col_ptr[j + 1] - col_ptr[j];
        }
    }
    h->h_owner = owner;

    // Device CSC, per owner: every
    // column padded to a multiple of 16 entries (128 bytes of fp64 P), so a
    // column is a whole number of cache-line pieces.  Padding slots hold row
    // index m (pi_s[m] == 0 on the device) and P == 0 for ever.
    std::vector<int> col_start(static_cast<size_t>(n)), col_len(static_cast<size_t>(n));
    std::vector<long long> gsize(static_cast<size_t>(groups), 0);
    for (int j = 0; j != n; ++j) {
        const int g = owner[j];
        col_start[j] = static_cast<int>(gsize[g]);
        col_len[j] = col_ptr[j + 1] - col_ptr[j];
        gsize[g] += std::max(16, (col_len[j] + 15) & ~15);
        if (gsize[g] * groups > 0x7fffffffLL)
            return fail(BXH_ERR_UNSUPPORTED, "more than 2^31-1 padded elements");
    }
    const int nnzp = static_cast<int>(*std::max_element(gsize.begin(), gsize.end()));
    h->nnzp = nnzp; // stride of one owner's slice
    std::vector<int> col_row(static_cast<size_t>(nnzp) * groups, m), row_pos(static_cast<size_t>(nnz));
    std::vector<signed char> fac_csr(pm1 ? nnz : 0);
    std::vector<int> fac32(zsolver ? nnz : 0), col_fac(zsolver ? static_cast<size_t>(nnzp) : 0, 0);
    {
        std::vector<int> fill(col_start);
        for (int k = 0; k != m; ++k)
            for (int e = pb->row_ptr[k]; e != pb->row_ptr[k + 1]; ++e) {
                const int j = pb->col[e];
                const int q = fill[j]++;
                const size_t at = static_cast<size_t>(owner[j]) * nnzp + q;
                col_row[at] = k;
                row_pos[e] = q;
                if (pm1) {
                    const int fe = pb->factor ? pb->factor[e] : 1; // (NULL = all +1)
                    fac_csr[e] = static_cast<signed char>(fe);
                    if (fe < 0)
                        col_row[at] |= 0x40000000;
                }
                if (zsolver) {
                    const int f = pb->factor ? pb->factor[e] : 1;
                    fac32[e] = f;
                    col_fac[at] = f < 0 ? -f : f;
                }
            }
    }
    // per-owner parts of every row (element ids in row order) and x bit masks
    std::vector<int> sub_ptr, sub_e;
    std::vector<unsigned> ownmask;
    int romax = rmax;
    if (groups > 1) {
        romax = 0;
        sub_ptr.assign(static_cast<size_t>(groups) * (m + 1), 0);
        sub_e.reserve(static_cast<size_t>(nnz));
        for (int g = 0; g != groups; ++g)
            for (int k = 0; k != m; ++k) {
                sub_ptr[static_cast<size_t>(g) * (m + 1) + k] = static_cast<int>(sub_e.size());
                for (int e = pb->row_ptr[k]; e != pb->row_ptr[k + 1]; ++e)
                    if (owner[pb->col[e]] == g)
                        sub_e.push_back(e);
                // longest columns first: a chunk of 16 loads and adds as many
                // 16-entry pieces as its longest column has, so like lengths
                // share a chunk (the order inside an owner's part is free: values
                // are exchanged by their position in the whole row)
                std::stable_sort(sub_e.begin() + sub_ptr[static_cast<size_t>(g) * (m + 1) + k],
                                 sub_e.end(), [&](int a, int b) {
                                     return ((col_len[pb->col[a]] + 15) >> 4) >
                                            ((col_len[pb->col[b]] + 15) >> 4);
                                 });
                const int len = static_cast<int>(sub_e.size()) -
                                sub_ptr[static_cast<size_t>(g) * (m + 1) + k];
                romax = std::max(romax, len);
                if (k == m - 1)
                    sub_ptr[static_cast<size_t>(g) * (m + 1) + m] = static_cast<int>(sub_e.size());
            }
        ownmask.assign(static_cast<size_t>(groups) * 2 * h->nwords, 0u);
        for (int j = 0; j != n; ++j)
            ownmask[static_cast<size_t>(owner[j]) * 2 * h->nwords + (j >> 5)] |= 1u << (j & 31);
    }
    h->romax = romax;
    if (std::getenv("BXH_DEBUG") && groups > 1) {
        long over = 0;
        for (int g = 0; g != groups; ++g)
            for (int k = 0; k != m; ++k)
                over += sub_ptr[static_cast<size_t>(g) * (m + 1) + k + 1] -
                          sub_ptr[static_cast<size_t>(g) * (m + 1) + k] > 128;
        std::fprintf(stderr, "bxh: groups=%d rmax=%d romax=%d parts longer than 128: %ld of %d\n",
                     groups, rmax, romax, over, groups * m);
    }

    // ---- bounds as each Solver constructor clamps them --------------------
    std::vector<int> bmin(m), bmax(m), bneg(m, 0);
    for (int k = 0; k != m; ++k) {
        const int rs = pb->row_ptr[k], re = pb->row_ptr[k + 1];
        const int len = re - rs;
        const int cmin = pb->cmin[k], cmax = pb->cmax[k];
        int lower = 0, upper = 0;
        for (int e = rs; e != re; ++e) {
            if ((pb->factor ? pb->factor[e] : 1) > 0)
                ++upper;
            else
                ++lower;
        }
        switch (kind) {
        case BXH_EQ_01: // itm-solver-equalities-01.cpp:84-96
            if (cmin != cmax)
                return fail(BXH_ERR_INVALID, "equalities-01 needs min == max");
            bmin[k] = bmax[k] = cmin;
            break;
        case BXH_INEQ_01: // itm-solver-inequalities-01.cpp:88-106
            if (cmin == cmax) {
                bmin[k] = cmin;
                bmax[k] = cmax;
            } else {
                bmin[k] = std::max(0, cmin);
                bmax[k] = std::min(len, cmax);
            }
            break;
        case BXH_EQ_101: // itm-solver-equalities-101.cpp:98-113
            if (cmin != cmax)
                return fail(BXH_ERR_INVALID, "equalities-101 needs min == max");
            bmin[k] = bmax[k] = cmin;
            bneg[k] = lower;
            break;
        case BXH_INEQ_101: // itm-solver-inequalities-101.cpp:99-129
            if (cmin == cmax) {
                bmin[k] = cmin;
                bmax[k] = cmax;
            } else {
                bmin[k] = std::max(-lower, cmin);
                bmax[k] = std::min(upper, cmax);
            }
            bneg[k] = lower;
            break;
        default: { // itm-solver-inequalities-Z.cpp:125-187 (bounds)
            int lower_sum = 0, upper_sum = 0;
            for (int e = rs; e != re; ++e) {
                const int f = pb->factor ? pb->factor[e] : 1;
                if (f > 0)
                    upper_sum += f;
                else
                    lower_sum += f;
            }
            if (cmin == cmax) {
                bmin[k] = cmin;
                bmax[k] = cmax;
            } else {
                bmin[k] = std::max(lower_sum, cmin);
                bmax[k] = std::min(upper_sum, cmax);
            }
            break;
        }
        }
        if (bmin[k] > bmax[k])
            return fail(BXH_ERR_INVALID, "constraint with min > max");
    }

    // ---- launch geometry ----------------------------------------------------
    // row_cs[e] = { first padded CSC slot of the column, padded length | the
    // largest piece count among the 16 descriptors of its staging chunk << 16 }:
    // a wave loads that many 16-entry pieces for every column of the chunk
    std::vector<int2> row_cs(static_cast<size_t>(nnz));
    for (int e = 0; e != nnz; ++e) {
        const int j = pb->col[e];
        row_cs[e] = make_int2(col_start[j], std::max(16, (col_len[j] + 15) & ~15));
    }
    auto mark_chunks = [&](const int* ids, int count) { // one row (or owner part)
        for (int c0 = 0; c0 < count; c0 += 16) {
            int mx = 1;
            for (int l = c0; l < std::min(count, c0 + 16); ++l)
                mx = std::max(mx, (row_cs[ids[l]].y & 0xffff) >> 4);
            for (int l = c0; l < std::min(count, c0 + 16); ++l)
                row_cs[ids[l]].y = (row_cs[ids[l]].y & 0xffff) | (mx << 16);
        }
    };
    if (groups > 1) {
        for (int g = 0; g != groups; ++g)
            for (int k = 0; k != m; ++k) {
                const size_t at = static_cast<size_t>(g) * (m + 1) + k;
                mark_chunks(sub_e.data() + sub_ptr[at], sub_ptr[at + 1] - sub_ptr[at]);
            }
    } else {
        std::vector<int> ids(static_cast<size_t>(rmax));
        for (int k = 0; k != m; ++k) {
            const int len = pb->row_ptr[k + 1] - pb->row_ptr[k];
            for (int l = 0; l != len; ++l)
                ids[l] = pb->row_ptr[k] + l;
            mark_chunks(ids.data(), len);
        }
    }

    const int lds_limit = 160 * 1024;
    const bxh::LdsPlan lp =
      bxh::lds_plan(m, h->nwords, rmax, static_cast<int>(h->fsize));
    if (lp.total > lds_limit && !wide_g)
        return fail(BXH_ERR_UNSUPPORTED,
                    "instance does not fit the one-workgroup-per-instance "
                    "kernel (pi, x and the row buffers must fit 160 KiB of "
                    "LDS)");
    h->lds_bytes_init = static_cast<size_t>(lp.total);
    h->version = 1;
    h->lds_bytes = static_cast<size_t>(lp.total);
    h->threads = std::min(1024, std::max(64, bxh::next_pow2(rmax)));
    if (const char* et = std::getenv("BXH_V1_THREADS")) // experiments: the one-workgroup kernels' size
        h->threads = std::max(64, std::min(1024, std::atoi(et) & ~63));
    // ---- inequalities-Z: sub-solver of every row (inequalities-Z.cpp:125-187)
    //      and the feasible patterns of the exhaustive rows
    //      (exhaustive-solver.hpp:111-167) --------------------------------------
    std::vector<signed char> zkind;
    std::vector<int> ex_start, ex_count, ex_flat;
    if (zsolver) {
        zkind.assign(static_cast<size_t>(m), 4);
        ex_start.assign(static_cast<size_t>(m), 0);
        ex_count.assign(static_cast<size_t>(m), 0);
        for (int k = 0; k != m; ++k) {
            const int rs = pb->row_ptr[k], len = pb->row_ptr[k + 1] - rs;
            int lower = 0, upper = 0;
            int z = 4; // linear
            for (int e = rs; e != rs + len; ++e) {
                const int f = fac32[e];
                if (f > 0)
                    upper += f;
                else
                    lower += f;
                if (f < -1 || f > 1)
                    z = 0; // branch and bound
            }
            if (z == 0 && len < 10) // maximum_factor_exhaustive_solver, :39
                z = 1;
            if (pb->cmin[k] != pb->cmax[k] && (lower >= pb->cmin[k] || upper <= pb->cmax[k])) {
                if (z == 0)
                    z = 2; // unconstrained branch and bound
                else if (z == 1)
                    z = 3; // unconstrained exhaustive
            }
            zkind[k] = static_cast<signed char>(z);
            if (z == 1 || z == 3) {
                // The reference walks a binary counter whose least significant
                // digit is the LAST element, starting at 0..01, and also
                // evaluates the states in the middle of every carry (so some
                // patterns appear more than once); that order decides ties.
                ex_start[k] = static_cast<int>(ex_flat.size());
                int nb = 0;
                std::vector<signed char> digit(static_cast<size_t>(len), 0);
                digit.back() = 1;
                auto visit = [&]() {
                    int sum = 0;
                    for (int jx = 0; jx != len; ++jx)
                        if (digit[jx])
                            sum += fac32[rs + jx];
                    if (bmin[k] <= sum && sum <= bmax[k]) {
                        ++nb;
                        for (int jx = 0; jx != len; ++jx)
                            ex_flat.push_back(digit[jx] ? fac32[rs + jx] : 0);
                    }
                };
                bool done = false;
                while (!done) {
                    int pos = len - 1;
                    for (;;) {
                        visit();
                        if (++digit[pos] <= 1)
                            break;
                        digit[pos] = 0;
                        if (pos == 0) {
                            done = true;
                            break;
                        }
                        --pos;
                    }
                }
                if (nb <= 0)
                    return fail(BXH_ERR_INVALID, "a constraint has no feasible assignment");
                ex_count[k] = nb;
            }
        }
        // wave-per-row groups (bxh_kernels.hpp: sweep_kernel_z<Float, true>): linear objective
        {
            // (the wave form holds a column's slots, their pi and the next group's descriptors in
            // registers: one workgroup per CU.  Beyond two workgroups per CU the plain form, two
            // to a CU, is ahead again -- c5 at 1024 restarts: 112 against 76 M row-updates/s.)
            const char* ez = std::getenv("BXH_ZWAVE");
            const int cus = device_cus(device);
            h->zwave = (pb->n_quad == 0 && batch <= 2 * cus) ? 1 : 0;
            if (ez) {
                h->zwave = (ez[0] != '0' && pb->n_quad == 0) ? 1 : 0;
                if (h->zwave && batch > 2 * cus)
                    std::fprintf(stderr, "libbxh: BXH_ZWAVE forces the wave form at %d instances on %d CUs; "
                                         "beyond two workgroups per CU the plain form is the faster one\n",
                                 batch, cus);
            }
            if (h->zwave &&
                bxh::lds_bytes_z(m, h->nwords, rmax, static_cast<int>(h->fsize), true) > lds_limit)
                h->zwave = 0;
            // (the wave form reads a slot's row and |coefficient| as one 32-bit word, and its
            // 32-byte descriptor keeps the padded column length in 16 bits: rows, coefficients
            // and the longest padded column must all stay below 65536 -- a column has at most
            // m entries, padded to the next multiple of 16)
            if (h->zwave && (m > 65535 || ((cmax + 15) & ~15) > 65535 ||
                             *std::max_element(col_fac.begin(), col_fac.end()) > 65535))
                h->zwave = 0;
        }
        const int zl = bxh::lds_bytes_z(m, h->nwords, rmax, static_cast<int>(h->fsize), h->zwave != 0);
        if (zl > lds_limit)
            return fail(BXH_ERR_UNSUPPORTED, "instance does not fit the Z kernel's LDS plan");
        h->version = 3;
        h->lds_bytes = static_cast<size_t>(zl);
        h->threads = std::min(h->threads, 256); // sweep_kernel_z's launch bound (longer rows loop)
        if (h->zwave)
            h->threads = 256; // four waves: up to four rows side by side
    }
    if (v2_ok) {
        const int np = cmax <= 64 ? 4 : (cmax <= 80 ? 5 : (cmax <= 96 ? 6 : 8));
        // 16 staging slots per wave; as many waves (up to 8) as LDS allows
        int tmax = 512, lds_cap = lds_limit;
        if (const char* et = std::getenv("BXH_THREADS"))
            tmax = std::max(128, std::min(512, std::atoi(et)));
        if (groups == 8) { // two resident workgroups share a CU's LDS
            tmax = std::min(tmax, 256);
            lds_cap = lds_limit / 2;
        }
        // a spare set of slots (the overflow chunk of a row, summed in the same
        // pass as wave 0's own chunk) where it fits beside a full set of waves
        if (h->pipe) {
            const int fs = static_cast<int>(h->fsize);
            const int lc = bxh::lds_plan2(m, h->nwords, rmax, rmax, fs, h->sw_chain * 16, np, false, 1, n).total;
            const int lpz = bxh::lds_plan2(m, h->nwords, rmax, rmax, fs, h->sw_prod * 16, np, false, 2, n).total;
            h->version = 2;
            h->nb = np;
            h->threads = 512;
            h->spare = 0;
            h->lds_bytes = static_cast<size_t>(std::max(lc, lpz));
        }
        for (int t = tmax; t >= 128 && h->version != 2; t >>= 1)
            for (int spare = 1; spare >= 0; --spare) {
                const bxh::LdsPlan2 lp2 = bxh::lds_plan2(m, h->nwords, rmax, romax,
                                                         static_cast<int>(h->fsize),
                                                         (t / 64 + spare) * 16, np, groups > 1);
                if (lp2.total <= lds_cap) {
                    h->version = 2;
                    h->nb = np;
                    h->threads = t;
                    h->spare = spare;
                    h->lds_bytes = static_cast<size_t>(lp2.total);
                    break;
                }
            }
    }
    int wide_chmax = 0;
    if (wide_g) {
        wide_chmax = (((rmax + wide_g - 1) / wide_g) + bxh::WIDE_TILE - 1) & ~(bxh::WIDE_TILE - 1);
        // the pi window: the largest power of two that LDS still holds, all of pi if it fits
        int ring = 1024;
        while (ring < m + 1 && bxh::wide_lds(wide_g, wide_chmax, static_cast<int>(h->fsize), ring * 2).total <= lds_limit)
            ring *= 2;
        h->wp.ring = ring;
        const bxh::WideLds wl = bxh::wide_lds(wide_g, wide_chmax, static_cast<int>(h->fsize), ring);
        if (wl.total > lds_limit)
            return fail(BXH_ERR_UNSUPPORTED,
                        "rows too long for the workgroups this batch leaves to an instance");
        h->version = 4;
        h->wide_g = wide_g;
        h->threads = bxh::WIDE_T;
        h->lds_bytes = static_cast<size_t>(wl.total);
        h->lds_bytes_init = 0;
    }
    if (swave) {
        h->version = 7;
        h->swave = 1;
        h->threads = 256;
        h->lds_bytes = static_cast<size_t>(bxh::sw_lds(m, h->nwords, rmax, static_cast<int>(h->fsize)).total);
    }
    if (narrow) {
        h->version = 5;
        h->threads = 64;
        h->narrow_rph = (rmax + 31) / 32;
        // four waves (one per SIMD of the CU) share a row of more than 64 elements;
        // BXH_NARROW_W=1 / 4 decides otherwise
        h->narrow_w = rmax > 64 ? 4 : 1;
        if (const char* ew = std::getenv("BXH_NARROW_W"))
            h->narrow_w = (std::atoi(ew) == 4 || h->narrow_rph > 4) ? 4 : 1; // (one wave: rows of up to 128)
        h->threads = 64 * h->narrow_w;
        h->lds_bytes = static_cast<size_t>(
          bxh::narrow_lds(m, n, h->nwords, rmax, static_cast<int>(h->fsize), h->narrow_rph).total);
        // more instances than CUs, and two LDS plans fit a CU: the 256-register build, two
        // workgroups per CU (BXH_NARROW_OCC=1 / 2 decides otherwise)
        {
            const int cus = device_cus(device);
            h->narrow_occ2 = h->narrow_w == 4 && h->narrow_rph >= 3 && batch > cus && 2 * h->lds_bytes <= 160 * 1024;
            if (const char* eo = std::getenv("BXH_NARROW_OCC"))
                h->narrow_occ2 = h->narrow_w == 4 && h->narrow_rph >= 3 && eo[0] == '2';
        }
    }
    if (std::getenv("BXH_DEBUG"))
        std::fprintf(stderr, "bxh: kernel v%d threads=%d lds=%zu np=%d groups=%d pipe=%d (chain %d, producer %d staging waves, lookahead %d)\n",
                     h->version, h->threads, h->lds_bytes, h->nb, groups, h->pipe, h->sw_chain,
                     h->sw_prod, h->lookahead);
    if (h->version != 2 && groups > 1)
        return fail(BXH_ERR_UNSUPPORTED, "instance does not fit the staged kernel's LDS plan");

    HIP_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;

    // ---- upload -------------------------------------------------------------
    HIP_TRY(h->row_ptr.upload(std::vector<int>(pb->row_ptr, pb->row_ptr + m + 1)));
    HIP_TRY(h->row_col.upload(std::vector<int>(pb->col, pb->col + nnz)));
    HIP_TRY(h->row_pos.upload(row_pos));
    HIP_TRY(h->col_start.upload(col_start));
    HIP_TRY(h->col_len.upload(col_len));
    HIP_TRY(h->col_row.upload(col_row));
    if (m < 32768) {
        std::vector<unsigned short> c16(col_row.size());
        for (size_t q = 0; q != col_row.size(); ++q)
            c16[q] = static_cast<unsigned short>((col_row[q] & 0x7fff) |
                                                 ((col_row[q] & 0x40000000) ? 0x8000 : 0));
        HIP_TRY(h->col_row16.upload(c16));
    }
    HIP_TRY(h->bmin.upload(bmin));
    HIP_TRY(h->bmax.upload(bmax));
    HIP_TRY(h->bneg.upload(bneg));
    HIP_TRY(h->fac_csr.upload(fac_csr));
    {
        std::vector<unsigned> pw(bxh::WIDE_LCG_POW); // (the short rows need 1026 of them)
        unsigned long long v = 1;
        for (size_t k = 0; k != pw.size(); ++k) {
            pw[k] = static_cast<unsigned>(v);
            v = v * 16807ull % 2147483647ull;
        }
        HIP_TRY(h->lcg_pow.upload(pw));
    }
    if (pb->n_quad > 0) {
        HIP_TRY(h->q_ptr.upload(std::vector<int>(pb->q_ptr, pb->q_ptr + n + 1)));
        HIP_TRY(h->q_id.upload(std::vector<int>(pb->q_id, pb->q_id + pb->n_quad)));
        std::vector<unsigned char> qf;
        if (float_type == BXH_FLOAT)
            to_float<float>(pb->q_factor, static_cast<size_t>(pb->n_quad), qf);
        else
            to_float<double>(pb->q_factor, static_cast<size_t>(pb->n_quad), qf);
        HIP_TRY(h->q_factor.upload(qf));
    }
    if (zsolver) {
        if (ex_flat.empty())
            ex_flat.push_back(0);
        HIP_TRY(h->fac32.upload(fac32));
        HIP_TRY(h->col_fac.upload(col_fac));
        HIP_TRY(h->zkind.upload(zkind));
    }
    if (zsolver || swave) {
        {
            // bit d - 1 of zconf[k]: row k + d holds a column of row k (d = 1 .. 64).  Rows
            // that share no column commute exactly (neither reads what the other writes), so
            // the kernel runs such neighbours of the row list side by side, a wave each.
            std::vector<unsigned long long> conf(static_cast<size_t>(m), 0ull);
            std::vector<std::vector<int>> recent(static_cast<size_t>(n));
            for (int k = 0; k != m; ++k)
                for (int e = pb->row_ptr[k]; e != pb->row_ptr[k + 1]; ++e) {
                    auto& rj = recent[static_cast<size_t>(pb->col[e])];
                    size_t keep = 0;
                    for (size_t t = 0; t != rj.size(); ++t)
                        if (k - rj[t] <= 64) {
                            if (rj[t] != k)
                                conf[static_cast<size_t>(rj[t])] |= 1ull << (k - rj[t] - 1);
                            rj[keep++] = rj[t];
                        }
                    rj.resize(keep);
                    if (rj.empty() || rj.back() != k)
                        rj.push_back(k);
                }
            HIP_TRY(h->zconf.upload(conf));
        }
    }
    if (zsolver) {
        HIP_TRY(h->ex_start.upload(ex_start));
        HIP_TRY(h->ex_count.upload(ex_count));
        HIP_TRY(h->ex_flat.upload(ex_flat));
    }
    HIP_TRY(h->sub_ptr.upload(sub_ptr));
    HIP_TRY(h->sub_e.upload(sub_e));
    {
        std::vector<int4> sd(sub_e.size());
        for (size_t t = 0; t != sub_e.size(); ++t) {
            const int e = sub_e[t];
            sd[t] = make_int4(row_cs[e].x, row_cs[e].y, row_pos[e], pb->col[e]);
        }
        HIP_TRY(h->sub_desc.upload(sd));
    }
    HIP_TRY(h->ownmask.upload(ownmask));
    if (h->pipe) {
        // one 16-byte record per CSR element, a row's records in one run: LDS-DMA moves
        // them into the chain's LDS two rows ahead (64 records per wave and instruction,
        // hence the padding behind the last row)
        // The neighbours of every entry in its column: how many rows up / down the column's
        // previous / next entry sits (clipped to 15, 0 = none) and, where that is the very
        // next row, the entry's index in it.  When a sweep walks consecutive rows (every full
        // sweep in the orders none / reversing), "which of the D rows before this one hold the
        // column" is this static gap, and neither the chain nor the producers need their
        // column marks (bxh_kernels.hpp: smode).  w = up | down << 4 | index up << 8 | index
        // down << 18; the producers find the two gaps in bits 24 .. 31 of row_cs[e].y.
        // (the two in-row indices take 10 bits each here, and the chain's merged mark is
        // `code << 10 | index` in 16 bits: both need rows of at most 1024 elements, which is
        // what the pipe gate above admits -- checked again where the packing depends on it)
        static_assert(bxh::PIPE_RING <= 63, "the marks' code space is 6 bits (position mod 63 + 1)");
        if (rmax > 1024)
            return fail(BXH_ERR_UNSUPPORTED, "pipelined sweep: rows of more than 1024 elements");
        std::vector<unsigned> nb(static_cast<size_t>(nnz), 0u);
        {
            std::vector<int> last(static_cast<size_t>(n), -1); // last element seen in column j
            for (int k = 0; k != m; ++k)
                for (int e = pb->row_ptr[k]; e != pb->row_ptr[k + 1]; ++e) {
                    const int j = pb->col[e];
                    const int pe_ = last[j];
                    if (pe_ >= 0) {
                        // the row of element pe_: its CSC slot is the one in front of e's
                        const int kp = col_row[static_cast<size_t>(row_pos[e]) - 1] & 0x3fffffff;
                        const unsigned gap = static_cast<unsigned>(std::min(k - kp, 15));
                        nb[e] |= gap;
                        nb[pe_] |= gap << 4;
                        if (gap == 1) {
                            nb[e] |= static_cast<unsigned>(pe_ - pb->row_ptr[kp]) << 8;
                            nb[pe_] |= static_cast<unsigned>(e - pb->row_ptr[k]) << 18;
                        }
                    }
                    last[j] = e;
                }
        }
        std::vector<int4> rec(static_cast<size_t>(nnz) + 64, make_int4(0, 16, 0, 0));
        for (int e = 0; e != nnz; ++e) {
            rec[e] = make_int4(row_cs[e].x, (row_cs[e].y & 0xffff) | ((row_pos[e] - row_cs[e].x) << 16),
                               pb->col[e], static_cast<int>(nb[e]));
            row_cs[e].y = (row_cs[e].y & 0x00ffffff) | static_cast<int>((nb[e] & 0xffu) << 24);
        }
        HIP_TRY(h->chain_rec.upload(rec));
    }
    HIP_TRY(h->row_cs.upload(row_cs)); // (behind the block above: it adds the gaps)
    if ((zsolver && h->zwave) || swave) {
        // what a lane of the wave form needs of its element, in ONE run of 32 bytes
        // ({first slot, padded length, own slot, column}, {coefficient, cost bits}) and of its
        // column's slots in one array ({row, |a|}): few base pointers, few loads
        std::vector<int4> zd(static_cast<size_t>(nnz) * 2);
        for (int e = 0; e != nnz; ++e) {
            const int j = pb->col[e];
            zd[2 * static_cast<size_t>(e)] = make_int4(row_cs[e].x, row_cs[e].y & 0xffff, row_pos[e], j);
            int lo = 0, hi = 0;
            if (float_type == BXH_FLOAT) {
                const float cf = static_cast<float>(pb->cost[j]);
                std::memcpy(&lo, &cf, 4);
            } else {
                const double cd = pb->cost[j];
                int two[2];
                std::memcpy(two, &cd, 8);
                lo = two[0];
                hi = two[1];
            }
            zd[2 * static_cast<size_t>(e) + 1] =
              make_int4(zsolver ? fac32[e] : (pb->factor ? pb->factor[e] : 1), lo, hi, col_len[j]); // (.w: the column's real entries)
        }
        HIP_TRY(h->zdesc.upload(zd));
        std::vector<int2> rf(static_cast<size_t>(nnzp));
        for (size_t q = 0; q != static_cast<size_t>(nnzp); ++q) {
            if (zsolver)
                rf[q] = make_int2(col_row[q], col_fac[q]);
            else { // 01 / 101: the row without its sign flag, the coefficient itself (0 in the padding)
                const int rq = col_row[q] & 0x3fffffff;
                rf[q] = make_int2(rq, rq >= m ? 0 : ((col_row[q] & 0x40000000) ? -1 : 1));
            }
        }
        HIP_TRY(h->zcolrf.upload(rf));
        if (zsolver) { // (bxh_create took the wave form only where both halves fit 16 bits)
            std::vector<unsigned> rp(static_cast<size_t>(nnzp));
            for (size_t q = 0; q != static_cast<size_t>(nnzp); ++q)
                rp[q] = (static_cast<unsigned>(col_row[q]) & 0xffffu) | (static_cast<unsigned>(col_fac[q]) << 16);
            HIP_TRY(h->zcolp.upload(rp));
        }
    }
    {
        std::vector<unsigned char> c;
        if (float_type == BXH_FLOAT)
            to_float<float>(pb->cost, n, c);
        else
            to_float<double>(pb->cost, n, c);
        HIP_TRY(h->cost.upload(c));
        // the same costs in descriptor order (per owner part, or CSR order
        // with one owner): an independent load next to the descriptor
        std::vector<double> dc(static_cast<size_t>(nnz));
        for (int t = 0; t != nnz; ++t)
            dc[t] = pb->cost[pb->col[groups > 1 ? sub_e[t] : t]];
        std::vector<unsigned char> c2;
        if (float_type == BXH_FLOAT)
            to_float<float>(dc.data(), dc.size(), c2);
        else
            to_float<double>(dc.data(), dc.size(), c2);
        HIP_TRY(h->desc_cost.upload(c2));
        if (narrow) {
            std::vector<int4> nd(static_cast<size_t>(nnz) * 2);
            for (int e = 0; e != nnz; ++e) {
                const int j = pb->col[e];
                const int neg = (pm1 && pb->factor && pb->factor[e] < 0) ? static_cast<int>(0x80000000u) : 0;
                nd[2 * static_cast<size_t>(e)] = make_int4(col_start[j], col_len[j], row_pos[e], j | neg);
                int lo = 0, hi = 0;
                if (float_type == BXH_FLOAT) {
                    const float cf = static_cast<float>(pb->cost[j]);
                    std::memcpy(&lo, &cf, 4);
                } else {
                    const double cd = pb->cost[j];
                    int w[2];
                    std::memcpy(w, &cd, 8);
                    lo = w[0];
                    hi = w[1];
                }
                nd[2 * static_cast<size_t>(e) + 1] = make_int4(lo, hi, 0, 0);
            }
            HIP_TRY(h->ndesc.upload(nd));
            h->npar.desc = h->ndesc.p;
            std::vector<int4> rc(static_cast<size_t>(m));
            for (int k = 0; k != m; ++k)
                rc[k] = make_int4(pb->row_ptr[k], (pb->row_ptr[k + 1] - pb->row_ptr[k]) | ((pm1 ? bneg[k] : 0) << 16),
                                  bmin[k], bmax[k]);
            HIP_TRY(h->nrowc.upload(rc));
            h->npar.rowc = h->nrowc.p;
        }
    }
    h->h_row_pos = std::move(row_pos);
    h->h_row_ptr.assign(pb->row_ptr, pb->row_ptr + m + 1);
    h->h_col_ptr = std::move(col_ptr);
    h->h_row_col.assign(pb->col, pb->col + nnz);
    {
        std::vector<double> rb(static_cast<size_t>(m));
        if (int rc = bxh_row_bytes(h, rb.data()))
            return rc;
        HIP_TRY(h->row_bytes.upload(rb));
    }

    const size_t B = static_cast<size_t>(batch);
    const size_t GG = static_cast<size_t>(groups);
    const size_t WG = static_cast<size_t>(wide_g ? wide_g : (h->pipe ? h->pipe : groups)); // workgroups per instance
    HIP_TRY(h->P.alloc(B * GG * nnzp * h->fsize));
    const size_t LC = static_cast<size_t>(std::max(m, n));
    h->lcap = static_cast<int>(LC);
    HIP_TRY(h->vwork.alloc(B * WG * LC));
    HIP_TRY(h->vpart.alloc(B * GG * m));
    {
        const size_t slots = h->pipe ? static_cast<size_t>(bxh::PIPE_RING) : 2;
        HIP_TRY(h->xbuf.alloc(B * slots * rmax * 2));
        HIP_TRY(hipMemset(h->xbuf.p, 0, B * slots * rmax * 2 * sizeof(unsigned long long)));
        HIP_TRY(h->msg.alloc(B * static_cast<size_t>(m) * 2));
        HIP_TRY(hipMemset(h->msg.p, 0, B * static_cast<size_t>(m) * 2 * sizeof(unsigned long long)));
    }
    HIP_TRY(h->cnt.alloc(2 * B + 1));
    HIP_TRY(hipMemset(h->cnt.p, 0, (2 * B + 1) * sizeof(unsigned)));
    HIP_TRY(h->pi.alloc(B * m * h->fsize));
    HIP_TRY(h->rstale.alloc(B * (rmax + 1) * h->fsize));
    HIP_TRY(h->x.alloc(B * 2 * h->nwords));
    HIP_TRY(h->vlist.alloc(B * LC));
    const size_t zbytes = zsolver ? ((bxh::zscratch_size(rmax, static_cast<int>(h->fsize)) + 255) & ~size_t{ 255 }) : 0;
    if (zsolver)
        HIP_TRY(h->zscratch.alloc(B * zbytes));
    {
        const size_t n2cap = static_cast<size_t>(bxh::next_pow2(static_cast<int>(std::max<size_t>(LC, 2))));
        HIP_TRY(h->mlist.alloc(B * LC));
        HIP_TRY(h->mtmp.alloc(B * LC));
        HIP_TRY(h->mcount.alloc(B));
        HIP_TRY(h->mrun.alloc(B * LC));
        HIP_TRY(h->slist.alloc(B * LC));
        HIP_TRY(h->vamt.alloc(B * m));
        HIP_TRY(h->skey.alloc(B * n2cap));
        HIP_TRY(h->sid.alloc(B * n2cap));
        HIP_TRY(h->order_b.alloc(B));
        HIP_TRY(hipMemset(h->order_b.p, 0, B * sizeof(int)));
        HIP_TRY(hipMemset(h->mlist.p, 0, B * LC * sizeof(int2)));
        HIP_TRY(hipMemset(h->mcount.p, 0, B * sizeof(int)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_order), B * sizeof(int),
                              hipHostMallocDefault));
        h->ostate.assign(B, 0);
    }
    HIP_TRY(h->vcount.alloc(B));
    HIP_TRY(h->vflag.alloc(B * GG * m));
    HIP_TRY(h->status.alloc(B));
    HIP_TRY(h->prof.alloc(B * 32));
    HIP_TRY(hipMemset(h->prof.p, 0, B * 32 * sizeof(unsigned long long)));
    HIP_TRY(h->kappa.alloc(B));
    HIP_TRY(h->active.alloc(B));
    HIP_TRY(hipMemset(h->P.p, 0, B * GG * nnzp * h->fsize));
    HIP_TRY(hipMemset(h->pi.p, 0, B * m * h->fsize));
    HIP_TRY(hipMemset(h->rstale.p, 0, B * (rmax + 1) * h->fsize));
    HIP_TRY(hipMemset(h->x.p, 0, B * 2 * h->nwords * sizeof(unsigned)));
    HIP_TRY(hipMemset(h->vcount.p, 0, B * sizeof(int)));
    HIP_TRY(hipMemset(h->status.p, 0, B * sizeof(bxh::DevStatus)));
    // (the state arrays above must be zeros even when BXH_POISON filled the
    // allocations; those memsets are null-stream work, which the handle's own
    // non-blocking stream does not wait for)
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_status),
                          B * sizeof(bxh::DevStatus),
                          hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(
      reinterpret_cast<void**>(&h->h_kappa), B * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(
      reinterpret_cast<void**>(&h->h_active), B, hipHostMallocDefault));
    {
        // every worker's engine starts as minstd_rand0(1) until bxh_set_rng
        std::vector<bxh::DevStatus> s(B);
        for (auto& v : s) {
            std::memset(&v, 0, sizeof v);
            v.rng = 1;
        }
        HIP_TRY(hipMemcpy(h->status.p,
                          s.data(),
                          B * sizeof(bxh::DevStatus),
                          hipMemcpyHostToDevice));
    }

    if (wide_g) {
        std::vector<int2> cd(static_cast<size_t>(n));
        for (int j = 0; j != n; ++j)
            cd[j] = make_int2(col_start[j], std::max(16, (col_len[j] + 15) & ~15));
        HIP_TRY(h->col_desc.upload(cd));
        const size_t Gz = static_cast<size_t>(wide_g);
        size_t hs = 1024;
        while (hs < 4 * static_cast<size_t>(rmax))
            hs <<= 1;
        const size_t n2big = rmax > bxh::WIDE_SMALL ? static_cast<size_t>(bxh::next_pow2(rmax)) : 0;
        HIP_TRY(h->piw.alloc((B * Gz * (static_cast<size_t>(m) + 1) + 2 * static_cast<size_t>(h->wp.ring) + 64) * h->fsize)); // (+ slack: the window's refill reads whole quarters)
        HIP_TRY(h->whash.alloc(B * hs));
        HIP_TRY(h->wrec.alloc(B * Gz * 8));
        HIP_TRY(h->wcnt.alloc(4 * B + 1));
        // empty hash slots and arrival counters are zeros (also under BXH_POISON):
        // cleared on the handle's stream, in order with its first launch
        HIP_TRY(hipMemsetAsync(h->whash.p, 0, B * hs * sizeof(unsigned long long), h->own_stream));
        HIP_TRY(hipMemsetAsync(h->wcnt.p, 0, (4 * B + 1) * sizeof(unsigned), h->own_stream));
        HIP_TRY(h->vbuf.alloc(B * static_cast<size_t>(rmax)));
        HIP_TRY(h->bkey.alloc(B * (Gz + 2) * n2big));
        HIP_TRY(h->bid.alloc(B * (Gz + 2) * n2big));
        HIP_TRY(h->vflag32.alloc(B * static_cast<size_t>(m)));
        if (batch <= 32 && !std::getenv("BXH_WIDE_OWN_SCAN"))
            HIP_TRY(h->xt.alloc(static_cast<size_t>(n)));
        {
            // the band scan: up to 16 instances of a 01 matrix whose rows list their columns in ascending order
            bool sorted = !pm1 && batch <= 16 && h->xt.p != nullptr;
            for (int k = 0; sorted && k != m; ++k)
                for (int e = pb->row_ptr[k] + 1; e < pb->row_ptr[k + 1]; ++e)
                    if (pb->col[e] <= pb->col[e - 1]) {
                        sorted = false;
                        break;
                    }
            const char* eb = std::getenv("BXH_WIDE_BAND_SCAN");
            if (sorted && !(eb && eb[0] == '0'))
                HIP_TRY(h->vsum.alloc(static_cast<size_t>(m) * 16));
        }
        HIP_TRY(h->wcount.alloc(B * hs)); // zeros: no key has a duplicate, no tied key is listed
        HIP_TRY(h->wdl.alloc(B * static_cast<size_t>(bxh::WIDE_TDL)));
        HIP_TRY(hipMemsetAsync(h->wcount.p, 0, B * hs * sizeof(unsigned), h->own_stream));
        HIP_TRY(hipMemsetAsync(h->wdl.p, 0, B * static_cast<size_t>(bxh::WIDE_TDL) * sizeof(int), h->own_stream));
        bxh::WideParams& wp = h->wp;
        wp.G = wide_g;
        wp.chmax = wide_chmax;
        wp.hmask = static_cast<unsigned>(hs - 1);
        wp.n2big = static_cast<int>(n2big);
        wp.rows_per_piece = static_cast<float>(static_cast<double>(m) /
                                               (static_cast<double>(nnzp) / 16.0 / static_cast<double>(n)));
        wp.col_desc = h->col_desc.p;
        wp.piw = h->piw.p;
        wp.whash = h->whash.p;
        wp.wrec = h->wrec.p;
        wp.wcnt = h->wcnt.p;
        wp.vbuf = h->vbuf.p;
        wp.bkey = h->bkey.p;
        wp.bid = h->bid.p;
        wp.vflag32 = h->vflag32.p;
        wp.xt = h->xt.p;
        wp.skip_scan = h->xt.p ? 1 : 0;
        wp.own_sort = std::getenv("BXH_WIDE_OWN_SORT") ? 1 : 0;
        wp.wcount = h->wcount.p;
        wp.wdl = h->wdl.p;
        {
            const char* et = std::getenv("BXH_WIDE_TIEDFAST"); // "0": every tied row is sorted (tests, A/B)
            wp.tied_fast = (et && et[0] == '0') ? 0 : 1;
        }
    }

    HIP_TRY(set_lds_limits(device));

    // The forms with several workgroups per instance wait for their partners
    // inside the launch: every workgroup of the grid must be resident at once.
    // Ask the runtime how many workgroups of this kernel one CU takes.
    if (wide_g || h->pipe || groups > 1) {
        int per_cu = 0;
        const int cus = device_cus(device);
        const void* fn = nullptr;
        if (wide_g)
            fn = float_type == BXH_FLOAT
                   ? (pm1 ? reinterpret_cast<const void*>(&bxh::sweep_kernel_wide<float, true>)
                          : reinterpret_cast<const void*>(&bxh::sweep_kernel_wide<float, false>))
                   : (pm1 ? reinterpret_cast<const void*>(&bxh::sweep_kernel_wide<double, true>)
                          : reinterpret_cast<const void*>(&bxh::sweep_kernel_wide<double, false>));
        else if (h->pipe) // (every NP instantiation has the same footprint: 512 threads, the same LDS request)
            fn = float_type == BXH_FLOAT
                   ? reinterpret_cast<const void*>(&bxh::sweep_kernel_v2<float, false, 8, false, true>)
                   : reinterpret_cast<const void*>(&bxh::sweep_kernel_v2<double, false, 8, false, true>);
        else
            fn = float_type == BXH_FLOAT
                   ? reinterpret_cast<const void*>(&bxh::sweep_kernel_v2<float, false, 8, true>)
                   : reinterpret_cast<const void*>(&bxh::sweep_kernel_v2<double, false, 8, true>);
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, h->threads, h->lds_bytes));
        const long long grid = static_cast<long long>(batch) * (wide_g ? wide_g : (h->pipe ? h->pipe : groups));
        if (static_cast<long long>(per_cu) * cus < grid)
            return fail(BXH_ERR_UNSUPPORTED,
                        "the cooperating workgroups of all instances would not be resident "
                        "together on this device");
    }

    bxh::DevProblem& dp = h->dp;
    dp.m = m;
    dp.n = n;
    dp.nnz = nnz;
    dp.nnzp = nnzp;
    dp.groups = groups;
    dp.romax = romax;
    dp.spare = h->spare;
    dp.pipe = h->pipe;
    dp.lookahead = h->lookahead;
    dp.sw_chain = h->sw_chain;
    dp.sw_prod = h->sw_prod;
    dp.static_late = 1;
    if (const char* es = std::getenv("BXH_STATIC_LATE")) // (diagnostic: 0 keeps the column marks on every row list)
        dp.static_late = std::atoi(es) != 0;
    dp.gather_other = 256;
    if (const char* eo = std::getenv("BXH_PIPE_GATHER"))
        dp.gather_other = std::max(0, std::atoi(eo));
    dp.sub_ptr = h->sub_ptr.p;
    dp.sub_e = h->sub_e.p;
    dp.sub_desc = h->sub_desc.p;
    dp.desc_cost = h->desc_cost.p;
    dp.chain_rec = h->chain_rec.p;
    dp.ownmask = h->ownmask.p;
    dp.rmax = rmax;
    dp.cmax = cmax;
    dp.nwords = h->nwords;
    dp.kind = kind;
    dp.maximize = h->maximize;
    dp.row_ptr = h->row_ptr.p;
    dp.row_col = h->row_col.p;
    dp.row_pos = h->row_pos.p;
    dp.row_cs = h->row_cs.p;
    dp.col_start = h->col_start.p;
    dp.col_len = h->col_len.p;
    dp.col_row = h->col_row.p;
    dp.col_row16 = h->col_row16.p;
    dp.fac_csr = h->fac_csr.p;
    dp.fac32 = h->fac32.p;
    dp.col_fac = h->col_fac.p;
    dp.zkind = h->zkind.p;
    dp.ex_start = h->ex_start.p;
    dp.ex_count = h->ex_count.p;
    dp.ex_flat = h->ex_flat.p;
    dp.zconf = h->zconf.p;
    dp.zdesc = h->zdesc.p;
    dp.zcolrf = h->zcolrf.p;
    dp.zcolp = h->zcolp.p;
    dp.zwave = h->zwave;
    dp.bmin = h->bmin.p;
    dp.bmax = h->bmax.p;
    dp.bneg = h->bneg.p;
    dp.cost = h->cost.p;
    dp.q_ptr = pb->n_quad > 0 ? h->q_ptr.p : nullptr;
    dp.q_id = h->q_id.p;
    dp.q_factor = h->q_factor.p;
    dp.row_bytes = h->row_bytes.p;
    dp.lcg_pow = h->lcg_pow.p;

    bxh::DevState& ds = h->ds;
    ds.P = h->P.p;
    ds.pi = h->pi.p;
    ds.x = h->x.p;
    ds.rstale = h->rstale.p;
    ds.vlist = h->vlist.p;
    ds.vcount = h->vcount.p;
    ds.vflag = h->vflag.p;
    ds.status = h->status.p;
    ds.prof = h->prof.p;
    ds.vwork = h->vwork.p;
    ds.lcap = h->lcap;
    ds.mlist = h->mlist.p;
    ds.mcount = h->mcount.p;
    ds.mrun = h->mrun.p;
    ds.mtmp = h->mtmp.p;
    ds.slist = h->slist.p;
    ds.vamt = h->vamt.p;
    ds.skey = h->skey.p;
    ds.sid = h->sid.p;
    ds.zscratch = h->zscratch.p;
    ds.zscratch_bytes = zbytes;
    ds.vpart = h->vpart.p;
    ds.xbuf = h->xbuf.p;
    ds.cnt = h->cnt.p;
    ds.msg = h->msg.p;

    *out = guard.release();
    return BXH_OK;
}

// The same constructor for a matrix that already lives on the device (and may
// be too large for any host loop: the XL shape has 2e9 elements): every array
// of `pb` is a DEVICE pointer on `device`.  The CSC view, the bounds and the
// accounting table are built by kernels (bxh_wide_build.hpp); the handle runs
// the wide kernel.  01 / 101 solver kinds, linear objective.
int
bxh_create_device(const bxh_problem* pb,
                  int kind,
                  int float_type,
                  int maximize,
                  int batch,
                  int device,
                  bxh_handle* out)
{
    if (!pb || !out)
        return fail(BXH_ERR_INVALID, "null argument");
    *out = nullptr;
    if (pb->m <= 0 || pb->n <= 0 || pb->nnz <= 0 || batch <= 0)
        return fail(BXH_ERR_INVALID, "empty problem or batch");
    if (!pb->row_ptr || !pb->col || !pb->cmin || !pb->cmax || !pb->cost)
        return fail(BXH_ERR_INVALID, "null problem array");
    if (kind < BXH_EQ_01 || kind > BXH_INEQ_101)
        return fail(BXH_ERR_UNSUPPORTED, "bxh_create_device: 01 and 101 solver kinds only");
    if (float_type != BXH_FLOAT && float_type != BXH_DOUBLE)
        return fail(BXH_ERR_UNSUPPORTED, "only float and double exist on the device");
    if (pb->n_quad != 0)
        return fail(BXH_ERR_UNSUPPORTED, "bxh_create_device: linear objective only");
    if (pb->nnz > 0x7fffffffLL)
        return fail(BXH_ERR_UNSUPPORTED, "more than 2^31-1 elements");
    hipError_t e0 = hipSetDevice(device);
    if (e0 != hipSuccess)
        return fail(BXH_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e0));
    // The caller's arrays may have been produced on any stream (a copy from
    // pageable host memory returns before the data is on the device): wait for
    // everything queued on the device before the build kernels, which run on the
    // handle's own non-blocking stream, read them.
    {
        hipError_t es = hipDeviceSynchronize();
        if (es != hipSuccess)
            return fail(BXH_ERR_DEVICE, std::string("hipDeviceSynchronize: ") + hipGetErrorString(es));
    }

    bxh_handle h = new (std::nothrow) bxh_handle_s;
    if (!h)
        return fail(BXH_ERR_OOM, "host allocation failed");
    std::unique_ptr<bxh_handle_s> guard(h);
    h->device = device;
    h->kind = kind;
    h->float_type = float_type;
    h->maximize = maximize ? 1 : 0;
    h->batch = batch;
    h->m = pb->m;
    h->n = pb->n;
    h->nnz = static_cast<int>(pb->nnz);
    h->nwords = pb->n / 64 + 1;
    h->fsize = float_type == BXH_FLOAT ? 4 : 8;
    h->device_built = true;
    h->groups = 1;
    const int m = h->m, n = h->n;
    const size_t nnz = static_cast<size_t>(pb->nnz);
    const bool pm1 = kind == BXH_EQ_101 || kind == BXH_INEQ_101;

    HIP_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    hipStream_t s = h->stream;

    // ---- the CSR as given (the handle keeps its own copy) -------------------
    HIP_TRY(h->row_ptr.alloc(static_cast<size_t>(m) + 1));
    HIP_TRY(hipMemcpyAsync(h->row_ptr.p, pb->row_ptr, (static_cast<size_t>(m) + 1) * 4,
                           hipMemcpyDeviceToDevice, s));
    HIP_TRY(h->row_col.alloc(nnz));
    HIP_TRY(hipMemcpyAsync(h->row_col.p, pb->col, nnz * 4, hipMemcpyDeviceToDevice, s));
    {
        int ends[2] = { -1, -1 };
        HIP_TRY(hipMemcpyAsync(&ends[0], pb->row_ptr, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&ends[1], pb->row_ptr + m, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (ends[0] != 0 || static_cast<long long>(ends[1]) != pb->nnz)
            return fail(BXH_ERR_INVALID, "row_ptr does not span [0, nnz]");
    }
    DevBuf<int> scal; // [0] rmax, [1] error bits
    HIP_TRY(scal.alloc(2));
    HIP_TRY(hipMemsetAsync(scal.p, 0, 8, s));
    hipLaunchKernelGGL(bxh::build::row_stats_kernel, dim3(2048), dim3(256), 0, s, h->row_ptr.p,
                       h->row_col.p, pb->factor, m, n, pm1 ? 1 : 0, scal.p, scal.p + 1);
    HIP_TRY(hipGetLastError());

    // ---- column lengths: counts per (row block, column) ---------------------
    const int nb = std::max(1, std::min(256, m / 16));
    const int rpb = (m + nb - 1) / nb;
    DevBuf<unsigned> cnt;
    HIP_TRY(cnt.alloc(static_cast<size_t>(nb) * n));
    HIP_TRY(hipMemsetAsync(cnt.p, 0, static_cast<size_t>(nb) * n * 4, s));
    hipLaunchKernelGGL(bxh::build::count_kernel, dim3(nb), dim3(1024), 0, s, h->row_ptr.p,
                       h->row_col.p, m, n, rpb, cnt.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(h->col_len.alloc(static_cast<size_t>(n)));
    hipLaunchKernelGGL(bxh::build::col_len_kernel, dim3((n + 255) / 256), dim3(256), 0, s, cnt.p, nb,
                       n, h->col_len.p);
    HIP_TRY(hipGetLastError());
    std::vector<int> col_len(static_cast<size_t>(n));
    HIP_TRY(hipMemcpyAsync(col_len.data(), h->col_len.p, static_cast<size_t>(n) * 4,
                           hipMemcpyDeviceToHost, s));
    int scal_h[2] = { 0, 0 };
    HIP_TRY(hipMemcpyAsync(scal_h, scal.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (scal_h[1] & 1)
        return fail(BXH_ERR_INVALID,
                    "empty row, column index out of range or a coefficient this solver kind does not allow");
    const bool rows_ascending = (scal_h[1] & 16) == 0;
    const int rmax = scal_h[0];
    h->rmax = rmax;
    // every column padded to a multiple of 16 entries (a whole number of
    // 128-byte lines of fp64 P), as in bxh_create
    std::vector<int2> cd(static_cast<size_t>(n));
    long long at = 0;
    int cmax = 0;
    for (int j = 0; j != n; ++j) {
        const int padded = std::max(16, (col_len[j] + 15) & ~15);
        cd[j] = make_int2(static_cast<int>(at), padded);
        at += padded;
        cmax = std::max(cmax, col_len[j]);
        if (at > 0x7fffffffLL)
            return fail(BXH_ERR_UNSUPPORTED, "more than 2^31-1 padded elements");
    }
    const size_t nnzp = static_cast<size_t>(at);
    h->nnzp = static_cast<int>(at);
    HIP_TRY(h->col_desc.upload(cd));

    // ---- the column view: rows ascending inside every column -----------------
    hipLaunchKernelGGL(bxh::build::bases_kernel, dim3((n + 255) / 256), dim3(256), 0, s, cnt.p, nb, n,
                       h->col_desc.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(h->col_row.alloc(nnzp));
    hipLaunchKernelGGL(bxh::build::fill_int_kernel, dim3(4096), dim3(256), 0, s, h->col_row.p, nnzp,
                       m); // padding: row m (pi[m] == 0)
    HIP_TRY(hipGetLastError());
    HIP_TRY(h->row_pos.alloc(nnz));
    if (pm1)
        HIP_TRY(h->fac_csr.alloc(nnz));
    hipLaunchKernelGGL(bxh::build::fill_kernel, dim3(nb), dim3(1024), 0, s, h->row_ptr.p, h->row_col.p,
                       pb->factor, m, n, rpb, cnt.p, h->col_row.p, h->row_pos.p,
                       pm1 ? h->fac_csr.p : nullptr);
    HIP_TRY(hipGetLastError());

    // ---- bounds, accounting, costs -------------------------------------------
    HIP_TRY(h->bmin.alloc(static_cast<size_t>(m)));
    HIP_TRY(h->bmax.alloc(static_cast<size_t>(m)));
    HIP_TRY(h->bneg.alloc(static_cast<size_t>(m)));
    HIP_TRY(h->row_bytes.alloc(static_cast<size_t>(m)));
    hipLaunchKernelGGL(bxh::build::bounds_kernel, dim3(2048), dim3(256), 0, s, h->row_ptr.p,
                       h->row_col.p, pb->factor, pb->cmin, pb->cmax, h->col_len.p, m, kind,
                       static_cast<double>(h->fsize), h->bmin.p, h->bmax.p, h->bneg.p, h->row_bytes.p,
                       scal.p + 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(h->cost.alloc(static_cast<size_t>(n) * h->fsize));
    if (float_type == BXH_FLOAT)
        hipLaunchKernelGGL(bxh::build::cost_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s,
                           pb->cost, reinterpret_cast<float*>(h->cost.p), n);
    else
        hipLaunchKernelGGL(bxh::build::cost_kernel<double>, dim3((n + 255) / 256), dim3(256), 0, s,
                           pb->cost, reinterpret_cast<double*>(h->cost.p), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(scal_h, scal.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (scal_h[1] & 2)
        return fail(BXH_ERR_INVALID, "an equalities solver needs min == max in every constraint");
    if (scal_h[1] & ~16)
        return fail(BXH_ERR_INVALID, "constraint with min > max");
    cnt.release(); // (1 GB at the XL shape: gone before the instances are allocated)
    {
        std::vector<unsigned> pw(bxh::WIDE_LCG_POW); // (the short rows need 1026 of them)
        unsigned long long v = 1;
        for (size_t k = 0; k != pw.size(); ++k) {
            pw[k] = static_cast<unsigned>(v);
            v = v * 16807ull % 2147483647ull;
        }
        HIP_TRY(h->lcg_pow.upload(pw));
    }

    // ---- geometry of the wide kernel -----------------------------------------
    const int cus = device_cus(device);
    int wide_g = std::max(1, std::min({ cus / batch, rmax / 256, 64 })); // (a small batch: at most 64 per instance)
    if (const char* eg = std::getenv("BXH_GROUPS"))
        if (eg[0] == 'w' && std::atoi(eg + 1) > 0)
            wide_g = std::atoi(eg + 1);
    if (batch * wide_g > cus || wide_g > 64 || wide_g < 1)
        return fail(BXH_ERR_UNSUPPORTED,
                    "the cooperating workgroups of all instances must be resident "
                    "together (batch * workgroups per instance <= CUs)");
    const int wide_chmax =
      (((rmax + wide_g - 1) / wide_g) + bxh::WIDE_TILE - 1) & ~(bxh::WIDE_TILE - 1);
    int ring = 1024; // the pi window: the largest power of two that LDS still holds
    while (ring < m + 1 &&
           bxh::wide_lds(wide_g, wide_chmax, static_cast<int>(h->fsize), ring * 2).total <= 160 * 1024)
        ring *= 2;
    const bxh::WideLds wl = bxh::wide_lds(wide_g, wide_chmax, static_cast<int>(h->fsize), ring);
    if (wl.total > 160 * 1024)
        return fail(BXH_ERR_UNSUPPORTED,
                    "rows too long for the workgroups this batch leaves to an instance");
    h->version = 4;
    h->wide_g = wide_g;
    h->threads = bxh::WIDE_T;
    h->lds_bytes = static_cast<size_t>(wl.total);
    if (std::getenv("BXH_DEBUG"))
        std::fprintf(stderr, "bxh: device build m=%d n=%d nnz=%zu nnzp=%zu rmax=%d cmax=%d wide g=%d lds=%zu ring=%d\n",
                     m, n, nnz, nnzp, rmax, cmax, wide_g, h->lds_bytes, ring);

    // ---- per-instance state ------------------------------------------------------
    const size_t B = static_cast<size_t>(batch), Gz = static_cast<size_t>(wide_g);
    const size_t LC = static_cast<size_t>(m); // (row lists only: no pi_sign_change here)
    h->lcap = static_cast<int>(LC);
    HIP_TRY(h->P.alloc(B * nnzp * h->fsize));
    HIP_TRY(hipMemsetAsync(h->P.p, 0, B * nnzp * h->fsize, s));
    HIP_TRY(h->pi.alloc(B * m * h->fsize));
    HIP_TRY(hipMemsetAsync(h->pi.p, 0, B * m * h->fsize, s));
    HIP_TRY(h->rstale.alloc(B * (static_cast<size_t>(rmax) + 1) * h->fsize));
    HIP_TRY(hipMemsetAsync(h->rstale.p, 0, B * (static_cast<size_t>(rmax) + 1) * h->fsize, s));
    HIP_TRY(h->x.alloc(B * 2 * h->nwords));
    HIP_TRY(hipMemsetAsync(h->x.p, 0, B * 2 * h->nwords * sizeof(unsigned), s));
    HIP_TRY(h->vlist.alloc(B * LC));
    HIP_TRY(h->vwork.alloc(B * Gz * LC));
    HIP_TRY(h->vcount.alloc(B));
    HIP_TRY(hipMemsetAsync(h->vcount.p, 0, B * sizeof(int), s));
    HIP_TRY(h->mlist.alloc(B * LC));
    HIP_TRY(h->mcount.alloc(B));
    HIP_TRY(h->order_b.alloc(B));
    HIP_TRY(hipMemsetAsync(h->order_b.p, 0, B * sizeof(int), s));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_order), B * sizeof(int), hipHostMallocDefault));
    h->ostate.assign(B, 0);
    HIP_TRY(h->status.alloc(B));
    HIP_TRY(h->kappa.alloc(B));
    HIP_TRY(h->active.alloc(B));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_status), B * sizeof(bxh::DevStatus),
                          hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_kappa), B * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_active), B, hipHostMallocDefault));
    {
        std::vector<bxh::DevStatus> st(B);
        for (auto& v : st) {
            std::memset(&v, 0, sizeof v);
            v.rng = 1;
        }
        HIP_TRY(hipMemcpy(h->status.p, st.data(), B * sizeof(bxh::DevStatus), hipMemcpyHostToDevice));
    }
    size_t hs = 1024;
    while (hs < 4 * static_cast<size_t>(rmax))
        hs <<= 1;
    const size_t n2big = rmax > bxh::WIDE_SMALL ? static_cast<size_t>(bxh::next_pow2(rmax)) : 0;
    HIP_TRY(h->piw.alloc((B * Gz * (static_cast<size_t>(m) + 1) + 2 * static_cast<size_t>(ring) + 64) * h->fsize)); // (+ slack: the window's refill reads whole quarters)
    HIP_TRY(h->whash.alloc(B * hs));
    HIP_TRY(hipMemsetAsync(h->whash.p, 0, B * hs * sizeof(unsigned long long), s));
    HIP_TRY(h->wrec.alloc(B * Gz * 8));
    HIP_TRY(h->wcnt.alloc(4 * B + 1));
    HIP_TRY(hipMemsetAsync(h->wcnt.p, 0, (4 * B + 1) * sizeof(unsigned), s));
    HIP_TRY(h->wcount.alloc(B * hs)); // zeros: no key has a duplicate, no tied key is listed
    HIP_TRY(h->wdl.alloc(B * static_cast<size_t>(bxh::WIDE_TDL)));
    HIP_TRY(hipMemsetAsync(h->wcount.p, 0, B * hs * sizeof(unsigned), s));
    HIP_TRY(hipMemsetAsync(h->wdl.p, 0, B * static_cast<size_t>(bxh::WIDE_TDL) * sizeof(int), s));
    HIP_TRY(h->vbuf.alloc(B * static_cast<size_t>(rmax)));
    HIP_TRY(h->bkey.alloc(B * (Gz + 2) * n2big));
    HIP_TRY(h->bid.alloc(B * (Gz + 2) * n2big));
    HIP_TRY(h->vflag32.alloc(B * static_cast<size_t>(m)));
        if (batch <= 32 && !std::getenv("BXH_WIDE_OWN_SCAN"))
            HIP_TRY(h->xt.alloc(static_cast<size_t>(n)));
    {
        const char* eb = std::getenv("BXH_WIDE_BAND_SCAN"); // "0": the scan that gathers x out of L2
        if (rows_ascending && !pm1 && batch <= 16 && h->xt.p && !(eb && eb[0] == '0'))
            HIP_TRY(h->vsum.alloc(static_cast<size_t>(m) * 16));
    }
    HIP_TRY(hipStreamSynchronize(s));

    bxh::WideParams& wp = h->wp;
    wp.G = wide_g;
    wp.chmax = wide_chmax;
    wp.hmask = static_cast<unsigned>(hs - 1);
    wp.n2big = static_cast<int>(n2big);
    wp.ring = ring;
    wp.rows_per_piece = static_cast<float>(static_cast<double>(m) /
                                           (static_cast<double>(nnzp) / 16.0 / static_cast<double>(n)));
    wp.col_desc = h->col_desc.p;
    wp.piw = h->piw.p;
    wp.whash = h->whash.p;
    wp.wrec = h->wrec.p;
    wp.wcnt = h->wcnt.p;
    wp.vbuf = h->vbuf.p;
    wp.bkey = h->bkey.p;
    wp.bid = h->bid.p;
    wp.vflag32 = h->vflag32.p;
    wp.xt = h->xt.p;
    wp.skip_scan = h->xt.p ? 1 : 0;
    wp.own_sort = std::getenv("BXH_WIDE_OWN_SORT") ? 1 : 0;
    wp.wcount = h->wcount.p;
    wp.wdl = h->wdl.p;
    {
        const char* et = std::getenv("BXH_WIDE_TIEDFAST"); // "0": every tied row is sorted (tests, A/B)
        wp.tied_fast = (et && et[0] == '0') ? 0 : 1;
    }
    HIP_TRY(set_lds_limits(h->device));

    bxh::DevProblem& dp = h->dp;
    dp.m = m;
    dp.n = n;
    dp.nnz = h->nnz;
    dp.nnzp = h->nnzp;
    dp.groups = 1;
    dp.rmax = rmax;
    dp.cmax = cmax;
    dp.nwords = h->nwords;
    dp.kind = kind;
    dp.maximize = h->maximize;
    dp.row_ptr = h->row_ptr.p;
    dp.row_col = h->row_col.p;
    dp.row_pos = h->row_pos.p;
    dp.col_len = h->col_len.p;
    dp.col_row = h->col_row.p;
    dp.fac_csr = h->fac_csr.p;
    dp.bmin = h->bmin.p;
    dp.bmax = h->bmax.p;
    dp.bneg = h->bneg.p;
    dp.cost = h->cost.p;
    dp.row_bytes = h->row_bytes.p;
    dp.lcg_pow = h->lcg_pow.p;
    bxh::DevState& ds = h->ds;
    ds.P = h->P.p;
    ds.pi = h->pi.p;
    ds.x = h->x.p;
    ds.rstale = h->rstale.p;
    ds.vlist = h->vlist.p;
    ds.vcount = h->vcount.p;
    HIP_TRY(h->prof.alloc(B * 32)); // (diagnostic builds stamp phases here; zeros otherwise)
    HIP_TRY(hipStreamSynchronize(nullptr));
    ds.prof = h->prof.p;
    ds.status = h->status.p;
    ds.vwork = h->vwork.p;
    ds.lcap = h->lcap;
    ds.mlist = h->mlist.p;
    ds.mcount = h->mcount.p;

    *out = guard.release();
    return BXH_OK;
}

int
bxh_destroy(bxh_handle h)
{
    if (!h)
        return BXH_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    delete h;
    return BXH_OK;
}

int
bxh_set_stream(bxh_handle h, void* s)
{
    if (!h)
        return fail(BXH_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc = drain_events(h))
        return rc;
    h->stream = s ? static_cast<hipStream_t>(s) : h->own_stream;
    return BXH_OK;
}

int
bxh_batch(bxh_handle h)
{
    return h ? h->batch : BXH_ERR_INVALID;
}
int
bxh_nwords(bxh_handle h)
{
    return h ? h->nwords : BXH_ERR_INVALID;
}
int
bxh_rmax(bxh_handle h)
{
    return h ? h->rmax : BXH_ERR_INVALID;
}

int
bxh_sweep_includes_scan(bxh_handle h)
{
    if (!h)
        return BXH_ERR_INVALID;
    return (h->version == 4 && h->wp.skip_scan) ? 0 : 1;
}

int
bxh_kernel_form(bxh_handle h)
{
    if (!h)
        return BXH_ERR_INVALID;
    if (h->version == 3)
        return h->zwave ? 6 : 3;
    if (h->version == 7)
        return 7;
    if (h->version == 4)
        return 100 + h->wide_g;
    if (h->version == 5)
        return 5;
    if (h->version != 2)
        return 1;
    if (h->pipe)
        return 40 + h->pipe;
    return h->groups > 1 ? 20 + h->groups : 2;
}

int
bxh_set_x(bxh_handle h, int inst, const uint64_t* words)
{
    if (int rc = check_inst(h, inst))
        return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h->x.p + static_cast<size_t>(inst) * 2 * h->nwords,
                      words,
                      static_cast<size_t>(h->nwords) * 8,
                      hipMemcpyHostToDevice));
    return BXH_OK;
}

int
bxh_get_x(bxh_handle h, int inst, uint64_t* words)
{
    if (int rc = check_inst(h, inst))
        return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(words,
                      h->x.p + static_cast<size_t>(inst) * 2 * h->nwords,
                      static_cast<size_t>(h->nwords) * 8,
                      hipMemcpyDeviceToHost));
    return BXH_OK;
}

int
bxh_set_x_all(bxh_handle h, const uint64_t* words)
{
    if (!h || !words)
        return fail(BXH_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    // pageable host memory: the copy is staged before this returns; a device
    // pointer (the XL bench builds its restart points on the device) is copied
    // in stream order
    HIP_TRY(hipMemcpyAsync(h->x.p,
                           words,
                           static_cast<size_t>(h->batch) * h->nwords * 8,
                           hipMemcpyDefault,
                           h->stream));
    return BXH_OK;
}

int
bxh_get_x_all(bxh_handle h, uint64_t* words)
{
    if (!h || !words)
        return fail(BXH_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(words, h->x.p, static_cast<size_t>(h->batch) * h->nwords * 8,
                      hipMemcpyDeviceToHost));
    return BXH_OK;
}

int
bxh_set_rng_all(bxh_handle h, const uint32_t* state)
{
    if (!h || !state)
        return fail(BXH_ERR_INVALID, "null argument");
    for (int b = 0; b != h->batch; ++b)
        if (state[b] == 0 || state[b] >= 2147483647u)
            return fail(BXH_ERR_INVALID, "minstd_rand0 state must be in [1, 2^31-2]");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    // (the engine state is a field of the per-instance status block)
    HIP_TRY(hipMemcpy2D(&h->status.p[0].rng, sizeof(bxh::DevStatus), state, sizeof(uint32_t),
                        sizeof(uint32_t), static_cast<size_t>(h->batch), hipMemcpyHostToDevice));
    return BXH_OK;
}

int
bxh_pop_create(bxh_handle h, int slots)
{
    if (!h || slots <= 0)
        return fail(BXH_ERR_INVALID, "bxh_pop_create: null handle or no slots");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(h->pop.alloc((static_cast<size_t>(slots) + 2) * 2 * h->nwords));
    h->pop_slots = slots;
    return BXH_OK;
}

static int
check_slot(bxh_handle h, int slot)
{
    if (!h)
        return fail(BXH_ERR_INVALID, "null handle");
    if (h->pop_slots < 0)
        return fail(BXH_ERR_INVALID, "no population: call bxh_pop_create first");
    if (slot < 0 || slot > h->pop_slots + 1)
        return fail(BXH_ERR_INVALID, "population slot out of range");
    return BXH_OK;
}

int
bxh_pop_set(bxh_handle h, int slot, const uint64_t* words)
{
    if (int rc = check_slot(h, slot))
        return rc;
    if (!words)
        return fail(BXH_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    // after what is queued (restarts that read the slot as it was), and over when
    // the call returns (the caller's vector may change)
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h->pop.p + static_cast<size_t>(slot) * 2 * h->nwords, words,
                      static_cast<size_t>(h->nwords) * 8, hipMemcpyHostToDevice));
    return BXH_OK;
}

int
bxh_pop_get(bxh_handle h, int slot, uint64_t* words)
{
    if (int rc = check_slot(h, slot))
        return rc;
    if (!words)
        return fail(BXH_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(words, h->pop.p + static_cast<size_t>(slot) * 2 * h->nwords,
                      static_cast<size_t>(h->nwords) * 8, hipMemcpyDeviceToHost));
    return BXH_OK;
}

int
bxh_pop_store(bxh_handle h, int slot, int inst)
{
    if (int rc = check_slot(h, slot))
        return rc;
    if (int rc = check_inst(h, inst))
        return rc;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->pop.p + static_cast<size_t>(slot) * 2 * h->nwords,
                           h->x.p + static_cast<size_t>(inst) * 2 * h->nwords,
                           static_cast<size_t>(h->nwords) * 8, hipMemcpyDeviceToDevice, h->stream));
    return BXH_OK;
}

int
bxh_reinit(bxh_handle h, int njobs, const bxh_reinit_job* jobs)
{
    if (!h || (njobs > 0 && !jobs))
        return fail(BXH_ERR_INVALID, "null argument");
    if (njobs <= 0)
        return BXH_OK;
    static_assert(sizeof(bxh_reinit_job) == sizeof(bxh::ReinitJob), "job layouts differ");
    for (int q = 0; q != njobs; ++q) {
        const bxh_reinit_job& jb = jobs[q];
        if (jb.inst < 0 || jb.inst >= h->batch)
            return fail(BXH_ERR_INVALID, "bxh_reinit: instance out of range");
        if (jb.cross < 0 || jb.cross > 1)
            return fail(BXH_ERR_INVALID, "bxh_reinit: unknown crossover kind");
        if (jb.cross && (h->pop_slots < 0 || jb.first < 0 || jb.first > h->pop_slots + 1 ||
                         jb.second < 0 || jb.second > h->pop_slots + 1))
            return fail(BXH_ERR_INVALID, "bxh_reinit: population slot out of range");
        const uint32_t s0 = jb.cross ? jb.rng_cross : jb.rng_mut;
        if (s0 == 0 || s0 >= 2147483647u || jb.rng_mut == 0 || jb.rng_mut >= 2147483647u)
            return fail(BXH_ERR_INVALID, "minstd_rand0 state must be in [1, 2^31-2]");
        if (jb.mutate && !(jb.var_p > 0.0 && jb.var_p < 1.0 && jb.val_p >= 0.0 && jb.val_p <= 1.0))
            return fail(BXH_ERR_INVALID, "bxh_reinit: mutation probabilities out of range");
    }
    HIP_TRY(hipSetDevice(h->device));
    if (h->jobs_d.count < static_cast<size_t>(njobs)) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(h->jobs_d.alloc(static_cast<size_t>(std::max(njobs, h->batch))));
    }
    // (the jobs of the previous call have been read: the stream is in order and the
    // copy below is a blocking one from pageable memory)
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h->jobs_d.p, jobs, static_cast<size_t>(njobs) * sizeof(bxh_reinit_job),
                      hipMemcpyHostToDevice));
    hipLaunchKernelGGL(bxh::reinit_kernel, dim3(njobs), dim3(64), 0, h->stream, h->dp, h->ds, h->jobs_d.p,
                       h->pop.p);
    HIP_TRY(hipGetLastError());
    h->known_empty.resize(static_cast<size_t>(h->batch), 0);
    for (int q = 0; q != njobs; ++q)
        h->known_empty[static_cast<size_t>(jobs[q].inst)] = 0;
    return BXH_OK;
}

int
bxh_set_pi(bxh_handle h, int inst, const double* v)
{
    if (int rc = check_inst(h, inst))
        return rc;
    return set_vec(h, h->pi.p + static_cast<size_t>(inst) * h->m * h->fsize, v, h->m);
}

int
bxh_get_pi(bxh_handle h, int inst, double* v)
{
    if (int rc = check_inst(h, inst))
        return rc;
    return get_vec(h, h->pi.p + static_cast<size_t>(inst) * h->m * h->fsize, v, h->m);
}

int
bxh_set_P(bxh_handle h, int inst, const double* v)
{
    if (int rc = check_inst(h, inst))
        return rc;
    if (h->device_built) { // element order -> padded CSC order on the device, chunk by chunk
        const size_t chunk = size_t{ 1 } << 24, total = static_cast<size_t>(h->nnz);
        if (!h->stage_d.p)
            HIP_TRY(h->stage_d.alloc(chunk));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (size_t e0 = 0; e0 < total; e0 += chunk) {
            const size_t c = std::min(chunk, total - e0);
            HIP_TRY(hipMemcpy(h->stage_d.p, v + e0, c * sizeof(double), hipMemcpyHostToDevice));
            if (h->float_type == BXH_FLOAT)
                hipLaunchKernelGGL(bxh::build::scatter_P_kernel<float>, dim3(1024), dim3(256), 0, h->stream,
                                   reinterpret_cast<float*>(h->P.p) + static_cast<size_t>(inst) * h->nnzp,
                                   h->row_pos.p, e0, c, h->stage_d.p);
            else
                hipLaunchKernelGGL(bxh::build::scatter_P_kernel<double>, dim3(1024), dim3(256), 0, h->stream,
                                   reinterpret_cast<double*>(h->P.p) + static_cast<size_t>(inst) * h->nnzp,
                                   h->row_pos.p, e0, c, h->stage_d.p);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        return BXH_OK;
    }
    const size_t slice = static_cast<size_t>(h->nnzp), all = slice * h->groups;
    std::vector<double> csc(all, 0.0); // padding stays 0
    for (int e = 0; e != h->nnz; ++e)
        csc[static_cast<size_t>(h->h_owner[h->h_row_col[e]]) * slice + h->h_row_pos[e]] = v[e];
    return set_vec(h, h->P.p + static_cast<size_t>(inst) * all * h->fsize, csc.data(), all);
}

int
bxh_get_P(bxh_handle h, int inst, double* v)
{
    if (int rc = check_inst(h, inst))
        return rc;
    if (h->device_built) {
        const size_t chunk = size_t{ 1 } << 24, total = static_cast<size_t>(h->nnz);
        if (!h->stage_d.p)
            HIP_TRY(h->stage_d.alloc(chunk));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (size_t e0 = 0; e0 < total; e0 += chunk) {
            const size_t c = std::min(chunk, total - e0);
            if (h->float_type == BXH_FLOAT)
                hipLaunchKernelGGL(bxh::build::gather_P_kernel<float>, dim3(1024), dim3(256), 0, h->stream,
                                   reinterpret_cast<const float*>(h->P.p) + static_cast<size_t>(inst) * h->nnzp,
                                   h->row_pos.p, e0, c, h->stage_d.p);
            else
                hipLaunchKernelGGL(bxh::build::gather_P_kernel<double>, dim3(1024), dim3(256), 0, h->stream,
                                   reinterpret_cast<const double*>(h->P.p) + static_cast<size_t>(inst) * h->nnzp,
                                   h->row_pos.p, e0, c, h->stage_d.p);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(h->stream));
            HIP_TRY(hipMemcpy(v + e0, h->stage_d.p, c * sizeof(double), hipMemcpyDeviceToHost));
        }
        return BXH_OK;
    }
    const size_t slice = static_cast<size_t>(h->nnzp), all = slice * h->groups;
    std::vector<double> csc(all);
    if (int rc = get_vec(h, h->P.p + static_cast<size_t>(inst) * all * h->fsize, csc.data(), all))
        return rc;
    for (int e = 0; e != h->nnz; ++e)
        v[e] = csc[static_cast<size_t>(h->h_owner[h->h_row_col[e]]) * slice + h->h_row_pos[e]];
    return BXH_OK;
}

int
bxh_set_rng(bxh_handle h, int inst, uint32_t state)
{
    if (int rc = check_inst(h, inst))
        return rc;
    if (state == 0 || state >= 2147483647u)
        return fail(BXH_ERR_INVALID, "minstd_rand0 state must be in [1, 2^31-2]");
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(&h->status.p[inst].rng, &state, 4, hipMemcpyHostToDevice));
    return BXH_OK;
}

int
bxh_get_rng(bxh_handle h, int inst, uint32_t* state)
{
    if (int rc = check_inst(h, inst))
        return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(state, &h->status.p[inst].rng, 4, hipMemcpyDeviceToHost));
    return BXH_OK;
}

int
bxh_reset(bxh_handle h, int inst)
{
    if (int rc = check_inst(h, inst))
        return rc;
    HIP_TRY(hipSetDevice(h->device));
    // (stream-ordered like every sweep: a null-stream memset would not be waited
    // for by the handle's non-blocking stream)
    HIP_TRY(hipMemsetAsync(h->P.p + static_cast<size_t>(inst) * h->nnzp * h->groups * h->fsize,
                           0,
                           static_cast<size_t>(h->nnzp) * h->groups * h->fsize,
                           h->stream));
    HIP_TRY(hipMemsetAsync(h->pi.p + static_cast<size_t>(inst) * h->m * h->fsize,
                           0,
                           static_cast<size_t>(h->m) * h->fsize,
                           h->stream));
    return BXH_OK;
}

// Solver::reset() of every instance, stream-ordered
int
bxh_reset_all(bxh_handle h)
{
    if (!h)
        return fail(BXH_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    const size_t B = static_cast<size_t>(h->batch);
    HIP_TRY(hipMemsetAsync(h->P.p, 0, B * h->nnzp * h->groups * h->fsize, h->stream));
    HIP_TRY(hipMemsetAsync(h->pi.p, 0, B * h->m * h->fsize, h->stream));
    return BXH_OK;
}

// compute_order's constructor (itm-common.hpp:640-658): `cycle` starts as
// `none`; both lists are sized with the number of variables, m_order with
// value-initialised pairs (which a `cycle` run meets before they are first
// computed).
int
bxh_order_create(bxh_handle h, int order)
{
    if (!h)
        return fail(BXH_ERR_INVALID, "null handle");
    if (order < BXH_ORDER_NONE || order > BXH_ORDER_CYCLE)
        return fail(BXH_ERR_INVALID, "unknown constraint order");
    if ((order == BXH_ORDER_PI_SIGN_CHANGE || order == BXH_ORDER_CYCLE) && h->n > h->m)
        return fail(BXH_ERR_UNSUPPORTED,
                    "pi_sign_change walks row ids 0..n-1 (itm-common.hpp:672-674): "
                    "needs n <= m");
    if (h->version == 4 && order > BXH_ORDER_RANDOM_SORTING)
        return fail(BXH_ERR_UNSUPPORTED,
                    "the wide kernel walks the rows in the orders none / reversing / "
                    "random_sorting only");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->policy = order;
    h->use_cycle = order == BXH_ORDER_CYCLE;
    h->known_empty.assign(static_cast<size_t>(h->batch), 0);
    h->skipped.assign(static_cast<size_t>(h->batch), 0);
    std::fill(h->ostate.begin(), h->ostate.end(), h->use_cycle ? BXH_ORDER_NONE : order);
    const size_t B = static_cast<size_t>(h->batch), LC = static_cast<size_t>(h->lcap);
    HIP_TRY(hipMemsetAsync(h->mlist.p, 0, B * LC * sizeof(int2), h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<int> cnt(B, h->n);
    HIP_TRY(hipMemcpy(h->mcount.p, cnt.data(), B * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->order_b.p, h->ostate.data(), B * sizeof(int), hipMemcpyHostToDevice));
    h->ostate_dev = h->ostate;
    return BXH_OK;
}

// compute_order::init of the instances `active` selects (all when null)
int
bxh_order_init_masked(bxh_handle h, int order, const uint8_t* active, int32_t* remaining)
{
    if (!h)
        return fail(BXH_ERR_INVALID, "null handle");
    if (order != h->policy)
        if (int rc = bxh_order_create(h, order))
            return rc;
    HIP_TRY(hipSetDevice(h->device));
    h->known_empty.resize(static_cast<size_t>(h->batch), 0);
    h->skipped.resize(static_cast<size_t>(h->batch), 0);
    for (int b = 0; b != h->batch; ++b)
        if (!active || active[b]) {
            h->known_empty[static_cast<size_t>(b)] = 0;
            h->skipped[static_cast<size_t>(b)] = 0;
        }
    // (the device copy of the per-instance orders: written by bxh_order_create,
    // and by a sweep whenever a `cycle` order moved on)
    if (int rc = sync_order_b(h))
        return rc;
    const unsigned char* dev_active = nullptr;
    if (active) {
        HIP_TRY(hipStreamSynchronize(h->stream)); // the pinned buffer is reused
        std::memcpy(h->h_active, active, static_cast<size_t>(h->batch));
        HIP_TRY(hipMemcpyAsync(h->active.p, h->h_active, static_cast<size_t>(h->batch),
                               hipMemcpyHostToDevice, h->stream));
        dev_active = h->active.p;
    }
    if (h->version == 4) { // x, pi beyond LDS: flags from global memory, then the compaction
        if (h->xt.p) {
            if (int rc = launch_wide_scan(h, dev_active))
                return rc;
        } else {
            int parts = 1;
            while (h->batch * parts * 2 <= 1024 && h->m / (parts * 2) >= 8)
                parts *= 2;
            hipLaunchKernelGGL(bxh::wide_flags_kernel, dim3(h->batch * parts), dim3(512), 0, h->stream,
                               h->dp, h->ds, h->wp, dev_active, parts);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(bxh::wide_compact_kernel, dim3(h->batch), dim3(512), 0, h->stream, h->dp,
                               h->ds, h->wp, dev_active);
            HIP_TRY(hipGetLastError());
        }
        if (remaining) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            HIP_TRY(hipMemcpy(h->h_status, h->status.p,
                              static_cast<size_t>(h->batch) * sizeof(bxh::DevStatus),
                              hipMemcpyDeviceToHost));
            for (int b = 0; b != h->batch; ++b) {
                remaining[b] = h->h_status[b].remaining;
                h->known_empty[static_cast<size_t>(b)] =
                  h->h_status[b].remaining == 0 && h->ostate[static_cast<size_t>(b)] <= BXH_ORDER_RANDOM_SORTING;
            }
        }
        return BXH_OK;
    }
    // the row scan on several workgroups per instance when the batch alone
    // does not fill the device, then the ordered compaction per instance
    int parts = 1;
    while (parts < 8 && h->batch * parts * 2 <= 512 && h->m / (parts * 2) >= 64)
        parts *= 2;
    if (parts > 1) {
        hipLaunchKernelGGL(bxh::order_flags_kernel, dim3(h->batch * parts), dim3(1024),
                           static_cast<size_t>(h->nwords) * 8, h->stream, h->dp, h->ds,
                           h->order_b.p, dev_active, parts);
        HIP_TRY(hipGetLastError());
    }
    if (h->float_type == BXH_FLOAT)
        hipLaunchKernelGGL((bxh::order_init_kernel<float>),
                           dim3(h->batch),
                           dim3(h->threads),
                           h->lds_bytes_init,
                           h->stream,
                           h->dp,
                           h->ds,
                           h->order_b.p,
                           dev_active,
                           parts > 1 ? 1 : 0);
    else
        hipLaunchKernelGGL((bxh::order_init_kernel<double>),
                           dim3(h->batch),
                           dim3(h->threads),
                           h->lds_bytes_init,
                           h->stream,
                           h->dp,
                           h->ds,
                           h->order_b.p,
                           dev_active,
                           parts > 1 ? 1 : 0);
    HIP_TRY(hipGetLastError());
    if (remaining) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(h->h_status, h->status.p,
                          static_cast<size_t>(h->batch) * sizeof(bxh::DevStatus),
                          hipMemcpyDeviceToHost));
        for (int b = 0; b != h->batch; ++b) {
            remaining[b] = h->h_status[b].remaining;
            h->known_empty[static_cast<size_t>(b)] =
              h->h_status[b].remaining == 0 && h->ostate[static_cast<size_t>(b)] <= BXH_ORDER_RANDOM_SORTING;
        }
    }
    return BXH_OK;
}

int
bxh_order_init(bxh_handle h, int order, int32_t* remaining)
{
    return bxh_order_init_masked(h, order, nullptr, remaining);
}

// the list compute_order would walk next: m_order's rows for the
// infeasibility orders, R otherwise
int
bxh_get_violated(bxh_handle h, int inst, int32_t* rows, int32_t* count)
{
    if (int rc = check_inst(h, inst))
        return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int st = h->ostate[static_cast<size_t>(inst)];
    const bool by_m = st == BXH_ORDER_INFEASIBILITY_DECR || st == BXH_ORDER_INFEASIBILITY_INCR;
    int c = 0;
    HIP_TRY(hipMemcpy(&c, (by_m ? h->mcount.p : h->vcount.p) + inst, sizeof(int),
                      hipMemcpyDeviceToHost));
    if (count)
        *count = c;
    if (rows && c > 0) {
        if (by_m) {
            std::vector<int2> tmp(static_cast<size_t>(c));
            HIP_TRY(hipMemcpy(tmp.data(), h->mlist.p + static_cast<size_t>(inst) * h->lcap,
                              static_cast<size_t>(c) * sizeof(int2), hipMemcpyDeviceToHost));
            for (int i = 0; i != c; ++i)
                rows[i] = tmp[static_cast<size_t>(i)].x;
        } else {
            HIP_TRY(hipMemcpy(rows,
                              h->vlist.p + static_cast<size_t>(inst) * h->lcap,
                              static_cast<size_t>(c) * sizeof(int),
                              hipMemcpyDeviceToHost));
        }
    }
    return BXH_OK;
}

int
bxh_sweep_async(bxh_handle h, const bxh_sweep_args* a)
{
    if (!h || !a || !a->kappa)
        return fail(BXH_ERR_INVALID, "null argument");
    if (a->order != h->policy) // a sweep with another order: a new compute_order
        if (int rc = bxh_order_create(h, a->order))
            return rc;
    for (int i = 0; i != h->batch; ++i)
        if (!(a->kappa[i] >= 0.0)) // (the reference lets kappa pass 1 when kappa_max does)
            return fail(BXH_ERR_INVALID, "kappa must not be negative");
    HIP_TRY(hipSetDevice(h->device));

    bxh::DevSweep s;
    s.delta = a->delta;
    s.theta = a->theta;
    s.obj_amp = a->obj_amp;
    s.is_push = a->is_push;
    // push_and_run of a `cycle` order first moves to the next order
    // (itm-common.hpp:686-687); every instance keeps its own state
    bool need_prep = false, need_post = false;
    for (int b = 0; b != h->batch; ++b) {
        if (a->active && !a->active[b])
            continue;
        int& st = h->ostate[static_cast<size_t>(b)];
        if (h->use_cycle && a->is_push)
            st = st + 1 >= BXH_ORDER_CYCLE ? 0 : st + 1;
        need_prep = need_prep || (st >= BXH_ORDER_INFEASIBILITY_DECR && st <= BXH_ORDER_LAGRANGIAN_INCR);
        need_post = need_post || st == BXH_ORDER_INFEASIBILITY_DECR || st == BXH_ORDER_INFEASIBILITY_INCR;
    }
    // A sweep over an empty list R changes nothing (compute_order::run walks no
    // row and finds the same empty list again): when that is known of every
    // active instance -- the steady state of optimize()'s push phases -- nothing
    // is launched.
    {
        h->known_empty.resize(static_cast<size_t>(h->batch), 0);
        h->skipped.resize(static_cast<size_t>(h->batch), 0);
        bool all_empty = !std::getenv("BXH_NO_SKIP");
        for (int b = 0; b != h->batch && all_empty; ++b)
            if ((!a->active || a->active[b]) &&
                !(h->known_empty[static_cast<size_t>(b)] &&
                  h->ostate[static_cast<size_t>(b)] <= BXH_ORDER_RANDOM_SORTING))
                all_empty = false;
        for (int b = 0; b != h->batch; ++b)
            if (!a->active || a->active[b]) {
                h->skipped[static_cast<size_t>(b)] = all_empty ? 1 : 0;
                if (!all_empty)
                    h->known_empty[static_cast<size_t>(b)] = 0; // until the outcome is read back
            }
        if (all_empty)
            return BXH_OK;
    }
    // The call's uploads: kappa and the active mask in ONE copy from the next
    // pinned slot of the ring (the slot's previous copy, kStageSlots calls ago,
    // must have been consumed: no wait for the previous LAUNCH); the
    // per-instance orders only when they changed (a `cycle` order's push).
    {
        const size_t B = static_cast<size_t>(h->batch);
        if (!h->sargs.p) {
            HIP_TRY(h->sargs.alloc(B * 9));
            for (int i = 0; i != bxh_handle_s::kStageSlots; ++i) {
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_stage[i]), B * 9, hipHostMallocDefault));
                HIP_TRY(hipEventCreateWithFlags(&h->stage_done[i], hipEventDisableTiming));
            }
        }
        const int slot = static_cast<int>(h->stage_seq++ % bxh_handle_s::kStageSlots);
        if (h->stage_busy[slot])
            HIP_TRY(hipEventSynchronize(h->stage_done[slot]));
        unsigned char* hs = h->h_stage[slot];
        std::memcpy(hs, a->kappa, B * sizeof(double));
        size_t bytes = B * sizeof(double);
        s.active = nullptr;
        if (a->active) {
            std::memcpy(hs + bytes, a->active, B);
            s.active = h->sargs.p + bytes;
            bytes += B;
        }
        HIP_TRY(hipMemcpyAsync(h->sargs.p, hs, bytes, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipEventRecord(h->stage_done[slot], h->stream));
        h->stage_busy[slot] = true;
        s.kappa = reinterpret_cast<const double*>(h->sargs.p);
        if (int rc = sync_order_b(h))
            return rc;
        if (h->events_used >= 256) { // bound the list of timing events: fold the finished ones
            HIP_TRY(hipStreamSynchronize(h->stream));
            if (int rc = drain_events(h))
                return rc;
        }
    }
    s.order = 0;
    s.order_b = h->order_b.p;
    if (h->version == 4) // arrival counters of the wide kernel's workgroups (the error word stays)
        HIP_TRY(hipMemsetAsync(h->wcnt.p, 0, 4 * static_cast<size_t>(h->batch) * sizeof(unsigned),
                               h->stream));
    if (h->groups > 1 || h->pipe) { // arrival counters of the cooperating workgroups
        HIP_TRY(hipMemsetAsync(h->cnt.p, 0, 2 * static_cast<size_t>(h->batch) * sizeof(unsigned),
                               h->stream));
        if ((++h->epoch & 0x1ffffu) == 0) { // tag space wrapped: forget old granules
            HIP_TRY(hipMemsetAsync(h->xbuf.p, 0, h->xbuf.count * sizeof(unsigned long long),
                                   h->stream));
            HIP_TRY(hipMemsetAsync(h->msg.p, 0, h->msg.count * sizeof(unsigned long long),
                                   h->stream));
        }
    }
    s.epoch = h->epoch;
    if (need_prep) { // the sorts of the infeasibility / lagrangian orders
        if (h->float_type == BXH_FLOAT)
            hipLaunchKernelGGL((bxh::order_prep_kernel<float>), dim3(h->batch), dim3(1024), 0,
                               h->stream, h->dp, h->ds, s);
        else
            hipLaunchKernelGGL((bxh::order_prep_kernel<double>), dim3(h->batch), dim3(1024), 0,
                               h->stream, h->dp, h->ds, s);
        HIP_TRY(hipGetLastError());
    }
    if (int rc = h->float_type == BXH_FLOAT ? launch_sweep<float>(h, s)
                                             : launch_sweep<double>(h, s))
        return rc;
    if (h->version == 4 && h->wp.skip_scan) // compute_violated_constraints, all instances in one pass
        if (int rc = launch_wide_scan(h, s.active))
            return rc;
    if (need_post) { // m_order of the new x (itm-common.hpp:893-914)
        if (h->float_type == BXH_FLOAT)
            hipLaunchKernelGGL((bxh::order_post_kernel<float>), dim3(h->batch), dim3(h->threads),
                               h->lds_bytes_init, h->stream, h->dp, h->ds, s);
        else
            hipLaunchKernelGGL((bxh::order_post_kernel<double>), dim3(h->batch), dim3(h->threads),
                               h->lds_bytes_init, h->stream, h->dp, h->ds, s);
        HIP_TRY(hipGetLastError());
    }
    return BXH_OK;
}

int
bxh_run_supported(bxh_handle h)
{
    return (h && (h->version == 5 || h->version == 7)) ? 1 : 0;
}

int
bxh_run(bxh_handle h, const bxh_run_args* a, bxh_status* status_out, bxh_run_status* run_out)
{
    if (!h || !a || !a->kappa || !a->kappa_inc)
        return fail(BXH_ERR_INVALID, "null argument");
    if (h->version != 5 && h->version != 7)
        return fail(BXH_ERR_UNSUPPORTED, "bxh_run: only the short-row kernels run several sweeps per launch");
    if (a->order > BXH_ORDER_RANDOM_SORTING)
        return fail(BXH_ERR_UNSUPPORTED, "bxh_run: row orders none / reversing / random_sorting only");
    if (a->max_sweeps <= 0)
        return fail(BXH_ERR_INVALID, "bxh_run: max_sweeps must be positive");
    HIP_TRY(hipSetDevice(h->device));
    const size_t B = static_cast<size_t>(h->batch);
    if (!h->run_out.p) {
        HIP_TRY(h->run_out.alloc(B * 3));
        HIP_TRY(h->run_inc.alloc(static_cast<size_t>(h->m) + 1));
        HIP_TRY(h->run_xbest.alloc(B * 2 * h->nwords));
    }
    {
        const size_t ni = static_cast<size_t>(h->m) + 1;
        if (h->run_inc_host.size() != ni || std::memcmp(h->run_inc_host.data(), a->kappa_inc, ni * sizeof(double)) != 0) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            HIP_TRY(hipMemcpy(h->run_inc.p, a->kappa_inc, ni * sizeof(double), hipMemcpyHostToDevice));
            h->run_inc_host.assign(a->kappa_inc, a->kappa_inc + ni);
        }
    }
    h->nrun.max_sweeps = a->max_sweeps;
    h->nrun.first_i = a->first_i;
    h->nrun.w_limit = a->w_limit;
    h->nrun.kappa_max = a->kappa_max;
    h->nrun.kappa_inc = h->run_inc.p;
    h->nrun.out = reinterpret_cast<int*>(h->run_out.p);
    h->nrun.kappa_out = h->run_out.p + 2 * B;
    h->nrun.xbest = h->run_xbest.p;
    bxh_sweep_args sa{};
    sa.kappa = a->kappa;
    sa.delta = a->delta;
    sa.theta = a->theta;
    sa.obj_amp = 0.0;
    sa.is_push = 0;
    sa.order = a->order;
    sa.active = a->active;
    // (a launch of several sweeps is never skipped: its first list may be empty, the lists
    // behind it are not known here)
    h->known_empty.assign(B, 0);
    int rc = bxh_sweep_async(h, &sa);
    h->nrun = bxh::NarrowRun{};
    if (rc)
        return rc;
    if (int rc2 = bxh_status_sync(h, status_out))
        return rc2;
    if (run_out) {
        std::vector<double> blob(B * 3);
        HIP_TRY(hipMemcpy(blob.data(), h->run_out.p, B * 3 * sizeof(double), hipMemcpyDeviceToHost));
        const int* ro = reinterpret_cast<const int*>(blob.data());
        const double* ko = blob.data() + 2 * B;
        for (size_t b = 0; b != B; ++b) {
            run_out[b].sweeps = ro[b * 4];
            run_out[b].best_remaining = ro[b * 4 + 1];
            run_out[b].best_i = ro[b * 4 + 2];
            run_out[b].stop = ro[b * 4 + 3];
            run_out[b].kappa = ko[b];
        }
    }
    return BXH_OK;
}

int
bxh_get_best_x(bxh_handle h, int inst, uint64_t* words)
{
    if (int rc = check_inst(h, inst))
        return rc;
    if (!words || !h->run_xbest.p)
        return fail(BXH_ERR_INVALID, "bxh_get_best_x: no run yet");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(words, h->run_xbest.p + static_cast<size_t>(inst) * 2 * h->nwords,
                      static_cast<size_t>(h->nwords) * 8, hipMemcpyDeviceToHost));
    return BXH_OK;
}

int
bxh_status_sync(bxh_handle h, bxh_status* out)
{
    if (!h)
        return fail(BXH_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->h_status,
                           h->status.p,
                           static_cast<size_t>(h->batch) * sizeof(bxh::DevStatus),
                           hipMemcpyDeviceToHost,
                           h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc = drain_events(h))
        return rc;
    if (h->version == 4) {
        unsigned err = 0;
        HIP_TRY(hipMemcpy(&err, h->wcnt.p + 4 * h->batch, sizeof err, hipMemcpyDeviceToHost));
        if (err)
            return fail(BXH_ERR_DEVICE,
                        "a workgroup timed out waiting for its partners (the "
                        "cooperating workgroups were not all resident)");
    }
    if (h->groups > 1 || h->version == 3 || h->pipe) {
        unsigned err = 0;
        HIP_TRY(hipMemcpy(&err, h->cnt.p + 2 * h->batch, sizeof err, hipMemcpyDeviceToHost));
        if (err == 2u)
            return fail(BXH_ERR_DEVICE,
                        "the branch-and-bound search of a constraint exceeded its "
                        "node budget (branch-and-bound-solver.hpp has none)");
        if (err)
            return fail(BXH_ERR_DEVICE,
                        "a workgroup timed out waiting for its partners (the "
                        "cooperating workgroups were not all resident)");
    }
    h->known_empty.resize(static_cast<size_t>(h->batch), 0);
    h->skipped.resize(static_cast<size_t>(h->batch), 0);
    for (int b = 0; b != h->batch; ++b) {
        if (h->skipped[static_cast<size_t>(b)]) { // its last sweep was the no-op that was not launched
            h->h_status[b].pi_changed = 0;
            h->h_status[b].rows_in_sweep = 0;
        }
        h->known_empty[static_cast<size_t>(b)] =
          h->h_status[b].remaining == 0 && static_cast<size_t>(b) < h->ostate.size() &&
          h->ostate[static_cast<size_t>(b)] <= BXH_ORDER_RANDOM_SORTING;
    }
    if (out) {
        static_assert(sizeof(bxh_status) == sizeof(bxh::DevStatus), "ABI");
        std::memcpy(out, h->h_status, static_cast<size_t>(h->batch) * sizeof(bxh_status));
    }
    return BXH_OK;
}

int
bxh_sweep(bxh_handle h, const bxh_sweep_args* a, bxh_status* out)
{
    if (int rc = bxh_sweep_async(h, a))
        return rc;
    return bxh_status_sync(h, out);
}

int
bxh_set_objective(bxh_handle h, const double* cost_n, double constant, int maximize)
{
    if (!h || !cost_n)
        return fail(BXH_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = static_cast<size_t>(h->n), B = static_cast<size_t>(h->batch);
    if (!h->obj_cost.p) {
        HIP_TRY(h->obj_cost.alloc(n));
        HIP_TRY(h->inc_value.alloc(B));
        HIP_TRY(h->inc_rec.alloc(1));
        HIP_TRY(h->inc_x.alloc(2 * static_cast<size_t>(h->nwords)));
        HIP_TRY(hipStreamSynchronize(nullptr)); // (DevBuf::alloc's fills run on the null stream)
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h->obj_cost.p, cost_n, n * sizeof(double), hipMemcpyHostToDevice));
    h->obj_constant = constant;
    h->obj_maximize = maximize ? 1 : 0;
    h->obj_set = true;
    return BXH_OK;
}

int
bxh_incumbent(bxh_handle h, int rank, void* record_device, void* x_device, bxh_incumbent_record* record_host)
{
    static_assert(sizeof(bxh_incumbent_record) == sizeof(bxh::IncumbentRecord), "ABI");
    if (!h)
        return fail(BXH_ERR_INVALID, "null handle");
    if (!h->obj_set)
        return fail(BXH_ERR_INVALID, "bxh_incumbent: bxh_set_objective first");
    if (rank < 0 || rank >= (1 << bxh::kIncRankBits))
        return fail(BXH_ERR_INVALID, "bxh_incumbent: the key has room for ranks 0 .. 63");
    HIP_TRY(hipSetDevice(h->device));
    bxh::IncumbentRecord* rec =
      record_device ? static_cast<bxh::IncumbentRecord*>(record_device) : h->inc_rec.p;
    unsigned* xo = x_device ? static_cast<unsigned*>(x_device) : h->inc_x.p;
    hipLaunchKernelGGL(bxh::incumbent_value_kernel, dim3(h->batch), dim3(256), 0, h->stream, h->x.p, h->nwords,
                       h->n, h->obj_cost.p, h->obj_constant, h->inc_value.p);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(bxh::incumbent_pick_kernel, dim3(1), dim3(256), 0, h->stream, h->status.p,
                       h->inc_value.p, h->x.p, h->batch, h->nwords, h->obj_maximize, rank, rec, xo);
    HIP_TRY(hipGetLastError());
    if (record_host) {
        HIP_TRY(hipMemcpyAsync(record_host, rec, sizeof(bxh::IncumbentRecord), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return BXH_OK;
}

int
bxh_kernel_time(bxh_handle h, double* total_ms, uint64_t* launches, int reset)
{
    if (!h)
        return fail(BXH_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rc = drain_events(h))
        return rc;
    if (total_ms)
        *total_ms = h->kernel_ms;
    if (launches)
        *launches = h->kernel_launches;
    if (reset) {
        h->kernel_ms = 0.0;
        h->kernel_launches = 0;
    }
    return BXH_OK;
}

/* Diagnostic builds only (-DBXH_PROFILE): per instance 16 phase accumulators,
 * 4 barrier skews (last - first wave) and 4 waits of wave 0, in ticks of the
 * 100 MHz wall clock; zeros otherwise. */
int
bxh_debug_profile(bxh_handle h, uint64_t* ticks_batch_x32, int reset)
{
    if (!h || !ticks_batch_x32)
        return fail(BXH_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(ticks_batch_x32,
                      h->prof.p,
                      static_cast<size_t>(h->batch) * 32 * sizeof(unsigned long long),
                      hipMemcpyDeviceToHost));
    if (reset) {
        HIP_TRY(hipMemsetAsync(h->prof.p, 0, static_cast<size_t>(h->batch) * 32 * sizeof(unsigned long long), h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return BXH_OK;
}

int
bxh_row_bytes(bxh_handle h, double* bytes_m)
{
    if (!h || !bytes_m)
        return fail(BXH_ERR_INVALID, "null argument");
    if (h->device_built) { // computed by bounds_kernel
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpy(bytes_m, h->row_bytes.p, static_cast<size_t>(h->m) * sizeof(double),
                          hipMemcpyDeviceToHost));
        return BXH_OK;
    }
    const double F = static_cast<double>(h->fsize);
    for (int k = 0; k != h->m; ++k) {
        const int rs = h->h_row_ptr[k], re = h->h_row_ptr[k + 1];
        double b = 0.0;
        for (int e = rs; e != re; ++e) {
            const int j = h->h_row_col[e];
            b += static_cast<double>(h->h_col_ptr[j + 1] - h->h_col_ptr[j]) * (4.0 + 2.0 * F);
        }
        const double r = static_cast<double>(re - rs);
        if (h->kind == BXH_INEQ_Z) // the repair pass walks every column again; 4 B per coefficient
            b = 2.0 * b + 4.0 * r;
        b += r * (4.0 + 5.0 * F) + 2.0 * F + static_cast<double>((re - rs + 7) / 8) * 2.0;
        bytes_m[k] = b;
    }
    return BXH_OK;
}

} // extern "C"
