// baryonyx_amd/csrc/bxh_rng.hpp -- std::minstd_rand0 and the libstdc++
// distributions the evolutionary layer of optimize draws from it
// (lib/src/itm-optimizer-common.hpp:57-63,359-371,494-526), restated so that
// the device reproduces the host's stream draw for draw.  From GCC 11's
// <bits/random.h> / <bits/random.tcc> (generate_canonical, bernoulli_distribution)
// and <bits/uniform_int_dist.h> (the up-scaling branch: the engine's range,
// 2^31 - 2 values, is smaller than the distribution's).  Plain C++ when not
// compiled by hipcc: tests/test_rng_restatement.py checks it against libstdc++
// itself on the host.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define BXH_HD __host__ __device__
#else
#define BXH_HD
#endif

namespace bxh {

struct Lcg // std::minstd_rand0
{
    uint32_t s;
    BXH_HD uint32_t next()
    {
        s = (uint32_t)(((uint64_t)s * 16807ull) % 2147483647ull);
        return s;
    }
};

// std::generate_canonical<double, 53>(g): two draws (log2 of the engine's range is 30.99..)
BXH_HD inline double
lcg_canonical(Lcg& g)
{
    double sum = 0.0;
    double tmp = 1.0;
    sum += (double)(g.next() - 1u) * tmp;
    tmp *= 2147483646.0;
    sum += (double)(g.next() - 1u) * tmp;
    tmp = 4611686009837453312.0; // double((long double)r * r)
    double ret = sum / tmp;
    if (ret >= 1.0)
        ret = 0.99999999999999988897769753748434595763683319091796875;
    return ret;
}

// std::bernoulli_distribution(p)(g)
BXH_HD inline bool
lcg_bernoulli(Lcg& g, double p)
{
    return lcg_canonical(g) < p;
}

// std::uniform_int_distribution<uint64_t>{0, urange}(g)
BXH_HD inline uint64_t
lcg_uniform_u64(Lcg& g, uint64_t urange)
{
    const uint64_t urngrange = 2147483645ull; // max() - min()
    uint64_t ret;
    if (urngrange > urange) {
        const uint64_t uerange = urange + 1;
        const uint64_t scaling = urngrange / uerange;
        const uint64_t past = uerange * scaling;
        do
            ret = (uint64_t)g.next() - 1ull;
        while (ret >= past);
        ret /= scaling;
    } else if (urngrange < urange) {
        uint64_t tmp; // wraparound control
        do {
            const uint64_t uerngrange = urngrange + 1;
            tmp = uerngrange * lcg_uniform_u64(g, urange / uerngrange);
            ret = tmp + ((uint64_t)g.next() - 1ull);
        } while (ret > urange || ret < tmp);
    } else
        ret = (uint64_t)g.next() - 1ull;
    return ret;
}

} // namespace bxh
