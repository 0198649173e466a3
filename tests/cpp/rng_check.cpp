#include "bxh_rng.hpp"
#include <random>
#include <cstdio>
#include <cstdint>
int main(){
  long bad=0;
  for (uint32_t seed : {1u, 12345u, 2147483646u, 987654321u, 42u}) {
    std::minstd_rand0 e(seed); bxh::Lcg g{ (uint32_t)(seed % 2147483647u) }; if (g.s==0) g.s=1;
    std::uniform_int_distribution<std::size_t> d64(0);
    for (int i=0;i<2000000;i++){ uint64_t a=d64(e), b=bxh::lcg_uniform_u64(g, ~0ull); if(a!=b){bad++; if(bad<5) printf("u64 mismatch seed %u i %d\n",seed,i);} }
    for (double p : {0.5, 0.0001, 0.3, 0.999, 1e-9}) { std::bernoulli_distribution bd(p);
      for (int i=0;i<300000;i++){ bool a=bd(e), b=bxh::lcg_bernoulli(g,p); if(a!=b){bad++; if(bad<5) printf("bern mismatch\n");} } }
    std::uniform_int_distribution<std::size_t> dsmall(0, 8589934596ull);
    for (int i=0;i<300000;i++){ uint64_t a=dsmall(e), b=bxh::lcg_uniform_u64(g, 8589934596ull); if(a!=b) bad++; }
    // engine states agree
    std::minstd_rand0 e2 = e; uint32_t se = (uint32_t)e2(); uint32_t sg = g.next(); if (se!=sg) { bad++; printf("state mismatch\n"); }
  }
  printf("bad %ld\n", bad); return bad!=0;
}
