// Stage timing of the teacher's row-chain kernel (csrc/chain.hip built with -DCHAIN_TIMING): wall_clock64 (100 MHz) marks of tile 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCHAIN_TIMING -I vln-magic_amd/csrc profiles/micro/chain_timing.hip vln-magic_amd/csrc/optim.o... (see below)
// Built as ONE translation unit with a stub for the group recorder:
#include "../../vln-magic_amd/csrc/chain.hip"
#include <cstdio>
GroupState& group_state() { static thread_local GroupState g = {}; return g; }
static void* dmalloc(size_t n, int fill) { void* p; hipMalloc(&p, n); hipMemset(p, fill, n); return p; }
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 3840;
  ChainParams P; memset(&P, 0, sizeof(P));
  P.M = M; P.ld_in = CH; P.Np = 3 * CH; P.eps = 1e-12f;
  P.in = dmalloc((size_t)M * CH * 2, 0x11); P.res = dmalloc((size_t)M * CH * 2, 0x11);
  P.Wa = dmalloc(CH * CH * 2, 0x11); P.ba = (float*)dmalloc(CH * 4, 0); P.g1 = (float*)dmalloc(CH * 4, 0x3c); P.b1 = (float*)dmalloc(CH * 4, 0);
  P.W1 = dmalloc(CI * CH * 2, 0x11); P.bi = (float*)dmalloc(CI * 4, 0); P.W2 = dmalloc(CI * CH * 2, 0x11); P.bo2 = (float*)dmalloc(CH * 4, 0);
  P.g2 = (float*)dmalloc(CH * 4, 0x3c); P.b2 = (float*)dmalloc(CH * 4, 0); P.y2 = dmalloc((size_t)M * CH * 2, 0);
  P.Wp = dmalloc(3 * CH * CH * 2, 0x11); P.bp = (float*)dmalloc(3 * CH * 4, 0); P.proj = dmalloc((size_t)M * 3 * CH * 2, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rows = 32; rows <= 64; rows += 32)
  for (int variant = 0; variant < 3; ++variant) {
    magic_chain_tile_rows(rows);
    ChainParams Q = P;
    if (variant == 1) { Q.Wp = nullptr; }                                  // no projection
    if (variant == 2) { Q.W1 = nullptr; Q.Np = CH; Q.y1 = Q.y2; }          // mini chain: stage 1 + H-wide projection
    for (int it = 0; it < 3; ++it) magic_chain_fwd(DT_BF16, &Q, sizeof(Q), nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) magic_chain_fwd(DT_BF16, &Q, sizeof(Q), nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long t[16]; hipMemcpyFromSymbol(t, HIP_SYMBOL(chain_ticks), sizeof(t));
    if (rows == 64) {
      printf("M %d rows 64 variant %d: %.1f us per launch; tile 0 (10 ns ticks): prologue %lld, stage-1 products %lld, LayerNorm 1 %lld, FFN (4 chunks) %lld, LayerNorm 2 %lld, y2 store + projection %lld, total %lld\n",
             M, variant, ms * 1000 / 20, t[1] - t[0], t[7] - t[1], t[2] - t[7], t[3] - t[2], t[4] - t[3], t[6] - t[4], t[6] - t[0]);
      continue;
    }
    printf("M %d variant %d (%s): %.1f us per launch; tile 0 (10 ns ticks): rows+params+ring prologue %lld, stage 1 %lld, 2a (8 chunks) %lld, 2b (8 chunks + LN) %lld, y2 store + 3 %lld, tail %lld, total %lld\n",
           M, variant, variant == 0 ? "full" : variant == 1 ? "no projection" : "mini", ms * 1000 / 20,
           t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[6] - t[0]);
  }
  return 0;
}
