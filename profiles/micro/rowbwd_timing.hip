// Stage timing of the row-block backward (csrc/encbwd.hip built with -DRBW_TIMING): wall_clock64 (100 MHz) marks of workgroup 0, one text block
// of the headline step (M = 3840 rows: 240 workgroups of 16 rows; with an argument >= 4097 the launch takes 32-row tiles).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DRBW_TIMING -I vln-magic_amd/csrc profiles/micro/rowbwd_timing.hip -o profiles/micro/bin/rowbwd_timing
#include "../../vln-magic_amd/csrc/encbwd.hip"
#include <cstdio>
static void* dmalloc(size_t n, int fill) { void* p; hipMalloc(&p, n); hipMemset(p, fill, n); return p; }
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 3840;
  const float pdrop = argc > 2 ? atof(argv[2]) : 0.1f;
  RbwParams P; memset(&P, 0, sizeof(P));
  RbwSeg& s = P.seg[0];
  typedef bf16 T;
  s.M = M; s.kt = 12;
  s.dqkv_n = (T*)dmalloc((size_t)M * 384 * 2, 0x11); s.WqkvT_n = (T*)dmalloc(384 * 128 * 2, 0x11); s.dao_n = (T*)dmalloc((size_t)M * 128 * 2, 0x11);
  s.y2 = (T*)dmalloc((size_t)M * 128 * 2, 0x11); s.rstd2 = (float*)dmalloc((size_t)M * 4, 0x3c); s.g2 = (float*)dmalloc(512, 0x3c); s.b2 = (float*)dmalloc(512, 0);
  s.dg2 = (float*)dmalloc(512 * 512, 0); s.db2 = (float*)dmalloc(512 * 512, 0);
  s.z = (T*)dmalloc((size_t)M * 512 * 2, 0x11); s.W2T = (T*)dmalloc(128 * 512 * 2, 0x11); s.W1T = (T*)dmalloc(128 * 512 * 2, 0x11);
  s.y1 = (T*)dmalloc((size_t)M * 128 * 2, 0x11); s.rstd1 = (float*)dmalloc((size_t)M * 4, 0x3c); s.g1 = (float*)dmalloc(512, 0x3c); s.b1 = (float*)dmalloc(512, 0);
  s.dg1 = (float*)dmalloc(512 * 512, 0); s.db1 = (float*)dmalloc(512 * 512, 0);
  s.WoT = (T*)dmalloc(128 * 128 * 2, 0x11);
  s.dfo = (T*)dmalloc((size_t)M * 128 * 2, 0); s.dfod = (T*)dmalloc((size_t)M * 128 * 2, 0); s.dz = (T*)dmalloc((size_t)M * 512 * 2, 0);
  s.daod = (T*)dmalloc((size_t)M * 128 * 2, 0); s.dao = (T*)dmalloc((size_t)M * 128 * 2, 0); s.dctx = (T*)dmalloc((size_t)M * 128 * 2, 0);
  s.site_out = 3; s.site_ao = 4;
  P.nseg = 1; P.p_hidden = pdrop; P.pad1 = 1;            // partial-row LayerNorm gradients (the product's form)
  unsigned* seed = (unsigned*)dmalloc(16, 0x5a); P.seed = pdrop > 0.f ? seed : nullptr;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) if (magic_rowbwd(DT_BF16, &P, sizeof(P), nullptr)) { printf("launch failed\n"); return 1; }
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 50; ++it) magic_rowbwd(DT_BF16, &P, sizeof(P), nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long t[16]; hipMemcpyFromSymbol(t, HIP_SYMBOL(rbw_ticks), sizeof(t));
  printf("M %d (rows per workgroup %d), dropout %.2f: %.1f us per launch; workgroup 0 (10 ns ticks): stage rows + first weights %lld | tail product %lld | LayerNorm backward 1 + z image %lld | "
         "d_fo / d_fod out %lld | first FFN product + gelu' %lld | d_z out %lld | second FFN product %lld | LayerNorm backward 2 %lld | d_ao / d_aod out %lld | "
         "output projection %lld | d_ctx out %lld | total %lld\n",
         M, magic_rowbwd_rows(M), pdrop, ms * 1000 / 50, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6], t[8] - t[7], t[9] - t[8],
         t[10] - t[9], t[11] - t[10], t[11] - t[0]);
  return 0;
}
