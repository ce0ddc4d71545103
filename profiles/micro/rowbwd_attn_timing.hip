// Stage timing of the row-block backward WITH the attention backward of the block above inside (csrc/encbwd.hip mode 1, built with -DRBW_TIMING):
// wall_clock64 (100 MHz) marks of workgroup 0 for one text block of the headline step (48 samples x 80 rows: 240 workgroups).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DRBW_TIMING -I vln-magic_amd/csrc profiles/micro/rowbwd_attn_timing.hip -o profiles/micro/bin/rowbwd_attn_timing
#include "../../vln-magic_amd/csrc/encbwd.hip"
#include <cstdio>
static void* dmalloc(size_t n, int fill) { void* p; hipMalloc(&p, n); hipMemset(p, fill, n); return p; }
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 48, N = argc > 2 ? atoi(argv[2]) : 80;
  const float pdrop = argc > 3 ? atof(argv[3]) : 0.1f;
  const int mode = argc > 4 ? atoi(argv[4]) : 1;
  const int M = B * N, ldp = (N + 7) / 8 * 8;
  RbwParams P; memset(&P, 0, sizeof(P));
  RbwSeg& s = P.seg[0];
  typedef bf16 T;
  s.M = M; s.kt = 12;
  s.WqkvT_n = (T*)dmalloc(384 * 128 * 2, 0x11); s.dao_n = (T*)dmalloc((size_t)M * 128 * 2, 0x11);
  s.y2 = (T*)dmalloc((size_t)M * 128 * 2, 0x11); s.rstd2 = (float*)dmalloc((size_t)M * 4, 0x3c); s.g2 = (float*)dmalloc(512, 0x3c); s.b2 = (float*)dmalloc(512, 0);
  s.dg2 = (float*)dmalloc(512 * 512, 0); s.db2 = (float*)dmalloc(512 * 512, 0);
  if (mode == 1) { s.z = (T*)dmalloc((size_t)M * 512 * 2, 0x11); s.W2T = (T*)dmalloc(128 * 512 * 2, 0x11); s.W1T = (T*)dmalloc(128 * 512 * 2, 0x11); }
  s.y1 = (T*)dmalloc((size_t)M * 128 * 2, 0x11); s.rstd1 = (float*)dmalloc((size_t)M * 4, 0x3c); s.g1 = (float*)dmalloc(512, 0x3c); s.b1 = (float*)dmalloc(512, 0);
  s.dg1 = (float*)dmalloc(512 * 512, 0); s.db1 = (float*)dmalloc(512 * 512, 0);
  s.WoT = (T*)dmalloc(128 * 128 * 2, 0x11);
  s.dfo = (T*)dmalloc((size_t)M * 128 * 2, 0); s.dfod = (T*)dmalloc((size_t)M * 128 * 2, 0); s.dz = (T*)dmalloc((size_t)M * 512 * 2, 0);
  s.daod = (T*)dmalloc((size_t)M * 128 * 2, 0); s.dao = (T*)dmalloc((size_t)M * 128 * 2, 0); s.dctx = (T*)dmalloc((size_t)M * 128 * 2, 0);
  s.site_out = 3; s.site_ao = 4;
  s.mode = mode; s.N = N; s.ntile = (N + 15) / 16; s.ldp = ldp;
  s.qkv_a = (T*)dmalloc((size_t)M * 384 * 2, 0x11); s.P_a = (T*)dmalloc((size_t)B * 2 * N * ldp * 2, 0x11); s.o_a = (T*)dmalloc((size_t)M * 128 * 2, 0x11);
  s.dctx_a = (T*)dmalloc((size_t)M * 128 * 2, 0x11); s.dqkv_out = (T*)dmalloc((size_t)M * 384 * 2, 0); s.site_attn = 5;
  P.nseg = 1; P.p_hidden = pdrop; P.p_attn = pdrop; P.scale = 0.125f; P.pad1 = 1;
  unsigned* seed = (unsigned*)dmalloc(16, 0x5a); P.seed = pdrop > 0.f ? seed : nullptr;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) if (magic_rowbwd(DT_BF16, &P, sizeof(P), nullptr)) { printf("launch failed\n"); return 1; }
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int it = 0; it < 50; ++it) magic_rowbwd(DT_BF16, &P, sizeof(P), nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long t[32]; hipMemcpyFromSymbol(t, HIP_SYMBOL(rbw_ticks), sizeof(t));
  printf("mode %d, %d samples x %d rows (%d workgroups), dropout %.2f: %.1f us per launch back to back\n", mode, B, N, B * s.ntile, pdrop, ms * 1000 / 50);
  printf("  attention stage of workgroup 0 (10 ns ticks): zero tiles + issue loads %lld | head 0: rows -> images + rs + barrier %lld, stage A %lld, stage B %lld | head 1: %lld, %lld, %lld | dQKV out %lld | stage total %lld\n",
         t[16] - t[0], t[17] - t[16], t[18] - t[17], t[19] - t[18], t[20] - t[19], t[21] - t[20], t[22] - t[21], t[23] - t[22], t[23] - t[0]);
  if (mode == 1)
    printf("  chain behind it: stage rows + first weights %lld | tail product %lld | LayerNorm backward 1 + z image %lld | d_fo / d_fod out %lld | first FFN product + gelu' %lld | d_z out %lld | "
           "second FFN product %lld | LayerNorm backward 2 %lld | d_ao / d_aod out %lld | output projection %lld | d_ctx out %lld | workgroup total %lld\n",
           t[1] - t[23], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6], t[8] - t[7], t[9] - t[8], t[10] - t[9], t[11] - t[10], t[11] - t[0]);
  return 0;
}
