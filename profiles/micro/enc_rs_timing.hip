// Stage timing of the ROW-SPLIT whole-encoder kernel (csrc/encoder.hip built with -DENC_TIMING): wall_clock64 (100 MHz) marks of tile 0 of
// sample 0 of the text segment, every layer.  Build here, run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DENC_TIMING -I vln-magic_amd/csrc profiles/micro/enc_rs_timing.hip -o gpurun_out/rs_timing
//   gpurun -- ./gpurun_out/rs_timing
#include "../../vln-magic_amd/csrc/encoder.hip"
#include <cstdio>
static void* dmalloc(size_t n, int fill) { void* p; hipMalloc(&p, n); hipMemset(p, fill, n); return p; }
int main() {
  const int B = 48, L = 80, Np = 290, V = 36;
  EncParams P; memset(&P, 0, sizeof(P));
  P.nseg = 2; P.eps = 1e-12f; P.scale = 0.125f;
  unsigned* seed = (unsigned*)dmalloc(8, 0x5a);
  auto fill_seg = [&](EncSeg& s, int ns, int N, int nl) {
    s.nsamp = ns; s.N = N; s.ldp = (N + 7) / 8 * 8; s.nlayers = nl;
    const size_t M = (size_t)ns * N;
    s.x = (const bf16*)dmalloc(M * EH * 2, 0x11); s.kmask = (const unsigned char*)dmalloc(M, 1);
    for (int l = 0; l < nl; ++l) {
      EncLayer& E = s.L[l];
      E.Wqkv = (const bf16*)dmalloc(3 * EH * EH * 2, 0x11); E.bqkv = (const float*)dmalloc(3 * EH * 4, 0);
      E.Wo = (const bf16*)dmalloc(EH * EH * 2, 0x11); E.bo = (const float*)dmalloc(EH * 4, 0);
      E.g1 = (const float*)dmalloc(EH * 4, 0x3c); E.be1 = (const float*)dmalloc(EH * 4, 0);
      E.W1 = (const bf16*)dmalloc(EI * EH * 2, 0x11); E.bi = (const float*)dmalloc(EI * 4, 0);
      E.W2 = (const bf16*)dmalloc(EH * EI * 2, 0x11); E.bo2 = (const float*)dmalloc(EH * 4, 0);
      E.g2 = (const float*)dmalloc(EH * 4, 0x3c); E.be2 = (const float*)dmalloc(EH * 4, 0);
      E.qkv = (bf16*)dmalloc(M * 3 * EH * 2, 0); E.P = (bf16*)dmalloc((size_t)ns * ENH * N * s.ldp * 2, 0);
      E.Pd = (bf16*)dmalloc((size_t)ns * ENH * N * s.ldp * 2, 0);
      E.ctx = (bf16*)dmalloc(M * EH * 2, 0); E.a = (bf16*)dmalloc(M * EH * 2, 0); E.z = (bf16*)dmalloc(M * EI * 2, 0);
      E.g = (bf16*)dmalloc(M * EI * 2, 0); E.out = (bf16*)dmalloc(M * EH * 2, 0);
      E.rstd_a = (float*)dmalloc(M * 4, 0); E.rstd_o = (float*)dmalloc(M * 4, 0);
      E.site_attn = 11 + l; E.site_ao = 21 + l; E.site_out = 31 + l;
    }
  };
  fill_seg(P.seg[0], B, L, 6);
  fill_seg(P.seg[1], Np, V, 2);
  P.sync_words = 4 + 6 * (B + Np) + 4; P.sync = (unsigned*)dmalloc(P.sync_words * 4, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int drop = 0; drop < 2; ++drop)
    for (int variant = 0; variant < 3; ++variant) {
      EncParams Q = P;
      Q.seed = drop ? seed : nullptr; Q.p_attn = drop ? 0.1f : 0.f; Q.p_hidden = drop ? 0.1f : 0.f;
      if (variant == 1) Q.nseg = 1;                                     // text only
      if (variant == 2) { Q.seg[0] = P.seg[1]; Q.nseg = 1; }            // panorama only
      for (int it = 0; it < 3; ++it) magic_encoder_fwd(DT_BF16, &Q, sizeof(Q), nullptr);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int it = 0; it < 20; ++it) magic_encoder_fwd(DT_BF16, &Q, sizeof(Q), nullptr);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long t[8][16]; hipMemcpyFromSymbol(t, HIP_SYMBOL(rs_ticks), sizeof(t));
      unsigned err; hipMemcpy(&err, P.sync, 4, hipMemcpyDeviceToHost);
      printf("dropout %d variant %d (%s): %.1f us per launch, err word %u\n", drop, variant,
             variant == 0 ? "text+pano" : variant == 1 ? "text only" : "pano only", ms * 1000 / 20, err);
      const char* nm[] = {"params+wait", "x load", "A kv/q", "B1 scores", "B2 softmax", "B3 PV", "C oproj+LN", "D ffn1", "E ffn2+LN", "handoff"};
      const int nl = variant == 2 ? 2 : 6;
      for (int l = 0; l < nl; ++l) {
        printf("  layer %d (10 ns ticks):", l);
        for (int i = 0; i < 10; ++i) printf(" %s=%lld", nm[i], t[l][i + 1] - t[l][i]);
        printf("  total=%lld\n", t[l][10] - t[l][0]);
      }
    }
  return 0;
}
