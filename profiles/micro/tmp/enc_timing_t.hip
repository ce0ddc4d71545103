// Stage timing of the whole-encoder kernel (csrc/encoder.hip built with -DENC_TIMING): wall_clock64 (100 MHz) marks of workgroup 0 of
// each segment in its LAST layer.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DENC_TIMING -I vln-magic_amd/csrc profiles/micro/enc_stage_timing.hip -o /tmp/enc_timing && /tmp/enc_timing
#include "encoder_t.hip"
#include <cstdio>
#include <vector>
static void* dmalloc(size_t n, int fill) { void* p; hipMalloc(&p, n); hipMemset(p, fill, n); return p; }
int main() {
  const int B = 48, L = 80, Np = 290, V = 36;
  EncParams P; memset(&P, 0, sizeof(P));
  P.nseg = 2; P.eps = 1e-12f; P.scale = 0.125f;
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
      E.qkv = (bf16*)dmalloc(M * 3 * EH * 2, 0); E.P = (bf16*)dmalloc((size_t)ns * ENH * N * s.ldp * 2, 0); E.Pd = nullptr;
      E.ctx = (bf16*)dmalloc(M * EH * 2, 0); E.a = (bf16*)dmalloc(M * EH * 2, 0); E.z = (bf16*)dmalloc(M * EI * 2, 0);
      E.g = (bf16*)dmalloc(M * EI * 2, 0); E.out = (bf16*)dmalloc(M * EH * 2, 0);
      E.rstd_a = (float*)dmalloc(M * 4, 0); E.rstd_o = (float*)dmalloc(M * 4, 0);
    }
  };
  fill_seg(P.seg[0], B, L, 6);
  fill_seg(P.seg[1], Np, V, 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 3; ++variant) {
    EncParams Q = P;
    if (variant == 1) Q.nseg = 1;                                     // text only
    if (variant == 2) { Q.seg[0] = P.seg[1]; Q.nseg = 1; }            // panorama only
    for (int it = 0; it < 3; ++it) magic_encoder_fwd(&Q, sizeof(Q), nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) magic_encoder_fwd(&Q, sizeof(Q), nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long t[2][16]; hipMemcpyFromSymbol(t, HIP_SYMBOL(enc_ticks), sizeof(t));
    printf("variant %d (%s): %.1f us per launch\n", variant, variant == 0 ? "text+pano" : variant == 1 ? "text only" : "pano only", ms * 1000 / 20);
    const char* nm[] = {"zero+barrier", "A qkv", "B attention (+qkv copy)", "ctx barrier", "C o-proj+LN (+ctx copy)", "D ffn1+gelu", "w2 issue+barrier", "E ffn2+LN (+g copy)"};
    for (int s = 0; s < (variant == 0 ? 2 : 1); ++s) {
      printf("  segment %d last layer (10 ns ticks):", s);
      for (int i = 0; i < 8; ++i) printf(" %s=%lld", nm[i], t[s][i + 1] - t[s][i]);
      printf("  total=%lld\n", t[s][8] - t[s][0]);
      if (s == 0) printf("    C detail (last add_norm call = stage E of last layer for 10-14): mfma=%lld | E: elementwise=%lld bar1=%lld mean=%lld bar2=%lld var+bar3=%lld\n", t[s][9]-t[s][4], t[s][10]-t[s][7], t[s][11]-t[s][10], t[s][12]-t[s][11], t[s][13]-t[s][12], t[s][14]-t[s][13]);
    }
  }
  return 0;
}
