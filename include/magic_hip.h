/* magic_hip.h -- C ABI of libmagic_hip.so (gfx950 / MI355X).
 *
 * The reference (CrystalSixone/VLN-MAGIC) has no FFI: its hot path is a Python nn.Module that launches stock
 * PyTorch ops (SURVEY.md S2.2 "Native-component conclusion").  This ABI is therefore build-defined:
 * each entry point below replaces a group of torch ops the withheld model would launch, and cites the reference
 * interface whose arithmetic it carries.  Conventions:
 *   - plain pointers (device memory owned by the caller), sizes, strides; no torch types
 *   - `stream` is a hipStream_t; every call is asynchronous on it, never synchronises, never allocates
 *   - return 0 on success, negative error code otherwise (MAGIC_ERR_*); never throws
 *   - dtype: 0 = fp32 ("parity mode", exact fp32 MFMA), 1 = bf16, 2 = fp16 (v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate; every
 *     16-bit kernel exists for both types: fp16 stores 11 significand bits against bf16's 8 and needs scaled gradient seeds)
 *   - weights / activations / embedding tables are in `dtype`; biases, LayerNorm params, losses, logits of the
 *     action heads, statistics and ALL parameter gradients are fp32
 */
#ifndef MAGIC_HIP_H
#define MAGIC_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define MAGIC_OK 0
#define MAGIC_ERR_ARG (-1)
#define MAGIC_ERR_LAUNCH (-2)
#define MAGIC_ERR_UNSUPPORTED (-3)

int magic_abi_version(void);
/* Identity of the source set this binary was built from: 16 hex digits + NUL copied into out (len >= 17), a content hash over every
 * csrc/*.hip with the shared headers and compile flags, gen_fastcall.py and host/lib.py (csrc/build_id.py computes the same number from a
 * source tree).  host/lib.py refuses a library whose id differs from its tree's; bench.py prints it. */
int magic_build_id(char* out, int len);
int magic_device_info(int* cu_count, int* clock_khz, char* arch, int arch_len);

/* Dense contraction C = epi(alpha * op(A) op(B) + bias) [+ residual], batched over (batch = nb*nh) with
 * offsets (b*s?b + h*s?h).  layout 0: NT A[M,K] B[N,K] (nn.Linear forward: every Linear of the model, e.g.
 * HF BertSelfAttention/BertIntermediate under train_r2r_magic.py:189-208 names); 1: NN A[M,K] B[K,N] (dX, PV,
 * dQ); 2: TN A[K,M] B[K,N] (dW with split-K fp32 atomics + fused bias gradient, dK, dV).
 * epilogue: 0 none, 1 GELU(erf), 2 ReLU, 3 *gelu'(aux), 4 *relu'(aux).  C2 (optional) receives the
 * pre-activation.  k-contiguous operands need ld % (16/sizeof) == 0 and zero padding up to that multiple. */
/* splitk < 0 (round 6): |splitk| K-splits, each STORING its partial into its own slab of C -- C is [|splitk|][M][ldc] fp32, no atomics; the consumer adds the
 * slabs in slab order (magic_ln_bwd_tail with the slab count in `act`): a deterministic split-K.  Plain product only (c_f32, no epilogue / residual / C2 /
 * bias_grad, batch = nh = 1); a split without k-tiles stores zeros. */
int magic_gemm(int dtype, int layout, int batch, int nh, int M, int N, int K,
               const void* A, int lda, long long sAb, long long sAh,
               const void* B, int ldb, long long sBb, long long sBh,
               void* C, int ldc, long long sCb, long long sCh, int c_f32, int accumulate,
               const float* bias, int epilogue, const void* aux, int ldaux,
               const void* residual, int ldr, void* C2, int ldc2,
               float alpha, int splitk, float* bias_grad, void* stream);
/* Block-tile selection of magic_gemm's single launches: 0 = 64x64 always (default), 1 = 128x128 when the problem has enough
 * such tiles and a long K loop, 2 = 128x128 whenever M, N >= 128.  Process-wide tuning knob (also MAGIC_GEMM_BIG). */
int magic_gemm_set_big(int mode);

/* Grouped weight-gradient GEMM: n <= 96 problems dW[N,K] (fp32, ldc) += dY[M,N]^T (lda) @ X[M,K] (ldb), db[N] += colsum(dY) in ONE
 * launch, split-K.  `d` is a HOST array of n descriptors holding device pointers.  Replaces the per-parameter `.grad` accumulation of
 * autograd's Linear backward (torch: `grad_weight = grad_output.t().mm(input)`, one GEMM per layer call).
 * ws == NULL: the K-splits add into dW with fp32 atomics (the sum then depends on arrival order).
 * ws != NULL: DETERMINISTIC -- every split stores its 64 x 64 partial into its own slot of `ws`, the workgroup that arrives last at a
 * tile adds the slots in slot order and does the one read-modify-write of dW; problems with the SAME dW pointer (one Linear called several
 * times in a step) share that dW's slots, so they too are summed in a fixed order and must agree in N, K, ldc, db.  `ws`: ws_floats >=
 * magic_gemm_dw_ws_need(...) fp32 words, contents undefined; `counters`: n_counters 32-bit words that are ZERO on entry and zero again
 * when the launch has finished (zero them once, at allocation); both owned by the caller until the launch completes, not shared with a
 * launch running concurrently on another stream. */
typedef struct magic_dw_desc { const void* dY; const void* X; float* dW; float* db; int M, N, K, lda, ldb, ldc, splitk; } magic_dw_desc;
int magic_gemm_dw_ws_need(int dtype, int n, const magic_dw_desc* d, long long* floats, int* counters);
int magic_gemm_dw_grouped(int dtype, int n, const magic_dw_desc* d, float* ws, long long ws_floats, unsigned* counters, int n_counters, void* stream);
/* Weight gradients over MANY row segments per Linear in one launch: dW_p[N, K] += sum_s dY_{p,s}[M_{p,s}, N]^T X_{p,s}[M_{p,s}, K] (and db_p += the column
 * sums of dY), p < n_prob <= 96, s < n_seg.  dy_tab / x_tab: DEVICE tables [n_prob][n_seg] of operand pointers (16-byte aligned rows: lda / ldb multiples of
 * 8 elements for the 16-bit types), m_tab: DEVICE [n_prob][n_seg] row counts (0 skips a segment).  One workgroup per 64 x 64 tile of a dW walks all of its
 * segments with the accumulators in registers and read-modify-writes dW once: no workspace, no atomics, sums in segment order (reproducible).  The
 * navigator iteration's ~38 calls of every Linear (agent_base.py:243-263: two rollouts, one backward) leave in one launch per <= 96 Linears. */
typedef struct magic_dwcat_prob { float* dW; float* db; int N, K, lda, ldb, ldc; } magic_dwcat_prob;
int magic_gemm_dw_cat(int dtype, int n_prob, const magic_dwcat_prob* probs, int n_seg, const void* const* dy_tab, const void* const* x_tab,
                      const int* m_tab, void* stream);

/* out = LayerNorm(x[M,K] W[H,K]^T + bias + residual): BertSelfOutput / BertOutput (dense -> add -> LayerNorm) in one launch;
 * H in {128, 256, 384} (a workgroup owns 32 full rows), otherwise MAGIC_ERR_UNSUPPORTED -> magic_gemm + magic_ln_fwd. */
int magic_linear_ln(int dtype, int M, int H, int K, const void* x, int lda, const void* W, int ldb, const float* bias,
                    const void* residual, int ldr, const float* gamma, const float* beta, float eps,
                    void* out, float* rstd, const void* drop_seed, float drop_p, unsigned drop_site, void* stream);
/* dense -> activation -> LayerNorm as one launch: out = LayerNorm(act(x W^T + bias)) (BertPredictionHeadTransform of the MLM head,
 * pretrain_src/model/pretrain_cmt.py; RegionClassification's Linear / ReLU / LayerNorm of the MRC head).  act: 1 = erf gelu, 2 = relu;
 * pre_out (optional): the pre-activation x W^T + bias in the storage dtype, pitch H, which the backward's act' reads; rstd (optional) as
 * magic_linear_ln.  H in {128, 256, 384} (MAGIC_ERR_UNSUPPORTED otherwise: the caller runs magic_gemm + magic_ln_fwd). */
int magic_linear_act_ln(int dtype, int M, int H, int K, const void* x, int lda, const void* W, int ldb, const float* bias, int act,
                        void* pre_out, const float* gamma, const float* beta, float eps, void* out, float* rstd, void* stream);

/* Backward twin: input-gradient GEMM + residual + LayerNorm BACKWARD in one launch (H in {128, 256}).
 *   v = x[M,K] @ W[K,H] + residual  (the gradient at the OUTPUT of the LayerNorm whose saved output is y);
 *   dx = LN-backward(v; y, gamma, beta, rstd);  dgamma/dbeta accumulated;  dxm = dx * dropout mask of the dense branch that fed
 * that LayerNorm (site 0 / p 0: dxm unused).  Replaces magic_gemm(NN, residual) + magic_ln_bwd on every block boundary of
 * the backward chain (FFN dX -> attention-output LN, QKV dX -> the previous block's output LN). */
int magic_linear_lnbwd(int dtype, int M, int H, int K, const void* x, int lda, const void* W, int ldb,
                       const void* residual, int ldr, const void* y, const float* gamma, const float* beta, const float* rstd,
                       void* dx, void* dxm, float* dgamma, float* dbeta,
                       const void* drop_seed, float drop_p, unsigned drop_site, void* stream);

/* Dropout (hidden_dropout_prob / attention_probs_dropout_prob of r2r_magic_model_config.json:2-3,6; active under
 * model.train(), train_r2r_magic.py:358) is counter-based: keep(seed[0..1], site, logical element index) is recomputed by
 * the backward kernels, no mask is stored.  `drop_seed` = 2 x uint32 in DEVICE memory (fresh per step, so a replayed HIP
 * graph draws new masks), `drop_p` in [0,1) (0 = off, seed may be NULL), `site` = id of the dropout module (0 = this
 * position is not dropped).  magic_dropout applies the same mask standalone: out = in * keep / (1-p), rows x cols logical,
 * row pitch ld (the Nk > 128 attention fallback, and the mask export the parity tests feed to the oracle). */
int magic_dropout(int dtype, long long rows, int cols, int ld, const void* in, void* out,
                  const void* drop_seed, float drop_p, unsigned site, void* stream);

/* out = [LayerNorm]( in0 + in1 + tab0[i0] + tab1[i1] + tab2[i2] ); table row = idx ? idx[r] : mod ? r%mod+off : off.
 * Carries BertEmbeddings (word + position(+2) + token-type -> LN), the image embedding sum, the map-node
 * input sum and every residual-add + LayerNorm of the BERT blocks. */
int magic_ln_fwd(int dtype, int M, int H, const void* in0, const void* in1,
                 const void* tab0, const int* idx0, int mod0, int off0,
                 const void* tab1, const int* idx1, int mod1, int off1,
                 const void* tab2, const int* idx2, int mod2, int off2,
                 const float* gamma, const float* beta, float eps, void* out, float* rstd, int do_ln,
                 const void* drop_seed, float drop_p, unsigned site_in0, unsigned site_out, void* out_drop, void* stream);
/* dropout: site_in0 drops in0 before the sum (dense -> dropout -> + residual); site_out writes dropout(out) to out_drop
 * while `out` keeps the clean y the backward recovers xhat from.  bwd: site_dy masks dy on load (forward dropped its
 * output), site_dx additionally writes dxm = dx * mask (the gradient of the dropped in0 branch). */
int magic_ln_bwd(int dtype, int M, int H, const void* dy, const void* y, const float* gamma, const float* beta,
                 const float* rstd, void* dx, float* dgamma, float* dbeta,
                 const int* idx0, int mod0, int off0, float* d0, int small0,
                 const int* idx1, int mod1, int off1, float* d1, int small1,
                 const int* idx2, int mod2, int off2, float* d2, int small2,
                 int do_ln, const void* drop_seed, float drop_p, unsigned site_dy, unsigned site_dx, void* dxm, int hot0, int pg_partial,
                 void* stream);
/* The LayerNorm backward of a prediction head's transform (BertPredictionHeadTransform: dense -> gelu -> LayerNorm -> decoder; the MLM head of
 * pretrain_src/model/pretrain_cmt.py via train_r2r_magic.py:441-467) together with its two neighbours in the backward chain: dy32 is the fp32
 * [M, H] accumulator the split-K input gradient of the vocabulary projection leaves (no cast launch), and dx = LayerNorm'(dy) x act'(act_pre)
 * (no activation-derivative launch; act: 1 = erf gelu, 2 = relu; act_pre: the dense output before the activation, storage dtype).  gamma / beta
 * gradients are added as magic_ln_bwd does without partial rows.  Replaces cast + magic_ln_bwd + magic_dact. */
/* act: bits 0..5 the activation in front of the LayerNorm (1 gelu, 2 relu); bit 6: dgamma / dbeta are partial buffers [magic_ln_bwd_blocks(M, H, 0)][H]; bits 8.. = S > 0: dy32 holds S slabs of M x H (magic_gemm with splitk = -S) that
 * are added in slab order on load (round 6). */
int magic_ln_bwd_tail(int dtype, int M, int H, const float* dy32, const void* y, const float* gamma, const float* beta, const float* rstd,
                      const void* act_pre, int act, void* dx, float* dgamma, float* dbeta, void* stream);
/* hot0 >= 0: a row of indexed table 0 that a large share of the input rows hit (the padding token id of the word-embedding lookup): its
 * gradient is summed per workgroup in LDS and added with one atomic per element and workgroup (else -1).
 * pg_partial != 0 (round 5): dgamma / dbeta point at PARTIAL buffers of magic_ln_bwd_blocks(M, H, any of d0 / d1 / d2 given) x H floats each, contents undefined: every
 * workgroup STORES its gamma / beta sums in its own row instead of adding them into the parameter gradients with same-address atomics (at
 * H = 768 the atomics were the launch: 17.6 us for 608 rows, 4 us without); magic_colsum_add_v adds the rows up in block order. */
int magic_ln_bwd_blocks(int M, int H, int has_tables);
/* dsts[j][c] += sum over b < nblks[j] of parts[j][b * strides[j] + c], c < lens[j], for n <= 96 jobs in one launch (host arrays of device pointers,
 * consumed before return).  The finisher of every partial-row epilogue (magic_ln_bwd pg_partial, magic_smallk_ln_bwd part); block order:
 * reproducible; a destination may occur once per launch.  (torch: `param.grad` accumulation of LayerNorm / Linear parameters.) */
int magic_colsum_add_v(int n, const float* const* parts, float* const* dsts, const int* nblks, const int* lens, const int* strides, void* stream);

/* gamma/beta gradients of one LayerNorm as a column reduction (used when magic_ln_bwd is called with dgamma = dbeta = NULL) */
int magic_ln_pgrad(int dtype, int M, int H, const void* dy, const void* y, const float* gamma, const float* beta,
                   float* dgamma, float* dbeta, void* stream);

/* y = LN(x[M,Kin<=16] W^T + b): loc_linear+loc_layer_norm, gmap_pos_embeddings, vp_pos_embeddings (App. B.2-B.3) */
int magic_smallk_ln_fwd(int dtype, int M, int H, int Kin, const float* x, const float* W, const float* b,
                        const float* gamma, const float* beta, float eps, void* out, float* rstd, void* stream);
/* Several independent CSR row gathers in one launch (n <= 4): out[r,:] (+)= sum_e w1[e] src1[idx1[e],:], then (optional second source)
 * += sum_e w2[e] src2[idx2[e],:], rounded to the storage type after each source exactly as two consecutive magic_csr_gather calls. */
typedef struct {
  int n_out, accumulate; const void* src1; const int* ptr1; const int* idx1; const float* w1;
  const void* src2; const int* ptr2; const int* idx2; const float* w2; void* out;
} magic_csr_prob;
int magic_csr_gather_multi(int dtype, int H, int n, const magic_csr_prob* d, void* stream);
/* two magic_smallk_ln_bwd problems (d[0], d[1]) in one launch */
typedef struct {
  int M, Kin; const float* x; const void* dy; const void* y; const float* gamma; const float* beta; const float* rstd;
  float* dW; float* db; float* dgamma; float* dbeta; float* part;
} magic_skb_prob;
int magic_smallk_ln_bwd_pair(int dtype, int H, const magic_skb_prob* d, void* stream);
/* Input stage of the cross-modal encoders as ONE launch for 1 or 2 encoders (the map encoder's gmap tokens, the local encoder's viewpoint
 * tokens): out = [CSR-gathered panorama embeddings src1 (+ src2)] or add0, + A, + tab[tab_idx]; A = LayerNorm(x[M,Kin] W^T + b) (saved, with
 * rstd, for magic_smallk_ln_bwd).  Same rounding points as magic_csr_gather -> magic_smallk_ln_fwd -> magic_ln_fwd(do_ln = 0). */
typedef struct {
  int M, Kin; const float* x; const float* W; const float* b; const float* gamma; const float* beta; float eps; int pad_;
  void* A; float* rstd; void* out;
  const void* add0;
  const void* src1; const int* ptr1; const int* idx1; const float* w1;
  const void* src2; const int* ptr2; const int* idx2; const float* w2;
  const void* tab; const int* tab_idx;
} magic_node_in;
int magic_node_in_fwd(int dtype, int H, int n, const magic_node_in* d, void* stream);

/* Input stage of the panorama encoder in ONE launch, together with the text-embedding rows (csrc/rowops.hip embed_in_fwd_kernel; round 4):
 * A1 = LN_img(P0), P0 = img_linear(view features); A2 = LN_loc(loc W^T + b); X0 = LN(A1 + A2 + nav_tab[nav_idx] + tok_tab[0]) and its dropped
 * copy X0d -- magic_ln_fwd -> magic_smallk_ln_fwd -> magic_ln_fwd of the per-op path with the same rounding points and summation order, so every
 * saved tensor is bit-identical and the backward kernels read them unchanged.  (The withheld model's `ImageEmbeddings`: img_linear /
 * img_layer_norm / loc_linear / loc_layer_norm / nav_type_embedding / layer_norm, names per train_r2r_magic.py:189-208; [LINEAGE] DUET.)
 * tx (may be NULL): a second, independent magic_ln_fwd problem served by the same launch (the text embedding: word + position + token-type
 * rows, LayerNorm, dropout).  drop: seed == NULL or p <= 0 -> off. */
typedef struct { const unsigned* seed; unsigned site; float p; } magic_drop_desc;
typedef struct magic_pano_in {
  int M, Kin; float eps; int pad_;
  const void* P0; const float* g1; const float* b1; void* A1; float* rstd1;
  const float* loc; const float* W; const float* b; const float* g2; const float* b2; void* A2; float* rstd2;
  const void* nav_tab; const int* nav_idx; const void* tok_tab;
  const float* g3; const float* b3; void* X0; float* rstd3; void* X0d; magic_drop_desc dout;
} magic_pano_in;
typedef struct magic_ln_in {
  int M, do_ln; const void* in0; const void* in1;
  const void* tab[3]; const int* idx[3]; int mod[3]; int off[3];
  const float* gamma; const float* beta; float eps; int pad_; void* out; float* rstd;
  const unsigned* drop_seed; float drop_p; unsigned site_in0, site_out, pad2_; void* out_drop;
} magic_ln_in;
int magic_embed_in_fwd(int dtype, int H, const magic_pano_in* pa, const magic_ln_in* tx, void* stream);

/* Backward of that stage in ONE launch (csrc/rowops.hip embed_in_bwd_kernel): magic_ln_bwd (sum LayerNorm: dsum, nav-type / token-type row
 * gradients) -> magic_ln_bwd (image LayerNorm: dP0, the operand of the image projection's weight gradient) -> magic_smallk_ln_bwd (location
 * LayerNorm + loc_linear gradients) of the per-op path with the same formulas and rounding points (dP0 bit-identical; parameter gradients to
 * fp32 summation order), every parameter gradient accumulated in registers and reduced once per block.  H = 128 or 256, Kin <= 8
 * (magic_embed_in_bwd_supported).  tx (may be NULL): a second, independent magic_ln_bwd problem (the text embedding's LayerNorm + table
 * scatters) served by the same launch.  dy: gradient of the stage's output; ddy: the output dropout (dy is masked on load). */
typedef struct magic_pano_in_bwd {
  int M, Kin, pad0_, pad1_;
  const void* dy; magic_drop_desc ddy;
  const void* X0; const float* rstd3; const float* g3; const float* b3; float* dg3; float* db3;
  const int* nav_idx; float* d_nav; float* d_tok;
  const void* A1; const float* rstd1; const float* g1; const float* b1; float* dg1; float* db1; void* dP0;
  const void* A2; const float* rstd2; const float* g2; const float* b2; float* dg2; float* db2;
  const float* loc; float* dW; float* dbl;
  /* round 6: != NULL -> every workgroup STORES its (11 + Kin) H sums in its own row of this buffer (pad0_ rows of pad1_ floats; row layout: dg3 | db3 |
   * d_nav[3 H] | d_tok | dg1 | db1 | dg2 | db2 | dbl | dW[H Kin], each as its destination is laid out) instead of adding them with atomics; the caller adds
   * rows 0 .. magic_embed_in_bwd_blocks(...) - 1 up in row order (magic_colsum_add_v).  NULL: the atomic form. */
  float* part;
} magic_pano_in_bwd;
typedef struct magic_ln_bwd_in {
  int M, do_ln; const void* dy; const void* y; const float* gamma; const float* beta; const float* rstd; void* dx; float* dgamma; float* dbeta;
  const int* idx[3]; int mod[3]; int off[3]; float* d[3]; int small[3];
  const unsigned* drop_seed; float drop_p; unsigned site_dy, site_dx; int hot0; void* dxm;
  int partial;   /* round 6: != 0 -> dgamma / dbeta are partial buffers [magic_ln_bwd_blocks(M, H, tables)][H], as magic_ln_bwd's `partial` */
} magic_ln_bwd_in;
int magic_embed_in_bwd_supported(int H, int Kin);
/* cs_*: n_cs <= 96 column-sum jobs (the arguments of magic_colsum_add; their vectors have H columns) served by extra workgroups of the same
 * launch: every magic_rowbwd launch of a backward pass precedes this one, so its partial LayerNorm gradients can be finished here. */
/* workgroups (= partial rows) of the panorama half for M rows beside nb_text workgroups of the text half (magic_ln_bwd_blocks); partial: with pa->part set */
int magic_embed_in_bwd_blocks(int M, int H, int nb_text, int partial);
int magic_embed_in_bwd(int dtype, int H, const magic_pano_in_bwd* pa, const magic_ln_bwd_in* tx,
                       int n_cs, const float* const* cs_parts, float* const* cs_dsts, const int* cs_nblks, void* stream);
int magic_smallk_ln_bwd(int dtype, int M, int H, int Kin, const float* x, const void* dy, const void* y,
                        const float* gamma, const float* beta, const float* rstd,
                        float* dW, float* db, float* dgamma, float* dbeta, float* part, void* stream);
/* part != NULL (round 5): magic_smallk_ln_bwd_blocks(M, H) x H (Kin + 3) floats, contents undefined: every workgroup stores its sums
 * [dW (H x Kin) | db | dgamma | dbeta] in its own row instead of H (Kin + 3) same-address atomics; finish with magic_colsum_add_v.  The pair
 * entry picks ONE tile shape from its larger problem: Mmax = that problem's rows (= M for the single-problem entry). */
int magic_smallk_ln_bwd_blocks(int M, int H, int Mmax);

/* P = softmax(scale*S + (kmask?0:-10000) + sprel_w*dist + sprel_b) over rows of S[B,nh,Nq,ldp] (HF additive
 * mask; graph_sprels bias r2r_magic_model_config.json:28).  bwd writes scale*dS and the 2 sprel_linear grads. */
int magic_softmax_fwd(int dtype, int B, int nh, int Nq, int Nk, int ldp, const float* S, void* P, float scale,
                      const unsigned char* kmask, const float* dist, const float* sprel_w, const float* sprel_b, void* stream);
int magic_softmax_bwd(int dtype, int B, int nh, int Nq, int Nk, int ldp, const void* P, const float* dP, void* dS, float scale,
                      const float* dist, float* dsprel_w, float* dsprel_b, void* stream);
int magic_head_mean_fwd(int dtype, int B, int nh, long long inner, const void* P, float* out, void* stream);
int magic_head_mean_bwd(int B, int nh, long long inner, const float* g, float* dP, int accumulate, void* stream);

/* Fused attention for Nk <= 128, head dim 64 (HF BertSelfAttention arithmetic: scores/sqrt(d) + additive mask -> softmax
 * -> probs @ V; graph_sprels bias as in magic_softmax_fwd).  P [B,nh,Nq,ldp] is written (needed by the backward and by the
 * attention-map distillation, agent.py:579-593).  bwd: dq/dk/dv given dctx (+ optional dP_init = dLoss/dP, fp32).
 * magic_attn_supported() tells the host whether a shape fits (LDS); otherwise use magic_gemm + magic_softmax_*.
 * With dropout the product uses P*keep/(1-p); P stays the clean softmax (backward), Pd (optional) receives the dropped
 * probabilities, which is what HF/METER BertSelfAttention returns as the attention map; dP_init is then dLoss/dPd. */
int magic_attn_supported(int dtype, int Nq, int Nk, int backward);
int magic_attn_fwd(int dtype, int B, int nh, int Nq, int Nk, const void* q, int ldq, const void* k, const void* v, int ldkv,
                   void* P, int ldp, void* ctx, int H, float scale, const unsigned char* kmask, const float* dist,
                   const float* sprel_w, const float* sprel_b,
                   const void* drop_seed, float drop_p, unsigned drop_site, void* Pd, void* stream);
/* dsprel_b NULL with dist and dsprel_w set (round 6): dsprel_w is a partial buffer [B nh][2]; every (sample, head) workgroup STORES its (weight, bias) sums
 * there instead of adding them with atomics, and the caller adds the pairs up in order (magic_colsum_add_v). */
int magic_attn_bwd(int dtype, int B, int nh, int Nq, int Nk, const void* q, int ldq, const void* k, const void* v, int ldkv,
                   const void* P, int ldp, const void* dctx, int H, float scale, const float* dP_init,
                   void* dq, int lddq, void* dk, void* dv, int lddkv,
                   const float* dist, float* dsprel_w, float* dsprel_b,
                   const void* drop_seed, float drop_p, unsigned drop_site, void* stream);
/* Long keys (128 < Nk <= 512; RxR-length instructions: `max_instr_len` 250, map_nav_src/scripts/run_rxr_kdl_valid.sh:37; text self-attention and
 * the cross-attention of map / viewpoint tokens to the instruction), 16-bit storage: magic_attn_fwd runs a key-split kernel (one workgroup per
 * (batch, head, 64-query tile), every wave owns a 64-key slab), and magic_attn_bwd_ks is the fused backward of the same products -- it replaces
 * the unfused chain (three batched magic_gemm + magic_softmax_bwd + two magic_dropout launches; torch: autograd of
 * BertSelfAttention.forward).  o = the forward's output ctx [B*Nq, H] (rowsum(P dP) = dO . O, also under dropout); no graph-distance bias, no
 * dP_init on this path; accumulate_kv != 0: dk / dv are ADDED to (what `txt_kv.grad` accumulation over an episode's steps does in torch).
 * magic_attn_supported(dtype, Nq, Nk, 2) tells whether it applies. */
int magic_attn_bwd_ks(int dtype, int B, int nh, int Nq, int Nk, const void* q, int ldq, const void* k, const void* v, int ldkv,
                      const void* P, int ldp, const void* o, const void* dctx, int H, float scale,
                      void* dq, int lddq, void* dk, void* dv, int lddkv, int accumulate_kv,
                      const void* drop_seed, float drop_p, unsigned drop_site, void* stream);

/* ClsPrediction tail (Linear->ReLU->LN->Linear(H,1)): logit = dot(LN(Y), w2) + b2 */
int magic_lndot_fwd(int dtype, int M, int H, const void* Y, const float* gamma, const float* beta, float eps,
                    const float* w2, const float* b2, float* logit, void* stream);
/* part (round 6; a trailing argument added to this entry point): != NULL -> every workgroup STORES its sums [dgamma | dbeta | dw2 | db2] (3 H + 1 floats) in row
 * `workgroup` of this buffer (magic_lndot_bwd_blocks(M) rows) instead of adding them with atomics; the caller adds the rows up in order (magic_colsum_add_v). */
int magic_lndot_bwd_blocks(int M);
int magic_lndot_bwd(int dtype, int M, int H, const void* Y, const float* gamma, const float* beta, float eps,
                    const float* w2, const float* dlogit, void* dZ, float* dgamma, float* dbeta, float* dw2, float* db2,
                    float* part, void* stream);

/* Rowwise CE with -inf masks + ignore_index (agent_base.py:152 criterion; validate_* of train_r2r_magic.py),
 * gradient coef*w*(softmax-onehot) in the same pass; w_out = exp(-w_rate*CE) = MKTD weights (agent.py:1013-1020,
 * kd_loss.py exponential_decay). */
int magic_ce_rows(int dtype, int M, int N, const void* logits, int ld, const int* labels, int ignore_index,
                  float coef, const float* row_w, float* loss_row, void* dlogits, int ldd, int accumulate,
                  float* w_out, float w_rate, void* stream);
/* Soft-target KL rows of the MRC head (validate_mrc, train_r2r_magic.py:483-485: F.kl_div(log_softmax(logits), targets)
 * summed over the 1000 classes): loss_row = sum_j t_j (log t_j - log p_j); dlogits = coef * ((sum_j t_j) p - t). */
int magic_softkl_rows(int dtype, int M, int N, const void* logits, int ld, const float* targets, int ldt, float coef,
                      const float* row_w /* NULL or [M]: dlogits row r scaled by coef * row_w[r] */, float* loss_row, void* dlogits, int ldd,
                      void* stream);

/* kd_loss (pretrain_src/optim/kd_loss.py:18-41, map_nav_src/utils/kd_loss.py:27-54): -inf -> -1e6, T-softmax KL * T^2 */
int magic_kd_rows(int M, int N, const float* s, const float* t, int ld, float temperature, const float* w, float norm,
                  float coef, const float* coef_dev, float* loss_row, float* ds, int accumulate, void* stream);
/* mse_loss (kd_loss.py:5-16 / :6-25): sum_b w_b (s-t)^2 * norm, grad 2*coef*norm*w*(s-t) */
int magic_mse(int dtype, int g_f32, long long outer, long long inner, const void* s, long long s_stride, const void* t,
              long long t_stride, const float* w, long long rows_per_w, float norm, float coef, const float* coef_dev, float* loss, void* ds,
              long long g_stride, int accumulate, void* stream);
/* n <= 10 independent magic_mse problems in ONE launch (the MAKD terms of a step, agent.py:546-719, sit back to back) */
typedef struct {
  int g_f32; long long outer, inner; const void* s; long long s_stride; const void* t; long long t_stride;
  const float* w; long long rows_per_w; float norm, coef; const float* coef_dev; float* loss; void* ds; long long g_stride; int accumulate;
  const int* valid_dev; const float* norm_dev; long long valid_mod;   /* magic_mse_multi only (shape-bucketed batches): device-side valid (outer, inner) <= the launch's (inner taken modulo valid_mod when > 0); norm *= norm_dev[0] */
} magic_mse_desc;
int magic_mse_multi(int dtype, int n, const magic_mse_desc* d, void* stream);
/* The three in-batch contrastive terms of the CFP task (train_r2r_magic.py:548-560), forward and backward in one launch: for a in {a0, a1, a2}
 * ([B,H] head outputs) sim = a txt^T / temperature; rows[2i][r] = CE(sim_i[r,:], r), rows[2i+1][c] = CE(sim_i[:,c], c) (unscaled);
 * d_i = G_i txt / temperature, dtxt = sum_i G_i^T a_i / temperature with G_i = coef (softmax_rows - I) + coef (softmax_cols - I).
 * B <= 64, H <= 256; d0 = d1 = d2 = dtxt = NULL: losses only.  part: fp32 scratch [3,B,H]; counter: one int32 that is 0 on entry and 0 again
 * on completion (the three workgroups' partial dtxt are summed in fixed order by the one that finishes last). */
int magic_cfp_loss(int dtype, int B, int H, const void* a0, const void* a1, const void* a2, const void* txt, float temperature, float coef,
                   float* rows, void* d0, void* d1, void* d2, void* dtxt, float* part, int* counter, void* stream);

/* out[n] (+)= sum_e w[e]*src[idx[e]]: map-node aggregation by viewpoint id (agent.py:905-924 semantics),
 * candidate-view / masked-token / CLS row selection; backward = same call on the transposed CSR. */
int magic_csr_gather(int dtype, int n_out, int H, const void* src, const int* ptr, const int* idx, const float* w,
                     void* out, int accumulate, void* stream);
/* adaptive_pano_fusion (r2r_magic_model_config.json:57): attention pooling of the V views.  P != NULL: the same launch also writes the
 * panorama encoder's attention map averaged over heads, pmean[n, r] = (1/nh) sum_h P[n, h, r], r < inner (what magic_head_mean_fwd computes) */
int magic_pano_fuse_fwd(int dtype, int N, int V, int H, const void* x, const int* lens, const float* wf, const float* bf,
                        void* fused, float* probs, const void* P, int nh, int inner, float* pmean, void* stream);
/* The SAP step's logit fusion + its row losses in ONE launch (one 64-thread workgroup per sample): magic_sap_fuse_fwd, three magic_ce_rows
 * (global / local / fused logits, gradients coef * (softmax - onehot)), the teacher-sample weights w = exp(-w_rate * CE(t_fused, label)) and the
 * action-distillation rows of magic_kd_rows (added into dfl) -- same arithmetic as those entry points.  K <= 512, Vp <= 128. */
typedef struct {
  int B, K, Vp, use_gate;
  const float *g_raw, *l_raw, *fuse_raw; const unsigned char *gmask, *lmask; const int* fsrc; const unsigned char* bwmask;
  float *gl, *ll, *fl;
  const int *glab, *llab; int ignore_index; float coef;
  float *rows, *dgl, *dll, *dfl;                 /* rows [3, B]; dgl / dll / dfl NULL: losses only */
  const float* t_fused; float w_rate; int pad_; float* w_out;      /* t_fused NULL: no teacher; w_out NULL: no sample weights */
  float T, kd_norm, kd_coef, pad2_; const float* kd_coef_dev; float* kd_rows;     /* kd_rows NULL: no distillation term */
} magic_sap_loss_params;
int magic_sap_fuse_loss(const void* params, int nbytes, void* stream);
/* dbf NULL (round 6): dwf is a partial buffer [magic_pano_fuse_bwd_blocks(N)][H + 1]; every workgroup STORES its weight-row and bias sums in its own row instead of
 * adding them with atomics, the caller adds the rows up in order (magic_colsum_add_v). */
int magic_pano_fuse_bwd_blocks(int N);
int magic_pano_fuse_bwd(int dtype, int N, int V, int H, const void* x, const float* probs, const float* wf, const void* dfused,
                        void* dx, float* dwf, float* dbf, void* stream);
/* global/local gate + -inf masks + local->global logit fusion (validate_sap contract :503-535) */
int magic_sap_fuse_fwd(int B, int K, int Vp, const float* g_raw, const float* l_raw, const float* fuse_raw,
                       const unsigned char* gmask, const unsigned char* lmask, const int* fsrc, const unsigned char* bwmask,
                       int use_gate, float* gl, float* ll, float* fl, void* stream);
int magic_sap_fuse_bwd(int B, int K, int Vp, const float* g_raw, const float* l_raw, const float* fuse_raw,
                       const unsigned char* gmask, const unsigned char* lmask, const int* fsrc, const unsigned char* bwmask,
                       int use_gate, const float* dgl, const float* dll, const float* dfl,
                       float* dg_raw, float* dl_raw, float* dfuse_raw, void* stream);

/* Step prologue: the per-step random scalars of a training step in one launch -- MKRW ability weights rw[5] = softmax(randn(5) / rw_temp) * 5
 * (map_nav_src/r2r/agent.py:866-871) and the two 31-bit words that key the counter-based dropout masks; `counter` (one device word) is
 * advanced here, so a replayed HIP graph draws fresh values each step.  seed_out / rw_out: either may be NULL. */
int magic_step_rng(unsigned long long base_seed, unsigned* counter, float rw_temp, int* seed_out, float* rw_out, float* zero_me,
                   float* scale_state, float growth, float backoff, int interval, void* stream);
/* scale_state (may be NULL) = {S, 1 / S, clean steps in a row, pending}: the DYNAMIC LOSS SCALE of fp16 storage, amp.GradScaler's rule
 * (train_r2r_magic.py:370-371) kept on the device so it works under HIP-graph replay with no host round trip: the prologue launch of a step
 * reads what the previous step's magic_adamw left in `pending` (1: weights updated, 2: update skipped, the gradient norm was not finite) --
 * 2: S *= backoff; 1: after `interval` updates in a row S *= growth -- and rewrites S and 1 / S before any loss kernel of the new step runs.
 * magic_seed_scale(scale_dev): registers (process-wide; NULL clears) the device word every gradient-SEEDING loss launch recorded from now
 * on multiplies its gradient coefficient by -- magic_ce_rows, magic_softkl_rows, magic_kd_rows, magic_mse, magic_mse_multi, magic_cfp_loss,
 * magic_sap_fuse_loss; loss VALUES are never scaled.  Pass &scale_state[0]. */
int magic_seed_scale(const float* scale_dev);
/* Loss assembly in one launch: sup = row_scale * sum rows[i] (* row_w[i]); slots[9] = sum kd_rows (optional); terms[i] = slots[i] * rw[ability(i)]
 * over the ten MAKD slots (agent.py:546-719: txt, txt, img, img, img, global, global, local, local, action); kdl = sum terms;
 * loss = alpha * kdl + (1 - alpha) * sup (agent.py:1110-1123; has_kd = 0: loss = sup).  out[13] = {sup, terms[10], kdl, loss}. */
int magic_loss_assemble(const float* rows, int n_rows, const float* row_w, float row_scale, const float* kd_rows, int n_kd,
                        float* slots, const float* rw, float alpha, int has_kd, float* out, void* stream);

/* Row dot / row gate: s[m] = x[m] . wx (+ e[m] . we) + b0 (+ b1) over rows of H elements (H a multiple of 128, <= 1024; wx / we / b fp32).
 * mode 0: out_s[m] = s -- the value head's output Linear(512, 1) (`Critic`, agent.py:30,39; [LINEAGE DUET] state2value); mode 1: g = sigmoid(s),
 * out[m, :] = e[m, :] g, gsave[m] = g -- the 'door' gate of the causal-intervention blocks (parser.py:129-142 do_add_method).  Backward: mode 0
 * dy fp32 [M] -> dx = dy wx; mode 1 dout [M, H] -> ds = (dout . e) g (1 - g), de = dout g + ds we, dx = ds wx; dwx / dwe [H], db0 / db1 [1]
 * accumulate (fp32, may be NULL); dx / de may be NULL. */
int magic_rowgate_fwd(int dtype, int M, int H, int mode, const void* x, const void* e, const float* wx, const float* we,
                      const float* b0, const float* b1, float* out_s, void* out, float* gsave, void* stream);
int magic_rowgate_bwd(int dtype, int M, int H, int mode, const void* x, const void* e, const float* wx, const float* we, const float* gsave,
                      const float* dy, const void* dout, void* dx, void* de, float* dwx, float* dwe, float* db0, float* db1, void* stream);

/* Flat-buffer optimizer: pretrain_src/optim/adamw.py:53-112 + clip_grad_norm_ (grad_norm, r2r_magic_pretrain.json:22) */
int magic_sumsq(long long n, const float* g, float* out, void* stream);
int magic_adamw(long long n, float* p, float* g, float* m, float* v, void* shadow, int shadow_dtype,
                float lr, float b1, float b2, float eps, float wd, float step_size,
                const float* sumsq, float max_norm, float gscale, const float* lr_ss, long long n_decay, int zero_grad,
                unsigned* overflow, float* scale_state, int* sched_step, int decay_first, void* stream);
/* decay_first != 0: torch.optim.AdamW's order -- p *= 1 - lr wd, then the Adam update (the navigator's optimizer, map_nav_src/r2r/agent_base.py:122-137;
 * pass step_size = lr sqrt(1 - b2^t) / (1 - b1^t) and eps sqrt(1 - b2^t) for its `sqrt(v / bc2) + eps` denominator); 0: pretrain_src/optim/adamw.py's. */
/* scale_state (may be NULL; needs sumsq): the dynamic loss scale's state (magic_step_rng): the gradient pre-scale is multiplied by its 1 / S and
 * `pending` is set to 1 (updated) / 2 (skipped).  sched_step (may be NULL): the device-side schedule's two step words (magic_sumsq_sched /
 * magic_sched_step: [0] = global_step, the lr schedule's; [1] = the optimizer's state step, the bias correction's); a SKIPPED update takes back word [1]'s
 * advance, as a skipped optimizer.step() under GradScaler leaves the state step, and leaves word [0] advanced, as the reference's global_step is. */
/* overflow (may be NULL): a device counter.  When the gradient norm in `sumsq` is not finite (fp16 storage under a static gradient scale)
 * the update is SKIPPED -- p, m, v untouched, g zeroed when zero_grad -- and the counter incremented: amp.GradScaler.step's behaviour
 * (train_r2r_magic.py:370-371) instead of NaN weights. */
/* zero_grad != 0: g is set to 0 after it has been consumed (the next step's accumulators start from zero without a fill launch).
 * magic_sumsq_sched: magic_sumsq that also advances the device-side schedule (magic_sched_step's arithmetic) in the same launch; `out`
 * must already be zero (magic_step_rng's zero_me at the top of the step) */
int magic_sumsq_sched(long long n, const float* g, float* out, int* step, float lr0, int warmup, int total, float b1, float b2,
                      float* lr_ss, void* stream);
/* n_decay: elements [0, n_decay) take the weight decay wd, the rest none (both parameter groups of optim/misc.py:13-22 in one launch);
 * < 0: all.  device-side lr schedule + Adam bias correction (optim/sched.py:17-30, adamw.py:97-100) for HIP-graph replay; coef_dev / lr_ss
 * arguments above are optional device scalars multiplied into / replacing the host values; zero_me (optional): one float set to 0 (the
 * gradient-norm accumulator of the step that begins) */
/* step: int[2] = {global_step, optimizer state step}; lr = lr0 * warmup_linear(global_step), step size = lr sqrt(1 - b2^t) / (1 - b1^t) with t = state step + 1 */
int magic_sched_step(int* step, float lr0, int warmup, int total, float b1, float b2, float* lr_ss, float* zero_me, void* stream);
/* shadow (optional): the 16-bit copy of the parameters the MFMA kernels read, rewritten in shadow_dtype (1 | 2).  magic_cast: dtype16 = 1 | 2 names
 * the 16-bit side; to16 != 0: fp32 x -> 16-bit y, else 16-bit x -> fp32 y */
int magic_cast(int dtype16, int to16, long long n, const void* x, void* y, void* stream);
/* y += xs[0] + ... + xs[count-1] (count <= 8 device tensors of n elements, 16-byte aligned): fp32 sum, one rounding */
int magic_add_n(int dtype, long long n, int count, const void* const* xs, void* y, void* stream);
int magic_add(int dtype, long long n, const void* x, void* y, void* stream);
int magic_dact(int dtype, int kind, long long n, const void* dy, const void* z, void* dz, void* stream);

/* Feature ingest.  The precomputed CLIP view features live once in HBM as a packed table
 * [n_viewpoints, 36, D] (the reference keeps them in a host dict keyed "{scan}_{vp}", dataset.py:246-254, and re-uploads
 * every batch); out[p, j, :] = table[vp_row[p], order[p, j], :], order < 0 -> zeros (padded slots of pad_tensors,
 * common.py:9).  `order` is the reference's token order: candidate views first, then the rest (dataset.py:742-756). */
int magic_view_gather(int dtype, int Np, int V, int D, const void* table, int n_viewpoints, const int* vp_row,
                      const int* order, void* out, void* stream);

/* Contraction arithmetic of the fp32 storage mode (dtype 0) in magic_gemm / magic_gemm_dw_grouped / magic_linear_ln /
 * magic_linear_lnbwd: mode 0 = the exact v_mfma_f32_16x16x4_f32 (default); mode 1 = "bf16x3": every fp32 operand split into
 * bf16 hi + lo, a.b = a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (~2^-17 relative error
 * per product, 3/16 of the matrix-pipe time).  Process-wide, synchronous; set it before capturing graphs. */
int magic_set_f32_mfma(int mode);
int magic_get_f32_mfma(void);

/* Whole self-attention encoders in one launch (csrc/encoder.hip): the 6-block text encoder and the 2-block panorama encoder of
 * MAGIC-S (the withheld model's `bert.lang_encoder` / `img_embeddings.pano_encoder`; HF BertLayer x n), one
 * 512-thread workgroup per sample for all layers, activations resident in LDS, weights streamed from L2 as MFMA B-fragments.
 * bf16, H = 128, 2 heads, FFN 512, <= 80 tokens per sample, <= 6 layers per encoder, 1 or 2 encoders ("segments") per launch.
 * Writes exactly what the per-op backward kernels read: qkv [M,3H], P (+ Pd under dropout) [B,2,N,ldp], ctx, a = attention-block
 * output + rstd_a, z = FFN pre-activation, g = GELU output, out + rstd_o; same rounding points and dropout masks as
 * magic_gemm / magic_attn_fwd / magic_linear_ln.  `params`: host copy of magic_enc_params, fully consumed before return.
 * Round 3: EVERY WEIGHT MATRIX of magic_encoder_fwd / magic_xencoder_fwd (Wqkv, Wo, Wq, Wkv, Woc, W1, W2) and every transposed matrix of
 * magic_rowbwd (WqkvT_n, WoT, W1T, W2T) is read in MFMA-FRAGMENT ORDER (magic_pack_frag_spans of the row-major matrix): the kernels load
 * B fragments straight from L2, and a fragment of a row-major matrix costs one cache line per weight row. */
typedef struct {
  const void* Wqkv; const float* bqkv; const void* Wo; const float* bo; const float* g1; const float* be1;
  const void* W1; const float* bi; const void* W2; const float* bo2; const float* g2; const float* be2;
  void *qkv, *P, *Pd, *ctx, *a, *z, *g, *out; float *rstd_a, *rstd_o;
  unsigned site_attn, site_ao, site_out, pad_;
} magic_enc_layer;
typedef struct { const void* x; const unsigned char* kmask; int nsamp, N, ldp, nlayers; magic_enc_layer L[6]; } magic_enc_seg;
/* sync != NULL selects the ROW-SPLIT form (round 3): one workgroup per (sample, 16-row tile) instead of one per sample -- an 80-token
 * instruction is five workgroups -- with the layer outputs handed between the tiles of a sample inside the launch (write-through stores +
 * one agent-scope arrival counter per (sample, layer)).  sync: >= 4 + 6 x (samples of all segments) 32-bit words of device memory,
 * 16-byte aligned, owned by this launch until it completes; the entry point zeroes them on `stream` (a memset node under capture);
 * word 0 is set to 1 if a bounded wait gave up (never in a healthy launch).  Same outputs as the per-sample form. */
typedef struct { magic_enc_seg seg[2]; int nseg; float p_attn, p_hidden, eps, scale; const unsigned* seed;
                 unsigned* sync; int sync_words, pad2_; } magic_enc_params;
int magic_encoder_supported(int dtype, int H, int I, int nh, int N, int nlayers);
int magic_encoder_params_bytes(void);
int magic_encoder_fwd(int dtype, const void* params, int nbytes, void* stream);
/* Scheduling aid for work that runs NEXT TO a training step on another stream (the frozen MAKD teacher's forward): parks `stream` -- one
 * sleeping wave -- until the next magic_encoder_fwd launch of this process has its last workgroup on a CU, or timeout_us (<= 100000) have
 * passed.  The whole-encoder launch wants every CU's LDS; side work that starts first delays its workgroups.  HIP does not promise that two
 * streams run concurrently, so the gate protects itself: a launch that became resident within the last recent_us opens it at once (the
 * gate came late), and after 3 consecutive timeouts it switches itself off (an empty launch from then on) until the caller zeroes
 * `stats` again.  stats: 8 x uint32 of zero-initialised device memory owned by the caller, updated by the gate:
 * [0] calls, [1] opened while waiting, [2] launch already resident at entry, [3] timeouts, [4] consecutive timeouts, [5] switched off,
 * [6] calls skipped while off, [7] reserved.  No reference counterpart (torch issues everything on one stream). */
int magic_encoder_start_gate(int timeout_us, int recent_us, unsigned* stats, void* stream);
/* Do launches on two streams of this process run side by side?  The runtime maps streams onto a few hardware queues; two streams that
 * share one execute in order, and work meant to overlap (the teacher's graph beside the student's, a rollout lane beside the other, a
 * gradient exchange beside the backward) then silently serialises.  Launches a bounded waiter (<= timeout_us, 1..100000) on stream_wait
 * and its release on stream_set; after synchronising both, word2[1] == 1 says the release overtook the waiter (side by side), 2 that the
 * waiter timed out (in order).  word2: 2 x uint32 of zeroed device memory; both streams idle at the call.  MAGIC_ERR_ARG for equal
 * streams.  No reference counterpart (torch issues everything on one stream). */
int magic_stream_probe(unsigned* word2, int timeout_us, void* stream_wait, void* stream_set);
/* Health of the row-split encoder launches (magic_encoder_fwd / magic_xencoder_fwd with sync != NULL): out[0] = bounded in-launch hand-off
 * waits that GAVE UP since the process started (sticky; any value but 0 means some activations were computed from rows that never arrived
 * and the caller must stop), out[1] = whole-encoder launches that became resident.  out: 2 x uint32 of device memory; one tiny launch. */
int magic_encoder_health(unsigned* out, void* stream);

/* Cross-modal encoders in one launch (csrc/encoder.hip, xencoder_fwd_kernel): the global (map) and local (viewpoint) co-attention
 * encoders, <= 3 METER BertCrossLayer blocks each (the withheld model's `bert.{global,local}_encoder.encoder.crossattention.N`,
 * names per train_r2r_magic.py:203-206), one workgroup per sample for all layers; same limits as magic_encoder_fwd, queries and
 * context <= 80 rows.  Saves what the per-op backward reads for the self-attention, cross-attention and FFN sub-blocks. */
typedef struct {
  const void* Wqkv; const float* bqkv; const void* Wo; const float* bo; const float* g1; const float* be1;
  const void* Wq; const float* bq; const void* Wkv; const float* bkv; const void* Woc; const float* boc; const float* gc; const float* bec;
  const void* W1; const float* bi; const void* W2; const float* bo2; const float* g2; const float* be2;
  void *qkv, *P, *Pd, *ctx, *a; float* rstd_a;
  void *q, *kv, *Pc, *Pdc, *cctx, *c; float* rstd_c;
  void *z, *g, *out; float* rstd_o;
  unsigned site_attn, site_ao, site_cattn, site_co, site_out, pad_;
} magic_xenc_layer;
typedef struct {
  const void* x; const void* cx; const unsigned char* qmask; const unsigned char* cmask;
  const float* dist; const float* sprel_w; const float* sprel_b;
  int nsamp, Nq, Nk, ldps, ldpc, nlayers;
  magic_xenc_layer L[3];
} magic_xenc_seg;
/* sync != NULL: the row-split form (one workgroup per (sample, 16-row query tile); see magic_enc_params) is taken when all tiles of the
 * launch are resident at once (tiles <= CUs); same outputs */
typedef struct { magic_xenc_seg seg[2]; int nseg; float p_attn, p_hidden, eps, scale; const unsigned* seed;
                 unsigned* sync; int sync_words, pad2_; } magic_xenc_params;
int magic_xencoder_supported(int dtype, int H, int I, int nh, int Nq, int Nk, int nlayers);
int magic_xencoder_params_bytes(void);
int magic_xencoder_fwd(int dtype, const void* params, int nbytes, void* stream);

/* Forward-only row chain of a post-LN block at the frozen teacher's width (csrc/chain.hip; H = 256, FFN 1024, bf16 / fp16): everything
 * of a block that is per token, between two attention products, in one launch on 16-row tiles --
 *   y1 = LayerNorm(in Wa^T + ba + res)                     BertSelfOutput / the cross-attention output block (HF BertSelfOutput)
 *   y2 = LayerNorm(gelu(y1 W1^T + bi) W2^T + bo2 + y1)     BertIntermediate + BertOutput; skipped when W1 == NULL
 *   proj = y_last Wp^T + bp  [M, Np], Np in {H, 2H, 3H}     the next attention's Q | K | V projection; skipped when Wp == NULL
 * No dropout, nothing saved for a backward pass: the MAKD teacher is frozen (kdl.train_teacher = false,
 * pretrain_src/config/r2r_magic_pretrain.json:62-87) and runs in eval mode.  Same rounding points as magic_linear_ln / magic_gemm (y1, the
 * GELU output, y2 and proj are rounded to the 16-bit type; everything in between is fp32).  Groupable (two chains = one launch).
 * Wa, W1, W2, Wp are given in MFMA-FRAGMENT ORDER (magic_pack_frag_spans): a lane's 16-byte operand of row-major W touches one cache line per
 * weight row, the packed form reads one contiguous KB per fragment. */
typedef struct {
  int M, ld_in, Np, pad_;
  const void* in; const void* res;
  const void* Wa; const float* ba; const float* g1; const float* b1; void* y1;      /* y1 NULL: not stored */
  const void* W1; const float* bi; const void* W2; const float* bo2; const float* g2; const float* b2; void* y2;
  const void* Wp; const float* bp; void* proj;
  float eps; int pad2_;
} magic_chain_params;
int magic_chain_supported(int dtype, int H, int I);
/* dst[offs[i] ..) = the [rows[i], cols[i]] row-major 16-bit matrix at src[offs[i] ..) in fragment order: chunk ((nt (cols/32) + ks) 64 + l) of
 * 8 elements = src[(16 nt + (l & 15)) cols + 32 ks + 8 (l >> 4) .. +7].  rows % 16 == 0, cols % 32 == 0, rows cols % 2048 == 0, offs % 8 == 0;
 * host arrays of n spans (element offsets), as magic_transpose_spans. */
int magic_pack_frag_spans(const void* src, void* dst, int n, const long long* offs, const int* rows, const int* cols, void* stream);
/* One launch for all layouts a training step refreshes after AdamW has rewritten the 16-bit shadow: for span i (rows, cols multiples of 64)
 * flags[i] & 1: dst_f[offs[i] ..) = W in fragment order; flags[i] & 2: dst_tf[offs[i] ..) = W^T ([cols, rows]) in fragment order. */
int magic_layout_spans(const void* src, void* dst_f, void* dst_tf, int n, const long long* offs, const int* rows, const int* cols,
                       const int* flags, void* stream);
int magic_chain_fwd(int dtype, const void* params, int nbytes, void* stream);
/* Rows per workgroup of magic_chain_fwd.  64 = the round-4 form (the 32 x 32 x 16 product, two A fragments per weight fragment: half the
 * workgroups and half the L2 -> CU weight traffic per row, 39 us per workgroup), 32 = the round-3 form on 16 x 16 x 32 (26 us per workgroup),
 * 1 = by launch size (the default): 64-row tiles for launches with more 32-row tiles than the device has CUs (they would run in two rounds),
 * 32-row tiles otherwise.  rows = 0: query; 1 | 32 | 64: set (process-wide, takes effect at the next launch: set it before capturing graphs);
 * returns the setting in force, MAGIC_ERR_ARG for anything else.  Both forms read the same fragment-order weights and have the same rounding
 * points; their LayerNorm statistics are summed in a different order (outputs agree to the 16-bit rounding).  MAGIC_CHAIN_ROWS=32 | 64 in the
 * environment fixes the form from the start. */
int magic_chain_tile_rows(int rows);

/* Backward of the per-token half of a post-LN self-attention block on 32-row blocks (csrc/encbwd.hip): [tail of the next block: dx =
 * dQKV Wqkv + d_ao -> LayerNorm backward through this block's output norm] -> FFN input gradients (x gelu') -> LayerNorm backward through the
 * attention-output norm -> d_ctx = d_aod Wo, one launch for 1 or 2 encoders ("segments"); bf16, H = 128, FFN 512.  Reads the TRANSPOSED
 * bf16 weights (W^T, see magic_transpose_spans); writes the dY operands of the deferred weight-gradient GEMMs (d_fod, d_z, d_aod), d_ao
 * (residual of the next tail) and d_ctx (input of magic_attn_bwd); gamma / beta gradients by atomics.  dqkv_n == NULL: no tail, the
 * (d_fo, d_fod) pair is given (top block of an encoder).  kt = k-steps of 32 of the tail product: 12 (dqkv_n is [M, 3H]) or 4 (a [M, H]
 * query gradient) or 0 (no product: dx = dao_n, the plain gradient wrt the output of an encoder's last block; dqkv_n / WqkvT_n any non-NULL).  z == NULL selects the SHORT chain of a cross-modal block's query side: tail (dQ Wq + d_co) -> LayerNorm backward
 * through the self-attention output norm (y2 / rstd2 / g2 / b2 / dg2 / db2, mask site_out) -> dfo = d_ao, dfod = d_aod -> dctx = d_aod Wo. */
typedef struct {
  int M, kt;
  const void* dqkv_n; const void* WqkvT_n; const void* dao_n; const void* dfo_in; const void* dfod_in;
  const void* y2; const float* rstd2; const float* g2; const float* b2; float* dg2; float* db2;
  const void* z; const void* W2T; const void* W1T;
  const void* y1; const float* rstd1; const float* g1; const float* b1; float* dg1; float* db1;
  const void* WoT;
  void *dfo, *dfod, *dz, *daod, *dao, *dctx;
  unsigned site_out, site_ao;
  /* round 6 -- the ATTENTION BACKWARD of the block above inside the launch (what magic_attn_bwd did as a launch of its own between two
   * magic_rowbwd launches: `loss.backward()` through BertSelfAttention, map_nav_src/r2r/agent_base.py:259-262).  mode 0: none (the fields below unused).
   * mode 1: every workgroup owns 16-row tile (blk % ntile) of sample (blk / ntile) of an encoder whose samples have N <= 96 rows each (M = nsamp x N),
   *   computes the dQKV rows of block j+1 for its rows from that block's saved qkv_a [M, 3H], clean probabilities P_a [nsamp, 2, N, ldp], attention
   *   output o_a [M, H] and the d_ctx rows dctx_a [M, H] the previous launch wrote (rowsum(P dP) = dO . O; dP_init: optional fp32 [nsamp, 2, N, ldp]
   *   gradient wrt the dropped probabilities; attention dropout p_attn / site_attn regenerated), stores them to dqkv_out [M, 3H] (the dY operand of
   *   dWqkv) and runs the chain above with them as the tail's dQKV (WqkvT_n, dao_n of block j+1 as before; dqkv_n unused).
   * mode 2: the same attention backward for block 0, then only dfo = dQKV Wqkv + d_ao: the gradient wrt the encoder's input (no LayerNorm). */
  int mode, N, ntile, ldp;
  const void* qkv_a; const void* P_a; const void* o_a; const void* dctx_a; const float* dP_init; void* dqkv_out;
  unsigned site_attn, pad_;
  /* graph-distance bias of the map encoder's self-attention (`graph_sprels`, r2r_magic_model_config.json:28): dist [nsamp, N, N] fp32 or NULL (then both
   * gradient pointers NULL); d sprel_linear.weight / .bias are ADDED (one atomic pair per workgroup), as magic_attn_bwd does */
  const float* dist; float* dsprel_w; float* dsprel_b;
} magic_rowbwd_seg;
typedef struct { magic_rowbwd_seg seg[2]; int nseg, blocks0; float p_hidden; int pad1; const unsigned* seed; float p_attn, scale; } magic_rowbwd_params;
int magic_rowbwd_supported(int dtype, int H, int I);
/* 1 when a segment may carry mode != 0: 16-bit storage, H = 128, FFN 512, 2 heads of 64, samples of N <= 96 rows */
int magic_rowbwd_attn_supported(int dtype, int H, int I, int nh, int N);
int magic_rowbwd_params_bytes(void);
int magic_rowbwd(int dtype, const void* params, int nbytes, void* stream);
/* params.pad1 != 0 (round 4): dg2 / db2 / dg1 / db1 of every segment point at PARTIAL buffers, ceil(M / magic_rowbwd_rows(total rows)) x H
 * floats each, contents undefined: every workgroup STORES its LayerNorm-gradient sums in its own row instead of adding them into the
 * parameter gradients with atomics; magic_colsum_add then adds the rows up in block order.  (torch: `LayerNorm.weight.grad` accumulation.) */
int magic_rowbwd_rows(long long total_rows);
/* dsts[j][c] += sum over b < nblks[j] of parts[j][b * H + c], c < H, for n <= 96 jobs in one launch (host arrays of device pointers, consumed
 * before return); the sum runs in block order, so the result is reproducible. */
int magic_colsum_add(int H, int n, const float* const* parts, float* const* dsts, const int* nblks, void* stream);
/* dst[off_i .. off_i + rows_i*cols_i) = transpose of the row-major [rows_i, cols_i] bf16 matrix at src[off_i ..), i < n (element
 * offsets into two congruent flat buffers; host arrays, consumed before return): the transposed weight shadow of magic_rowbwd. */
int magic_transpose_spans(const void* src, void* dst, int n, const long long* offs, const int* rows, const int* cols, void* stream);

/* Grouping: between magic_group_begin() and magic_group_end(stream) up to eight calls of magic_gemm / magic_attn_fwd /
 * magic_attn_bwd / magic_linear_ln / magic_linear_lnbwd / magic_ln_bwd / magic_chain_fwd are recorded instead of launched; magic_group_end launches ONE kernel serving
 * the problems of the same kind / dtype / variant: GEMMs as one grouped launch (<= 8 problems), other kinds as pairs.
 * Records must be independent of each other.  Thread-local state. */
int magic_group_begin(void);
int magic_group_end(void* stream);

#ifdef __cplusplus
}
#endif
#endif
