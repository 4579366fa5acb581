/* domainrag_hip.h — C ABI of libdomainrag_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for Domain-RAG's retrieve-then-generate hot path.  The
 * reference (LiYu0524/Domain-RAG) has no FFI of its own: all arithmetic happens inside
 * third-party wheels behind the Python call sites cited per entry point below
 * (paths are relative to the reference checkout).  Each function here replaces the vendor
 * kernels those call sites reach.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is owned by the caller (device memory
 *     unless stated), kernels never allocate; workspace sizes come from *_workspace_bytes().
 *   - asynchronous on the caller's HIP stream (`stream` is a hipStream_t passed as void*);
 *     no hidden synchronisation.
 *   - return 0 on success, negative on error; drag_last_error() returns the message
 *     (thread-local).  No exceptions cross the ABI.
 *   - bf16 tensors are raw uint16 bit patterns; "f32" is IEEE binary32.
 */
#ifndef DOMAINRAG_HIP_H
#define DOMAINRAG_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int drag_version(void);
const char* drag_last_error(void);
/* Measurement switches (A/B of kernel variants inside one process; every setting computes the same function):
 *   "attn_sched" 0 | 1 | 2 (schedule of the attention kernel's KV-tile loop), "attn_w4" 0 | 1 (128-query blocks at any
 *   length), "attn_tune" bit 0: static wave priority, bit 1: 16-byte epilogue stores, "attn_q64" 0 | 1 | 2 (the 4-wave x 64-query
 *   kernel: by policy — joint sequences of 4096 keys and more that fill the chip | whenever S >= 1024 | never), "attn_persist" 0 | 1 | n >= 3 (persistent attention experiment: off, one workgroup per
 *   CU, n per XCD), "gemm_kernel", "gemm_group_m", "ln_generic", "topk_grid" (workgroups at most of the top-k scan, 0 = 512),
 *   "topk_depth" 0 | 3 (LDS-DMA ring depth of the scan), "topk_qt" 0 | 2 | 4 (query tiles per scan workgroup), "topk_select" 0 | 256 | 1024,
 *   "topk_dense_sample" 0 | 1 (threshold from every sampled row instead of the group maxima), "topk_path" 0 | 1 | 2 (top-k call: two launches
 *   through the row groups' maxima by policy — k <= 128 and 8 192 < N <= 131 072, or N <= 524 288 with at most 4 queries | always the
 *   sampled-threshold form | the two-launch form wherever it applies: k <= 128, 8 192 < N <= 1 048 576), "gemm_pair" 0 | 1 | 2
 *   (drag_gemm_bf16_pair: merge unless both problems fill the chip alone | never | always), "topk_qreg" 0 | 1 (d = 512 scan: query tile in
 *   registers | read from LDS per corpus chunk), "gemm_w4" 0 | 1 | 2 | 3 (gemm_bf16_w4p by policy | never | wherever it can run | launches of
 *   >= 256 tiles), "attn_walk" 0 | 2 | n (the 64-query attention kernel: one workgroup per CU walks the (batch-head, query block) items when
 *   it can | one item per workgroup | n workgroups, n a multiple of 8), "gemm_splitk" 0 | 1 | n (split-K by policy | never | n slices wherever valid: see
 *   drag_gemm_set_workspace).
 * Initial values: $DRAG_ATTN_SCHED, $DRAG_ATTN_W4, $DRAG_ATTN_TUNE, $DRAG_ATTN_Q64, $DRAG_ATTN_PERSIST, $DRAG_GEMM_KERNEL,
 * $DRAG_GEMM_GROUP_M, $DRAG_LN_GENERIC, $DRAG_TOPK_GRID, $DRAG_TOPK_DEPTH, $DRAG_TOPK_QT, $DRAG_TOPK_SELECT, $DRAG_TOPK_DENSE_SAMPLE, $DRAG_TOPK_PATH, $DRAG_GEMM_PAIR, $DRAG_TOPK_QREG, $DRAG_GEMM_W4, $DRAG_ATTN_WALK, $DRAG_GEMM_SPLITK.  Returns 0, or -1 for an unknown name. */
int drag_set_option(const char* name, int32_t value);
/* 1 when the library was built with DRAG_EXPERIMENTS=1 and carries the kernels behind "attn_persist", "attn_sched" = 3 and "topk_qt"
 * (measured non-improvements kept for their A/B records), else 0.  A pure query: no option is touched. */
int drag_experiments_built(void);

/* activation codes used by epilogues */
#define DRAG_ACT_NONE 0
#define DRAG_ACT_GELU_TANH 1
#define DRAG_ACT_SILU 2
#define DRAG_ACT_QUICK_GELU 3
#define DRAG_ACT_GELU_ERF 4

/* ---------------------------------------------------------------------------------------
 * drag_gemm_bf16 — C = epi(A[M,K] · W[N,K]^T), bf16 in, fp32 accumulate on MFMA.
 * Replaces torch.nn.Linear inside FluxTransformer2DModel / SiglipVisionModel / ReduxImageEncoder /
 * clip VisionTransformer, reached from batch_generate_flux_kshot.py:459-474,
 * outpainting_updown_sampling_redux.py:1237-1257 and
 * retrieval/clip100_resnet_style_all_shots.py:171.
 *   logical row r of A lives at A + (r / a_rows_per_batch) * a_batch_stride + (r % a_rows_per_batch) * lda
 *   (elements); same for C / resid with the c_* fields; *_rows_per_batch <= 0 means "one batch".
 *   epilogue:  v = acc + bias[n];  v = act(v) for n >= act_n0  (act: NONE, GELU_TANH, SILU or QUICK_GELU);
 *              gate != NULL:  C = resid + gate[r / c_rows_per_batch, n] * v   (bf16-rounded like torch)
 *              else resid != NULL: C = resid + v
 *   Requires K % 64 == 0, N % 4 == 0, lda % 8 == 0, ldc % 4 == 0, A span < 2 GiB.
 */
typedef struct drag_gemm_args {
  const void* A;
  const void* W;
  void* C;
  const void* bias;
  const void* gate;
  const void* resid;
  int32_t M, N, K;
  int32_t lda, a_rows_per_batch;
  int64_t a_batch_stride;
  int32_t ldc, c_rows_per_batch;
  int64_t c_batch_stride;
  int32_t ldg;
  int32_t act, act_n0;
  int32_t out_f32;
  /* optional second destination (NULL = none): output columns >= n_split are written to C2 as dense rows of ldc2 elements
   * (column n -> C2[m * ldc2 + n - n_split]); columns < n_split go to C as usual.  n_split % 256 == 0; bf16 outputs, dense rows,
   * no gate / residual.  Two Linears over one input as ONE launch (Flux single block: to_q|k|v and proj_mlp + GELU via act_n0). */
  void* C2;
  int32_t ldc2, n_split;
} drag_gemm_args;
int drag_gemm_bf16(const drag_gemm_args* args, void* stream);
/* drag_gemm_bf16_pair — two Linears with their own operands but the same N, K, activation, output type and epilogue form as ONE
 * launch (the second problem's rows follow the first one's in the tile walk).  Bit-identical to drag_gemm_bf16(a); drag_gemm_bf16(b):
 * every output element keeps its MFMA chain and epilogue; what changes is the occupancy of launches that are small alone.  Replaces
 * the text-stream / image-stream Linear pairs of a FluxTransformerBlock (to_q|k|v + add_q|k|v_proj, to_out + to_add_out, ff + ff_context)
 * reached from outpainting_updown_sampling_redux.py:1246-1257 and batch_generate_flux_kshot.py:467-474.  No C2 destinations. */
int drag_gemm_bf16_pair(const drag_gemm_args* a, const drag_gemm_args* b, void* stream);
/* 1 when drag_gemm_bf16_pair runs these two problems as one launch, 0 when it issues them one after the other (both fill the chip
 * alone; option "gemm_pair": 1 = never merge, 2 = always) — for callers that account launches */
int drag_gemm_bf16_pair_merges(int M1, int M2, int N, int K);
/* which kernel a launch over M1 rows (M2 > 0: a merged pair of M1 + M2 rows) takes under the tile policy: 2 = the persistent 256x256
 * kernel, 0 = the 128x128 kernel, else 100 * (192-column tiles) + 10 * MI + ST of gemm_bf16_deep<MI, ST, NI>.  Every choice computes the
 * same bits; the policy is a function of the launch shape only. */
int drag_gemm_bf16_choice(int M1, int M2, int N, int K);
/* the kernel drag_conv3x3_bf16 dispatches M = B*Ho*Wo output pixels x Cout x (9*Cin) to: 2 = the 256x256 kernel, 0 = the 128x128 one */
int drag_conv3x3_bf16_choice(int64_t M, int Cout, int Cin);
/* the policy's cost model for that launch: (tile rounds on the busiest CU) x (tile rows + tile columns), i.e. proportional to what the
 * busiest CU pulls through its L2 -> LDS path per K-step; comparable between launches of equal K.  0 under a forced "gemm_kernel". */
int64_t drag_gemm_bf16_cost(int M1, int M2, int N, int K);
/* Split-K (round 5): a device buffer drag_gemm_bf16 may use for f32 partial products on the CURRENT device ((null, 0) takes it back).
 * With one registered, a Linear of at most 96 output tiles of 256 x 256 and K >= 12 288 without activation — M, N multiples of 256, one dense
 * A batch, bf16 output: the single blocks' proj_out and the ff down-projections at batch 1 (transformer_flux.py FluxSingleTransformerBlock /
 * FeedForward reached from the pipelines above at 512 x 512, batch 1) — runs as S stacked K slices in one launch plus one pass that adds
 * them in order and applies bias / gate / residual: same arithmetic per slice, the sum of S f32 chains instead of one (the only launch
 * whose last bit depends on the kernel choice; "gemm_splitk" = 1 switches it off, n >= 2 asks for n slices wherever they fit).  Needs
 * S * M * N * 4 bytes; launches that use the buffer must be ordered on one stream.  Nothing is allocated on the launch path. */
int drag_gemm_set_workspace(void* ptr, int64_t bytes);
/* the slices drag_gemm_bf16 would use for these arguments right now (0: one launch, not split) — host-only, for callers that account launches */
int drag_gemm_bf16_splitk_slices(const drag_gemm_args* args);
/* the same for drag_gemm_bf16_pair(a, b): a pair that splits runs ONE partial launch over both problems' rows (same row stride of A required)
 * and one reduce pass per problem */
int drag_gemm_bf16_pair_splitk_slices(const drag_gemm_args* a, const drag_gemm_args* b);

/* ---------------------------------------------------------------------------------------
 * drag_conv3x3_bf16 — 3x3 convolution as an implicit GEMM on the same MFMA main loop.
 * Replaces torch.nn.Conv2d(k=3) inside AutoencoderKL.encode/decode (Flux VAE; diffusers 0.33.1,
 * un-vendored; reached at the start/end of pipe(...) / pipe_fill(...):
 * batch_generate_flux_kshot.py:467-474, outpainting_updown_sampling_redux.py:1246-1257).
 *   x: NHWC bf16 with a zero halo, [B, Hp, Wp, Cin], Cin % 64 == 0 (zero-pad channels);
 *   w: bf16 [Cout, 3, 3, Cin] (KRSC);  y: rows m = (b, yo, xo) of ldy elements (NHWC, no halo);
 *   output pixel (yo, xo) reads input pixels (yo*stride + oy + r, xo*stride + ox + s), r,s in 0..2
 *   (pad=1 stride=1: halo 1, oy = ox = 0;  diffusers' Downsample2D pad (0,1,0,1) stride 2: oy = ox = 1).
 *   epilogue: y = act(conv + bias) or resid + (conv + bias)   (resid addressed like y).
 */
typedef struct drag_conv_args {
  const void* x;
  const void* w;
  void* y;
  const void* bias;
  const void* resid;
  int32_t B, Ho, Wo, Hp, Wp, Cin, Cout, ldy, stride, oy, ox, act;
} drag_conv_args;
int drag_conv3x3_bf16(const drag_conv_args* args, void* stream);

/* ---------------------------------------------------------------------------------------
 * drag_qk_norm_rope_vt_bf16 — per-head RMSNorm(128) on q,k + interleaved-pair RoPE, in place,
 * and V -> V^T repack for the attention kernel.  Replaces FluxAttnProcessor2_0's norm_q/norm_k/
 * norm_added_q/norm_added_k + apply_rotary_emb (diffusers 0.33.1, un-vendored; call sites as above).
 *   qkv: [B, S, ld] rows, q at column 0, k at column H*128, v at column 2*H*128.
 *   rows s < s_txt use (wq_txt, wk_txt), rows >= s_txt use (wq_img, wk_img) (bf16 [128] each);
 *   all four NULL = no RMSNorm.  rope_cos / rope_sin: f32 [S, 64]; both NULL = no RoPE (ViT encoders:
 *   only the V^T repack runs).
 *   vt: [B, H, 128, s_pad] bf16, s_pad = ceil(S/64)*64, keys permuted inside each group of 16
 *       (position 8h+j  <->  key 8*(j>>2) + 4h + (j&3)) to match the MFMA accumulator layout; the
 *       pad keys are written as zeros.
 */
int drag_qk_norm_rope_vt_bf16(void* qkv, void* vt, const void* wq_txt, const void* wk_txt,
                              const void* wq_img, const void* wk_img, const float* rope_cos,
                              const float* rope_sin, int32_t B, int32_t S, int32_t H, int32_t ld,
                              int32_t s_txt, float eps, void* stream);

/* drag_attention_bf16 — softmax(q k^T / sqrt(128)) v, flash-style on MFMA, head_dim 128, no mask.
 * Replaces F.scaled_dot_product_attention in FluxAttnProcessor2_0.
 *   q, k: [B, S, *] rows with row stride ld_qk (elements), head h at column h*128 from q / k.
 *   vt as written by drag_qk_norm_rope_vt_bf16.  out: [B, S, *] row stride ld_o, head h at column h*128.
 */
int drag_attention_bf16(const void* q, const void* k, const void* vt, void* out, int32_t B,
                        int32_t S, int32_t H, int32_t ld_qk, int64_t qk_batch_stride, int32_t ld_o,
                        int64_t o_batch_stride, float scale, void* stream);

/* The q third of drag_qk_norm_rope_vt_bf16 fused into the attention kernel (one read + one write of [M, D] less per
 * attention): drag_k_norm_rope_vt_bf16 normalises / rotates k and writes V^T only, and drag_attention_qprep_bf16 takes
 * the RAW q projection and applies norm_q / norm_added_q (rows < s_txt: wq_txt, else wq_img) and apply_rotary_emb
 * (rope_cos / rope_sin f32 [S, 64]) — all four required — to each query row while it loads its fragments,
 * with the rounding points of the separate pass.  FluxAttnProcessor2_0 as reached from batch_generate_flux_kshot.py:467-474,
 * outpainting_updown_sampling_redux.py:1246-1257. */
int drag_k_norm_rope_vt_bf16(void* qkv, void* vt, const void* wk_txt, const void* wk_img, const float* rope_cos,
                             const float* rope_sin, int32_t B, int32_t S, int32_t H, int32_t ld, int32_t s_txt,
                             float eps, void* stream);
int drag_attention_qprep_bf16(const void* q, const void* k, const void* vt, void* out, int32_t B, int32_t S, int32_t H,
                              int32_t ld_qk, int64_t qk_batch_stride, int32_t ld_o, int64_t o_batch_stride, float scale,
                              const void* wq_txt, const void* wq_img, const float* rope_cos, const float* rope_sin,
                              int32_t s_txt, float eps, void* stream);
/* The same attention over q | k | v AS THE LINEARS WROTE THEM: v is read row-major from the projection buffer (row stride
 * ld_qk, batch stride qk_batch_stride, like q and k) through the LDS transpose read of gfx950 — no V^T pass, no V^T buffer;
 * drag_k_norm_rope_vt_bf16 is then called with vt = NULL (k only).  The fused q preparation is optional: all four of wq_txt /
 * wq_img / rope_cos / rope_sin, or none.  Same MFMA operands as the V^T kernels: identical bits. */
int drag_attention_v_bf16(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t S, int32_t H, int32_t ld_qk,
                          int64_t qk_batch_stride, int32_t ld_o, int64_t o_batch_stride, float scale, const void* wq_txt,
                          const void* wq_img, const float* rope_cos, const float* rope_sin, int32_t s_txt, float eps, void* stream);

/* Which kernel a drag_attention*_bf16 call over S keys is dispatched to: 641 / 640 = attention_q64g_kernel (4 waves x 64 queries, the KV
 * loop one generated instruction stream: the DiT's calls) with / without the scale fold (with: only under the fused q preparation),
 * 64 = attention_q64_kernel (the hand-placed form it replaces; odd KV tile counts), 8 / 4 = attention_d128_kernel with 8 / 4 waves.
 * A pure query (no launch) for callers that account launches per kernel — bench.py's roofline.attention; nothing in the reference
 * corresponds (SDPA picks its backend inside torch). */
int drag_attention_bf16_choice(int32_t S, int32_t v_row_major, int32_t q_prep);

/* drag_layernorm_modulate_bf16 — y = LN(x) * (1 + scale[b]) + shift[b]   (no affine LN, eps given)
 * or, with gamma/beta != NULL, y = LN(x) * gamma + beta (affine LayerNorm, scale/shift NULL).
 * Replaces AdaLayerNormZero / AdaLayerNormZeroSingle / AdaLayerNormContinuous and nn.LayerNorm.
 *   x rows: batched-row addressing (rows_per_batch, x_batch_stride, ldx); y dense rows of ldy.
 *   scale, shift: bf16 [B, ld_mod] row b.
 */
int drag_layernorm_modulate_bf16(const void* x, void* y, const void* scale, const void* shift,
                                 const void* gamma, const void* beta, int32_t M, int32_t D,
                                 int32_t ldx, int32_t rows_per_batch, int64_t x_batch_stride,
                                 int32_t ldy, int32_t ld_mod, float eps, void* stream);

/* elementwise helpers (bf16) */
int drag_act_bf16(const void* x, void* y, int64_t n, int32_t act, void* stream);
/* sinusoidal embedding as diffusers get_timestep_embedding(t, dim, flip_sin_to_cos=True,
 * downscale_freq_shift=0): out[b] = [cos(t*f_i) | sin(t*f_i)], f_i = exp(-ln(1e4) * i / (dim/2)); bf16 out */
int drag_timestep_embedding_bf16(const float* t, void* out, int32_t B, int32_t dim, void* stream);
/* FlowMatchEulerDiscreteScheduler.step: x = bf16(float(x) + dt * float(v)) */
int drag_flow_euler_step_bf16(void* x, const void* v, float dt, int64_t n, void* stream);
/* y = a + b (bf16) */
int drag_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);
/* f32 <-> bf16 casts */
int drag_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);
int drag_cast_bf16_to_f32(const void* x, float* y, int64_t n, void* stream);

/* uint8 RGB [B,H,W,3] -> normalised patch rows [B*(H/P)*(W/P), ldo] bf16 ((u8/255 - mean)/std), column
 * k = c*P*P + py*P + px (= flattened Conv2d(3, D, P, stride=P) weight), extra columns zero.  Front end of
 * clip VisionTransformer.conv1 (retrieval/clip100_resnet_style_all_shots.py:171) and SiglipVisionEmbeddings
 * (inside pipe_prior_redux, outpainting_updown_sampling_redux.py:1237). mean3/std3 are HOST pointers. */
int drag_patchify_u8(const void* img, void* out, int32_t B, int32_t H, int32_t W, int32_t P, int32_t ldo,
                     const float* mean3, const float* std3, void* stream);
/* out[g] = sum_n scales[g*N+n] * x[g, n]  (bf16 ops as torch: `prompt_embeds *= scale; torch.sum(dim=0)`,
 * FluxPriorReduxPipeline.__call__; scales is a DEVICE f32 array [G*N]; x [G, N, elems], out [G, elems]) */
int drag_scale_sum_bf16(const void* x, const float* scales, void* out, int32_t G, int32_t N, int64_t elems,
                        void* stream);

/* ---------------------------------------------------------------------------------------
 * drag_cosine_topk_f32 — exact inner-product top-k (faiss.IndexFlatIP.add/search,
 * retrieval/clip100_resnet_style_all_shots.py:425-434).
 *   corpus f32 [N, d] (d % 64 == 0, d <= 1024), queries f32 [Q, d]; out_d f32 [Q, k] descending,
 *   out_i int64 [Q, k].  Score = fp32 fma chain in the fixed order documented in oracle/topk.c;
 *   ties -> lower index first; k <= 2048; if k > N the tail is (-FLT_MAX, -1) like faiss.
 *   Up to 64 queries share ONE pass over the corpus (more: ceil(Q / 64) passes; 32 or 16 per pass when N is so large that 64
 *   candidate regions would pass 1 GiB).  For k <= 128 and 8 192 < N <= 131 072 (up to 524 288 rows for at most 4 queries) the call is TWO launches: the corpus pass writes
 *   the scores and the best (score, index) of every group of 16 rows; one workgroup per query takes the k-th best group maximum as
 *   the bound (exactly k groups reach it and hold the whole answer), reads those groups' 16 k scores and ranks them.  Otherwise four
 *   launches: a strided
 *   8192-row sample -> its k-th best per query (a valid lower bound of the answer's k-th best) -> the corpus pass keeping only
 *   scores above it (per-wave candidate regions, no atomics, no [Q, N] score matrix) -> select + sort + decode on the candidates.
 *   Deterministic; results do not depend on Q or on how queries are grouped into calls.
 *   workspace: drag_cosine_topk_workspace_bytes(N, Q) bytes of device memory (candidate regions sized for all N rows per query
 *   of a pass: about 8 * N * min(Q, 64) bytes + 5 MiB, at most ~1 GiB + 128 * N).
 */
int64_t drag_cosine_topk_workspace_bytes(int64_t N, int32_t Q);
int drag_cosine_topk_f32(const float* corpus, const float* queries, int64_t N, int32_t d,
                         int32_t Q, int32_t k, float* out_d, int64_t* out_i, void* workspace,
                         void* stream);
/* Scan pass of the above alone (ONE pass over the corpus, Q <= 64): scores f32 [Q, ceil64(N)] = corpus . queries in the same fixed
 * summation order (what faiss computes before its heap; retrieval/clip100_resnet_style_all_shots.py:431).  Entries
 * N..ceil64(N) of a score row are unspecified. */
int drag_cosine_scores_f32(const float* corpus, const float* queries, int64_t N, int32_t d, int32_t Q,
                           float* scores, void* stream);
/* ResNet50 stem style vector: conv1 7x7/2 (no bias) + eval-BatchNorm (folded scale/shift) + ReLU + maxpool 3x3/2,
 * then per-channel mean and sqrt(unbiased var + eps) -> out f32 [B, 128] = [mean(64) | std(64)].
 * Replaces ResNetEncoder.forward + calc_mean_std (retrieval/clip100_resnet_style_all_shots.py:51-74,197-200).
 * img f32 [B,3,H,W] in [0,1]; workspace: drag_resnet_stem_style_workspace_bytes(B,H,W) bytes (pooled maps). */
int64_t drag_resnet_stem_style_workspace_bytes(int32_t B, int32_t H, int32_t W);
int drag_resnet_stem_style_f32(const float* img, const float* conv_w, const float* bn_scale, const float* bn_shift,
                               float* out, int32_t B, int32_t H, int32_t W, float eps, void* workspace, void* stream);
/* Pillow-exact 8-bit separable resample of interleaved HWC uint8 images (bit-identical to PIL.Image.resize for
 * BILINEAR / BICUBIC / LANCZOS on modes L, RGB and RGBX; PIL premultiplies alpha for RGBA, which this does not).  Replaces the PIL resize inside openai-CLIP's `preprocess` (Resize(224, BICUBIC) +
 * CenterCrop; retrieval/clip100_resnet_style_all_shots.py:209,171,270-287) and SiglipImageProcessor's 384x384 BICUBIC
 * inside FluxPriorReduxPipeline (batch_generate_flux_kshot.py:459-465, outpainting_updown_sampling_redux.py:1237-1243).
 * The caller supplies Pillow's fixed-point tables (device int32): per output index `bounds` = (first source index,
 * tap count) and `kk` = ksize weights of 22 fractional bits; out = clamp8(((1<<21) + sum pixel*kk) >> 22).  Both
 * passes: horizontal into `tmp` (rows [tmp_row0, tmp_row0+tmp_rows) of the source, batch*tmp_rows*out_w*channels
 * bytes), then vertical.  kx == NULL / ky == NULL skips that pass (that axis is then the window starting at
 * src_col0 / src_row0); a crop of the resized image = tables sliced to the crop.  All strides in bytes. */
typedef struct drag_resample_args {
  const uint8_t* src;
  uint8_t* dst;
  uint8_t* tmp;
  int32_t batch, channels;
  int32_t src_h, src_w;
  int64_t src_image_stride;
  int32_t src_row_stride;
  int32_t out_h, out_w;
  int64_t dst_image_stride;
  int32_t dst_row_stride;
  const int32_t* kx;
  const int32_t* bx;
  int32_t ksize_x;
  const int32_t* ky;
  const int32_t* by;
  int32_t ksize_y;
  int32_t tmp_row0, tmp_rows;
  int32_t src_col0, src_row0;
} drag_resample_args;
int drag_resample_u8(const drag_resample_args* args, void* stream);
/* float NCHW (already normalised, e.g. clip `preprocess` output) -> patch rows like drag_patchify_u8 */
int drag_patchify_f32_nchw(const float* img, void* out, int32_t B, int32_t H, int32_t W, int32_t P, int32_t ldo,
                           void* stream);
/* L2-normalise rows in place (image_embedding / image_embedding.norm(dim=-1), retrieval/...:172) */
int drag_l2_normalize_f32(float* x, int64_t rows, int32_t d, void* stream);

/* ---------------------------------------------------------------------------------------
 * Flux VAE helpers (AutoencoderKL.encode/decode, VaeImageProcessor, FluxFillPipeline.prepare_mask_latents,
 * FluxPipeline._pack_latents/_unpack_latents — diffusers 0.33.1, un-vendored; reached from
 * batch_generate_flux_kshot.py:467-474 and outpainting_updown_sampling_redux.py:1246-1257).
 * Activations are NHWC bf16; "haloed" buffers are [B, H+2, W+2, C] with a zero border that kernels never write.
 */
/* GroupNorm(groups=32) (+ SiLU) over NHWC [B,H,W,C] -> y, optionally into a haloed buffer (out_pad=1). */
int64_t drag_groupnorm_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C);
int drag_groupnorm_silu_bf16(const void* x, void* y, const void* gamma, const void* beta, int32_t B,
                             int32_t H, int32_t W, int32_t C, int32_t groups, int32_t out_pad,
                             int32_t silu, float eps, void* workspace, void* stream);
/* copy NHWC [B,H,W,C] into the interior of a haloed [B, up*H+2, up*W+2, C] buffer, nearest-upsampling by `upsample` (1|2) */
int drag_pad_copy_bf16(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t upsample,
                       void* stream);
/* y[r, :cols] = softmax(x[r] * scale), y[r, cols:ldy] = 0; f32 in (dense rows of `cols`), bf16 out (row stride ldy)
 * (VAE mid-block attention, softmax upcast; the zero tail is the K padding of the following P·V GEMM) */
int drag_softmax_rows_f32_bf16(const float* x, void* y, int64_t rows, int32_t cols, int32_t ldy, float scale,
                               void* stream);
/* packed tokens [B, h*w, ld] (64 features) -> haloed NHWC latents [B, 2h+2, 2w+2, C]: tok / scaling + shift */
int drag_unpack_latents_bf16(const void* tokens, void* y, int32_t B, int32_t h, int32_t w, int32_t ld, int32_t C,
                             float scaling, float shift, void* stream);
/* encoder moments NHWC [B,H,W,ldm] (mean|logvar) (+ noise NCHW [B,16,H,W], may be NULL = mode) ->
 * packed tokens [B,(H/2)(W/2), ld]: ((mean + std*noise) - shift) * scaling */
int drag_sample_pack_latents_bf16(const void* moments, const void* noise, void* tokens, int32_t B, int32_t H,
                                  int32_t W, int32_t ldm, int32_t ld, float scaling, float shift, void* stream);
/* uint8 RGB [B,H,W,3] (+ uint8 mask [B,H,W], may be NULL) -> haloed NHWC bf16: (2*u8/255-1) * (1-binarised mask) */
int drag_image_preprocess_u8(const void* img, const void* mask, void* y, int32_t B, int32_t H, int32_t W, int32_t C,
                             void* stream);
/* rows [npix, ld] bf16 (3 real channels) -> uint8 RGB [npix,3]: ((x/2+0.5).clamp(0,1)*255).round() */
int drag_image_postprocess_u8(const void* x, void* out, int64_t npix, int32_t ld, void* stream);
/* uint8 mask [B,H,W] -> 256 mask features per token (8x8 pixel-unshuffle then 2x2 pack), tokens [B,(H/16)(W/16), ld] */
int drag_mask_pack_u8(const void* mask, void* tokens, int32_t B, int32_t H, int32_t W, int32_t ld, void* stream);
/* strided forms on token rows: x[r, :cols] += dt * v[r, :cols];  x = sigma*noise + (1-sigma)*x (scale_noise) */
int drag_flow_euler_rows_bf16(void* x, const void* v, int64_t rows, int32_t cols, int32_t ldx, int32_t ldv, float dt,
                              void* stream);
int drag_scale_noise_rows_bf16(void* x, const void* noise, int64_t rows, int32_t cols, int32_t ldx, int32_t ldn,
                               float sigma, void* stream);

/* ---------------------------------------------------------------------------------------
 * LaMa inpainting stage (lama_inpaint/lama_inpaint.py:172-215 -> simple_lama_inpainting.SimpleLama.__call__, un-vendored:
 * prepare_img_and_mask, the big-lama FFCResNetGenerator, the blend of DefaultInpaintingTrainingModule.forward).
 * float32 throughout like the reference; activations NHWC f32.
 */
#define DRAG_PAD_ZERO 0
#define DRAG_PAD_REFLECT 1
#define DRAG_CONV_ACT_NONE 0
#define DRAG_CONV_ACT_RELU 1
#define DRAG_CONV_ACT_SIGMOID 2
#define DRAG_CONV_ACT_QUICK_GELU 3 /* x * sigmoid(1.702 x), openai-CLIP's QuickGELU */
typedef struct drag_conv2d_f32_args {
  const float* x;      /* NHWC [B, Hi, Wi, ldx]; the Cin channels read start at x (pre-offset the pointer for a channel slice).
                          ldx < Cin is allowed for KW = 1, pad = 0: the read runs on into the next pixels of the row ("packed row":
                          a KH x KW x C kernel given as KH x 1 x (KW*C)) */
  const float* w;      /* [Cout, KH, KW, Cin] (nn.Conv2d weight permuted 0,2,3,1; nn.ConvTranspose2d weight permuted 1,2,3,0) */
  float* y;            /* NHWC [B, Ho, Wo, ldy], Cout channels written from y */
  const float* scale;  /* [Cout] or NULL (= 1): eval-mode BatchNorm gamma / sqrt(var + eps) */
  const float* shift;  /* [Cout] or NULL (= 0): beta - mean * scale, or the conv bias */
  const float* addend; /* NHWC [B, Ho, Wo, ld_add] added BEFORE scale/shift (the other FFC branch), or NULL */
  const float* resid;  /* NHWC [B, Ho, Wo, ld_res] added AFTER the activation (FFCResnetBlock identity), or NULL */
  int32_t B, Hi, Wi, Cin, ldx, Ho, Wo, Cout, ldy, ld_add, ld_res;
  int32_t KH, KW, stride, pad, pad_mode, transposed, act;
} drag_conv2d_f32_args;
/* y = act((conv(x, w) + addend) * scale + shift) + resid.  transposed = 1: the gather form of nn.ConvTranspose2d
 * (out[o] += in[i] * w[k] where o = i*stride - pad + k); Ho / Wo then carry the output_padding. */
int drag_conv2d_f32(const drag_conv2d_f32_args* args, void* stream);
/* torch.fft.rfftn(x, dim=(-2,-1), norm="ortho") of C maps x NHWC [B,H,W,ldx] -> y [B,H,W/2+1,2C] with channel 2c = real, 2c+1 =
 * imaginary (FourierUnit's stacking).  tw_w / tw_h: float2 tables (cos, sin)(2 pi j / n) for n = W / H, float64-evaluated by the
 * caller; tmp: B*H*(W/2+1)*2C floats. */
int drag_rfft2_f32(const float* x, float* tmp, float* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx,
                   const float* tw_w, const float* tw_h, void* stream);
/* torch.fft.irfftn(f, s=(H,W), dim=(-2,-1), norm="ortho") of f [B,H,W/2+1,2C] -> y NHWC [B,H,W,ldy] (+ add [B,H,W,ld_add] or NULL) */
int drag_irfft2_f32(const float* f, float* tmp, float* y, const float* add, int32_t B, int32_t H, int32_t W, int32_t C,
                    int32_t ldy, int32_t ld_add, const float* tw_w, const float* tw_h, void* stream);
/* prepare_img_and_mask + the model's input: uint8 RGB [H,W,3], uint8 mask [H,W] -> NHWC f32 [Hp,Wp,4] =
 * (img/255 * (1-m), m), m = mask > 0, both padded bottom/right to (Hp,Wp) like np.pad(mode="symmetric") */
int drag_lama_prepare_u8(const void* img, const void* mask, float* x, int32_t H, int32_t W, int32_t Hp, int32_t Wp, void* stream);
/* m * pred + (1-m) * img/255 -> *255, clip, truncate -> uint8 RGB [Hp,Wp,3]; pred NHWC f32 [Hp,Wp,ld] */
int drag_lama_blend_u8(const float* pred, int32_t ld, const void* img, const void* mask, void* out, int32_t H, int32_t W,
                       int32_t Hp, int32_t Wp, void* stream);

/* ---------------------------------------------------------------------------------------
 * float32 CLIP ViT-B/32 image tower (clip.model.VisionTransformer.forward, openai/CLIP@dcba3cb, un-vendored; reached through
 * model.encode_image at retrieval/clip100_resnet_style_all_shots.py:171,284,337,948).  openai-CLIP computes in fp32 on the CPU
 * and fp16 on CUDA; this path is fp32 on the GPU.  Matrix products run on drag_conv2d_f32 (patch embedding = stride-P conv,
 * Linear = 1x1 conv); these are the remaining pieces.
 */
/* uint8 RGB pixels [npix,3] -> NHWC f32 [npix,4]: ((u/255 - mean)/std, 0)  (ToTensor + Normalize, two IEEE divisions) */
int drag_vit_prepare_u8(const void* img, float* out, int64_t npix, const float* mean3, const float* std3, void* stream);
/* normalised float NCHW [B,3,hw] (what clip's `preprocess` returns) -> NHWC f32 [B*hw,4] */
int drag_vit_prepare_f32(const float* img, float* out, int32_t B, int64_t hw, void* stream);
/* y[r,:D] = LayerNorm(x[r,:D]) * gamma + beta, rows at strides ldx / ldy (floats); D <= 1024 */
int drag_layernorm_f32(const float* x, float* y, const float* gamma, const float* beta, int64_t rows, int32_t D, int64_t ldx,
                       int64_t ldy, float eps, void* stream);
/* x[b,0] = ln_pre(class_embedding + pos[0]); x[b,1+p] = ln_pre(emb[b,p] + pos[1+p]); emb [B,T-1,D], x [B,T,D] */
int drag_clip_embed_ln_f32(const float* emb, const float* cls, const float* pos, const float* gamma, const float* beta, float* x,
                           int32_t B, int32_t T, int32_t D, float eps, void* stream);
/* out[b,t,h*hd:(h+1)*hd] = softmax(q k^T * scale) v per (image, head); qkv rows [B*T, ld] = (q | k | v), each H*head_dim wide;
 * T <= 64, head_dim <= 64 */
int drag_attention_small_f32(const float* qkv, float* out, int32_t B, int32_t T, int32_t H, int32_t head_dim, int32_t ld,
                             int32_t ldo, float scale, void* stream);

/* ---------------------------------------------------------------------------------------
 * Baseline JPEG decode on the GPU: a batch of FILES as one byte blob -> RGB pixels, byte-identical to
 * PIL.Image.open(f).convert("RGB") (libjpeg-turbo defaults: ISLOW IDCT, fancy upsampling, YCbCr -> RGB tables) — the decode
 * the reference pays per corpus image at retrieval/clip100_resnet_style_all_shots.py:270-281.
 *   drag_jpeg_parse: data = concatenated files, offsets int64 [n+1] (device); writes one descriptor per file (device).
 *     status 0 = decodable here; 1 not a JPEG, 2 truncated header, 3 progressive / arithmetic / lossless, 4 not 8-bit,
 *     5 not grey / YCbCr, 6 sampling other than 4:4:4 / 4:2:2 / 4:2:0, 7 multi-scan, 8 table problem, 9 chroma <= 2 samples
 *     wide, 11 more than 2^24 pixels.  Files with a non-zero status are skipped by drag_jpeg_decode_rgb (the caller decodes those few elsewhere).
 *   drag_jpeg_decode_rgb: plan int64 [n, 3] (device) = per file: offset into coef_ws (int16 elements), into plane_ws
 *     (bytes), into out_rgb (bytes); a file needs 64 * blocks int16 of coefficients and 64 * blocks bytes of planes where
 *     blocks = sum over components of (mcus_x * hs) * (mcus_y * vs), and width * height * 3 output bytes ([H, W, 3]).
 *     max_blocks / max_pixels: the largest per-file block / pixel count of the batch (launch geometry); coef_bytes: size of
 *     coef_ws (zeroed by the call); qtab_ws: n * 3 * 64 uint16.  n <= 65535.
 *     scan_status int32 [n] (device): 0 when the entropy-coded data ended at the EOI marker as a clean file's does, 1 otherwise
 *     (cut short, trailing data, damaged): such a file's pixels are what this decoder made of it, and the caller should let
 *     libjpeg / PIL decide instead (they warn or raise).
 */
typedef struct drag_jpeg_info {
  int32_t status;
  int32_t width, height, ncomp;
  int32_t hs[3], vs[3];          /* sampling factors per component */
  int32_t tq[3], td[3], ta[3];   /* quantisation / DC / AC table ids */
  int32_t hmax, vmax, mcus_x, mcus_y;
  int32_t restart_interval;
  int32_t scan_off;              /* first entropy-coded byte */
  int32_t dqt_off[4], dqt_16[4]; /* table offsets inside the file (-1 = absent), 16-bit flag */
  int32_t dht_off[8];            /* [class * 4 + id] */
  int32_t progressive;          /* 1 = SOF2 (scan_off then names the first SOS marker; td / ta are per scan) */
  int32_t cid[3];               /* component identifiers */
  int32_t reserved[3];
} drag_jpeg_info;
int drag_jpeg_parse(const void* data, const int64_t* offsets, int32_t n, drag_jpeg_info* info, void* stream);
int drag_jpeg_decode_rgb(const void* data, const int64_t* offsets, const drag_jpeg_info* info, const int64_t* plan, int32_t n,
                         int64_t max_blocks, int64_t max_pixels, void* coef_ws, int64_t coef_bytes, void* plane_ws,
                         void* qtab_ws, void* out_rgb, int32_t* scan_status, void* stream);

/* cv2.resize(img, (out_w, out_h)) (INTER_LINEAR, 8-bit path) of n differently sized RGB images + /255 -> f32 [n, 3, out_h, out_w]:
 * the input of the stem style vector (compute_resnet_features, retrieval/clip100_resnet_style_all_shots.py:186-196).  src: blob of
 * [h, w, 3] uint8 images at byte offsets src_off[n]; hw int32 [n, 2]; tab int32 [n, 8, L], L = max(out_h, out_w): per image the
 * rows sx, sx+1 (clamped), a0, a1 (per output column), y0, y1, b0, b1 (per output row) of OpenCV's fixed-point tap tables, built
 * by the host (retrieval.cv2_linear_tables) so that host and device agree bit for bit. */
int drag_cv_resize_linear_u8_f32(const void* src, const int64_t* src_off, const int32_t* hw, const int32_t* tab, float* dst,
                                 int32_t n, int32_t out_h, int32_t out_w, void* stream);

/* uint8 images in HBM -> complete PNG files in HBM; replaces the host-side `image.save(path)` of the reference's large RGB
 * results (batch_generate_flux_kshot.py:480 generated_image_rank{r}.png, outpainting_updown_sampling_redux.py:1262 / :1278
 * *_hires_result_* / *_final_result_*).  The contract is the PNG format's: any reader (PIL / libpng) decodes exactly the input
 * array.  Adaptive row filters (libpng's minimum-sum heuristic) + ONE dynamic-Huffman DEFLATE block of literals (no LZ77
 * matches) + Adler-32 + CRC-32, all on the device, stream-ordered, no host round trip.
 *   drag_png_plan:   sizes for a batch of n dense images [H, W, C] (C = 1 grey, 3 RGB; fewer than 2^26 filtered bytes each):
 *                    *workspace_bytes of scratch, *out_stride bytes per image in the output buffer (worst case).
 *   drag_png_encode: images uint8 [n, H, W, C] -> file i at out + i * out_stride, its length in sizes[i] (device int64);
 *                    workspace 256-byte aligned. */
int drag_png_plan(int32_t n, int32_t H, int32_t W, int32_t C, int64_t* workspace_bytes, int64_t* out_stride);
int drag_png_encode(const void* images, int32_t n, int32_t H, int32_t W, int32_t C, void* workspace, int64_t workspace_bytes,
                    void* out, int64_t out_stride, int64_t* sizes, void* stream);

/* Host-side batch file reader feeding drag_jpeg_* (no device work): native threads do the per-file system calls that cost the
 * interpreter ~80 us each.  drag_file_sizes: sizes[i] = bytes of paths[i] or -errno.  drag_read_files: paths[i] -> dst[offsets[i]
 * .. offsets[i+1]) in plain host (ideally pinned) memory; status[i] = 0, errno, or -1 for a file shorter than its slot. */
int drag_file_sizes(const char* const* paths, int64_t n, int64_t* sizes, int32_t threads);
int drag_read_files(const char* const* paths, int64_t n, void* dst, const int64_t* offsets, int32_t* status, int32_t threads);

#ifdef __cplusplus
}
#endif
#endif
