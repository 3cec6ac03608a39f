/* svi_hip.h — C ABI of libsvi_hip.so: the MI355X (gfx950) native backend for the
 * Stable-Video-Infinity rolling-window denoising hot path (Wan DiT block stack + flow-match
 * step + Wan 3-D causal VAE).
 *
 * The reference (vita-epfl/Stable-Video-Infinity) has no FFI: its boundary is a Python call
 * surface.  Each entry point below names the reference function it stands in for
 * (paths relative to the reference's diffsynth/ package); INTEGRATION.md shows the ctypes stub
 * that binds it underneath the unchanged reference call surface.
 *
 * Conventions
 *   - Plain C: opaque handles, raw DEVICE pointers, sizes.  No torch / HIP types in signatures
 *     (svi_stream is a hipStream_t passed as void*; NULL = the null stream).
 *   - Ownership: the caller owns every tensor (weights, inputs, outputs).  The library borrows
 *     pointers for the duration of a call; weight pointers for the lifetime of the binding
 *     (re-bind after a LoRA merge or any .to()/offload that moves storage).  The library owns
 *     only its workspace (allocated at first use for a given problem size, never in steady state).
 *   - All work is enqueued on the caller's stream; no internal synchronisation in steady state.
 *   - Errors: integer status, never abort/throw across the ABI; message via svi_last_error()
 *     (thread-local).  There is NO CPU fallback: with no usable GPU every compute call fails.
 *   - Activations are bf16 (row-major, innermost dim contiguous) unless stated; accumulation fp32.
 *   - Handles are not thread-safe; one handle per device, driven from one host thread at a time.  What the library owns
 *     outside the handles — the attention kernels' per-workgroup flag words, the scratch of the operator-level seams
 *     (svi_attention_fwd, svi_layernorm_modulate, svi_rmsnorm_rope), the per-kernel LDS attributes, the event profiler's
 *     records — is keyed by (device, stream) and guarded by one mutex: two handles on two streams, or two host threads
 *     driving two streams, do not share a buffer (work on ONE stream is ordered, which is what protects a buffer between
 *     consecutive calls).  A stream that is being captured must have seen each call once before the capture (buffers are
 *     created on first use; creating one under capture is refused with a message).
 */
#ifndef SVI_HIP_H
#define SVI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVI_HIP_ABI_VERSION 10

typedef enum {
    SVI_OK = 0,
    SVI_ERR_INVALID = 1,      /* bad argument / shape */
    SVI_ERR_UNBOUND = 2,      /* a required weight was never bound */
    SVI_ERR_HIP = 3,          /* HIP runtime error (message has hipGetErrorString) */
    SVI_ERR_UNSUPPORTED = 4,  /* valid in the reference, not implemented here */
    SVI_ERR_OOM = 5
} svi_status;

typedef enum { SVI_BF16 = 0, SVI_F32 = 1 } svi_dtype;

typedef void* svi_stream;
typedef struct svi_dit svi_dit;
typedef struct svi_vae svi_vae;

/* Constructor arguments of models/wan_video_dit.py:408-421 (WanModel.__init__). head_dim must be 128. */
typedef struct {
    int32_t dim, in_dim, ffn_dim, out_dim, text_dim, freq_dim;
    float eps;
    int32_t patch_t, patch_h, patch_w;
    int32_t num_heads, num_layers;
    int32_t has_image_input;
    int32_t enable_multitalk;   /* talk variant: per-block audio cross-attention + audio_proj (models/wan_video_dit.py:338-351,455-470) */
} svi_dit_config;

/* GEMM epilogues (fused; see svi_gemm_bf16). */
typedef enum {
    SVI_EPI_BIAS = 0,           /* C = bf16(acc + bias)                                   nn.Linear            */
    SVI_EPI_BIAS_GELU_TANH = 1, /* C = bf16(gelu_tanh(bf16(acc + bias)))                  ffn.0+GELU  dit:334  */
    SVI_EPI_BIAS_GATE_RES = 2,  /* C = bf16(res + bf16(gate[n] * bf16(acc + bias)))       dit:369,370,373      */
    SVI_EPI_BIAS_GELU_ERF = 3,  /* exact GELU                                             img_emb     dit:383  */
    SVI_EPI_BIAS_SILU = 4,      /* C = bf16(silu(bf16(acc+bias)))                         time_embedding       */
    SVI_EPI_BIAS_RELU = 5       /* C = bf16(relu(bf16(acc+bias)))                         AudioProjModel dit:97-106 */
} svi_epilogue;

const char* svi_last_error(void);
int32_t svi_abi_version(void);
/* Number of visible HIP devices (0 => every compute entry point will fail with SVI_ERR_HIP). */
int32_t svi_device_count(void);

/* A/B tooling: the library reads its environment switches (SVI_FLASH_KERNEL, SVI_FLASH_TWO_PASS, SVI_FLASH_M16, SVI_FLASH_SPLIT, SVI_GEMM_KERNEL, SVI_GEMM_GM, SVI_GEMM_PF,
 * SVI_CROSS_DEDUP, SVI_CROSS_FUSED, SVI_RMS_ROWS, SVI_QK_FUSED, SVI_MX8_FUSED, SVI_QK8_FUSED, SVI_VAE_EXACT_FP32, SVI_VAE_X2H, SVI_VAE_DMA, SVI_VAE_UP_PHASES, SVI_VAE_TILE_ORDER, SVI_VAE_PAIR, SVI_T5_BUCKETS — all of them select between kernels that
 * compute the same result, bit for bit or within the stated parity bounds — and SVI_WS_LIMIT_MB, a budget in MiB beyond which a DiT workspace is refused with
 * SVI_ERR_OOM as if the allocation had failed (callers that share the device; the stacked CFG pair then falls back to its unstacked form, same bits);
 * csrc/svi_common.h SviSwitches) once, at first use; tools that flip them inside one process
 * call this afterwards.  Switches that change results exist only in variant builds (-DSVI_ABLATIONS), never in the product — with ONE documented
 * exception, off by default: SVI_ATTN_QK8=1 selects the opt-in quantised-QK^T attention (every long-sequence attention call quantises Q and K to
 * MX e4m3, one E8M0 scale per 32 channels, and takes QK^T on v_mfma_scale_f32_32x32x64_f8f6f4; softmax and P·V unchanged).  Like svi_dit_ffn_mx8
 * below it is arithmetic the reference never performs — its attention dispatch (models/wan_video_dit.py:116-147) merely accepts a quantised-QK^T
 * backend as interchangeable — with its own oracle and stated tolerance (tests/test_gpu_attn_qk8.py) and its own bench line (bench.py --fp8-attn).
 * In that mode the DiT's RMSNorm + RoPE launch writes the e4m3 rows and block scales itself (SVI_QK8_FUSED=0: bf16 q | k + two quantiser launches; same bits). */
svi_status svi_switches_reload(void);
/* The value the library PARSED for a switch at its last (re)load — SVI_ATTN_QK8, SVI_CROSS_FUSED, SVI_CROSS_DEDUP, SVI_QK_FUSED, SVI_FLASH_TWO_PASS, SVI_FLASH_M16:
 * 1 / 0 — or -1 for any other name.  What the kernels do follows this, not the process environment of the moment. */
int32_t svi_switch_state(const char* name);

/* ------------------------------------------------------------------ DiT: whole model ------ */
/* WanModel(...) construction (weights are bound afterwards, by reference state-dict key). */
svi_status svi_dit_create(const svi_dit_config* cfg, svi_dit** out);
svi_status svi_dit_destroy(svi_dit* h);
/* name = reference state-dict key, e.g. "blocks.0.self_attn.q.weight" (models/wan_video_dit.py
 * module tree).  dtype must be SVI_BF16 (the pipelines run the DiT in bf16, pipelines/svi_video.py:259).
 * shape is checked against the config. */
svi_status svi_dit_bind_weight(svi_dit* h, const char* name, const void* dev_ptr, svi_dtype dtype,
                               const int64_t* shape, int32_t rank);
/* Returns SVI_OK when every parameter the config requires has been bound; otherwise
 * SVI_ERR_UNBOUND with the first missing key in svi_last_error(). */
svi_status svi_dit_check_bound(svi_dit* h);

/* model_fn_wan_video(dit, x, timestep, context, clip_feature, y, add_condition)
 * (pipelines/svi_video.py:74-137; same math as WanModel.forward, models/wan_video_dit.py:486-567).
 *   x            bf16 [B, 16, T, H, W]           latents
 *   timestep     f32  [B]            (device)    flow-match timestep (pipelines/svi_video.py:397)
 *   context      bf16 [B, Lc, text_dim]          T5 embeddings (un-projected)
 *   clip_feature bf16 [B, 257, 1280] or NULL     I2V only
 *   y            bf16 [B, in_dim-16, T, H, W] or NULL   I2V only (mask ‖ VAE latent)
 *   add_condition bf16 [B, L, dim] or NULL       added to patch tokens (dance pose embedder)
 *   out          bf16 [B, out_dim, T, H, W]      velocity prediction
 * TeaCache and USP hooks of the reference function are not part of this entry point. */
svi_status svi_dit_forward(svi_dit* h, const void* x, const float* timestep, const void* context,
                           const void* clip_feature, const void* y, const void* add_condition,
                           void* out, int32_t B, int32_t T, int32_t H, int32_t W, int32_t Lc,
                           svi_stream stream);

/* Talk variant — model_fn_wan_talk_video (pipelines/svi_video_talk.py:83-160) = WanModel.forward with audio_embed_tuple
 * (models/wan_video_dit.py:486-567): the audio windows are projected to 32 context tokens of width 768 per latent frame
 * (AudioProjModel, dit:44-115) and every block adds, after its text cross-attention,
 *     x += proj(attention_per_frame(q_linear(norm_x(x)), kv_linear(audio tokens of the frame)))     (dit:361-366, models/attention.py:318-371)
 * svi_dit_set_audio arms the handle for the following forwards (svi_dit_forward / _forward_tea; the CFG pair and the sequence-parallel
 * entry points refuse while audio is set) and NULL pointers disarm it:
 *   audio_first  bf16 [1, seq_len = 5, 12, 768]          the first frame's audio window          (audio_embed_tuple[0])
 *   audio_latter bf16 [T - 1, seq_len_vf = 8, 12, 768]   the later latent frames' windows        (audio_embed_tuple[1])
 * The pointers are borrowed until the next svi_dit_set_audio. */
svi_status svi_dit_set_audio(svi_dit* h, const void* audio_first, const void* audio_latter, int32_t n_latter);

/* TeaCache support (pipelines/svi_video.py:23-72 class TeaCache, :117-131 its use inside model_fn_wan_video).
 * svi_dit_time_mod: t_mod bf16 [B, 6, dim], the tensor TeaCache.check() compares between steps (host logic decides).
 * svi_dit_forward_tea: svi_dit_forward with tea_mode 0 = plain; 1 = run the blocks and write
 *   residual bf16 [B, L, dim] = bf16(x_after_blocks - x_before_blocks)          (TeaCache.store, :64-66);
 *   2 = skip the blocks: x = bf16(x_patchified + residual), then the head       (TeaCache.update, :68-70). */
svi_status svi_dit_time_mod(svi_dit* h, const float* timestep, void* t_mod_out, int32_t B, svi_stream stream);
svi_status svi_dit_forward_tea(svi_dit* h, const void* x, const float* timestep, const void* context,
                               const void* clip_feature, const void* y, const void* add_condition, void* out,
                               int32_t B, int32_t T, int32_t H, int32_t W, int32_t Lc, int32_t tea_mode, void* residual,
                               svi_stream stream);

/* The two forwards of one classifier-free-guidance step — model_fn_wan_video(dit, latents, timestep, **prompt_emb_posi, ...) and
 * the same call with prompt_emb_nega (pipelines/svi_video.py:401-408) — in one call.  They share latents and timestep, so the
 * timestep embedding, patchify and block 0's self-attention (everything that precedes the first use of the prompt) are
 * computed once.  out_cond / out_uncond are bit-identical to two svi_dit_forward calls with the respective context. */
svi_status svi_dit_forward_cfg_pair(svi_dit* h, const void* x, const float* timestep, const void* context_cond,
                                    const void* context_uncond, const void* clip_feature, const void* y,
                                    const void* add_condition, void* out_cond, void* out_uncond, int32_t B, int32_t T,
                                    int32_t H, int32_t W, int32_t Lc, svi_stream stream);

/* Hoists what depends only on the prompt out of the step loop (SURVEY §8 a2 / f N2): with the cache enabled
 * svi_dit_forward projects a context (text_embedding / img_emb, pipelines/svi_video.py:94-99) and computes every block's
 * cross-attention K / V^T (models/wan_video_dit.py:272-274) ONCE per distinct (context pointer, clip_feature pointer, Lc)
 * and reuses them; results are bit-identical to the uncached path.  The caller promises that the contents behind a cached
 * pointer do not change while the cache is on; svi_dit_context_cache(h, 0), or binding a weight, drops all entries
 * (4 entries, least recently used first). */
svi_status svi_dit_context_cache(svi_dit* h, int32_t enable);
/* The rolling window's clip boundary (test_svi.py:424-476: every clip re-enters SVIVideoPipeline.__call__ with the next prompt): the caller has
 * written the NEXT prompt's embedding (and CLIP feature) into the tensors a cache entry is keyed by; the entry is recomputed IN PLACE — projected
 * context, identical-suffix summary, every block's cross-attention K / V^T (models/wan_video_dit.py:272-274) — in the buffers it already owns.
 * No device address moves and svi_dit_generation does not, so a hipGraph of the step captured for the previous clip replays on the new prompt.
 * Stream-ordered.  Without an entry for (context, clip_feature, Lc) it is a first fill (allocates; the generation moves). */
svi_status svi_dit_context_refill(svi_dit* h, const void* context, const void* clip_feature, int32_t Lc, svi_stream stream);

/* Sequence-parallel (Ulysses) pieces of one forward — SURVEY §8e axis 3; the reference's USP path: the token chunk / all_gather of
 * pipelines/svi_video.py:119-135 and the all-to-all attention of distributed/xdit_context_parallel.py.  A rank owns token rows
 * [row0, row0 + nrows) of the (f h w) sequence for everything row-local and trades tokens for heads around self-attention; the
 * exchanges (RCCL all-to-all / all-gather) are the caller's, between these calls (svi_hip/sequence_parallel.py):
 *   svi_dit_sp_begin        timestep embedding, context (cache honoured), patchify of the rank's rows
 *   svi_dit_sp_block_qkv    block `layer`: LN + modulate, q | k projections, RMSNorm + RoPE at the rows' true positions (q pre-scaled by
 *                           softmax_scale*log2e) stored IN SEND ORDER: q_send / k_send bf16 [G][P][nrows][Dg] — destination rank j owns
 *                           channel block [j*Dp, (j+1)*Dp), Dp = dim/P, split into G head groups of Dg = Dp/G channels — so each
 *                           (operand, head group) is one contiguous all-to-all input; V^T -> vt_out bf16 [dim, ldvt] (cols >= nrows
 *                           untouched), whose row block j is rank j's piece.  No packing pass on either side of the exchange.
 *   svi_sp_unpack_vt        received V^T pieces [P(src)][Dp][lds] -> [Dp][L8] (token axis source-major)
 *   svi_sp_unpack_out       received attention pieces [G][P(src)][nrows][Dg] -> attn bf16 [nrows, dim]
 *   svi_attention_vt_fwd    attention of a head group on those layouts (after the exchange: all tokens, n = heads/P)
 *   svi_dit_sp_block_rest   attn bf16 [nrows, dim] (after the exchange back) -> output projection + gate + residual,
 *                           cross-attention, MLP of block `layer`
 *   svi_dit_sp_head         head rows bf16 [nrows, svi_dit_head_ld]; all-gathered, then svi_dit_unpatchify -> [out_dim, T, H, W]
 * With one rank (row0 = 0, nrows = L) the sequence is bit-identical to svi_dit_forward. */
svi_status svi_dit_sp_begin(svi_dit* h, const void* x, const float* timestep, const void* context, const void* clip_feature,
                            const void* y, const void* add_condition, int32_t T, int32_t H, int32_t W, int32_t Lc,
                            int32_t row0, int32_t nrows, svi_stream stream);
/* Both forwards of a CFG step on one shard, STACKED (svi_dit_forward_cfg_pair's form on a rank's rows; needs svi_dit_context_cache(h, 1)): the shard's
 * rows of the conditional branch on top of the unconditional branch's.  The calls that follow act on 2 nrows rows: svi_dit_sp_block_qkv stores
 * q_send / k_send as [G][P][nrows][branch][Dg] and V^T as [dim, ldvt >= 2 nrows] (columns [0, nrows) conditional, [nrows, 2 nrows) unconditional);
 * svi_dit_sp_block_rest takes attn [2 nrows, dim]; svi_dit_sp_head writes [2 nrows, svi_dit_head_ld].  After the exchange the two branches are twice as
 * many heads of one svi_attention_vt_fwd call per head group (row stride 2 Dg).
 * No CFG exchange between ranks; each row-local launch of a shard is twice as long.  Bit-identical to two svi_dit_sp_begin forwards. */
svi_status svi_dit_sp_begin_pair(svi_dit* h, const void* x, const float* timestep, const void* context_cond, const void* context_uncond,
                                 const void* clip_feature, const void* y, const void* add_condition, int32_t T, int32_t H, int32_t W, int32_t Lc,
                                 int32_t row0, int32_t nrows, svi_stream stream);
/* svi_dit_sp_block_qkv_part: the same in two pieces — part 1 = LN + modulate and the V^T projection, part 2 = the q | k projection with RMSNorm + RoPE
 * (0 = both: svi_dit_sp_block_qkv) — so that V^T can be on the wire while q | k are still being made (the gather mode of sequence_parallel.py). */
svi_status svi_dit_sp_block_qkv_part(svi_dit* h, int32_t layer, void* q_send, void* k_send, void* vt_out, int32_t ldvt, int32_t P, int32_t G,
                                     int32_t part, svi_stream stream);
svi_status svi_dit_sp_block_qkv(svi_dit* h, int32_t layer, void* q_send, void* k_send, void* vt_out, int32_t ldvt, int32_t P, int32_t G,
                                svi_stream stream);
/* nb = CFG branches stacked on the shard (1, or 2 after svi_dit_sp_begin_pair): a token's branches travel side by side, so what a rank receives per head
 * group is token-major [L, nb * Dg] and the branches' heads are nb x as many heads of ONE attention launch.  svi_sp_unpack_vt: a piece's columns hold branch b's
 * tokens at [b * Ls, (b + 1) * Ls); channel g * Dg + cg goes to row (g * nb + b) * Dg + cg of out [G * nb * Dg][L8] (Dg = channels per head group; nb = 1:
 * row = channel).  svi_sp_unpack_out: pieces [G][P(src)][Ls][nb][Dg] -> attn [nb * Ls][dim], branch b's rows at [b * Ls, (b + 1) * Ls). */
svi_status svi_sp_unpack_vt(const void* recv, void* out, int32_t P, int32_t Dp, int32_t Ls, int32_t lds, int32_t L8, int32_t nb, int32_t Dg, svi_stream stream);
svi_status svi_sp_unpack_out(const void* recv, void* out, int32_t P, int32_t G, int32_t Ls, int32_t Dg, int32_t nb, svi_stream stream);
svi_status svi_dit_sp_block_rest(svi_dit* h, int32_t layer, const void* attn, svi_stream stream);
/* TeaCache on a shard's rows (the reference combines TeaCache and USP, pipelines/svi_video.py:112-131): mode 0 snapshot before the blocks,
 * 1 residual bf16 [nrows, dim] = x_after - x_before, 2 x += residual in place of the blocks. */
svi_status svi_dit_sp_tea(svi_dit* h, int32_t mode, void* residual, svi_stream stream);
svi_status svi_dit_sp_head(svi_dit* h, void* head_rows_out, svi_stream stream);
svi_status svi_dit_unpatchify(svi_dit* h, const void* head_rows, void* out, int32_t T, int32_t H, int32_t W, svi_stream stream);
int32_t svi_dit_head_ld(svi_dit* h);
/* Moves whenever device state that a captured hipGraph of this handle's forwards may have baked in stops being valid (workspace
 * growth, a context-cache entry filled or evicted, svi_dit_context_cache, a weight re-bound, a per-stream library buffer freed).  Read it
 * right after a capture; replay only while it is unchanged (svi_hip.DenoiseLoop does). */
int64_t svi_dit_generation(svi_dit* h);
/* The library keeps small device buffers per (device, stream) — the attention kernels' flag words, split-key partial sums — for as long as
 * the process lives.  A caller that retires a stream (a hipGraph capture stream) releases them here; all_streams != 0: those of every stream
 * of the current device.  Drains the device.  Graphs captured on the stream must not be replayed afterwards (svi_dit_generation moves). */
svi_status svi_stream_buffers_release(svi_stream stream, int32_t all_streams);
svi_status svi_attention_vt_fwd(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* vt, int32_t ldvt, void* out,
                                int32_t ldo, int32_t s_q, int32_t s_kv, int32_t n, int32_t q_prescaled, svi_stream stream);

/* CrossAttention.forward's query path (models/wan_video_dit.py:296,299: q = norm_q(self.q(x)); x = attn(q, k, v)) as the DiT block runs it by default:
 *   svi_linear_row_stats    C = bf16(A W^T + bias) [M, N] AND the statistic of RMSNorm over the full width (dit:192-197): row_sumsq [N/64][ldss] = sums of
 *                           squares of the rounded results per aligned 64-column group (left by the GEMM's epilogue: fixed summation tree, the same in every
 *                           tile kernel), rs_out[m] = rsqrt(sum_g row_sumsq[g][m] / N + eps).
 *   svi_cross_attention_fwd softmax(q' k^T) v over a SHORT key axis (the prompt) with q' = bf16(bf16(bf16(q q_rs[row]) q_gain) q_out_scale) formed as the rows
 *                           are read (q_rs NULL: q is used as it is; either way q carries softmax_scale*log2e already); vt = V transposed [n*128, ldvt];
 *                           key_tail as in svi_dit's cross-attention (device {n, m}: keys n-1.. identical, counted m times) or NULL.  One [s_q, n*128] read and
 *                           one write per call — the separately normalised q never exists in memory. */
svi_status svi_linear_row_stats(const void* A, int32_t lda, const void* W, int32_t ldw, void* C, int32_t ldc, int32_t M, int32_t N, int32_t K,
                                const void* bias, float eps, float* row_sumsq, int32_t ldss, float* rs_out, svi_stream stream);
svi_status svi_cross_attention_fwd(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* vt, int32_t ldvt, void* out, int32_t ldo,
                                   int32_t s_q, int32_t s_kv, int32_t n, const int32_t* key_tail, const float* q_rs, const void* q_gain,
                                   float q_out_scale, svi_stream stream);

/* DiTBlock.forward(x, context, t_mod, freqs) for block `layer` (models/wan_video_dit.py:354-374).
 *   x_inout bf16 [L, dim] (L = f*h*w, updated in place); context bf16 [Lc(+257), dim] ALREADY
 *   projected by text_embedding/img_emb; t_mod bf16 [6, dim]; freqs implied by the (f,h,w) grid. */
svi_status svi_dit_block_forward(svi_dit* h, int32_t layer, void* x_inout, const void* context,
                                 const void* t_mod, int32_t f, int32_t hh, int32_t ww, int32_t Lc,
                                 svi_stream stream);

/* ------------------------------------------------------------------ operator seams -------- */
/* flash_attention(q, k, v, num_heads) (models/wan_video_dit.py:116-147): layout [b, s, (n d)],
 * unmasked softmax(q k^T / sqrt(d)) v, d must be 128.  out may not alias the inputs. */
svi_status svi_attention_fwd(const void* q, const void* k, const void* v, void* out, int32_t b,
                             int32_t s_q, int32_t s_kv, int32_t n, int32_t d, svi_stream stream);
/* Diagnostics for the long-sequence attention (keys >= 2048): one call = an optimistic pass that fixes each row's reference maximum
 * after the first key tile + a second pass in which the complete kernel recomputes exactly the workgroups whose row sums left the
 * range the fixed reference covers.  Reports, for the LAST such call enqueued on `stream`, how many workgroups were recomputed and
 * how many the launch had (drains the stream; tests use it to prove that adversarial operands take the second pass). */
svi_status svi_attention_last_flagged(svi_stream stream, int32_t* flagged_out, int32_t* workgroups_out);
/* Launch planners: pure arithmetic on sizes and the environment switches, no device work (they run on a machine without a GPU).
 * svi_gemm_plan: the kernel svi_gemm_bf16 takes for [M, K] x [N, K]^T — 0 = weight-streaming skinny kernel, 128 = 128^2 tile, 192 = 256 x 192 tile
 * (where 192-wide tiles fill the chip's rounds better: sequence-parallel shards), 259 / 260 = the 256^2 tile (four / two phases per K tile); `compute_units` = CUs of the part
 * the round arithmetic is done for (256 on MI355X; the launcher asks the device).
 * svi_attention_plan: out4 = {kernel (1 = short key axes, 2 = long-sequence kernel), work items run whole, pieces per remaining item, workgroups}:
 * the items of a partly filled last round are cut along the key axis (csrc/svi_attention.hip flash_splits). */
svi_status svi_gemm_plan(int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t skinny, int32_t compute_units, int32_t* kernel_out);
svi_status svi_attention_plan(int32_t s_q, int32_t s_kv, int32_t heads, int32_t compute_units, int32_t* out4);

/* nn.LayerNorm(eps) [+ affine w,b] [+ modulate(x, shift, scale)] over rows of x[rows, dim]
 * (models/wan_video_dit.py:150-151,331-333,358,372).  w,b,shift,scale are bf16 [dim] or NULL. */
svi_status svi_layernorm_modulate(const void* x, void* out, int32_t rows, int32_t dim, float eps,
                                  const void* w, const void* b, const void* shift, const void* scale,
                                  svi_stream stream);

/* RMSNorm over the full model dim, then (optionally) 3-D RoPE per head, in place on x[rows, ld]
 * (models/wan_video_dit.py:186-197 then :178-183).  grid f*h*w must equal rows when rope != 0. */
svi_status svi_rmsnorm_rope(void* x, int32_t ld, int32_t rows, int32_t dim, const void* weight, float eps,
                            int32_t rope, int32_t num_heads, int32_t f, int32_t h, int32_t w,
                            svi_stream stream);

/* C[M,N] = epilogue(A[M,K] · W[N,K]^T)  — nn.Linear with the weight in its native [out,in] layout.
 * bias bf16 [N] (or [M] when bias_along_m, used to emit V^T); gate f32 [N] or NULL; res bf16 [M,ldres]
 * (may alias C).  K multiple of 8; lda/ldw/ldc multiples of 8 elements. */
svi_status svi_gemm_bf16(const void* A, int32_t lda, const void* W, int32_t ldw, void* C, int32_t ldc,
                         int32_t M, int32_t N, int32_t K, const void* bias, int32_t bias_along_m,
                         int32_t epilogue, const float* gate, const void* res, int32_t ldres,
                         svi_stream stream);

/* ---- MX-fp8 (opt-in; north_star "bf16/fp8 MFMA").  The reference computes in bf16 and only STORES weights as float8_e4m3fn
 * (test_svi.py:337, vram_management/layers.py:65-71): what follows is arithmetic it never performs — own tolerance, own bench line.
 *   svi_mx8_quantize   x bf16 [rows, ldx] -> q e4m3 [rows, ldq] (OCP e4m3fn, round-to-nearest-even, saturating) with one E8M0 scale per
 *                      32 consecutive K elements (OCP MX: 2^(floor(log2 amax) - 8)); scales as dwords [K/128][sc_rows], byte b of dword
 *                      [kt][m] = block 4 kt + b of row m; K % 128 == 0, sc_rows >= rows.
 *   svi_gemm_mx8       C[M,N] = epilogue(A8[M,K] · W8[N,K]^T) on v_mfma_scale_f32_32x32x64_f8f6f4: A8 with the scales above (sc_rows a
 *                      multiple of 256 covering M), W8 e4m3 with unit scales (the reference's stored bytes); bias / epilogue / gate / res
 *                      as svi_gemm_bf16; lda / ldw in bytes, multiples of 16.
 *   svi_dit_bind_ffn_fp8 / svi_dit_ffn_mx8   hand blocks.<layer>.ffn.<0|2>.weight over as stored e4m3 bytes [out, in] and route both MLP
 *                      GEMMs of every block through svi_gemm_mx8 (activations quantised in front of each). */
svi_status svi_mx8_quantize(const void* x, int32_t ldx, int32_t rows, int32_t K, void* q, int32_t ldq, void* scales, int32_t sc_rows, svi_stream stream);
svi_status svi_gemm_mx8(const void* A8, int32_t lda, const void* a_scales, int32_t sc_rows, const void* W8, int32_t ldw, void* C, int32_t ldc,
                        int32_t M, int32_t N, int32_t K, const void* bias, int32_t epilogue, const float* gate, const void* res, int32_t ldres,
                        svi_stream stream);
svi_status svi_dit_bind_ffn_fp8(svi_dit* h, int32_t layer, int32_t which, const void* e4m3_weight);
svi_status svi_dit_ffn_mx8(svi_dit* h, int32_t enable);
/* ABI v10 (round 6) — the same opt-in arithmetic on the block's other six projections ("QKV/out-proj ... on bf16/fp8 MFMA" of the north star):
 *   svi_gemm_mx8_wscaled   C[M,N] = bf16(A8[M,K] · dequant(W8[N,K])^T + bias): the block scales ([K/128][sc_rows], sc_rows a multiple of 256 covering N) belong to
 *                          the W operand's rows, A carries unit scales — the transposed value projection V^T = Wv · X^T (A8 = the stored e4m3 weight, W8 = the
 *                          quantised activation rows, bias along M).
 *   svi_dit_bind_ffn_fp8   also takes which = 10 self_attn.q, 11 self_attn.k, 12 self_attn.v, 13 self_attn.o, 14 cross_attn.q, 15 cross_attn.o
 *                          (blocks.<layer>.<module>.weight as stored e4m3 bytes [out, in]).
 *   svi_dit_proj_mx8       route those six GEMMs of every block through the MX fp8 kernels (plain forwards and the stacked CFG pair; sequence-parallel shards keep
 *                          bf16): the LayerNorm output is quantised once for q, k and V^T, the attention outputs once for each output projection, the cross-attention
 *                          query's row statistic comes from the fp8 GEMM's epilogue as it does from the bf16 one.  The prompt-side K / V (context cache) stay bf16.
 *                          Needs all six weights of every block bound; own oracle and tolerance (tests/test_gpu_mx8.py), bench.py --fp8-all; never a default. */
svi_status svi_gemm_mx8_wscaled(const void* A8, int32_t lda, const void* W8, int32_t ldw, const void* w_scales, int32_t sc_rows, void* C, int32_t ldc,
                                int32_t M, int32_t N, int32_t K, const void* bias, int32_t bias_along_m, svi_stream stream);
svi_status svi_dit_proj_mx8(svi_dit* h, int32_t enable);

/* Classifier-free-guidance combine + FlowMatchScheduler.step, fused (pipelines/svi_video.py:410,420;
 * schedulers/flow_match.py:53-64):  lat += (uncond + s*(cond-uncond)) * (sigma_next - sigma), bf16,
 * rounded after every op exactly like the reference's bf16 tensor arithmetic.  uncond may be NULL
 * (cfg_scale == 1 path). */
svi_status svi_cfg_step(void* latents, const void* cond, const void* uncond, int64_t n, float cfg_scale,
                        float dsigma, svi_stream stream);

/* Three-way guidance of the talk sampler + FlowMatchScheduler.step (pipelines/svi_video_talk.py:455-461):
 *   v = uncond + s_text*(cond - drop_text) + s_audio*(drop_text - uncond);  lat += v * (sigma_next - sigma)
 * bf16, rounded after every op in the reference's evaluation order. */
svi_status svi_cfg3_step(void* latents, const void* cond, const void* uncond, const void* drop_text, int64_t n, float s_text,
                         float s_audio, float dsigma, svi_stream stream);

/* FP8 weight storage: the reference's `torch_dtype=torch.float8_e4m3fn` mode (test_svi.py:337) keeps parameters as OCP e4m3fn and
 * casts them to bf16 in front of every use (vram_management/layers.py:65-71, cast_to).  The cast is exact, so it is done once:
 * out bf16 [n] = bf16(in e4m3fn [n]); bind the result with svi_dit_bind_weight.  (svi_hip.WanDiT.bind does this for fp8 tensors.) */
svi_status svi_fp8_e4m3_to_bf16(const void* in, void* out, int64_t n, svi_stream stream);

/* ------------------------------------------------------------------ measurement ----------- */
/* Per-kernel timing with HIP events recorded on the launch stream (so it measures the kernels where
 * they run, inside the caller's timed region).  Off by default; when on, every tagged launch inside
 * svi_dit_forward / svi_vae_* is bracketed by an event pair.  svi_prof_summary synchronises the
 * recorded events and writes one JSON object {"tag": {"count": n, "ms": total_ms}, ...} into buf. */
svi_status svi_prof_enable(int32_t on);
svi_status svi_prof_summary(char* buf, int64_t buflen);
/* Restrict recording to some tags: comma-separated names as svi_prof_summary prints them ("flash_self,gemm_ffn1"); NULL or "" = all (the
 * default).  Every event record is a packet between two kernels of the stream it measures (~1440 per C2 step with all tags, ~1 % of the
 * step); bench.py records the dominant kernel alone over its timed region and takes the full breakdown in a short pass behind it. */
svi_status svi_prof_select(const char* tags);

/* ------------------------------------------------------------------ VAE ------------------- */
/* WanVideoVAE() (models/wan_video_vae.py:599-618): fixed architecture (dim 96, z 16, mult 1,2,4,4). */
svi_status svi_vae_create(svi_vae** out);
svi_status svi_vae_destroy(svi_vae* h);
/* name = reference state-dict key ("model.decoder.conv1.weight", ...); dtype SVI_F32
 * (the pipelines run the VAE in fp32: pipelines/svi_video.py:386-387,303-309).  Convolution weights are re-packed for the
 * kernels at bind time by a launch on the NULL stream: the tensor must be complete with respect to that stream when it is
 * bound; the first encode / decode after a bind waits for the packing before it enqueues on the caller's stream. */
svi_status svi_vae_bind_weight(svi_vae* h, const char* name, const void* dev_ptr, svi_dtype dtype,
                               const int64_t* shape, int32_t rank);
svi_status svi_vae_check_bound(svi_vae* h);
/* The 8-bit frame hand-off between the clips of a stream, on the device.
 * svi_video_to_u8: SVIVideoPipeline.tensor2video (pipelines/svi_video.py:366-370): video f32 [3, T, H, W] in [-1, 1] ->
 *   frames u8 [T, H, W, 3] = uint8(clip((x + 1) * 127.5, 0, 255)).
 * svi_u8_to_video: BasePipeline.preprocess_image (pipelines/base.py:44-45): frames u8 [n, H, W, 3] -> f32 [n, 3, H, W] =
 *   float32(x) * (2 / 255) - 1.  Both bit-identical to the reference's host arithmetic. */
svi_status svi_video_to_u8(const float* video, uint8_t* frames, int32_t T, int32_t H, int32_t W, svi_stream stream);
svi_status svi_u8_to_video(const uint8_t* frames, float* video, int32_t n, int32_t H, int32_t W, svi_stream stream);

/* WanVideoVAE.decode -> single_decode -> VideoVAE_.decode (models/wan_video_vae.py:777-789,753-756,552-575)
 *   latents f32 [16, T, h, w]  ->  video f32 [3, 1+4(T-1), 8h, 8w], clamped to [-1,1]. */
svi_status svi_vae_decode(svi_vae* h, const float* latents, float* video, int32_t T, int32_t hh, int32_t ww,
                          svi_stream stream);
/* WanVideoVAE.encode -> single_encode -> VideoVAE_.encode (models/wan_video_vae.py:759-774,747-750,525-550)
 *   video f32 [3, 1+4k, H, W]  ->  latents f32 [16, 1+k, H/8, W/8]  (normalised mean). */
svi_status svi_vae_encode(svi_vae* h, const float* video, float* latents, int32_t T, int32_t H, int32_t W,
                          svi_stream stream);

/* WanVideoVAE.tiled_decode / tiled_encode (models/wan_video_vae.py:643-693 / :696-744, masks :621-640): the spatial tiling the
 * pipelines switch on with `tiled=True` (the default of SVIVideoPipeline.__call__, pipelines/svi_video.py:439).  Tiles start every
 * tile_stride, are clipped at the far edge, and a start is dropped once the previous tile reaches the edge; each tile is read in
 * place from the caller's tensor, run through the same graph as svi_vae_decode / _encode, multiplied by the linear-ramp mask
 * (ramp width = (tile_size - tile_stride), in output elements) and accumulated in task order; the result is values / weight,
 * clamped to [-1,1] for decode only AFTER the blend (tile values are not clamped, as in the reference).  `video` / `latents` serve
 * as the accumulator.  Decode: sizes and strides in LATENT pixels; encode: in VIDEO pixels, multiples of 8 (the reference's
 * encode() multiplies its latent-unit arguments by 8 before the call, :765-767). */
svi_status svi_vae_tiled_decode(svi_vae* h, const float* latents, float* video, int32_t T, int32_t hh, int32_t ww,
                                int32_t size_h, int32_t size_w, int32_t stride_h, int32_t stride_w, svi_stream stream);
svi_status svi_vae_tiled_encode(svi_vae* h, const float* video, float* latents, int32_t T, int32_t H, int32_t W,
                                int32_t size_h, int32_t size_w, int32_t stride_h, int32_t stride_w, svi_stream stream);

/* ------------------------------------------------------------------ dance variant: pose embedder ------- */
/* The `dwpose_embedding` of SVIDanceVideoPipeline (pipelines/svi_video_dance.py:255-269): seven fp32 Conv3d layers with SiLU between
 * them, 3 -> hidden (16) channels at full resolution down to dim channels at the DiT's token grid.  svi_pose_forward also does what
 * :527-530 does around it: the pose video's first frame is repeated three more times in front, values are divided by 255, the result
 * is cast to bf16 and laid out 'b c f h w -> b (f h w) c' — exactly the `add_condition` input of svi_dit_forward (:423, :103-104).
 *   pose  f32 [3, F, H, W] (values 0..255)   ->   out bf16 [f*h*w, dim],  (f, h, w) from svi_pose_tokens (F = 81, 480x832: 21, 30, 52)
 * Weight names are the nn.Sequential's state-dict keys "0.weight", "0.bias", "2.weight", ... "12.bias" (fp32). */
typedef struct svi_pose svi_pose;
svi_status svi_pose_create(int32_t hidden, int32_t dim, svi_pose** out);
svi_status svi_pose_destroy(svi_pose* h);
svi_status svi_pose_bind_weight(svi_pose* h, const char* name, const void* dev_ptr, svi_dtype dtype, const int64_t* shape, int32_t rank);
svi_status svi_pose_check_bound(svi_pose* h);
svi_status svi_pose_tokens(svi_pose* h, int32_t F, int32_t H, int32_t W, int32_t* f, int32_t* hh, int32_t* ww);
svi_status svi_pose_forward(svi_pose* h, const float* pose, void* out, int32_t F, int32_t H, int32_t W, svi_stream stream);

/* ------------------------------------------------------------------ prompt-side encoders ------- */
/* WanTextEncoder (models/wan_video_text_encoder.py:209-256): the umT5 encoder behind WanPrompter.encode_prompt
 * (prompters/wan_prompter.py:99-112), bf16 as the pipeline keeps it.  Per block: T5LayerNorm, attention WITHOUT the 1/sqrt(d) scale and
 * with a relative-position bias (per layer unless shared_pos), gated-GELU feed-forward; dropout is the identity (eval).
 * Weight names are the module's state-dict keys ("token_embedding.weight", "blocks.<i>.attn.q.weight", "blocks.<i>.ffn.gate.0.weight",
 * "blocks.<i>.pos_embedding.embedding.weight", "norm.weight", ...), bf16, borrowed.
 *   svi_t5_forward: ids int64 [L] (device); the reference's mask is the tokenizer's prefix mask, given as n_valid = mask.sum():
 *   keys = positions < n_valid.  Query rows < `rows` (n_valid <= rows <= L) are computed, the rest of out bf16 [L, dim] is zero:
 *   rows = L reproduces text_encoder(ids, mask); rows = n_valid reproduces encode_prompt (it zeroes rows >= n_valid, :110-111).
 *   svi_t5_relative_buckets: host-only; T5RelativeEmbedding._relative_position_bucket (:175-194, bidirectional) for
 *   rel = key - query in -(len-1) .. len-1, out[rel + len - 1], in the host's fp32 arithmetic (what the module computes on a CPU).
 *   svi_t5_device_buckets: the table svi_t5_forward actually uses — the same function evaluated on the device in the arithmetic
 *   the module's tensor ops perform THERE (scalar division = multiplication by the fp32 reciprocal, device logf): the two can differ
 *   where the log ratio is an exact integer (|rel| = 16, 32, 64).  Copied to the host for inspection. */
typedef struct svi_t5_config {
    int32_t vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, max_dist, shared_pos;
} svi_t5_config;
typedef struct svi_t5 svi_t5;
svi_status svi_t5_create(const svi_t5_config* cfg, svi_t5** out);
svi_status svi_t5_destroy(svi_t5* h);
svi_status svi_t5_bind_weight(svi_t5* h, const char* name, const void* dev_ptr, svi_dtype dtype, const int64_t* shape, int32_t rank);
svi_status svi_t5_check_bound(svi_t5* h);
svi_status svi_t5_forward(svi_t5* h, const int64_t* ids, int32_t L, int32_t n_valid, int32_t rows, void* out, svi_stream stream);
svi_status svi_t5_relative_buckets(int32_t num_buckets, int32_t max_dist, int32_t len, int32_t* out);
svi_status svi_t5_device_buckets(svi_t5* h, int32_t len, int32_t* out);

/* WanImageEncoder.encode_image (models/wan_video_image_encoder.py:864-880): bicubic resize to image_size^2 (align_corners=False),
 * v*0.5+0.5, CLIP mean/std normalisation, then VisionTransformer.forward(use_31_block=True) (:456-478): bias-free patch embedding,
 * class token, positional embedding, pre_norm, and the first layers_used (= num_layers - 1) pre-norm blocks (fused qkv, softmax
 * attention with 1/sqrt(d), exact GELU MLP).  fp32 throughout, as SVI runs this module (pipelines/svi_video.py:307-309) — on the exact
 * fp32 MFMA kernel.  Weight names are VisionTransformer's state-dict keys (WanImageEncoder's "model.visual." prefix stripped), fp32,
 * borrowed; post_norm.*, head and blocks >= layers_used are accepted and ignored.
 *   images f32 [B, 3, H, W] in [-1, 1]  ->  out f32 [B, tokens, dim]   (tokens = (image_size/patch_size)^2 + 1: 257 x 1280 for ViT-H/14) */
typedef struct svi_clip_config {
    int32_t image_size, patch_size, dim, mlp_ratio, num_heads, num_layers, layers_used;
    float norm_eps;
} svi_clip_config;
typedef struct svi_clip svi_clip;
svi_status svi_clip_create(const svi_clip_config* cfg, svi_clip** out);
svi_status svi_clip_destroy(svi_clip* h);
svi_status svi_clip_bind_weight(svi_clip* h, const char* name, const void* dev_ptr, svi_dtype dtype, const int64_t* shape, int32_t rank);
svi_status svi_clip_check_bound(svi_clip* h);
svi_status svi_clip_tokens(svi_clip* h, int32_t* tokens, int32_t* dim);
svi_status svi_clip_encode_image(svi_clip* h, const float* images, int32_t B, int32_t H, int32_t W, float* out, svi_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* SVI_HIP_H */
