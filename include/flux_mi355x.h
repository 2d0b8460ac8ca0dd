/*
 * flux_mi355x.h — C-ABI of the MI355X-native FLUX.1 denoise path (libflux_mi355x.so).
 *
 * This is the drop-in boundary for the hot path of EricLBuehler/diffusion-rs
 * (SURVEY.md §8b).  Everything here is `extern "C"`, plain pointers and sizes, no torch /
 * candle types.  A Rust `extern "C"` block (or cgo / ctypes) binds it 1:1; INTEGRATION.md
 * shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - Every function returning `int` returns FMI_OK (0) or a negative fmi_status; the
 *     message is available from fmi_last_error() (thread-local).  The reference's only
 *     native library (diffusion_rs_backend/kernels/bitsandbytes/dequant.cu:172-232) has
 *     void returns and no error channel; the `dequantize_*` entry points below keep that
 *     exact signature so the reference's ffi.rs binds them unchanged.
 *   - All tensors are row-major, contiguous unless a leading dimension is given.
 *   - Data pointers are DEVICE pointers borrowed for the call unless the parameter name
 *     ends in `_host` or the doc says "host or device" (those go through
 *     hipMemcpyDefault).  The caller owns every output buffer (same ownership rule as
 *     the reference's CustomOp::cuda_fwd call sites, bitsandbytes/op.rs:204-228).
 *   - `stream` is a hipStream_t passed as void*; NULL = the null stream.  All compute is
 *     asynchronous on that stream unless stated.
 *   - Not thread-safe per handle (the reference serialises on a Mutex too,
 *     diffusion_rs_core/src/pipelines/mod.rs:110-113); distinct handles are independent.
 */
#ifndef FLUX_MI355X_H
#define FLUX_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared here are exported. */
#pragma GCC visibility push(default)

/* 4 (round 4): + fmi_build_id, fmi_comm_probe, fmi_flux_set_attention_kernel; fmi_flux_set_modulation_gemm accepts 2; fmi_set_attention_kernel accepts 5.
 * 5 (round 5): + the caller-owned-workspace forms fmi_sdpa_bf16_ws / fmi_sdpa_fp8qk_ws / fmi_linear_fp8_ws / fmi_linear_i8_ws and their size queries;
 *   fmi_sdpa_* / fmi_linear_fp8 / fmi_linear_i8 no longer synchronise the stream or call hipMalloc; + fmi_quantize_rows_i8_asym, fmi_rowsum_i8,
 *   fmi_gemm_i8_asym (the int8 mode's post-GELU operand form); fmi_flux_set_quant_dense_cache accepts -1 (default: by memory) and 3.
 *   + fmi_sdpa_fp8 / fmi_sdpa_fp8_ws (e4m3 P and V as well; op-level only: fmi_flux_set_fp8_attention still accepts 0..2).
 * 6 (round 6): + fmi_release_scratch; + the small f32 seams fmi_timestep_embedding, fmi_rope_table, fmi_rmsnorm_rope; + fmi_flux_calibrate_int8
 *   (per-channel smoothing of the int8 mode from a calibration; fmi_flux_quantize_int8 without one is unchanged).  The op-level entries that use
 *   the library's per-stream scratch (fmi_sdpa_*, fmi_linear_fp8 / _i8, fmi_groupnorm_nhwc) now enqueue their kernels under one lock: host threads
 *   may share a stream.
 * Additions only: a host bound against version 3 keeps working. */
#define FMI_ABI_VERSION 6

typedef enum fmi_status {
  FMI_OK = 0,
  FMI_ERR_INVALID = -1,     /* bad argument / shape / unknown tensor name */
  FMI_ERR_HIP = -2,         /* a HIP runtime call failed */
  FMI_ERR_STATE = -3,       /* missing weights, wrong call order */
  FMI_ERR_UNSUPPORTED = -4, /* valid request this build does not implement */
  FMI_ERR_NOMEM = -5
} fmi_status;

/* Element types of tensors crossing the boundary (host side of set_tensor, or device
 * activations).  NF4/FP4/INT8 are the bitsandbytes storage types of
 * diffusion_rs_backend/src/bitsandbytes/mod.rs:26-31. */
typedef enum fmi_dtype {
  FMI_F32 = 0,
  FMI_F16 = 1,
  FMI_BF16 = 2,
  FMI_U8 = 3,
  FMI_I8 = 4
} fmi_dtype;

/* Mirrors diffusion_rs_core::ModelDType (util/auto_dtype.rs) — the compute dtype of the
 * DiT/VAE weights and activations.  This build computes in BF16 on MFMA with f32
 * accumulation; Auto resolves to BF16.  F16/F32 are rejected with FMI_ERR_UNSUPPORTED. */
typedef enum fmi_model_dtype { FMI_MODEL_AUTO = 0, FMI_MODEL_BF16 = 1, FMI_MODEL_F16 = 2, FMI_MODEL_F32 = 3 } fmi_model_dtype;

const char* fmi_last_error(void);
int fmi_abi_version(void);
/* Select the HIP device for this thread (hipSetDevice) and warm the runtime. */
int fmi_init(int device_ordinal);
/* Device / build info as a JSON string (static storage, valid until next call); includes "build_id" and the count of fp8-QK^T
 * attention launches that fell back from the one-wave stream to the 8-wave kernel ("fp8_attention_fallbacks"). */
const char* fmi_device_info(void);
/* Source identity of this binary: the first 16 hex digits of sha256 over every file of diffusion-rs_amd/csrc/, include/flux_mi355x.h and the Makefile
 * (sorted by path, concatenated) at build time.  __graft_entry__.build() recomputes it from the tree and rebuilds on a mismatch,
 * so a stale prebuilt .so cannot stand in for the sources next to it. */
const char* fmi_build_id(void);
/* 1 = this is the TEST build (libflux_mi355x_alt.so, `make alt`: compiled with -DFMI_ALT_KERNELS=1), which carries the superseded kernels as well —
 * attention kernels 0, 2, 3, 4 of fmi_set_attention_kernel, the dense 4-wave GEMM (FMI_GEMM_W4=1) — for the bit-identity cross-checks of tests/.
 * The product library returns 0, carries kernels 5 and 1, and answers a request for another with FMI_ERR_UNSUPPORTED. */
int fmi_has_alt_kernels(void);

/* ------------------------------------------------------------------------------------
 * FLUX DiT — replaces diffusion_rs_core::models::flux::Flux (model.rs:709-838)
 * ---------------------------------------------------------------------------------- */

/* Mirrors `Config` (model.rs:21-31) + the model constants (model.rs:16-19).  hidden_size
 * must be num_attention_heads*128 (pe_dim = sum(axes_dim) = 128); mlp_ratio is fixed 4. */
typedef struct fmi_flux_config {
  int in_channels;           /* 64 */
  int pooled_projection_dim; /* 768 */
  int joint_attention_dim;   /* 4096 */
  int num_attention_heads;   /* 24 */
  int num_layers;            /* 19 double-stream blocks */
  int num_single_layers;     /* 38 single-stream blocks */
  int guidance_embeds;       /* 1 = dev, 0 = schnell */
  int axes_dim[3];           /* {16,56,56} */
  int theta;                 /* 10000 */
} fmi_flux_config;

typedef struct fmi_flux fmi_flux; /* opaque */

/* Fill `cfg` with the public FLUX.1 config (dev if guidance_embeds else schnell). */
void fmi_flux_default_config(fmi_flux_config* cfg, int guidance_embeds);

/* Allocate the model: one bf16 weight arena in HBM sized from cfg (23.8 GB for FLUX.1),
 * zero-initialised.  == Flux::new (model.rs:722-787) minus the tensor reads. */
int fmi_flux_create(const fmi_flux_config* cfg, fmi_model_dtype dtype, fmi_flux** out);
void fmi_flux_destroy(fmi_flux*);

/* Provide one tensor by its diffusers name, exactly the names VarBuilder looks up in
 * Flux::new (model.rs:165-772), e.g.
 *   "x_embedder.weight", "context_embedder.bias",
 *   "time_text_embed.timestep_embedder.linear_1.weight",
 *   "transformer_blocks.3.attn.to_q.weight", "transformer_blocks.3.norm1.linear.bias",
 *   "transformer_blocks.3.attn.norm_added_k.weight", "transformer_blocks.3.ff.net.0.proj.weight",
 *   "single_transformer_blocks.7.proj_mlp.weight", "single_transformer_blocks.7.attn.norm_q.weight",
 *   "norm_out.linear.weight", "proj_out.bias".
 * `data` may be a host or device pointer; `dtype` F32/F16/BF16 is converted to bf16 (RNE)
 * and placed into the fused layouts the kernels read (q|k|v|proj_mlp rows concatenated,
 * all modulation linears in one matrix).  Shape is checked against cfg. Synchronous. */
int fmi_flux_set_tensor(fmi_flux*, const char* name, const void* data, fmi_dtype dtype,
                        const int64_t* shape, int rank);

/* bitsandbytes 4-bit weight for linear `prefix` (e.g. "transformer_blocks.0.attn.to_q"):
 * packed u8 (out*in/2), f32 absmax (out*in/blocksize), quant type 1=fp4 2=nf4
 * (bitsandbytes/mod.rs:137-222; nested absmax must be resolved by the caller with
 * fmi_dequantize_blockwise first, as BnbLinear::dequantize_4bit does, mod.rs:230-239).
 * The layer then runs through the fused dequant-GEMM. host or device pointers. */
int fmi_flux_set_linear_bnb4(fmi_flux*, const char* prefix, const uint8_t* packed,
                             const float* absmax, int blocksize, int quant_type,
                             int out_features, int in_features);

/* Number of tensors still missing (never set); names via fmi_flux_missing_name(i). */
/* LLM.int8 linear (BnbLinear::Int8, bitsandbytes/mod.rs:104-134,293-300): weight int8 (out,in) row
 * major + SCB f32 (out); effective weight = w * SCB[row] / 127 (dequant.cu:205-214), expanded to
 * bf16 inside the layer's GEMM (small launches) or right before it (see fmi_flux_set_quant_dense_cache).
 * Same prefix rules as fmi_flux_set_linear_bnb4. */
int fmi_flux_set_linear_int8(fmi_flux*, const char* prefix, const int8_t* weight, const float* scb, int out_features, int in_features);
/* Quantised block / modulation linears (BnbLinear::forward = dequantize, then matmul: bitsandbytes/mod.rs:293-312) — where the expanded
 * weights live.  Launches of up to 383 rows (nf4 / fp4) or 256 rows (LLM.int8) always multiply straight from the packed codes with the
 * fused dequant-GEMM (the expansion is an LDS stage of the GEMM; the packed read wins there).  For larger launches the dense kernel on
 * expanded weights is measured faster, and `mode` says where those come from:
 *   -1 (default) by memory: 3 if, when the first such launch comes, the device has at least twice the dense block arena free (it does
 *      on a 288 GB part: an nf4 FLUX.1-dev then holds 7 GB of codes + 16 GB of expanded block matrices and runs at bf16 speed), else 0;
 *    0 packed only: the matrix is expanded per call into a reusable scratch (2 x the largest fused matrix); ~7 GB resident for FLUX.1-dev;
 *    1 every quantised matrix, the 6.5 GB modulation matrix included, is expanded ONCE into its slot of the bf16 arenas (+23 GB);
 *    2 every launch on the fused kernels;
 *    3 as 0, but a matrix that 0 would expand per call is expanded once into its dense slot (small launches stay on the codes).
 * All of them produce the same bits. */
int fmi_flux_set_quant_dense_cache(fmi_flux*, int mode);
/* Single-image sequence parallelism.  The reference runs one image on one device
 * (pipelines/mod.rs:214-217 "This will need to be updated!"); here the tokens of ONE image are sharded over the
 * world_size ranks of a group: rank r passes ONLY its token shard to fmi_flux_forward / fmi_flux_denoise
 * (txt rows [r*T/N, (r+1)*T/N), img rows [r*S/N, (r+1)*S/N), the matching txt_ids / img_ids rows; B = 1) and gets
 * its shard of the prediction / latents back.  Everything per token runs on the local rows; the joint attention
 * (model.rs:540-552) is made head-local by two all-to-alls per block, which the library asks the caller to run:
 *   a2a(user, send, recv, bytes_per_peer, stream): block p (bytes_per_peer bytes) of `send` goes to rank p, block p
 *   of `recv` comes from rank p; device buffers owned by the library; the exchange must be ordered on `stream`
 *   (hipStream_t) like a kernel launch (RCCL: all_to_all on that stream).  Returns 0 on success.
 * Needs heads % world_size == 0 and equal shards (T, S divisible by world_size); bf16 mode.  world_size 1 (or a
 * null callback) switches it off.  Results are bit-identical to the single-device forward. */
/* Latency mode for small launches (default off).  A residual projection (attention output projection, MLP down projection,
 * single-block linear2) launched on so few rows that it has fewer than 128 tiles of 256 x 256 — the shards of sequence
 * parallelism — is split along K into up to 8 parts of one grouped launch and reduced in a fixed order by a second kernel;
 * and the sequence-parallel attention of few heads walks up to 4 key ranges in parallel workgroups whose partial outputs
 * are merged by their log-sum-exp.  Deterministic, but sums are associated differently from the unsplit launches: results
 * agree to rounding, not bit for bit (which is why it is opt-in).  Per-rank effect: profiles/r02_sp_rank_time.txt. */
int fmi_flux_set_split_k(fmi_flux*, int enable);
typedef int (*fmi_all_to_all_fn)(void* user, const void* send, void* recv, size_t bytes_per_peer, void* stream);
int fmi_flux_set_sequence_parallel(fmi_flux*, int rank, int world_size, fmi_all_to_all_fn a2a, void* user);

/* ------------------------------------------------------------------------------------
 * RCCL communicator (rccl_comm.hip): the collectives of the multi-GPU path behind plain C — the counterpart a host
 * without torch.distributed needs (the reference itself is single-device, pipelines/mod.rs:214-217).  librccl.so.1 is
 * opened with dlopen on first use (FMI_ERR_UNSUPPORTED if absent).  One fmi_comm per process / GPU; rank 0 calls
 * fmi_comm_unique_id and ships the FMI_COMM_ID_BYTES bytes to the other ranks by any host channel; fmi_comm_create is
 * collective (ncclCommInitRank) and binds the comm to the CURRENT device.  All operations are enqueued on `stream`
 * (hipStream_t) like a kernel launch; none synchronises the host.
 *   fmi_comm_all_to_all  has the fmi_all_to_all_fn signature: pass (fmi_comm_all_to_all, comm) to
 *                        fmi_flux_set_sequence_parallel and the two exchanges per block are one ncclAllToAll each.
 *   fmi_comm_broadcast   in place, e.g. on every fmi_flux_state_buffer (weights from rank 0).
 *   fmi_comm_gather      rank r's `bytes` land at recv + r*bytes on root (decoded u8 images to rank 0). */
#define FMI_COMM_ID_BYTES 128
typedef struct fmi_comm fmi_comm;
/* Non-collective: FMI_OK iff librccl is usable and the thread has a current device.  Call it on every rank and agree on the
 * results over the host channel BEFORE fmi_comm_create, which blocks in ncclCommInitRank until every rank has arrived. */
int fmi_comm_probe(void);
int fmi_comm_unique_id(void* id_out /* FMI_COMM_ID_BYTES */);
int fmi_comm_create(const void* id, int rank, int world_size, fmi_comm** out);
void fmi_comm_destroy(fmi_comm*);
int fmi_comm_rank(const fmi_comm*);
int fmi_comm_world_size(const fmi_comm*);
int fmi_comm_stats(const fmi_comm*, unsigned long long* calls, unsigned long long* bytes_sent);
int fmi_comm_all_to_all(void* comm, const void* send, void* recv, size_t bytes_per_peer, void* stream);
int fmi_comm_broadcast(fmi_comm*, void* buf, size_t bytes, int root, void* stream);
int fmi_comm_gather(fmi_comm*, const void* send, void* recv, size_t bytes, int root, void* stream);
/* PROCESS-WIDE test / ablation hooks — fmi_set_bnb4_onewave_min_rows, fmi_set_attention_kernel (below; a model handle can override it for itself: fmi_flux_set_attention_kernel) and the environment
 * variables FMI_GEMM_W4 / FMI_ATT_W4 / FMI_ATT_W16 / FMI_ATT_W32 / FMI_ATT_W16L (read once at load: the initial state of fmi_set_attention_kernel's switches), FMI_TWO_STREAMS=1 (read
 * by fmi_flux_create: the text chain of a double block on an internal second stream — an experiment measured at -0.3 %, same bits, default off) FMI_GEMM_W4_QKV_MIN_N=<n> (read at load, with FMI_GEMM_W4=1: dense q|k|v launches at least n wide on the 4-wave GEMM kernel too — measured 5 % slower in round 4, default never) and FMI_GEMM_BAND=<n> (read at the first GEMM launch: pins the band height of the
 * GEMM tile order, a bijection of the tiles whatever its value) — are shared by every handle and thread of the process: they pick between
 * kernels that produce identical bits (the attention kernels: identical within a family, equal to rounding across, see
 * fmi_set_attention_kernel), so they move time, not results.  Set them before starting work on other threads.
 * Everything that changes results (fp8 / int8 mode and their attention operands, split-K latency mode, the quantised-weight policy, sequence parallelism) is per handle.
 * Rows from which 4-bit GEMMs use the one-wave-per-SIMD fused kernel (default 256): */
int fmi_set_bnb4_onewave_min_rows(int rows);
/* Test hook of the attention kernel's deferred-rescale branch, a BOOLEAN: 0 = rescale the accumulator on every key tile,
 * any other value = the default (rescale only when the running maximum grew by more than 6.0 in log2 units; the kernels are
 * instantiated for these two settings only). */
int fmi_flux_set_attention_rescale_threshold(fmi_flux*, int thr_x16);
/* PER HANDLE: the attention kernel this model uses, in fmi_set_attention_kernel's numbering (0 .. 5; see there), or -1 (default) = follow
 * the process-wide switch.  Two handles of one process can run different kernels; the op-level fmi_sdpa_* entry points have no handle and
 * keep following the process-wide switch. */
int fmi_flux_set_attention_kernel(fmi_flux*, int kind);
/* The weights as flat device buffers — the unit of the multi-GPU broadcast (north star: "RCCL broadcast of
 * weights").  Rank 0 loads a checkpoint, fmi_flux_state_export() fills a small host blob saying which
 * arenas exist and how every fused matrix is stored (pass blob_host = NULL to query *len); the other ranks
 * fmi_flux_state_adopt() it (same arenas allocated, every tensor marked present) and receive the bytes of
 * buffers 0 .. fmi_flux_state_buffer_count()-1 (fmi_flux_state_buffer: device pointer + size, NULL / 0 for
 * an arena this checkpoint does not use).  LLM.int8 matrices are not covered (FMI_ERR_UNSUPPORTED).
 * The reference is single-device (pipelines/mod.rs:214-217). */
int fmi_flux_state_buffer_count(void);
int fmi_flux_state_export(fmi_flux*, uint8_t* blob_host, size_t cap, size_t* len);
int fmi_flux_state_adopt(fmi_flux*, const uint8_t* blob_host, size_t len);
int fmi_flux_state_buffer(fmi_flux*, int index, void** ptr, size_t* bytes);
/* fmi_flux_denoise computes every step's modulation vectors (they depend on t, guidance and y only) before
 * the loop: 1 (default) = one MFMA GEMM (n_steps*B, D) x (n_mod, D)^T with silu(vec) rounded to bf16, the
 * 6.5 GB modulation matrix read once per image, taken when n_steps * B > 4; 0 = f32 GEMV passes of 4 rows (one pass per
 * 4 steps); 2 = the GEMM at any row count (test hook: lets a 2-step run at full size go through the one-GEMM path). */
int fmi_flux_set_modulation_gemm(fmi_flux*, int enable);
/* fp8 inference mode (BASELINE.json configs[4]; the reference has no fp8 path, SURVEY.md §8d — the recipe
 * is this library's own, restated in oracle/flux_oracle.cpp:orc_quantize_rows_fp8).  Call once after all
 * tensors are set: every DiT block Linear (q|k|v, attention out, MLP, single-block linear1/linear2) is
 * quantised from its bf16 values to OCP e4m3 with one f32 scale per output channel; from then on their
 * GEMMs run on v_mfma_f32_32x32x64_f8f6f4 with activations quantised per token by the producing kernel
 * (AdaLN-modulate) or by one row pass (attention output, GELU(MLP)).  Attention, the residual stream,
 * modulation, embedders and the final layer stay bf16/f32.  Not combinable with bnb-quantised linears. */
int fmi_flux_quantize_fp8(fmi_flux*, void* stream);
/* int8 inference mode (round 4; this library's own recipe like the fp8 one, restated in oracle/flux_oracle.cpp: orc_quantize_rows_i8 and
 * lin_blk mode 5).  The same per-row scheme on symmetric int8 codes — scale = absmax / 127 per output channel (weights, once) and per
 * token (activations, by the producing AdaLN-modulate kernel or one row pass), codes = clamp(rint(x * 127 / absmax), -127, 127) — and
 * the GEMMs on v_mfma_i32_32x32x32_i8: the sum over k is an exact int32, converted once and multiplied by scale_a[m] * scale_w[n].
 * Why it exists: an e4m3 operand carries 2.65e-2 rms relative noise whatever the scale granularity (3-bit mantissa), an int8 one
 * 8.5e-3 (uniform steps over a Gaussian row), at the same matrix-pipe rate; with the default mask the full model stays within the
 * tolerance stated for 8-bit modes where the e4m3 mode is a factor three outside it.
 * `linear_mask` says WHICH block linears are quantised (the others keep their bf16 GEMM): FMI_INT8_DEFAULT_MASK = all but the double
 * blocks' MLP, the two linears that carry most of the error per millisecond saved (DESIGN.md 4.3c's table).  Attention, residual
 * stream, modulation, embedders and final layer stay bf16 / f32 — except that, as in the fp8 mode and under the same switch
 * (fmi_flux_set_fp8_attention, default on), the blocks whose q|k|v linear is in the mask hand q and k to the attention as e4m3 with the
 * static scales and QK^T runs on the fp8 MFMA (measured cost: 5.9e-3 alone, 2.40e-2 -> 2.50e-2 with the default mask at 6 + 12 blocks; P, V
 * and the accumulation are unchanged).  Call once after all tensors are set; not combinable with bnb-quantised linears or with
 * fmi_flux_quantize_fp8 (FMI_ERR_STATE). */
#define FMI_Q8_DOUBLE_QKV 1u      /* double blocks: q|k|v of both streams */
#define FMI_Q8_DOUBLE_OUT 2u      /* double blocks: attention output projections */
#define FMI_Q8_DOUBLE_MLP_IN 4u   /* double blocks: MLP linear 1 (+ GELU) */
#define FMI_Q8_DOUBLE_MLP_OUT 8u  /* double blocks: MLP linear 2 */
#define FMI_Q8_SINGLE_LINEAR1 16u /* single blocks: q|k|v|proj_mlp */
#define FMI_Q8_SINGLE_LINEAR2 32u /* single blocks: proj_out over cat(attention, gelu(mlp)) */
#define FMI_INT8_DEFAULT_MASK (FMI_Q8_DOUBLE_QKV | FMI_Q8_DOUBLE_OUT | FMI_Q8_SINGLE_LINEAR1 | FMI_Q8_SINGLE_LINEAR2)
int fmi_flux_quantize_int8(fmi_flux*, unsigned linear_mask, void* stream);
/* Calibration of the SMOOTHED int8 recipe (ABI 6, round 6).  The per-token grid above is sized by a row's largest element; real DiT activations have a
 * few hidden channels two orders of magnitude above the rest (their AdaLN (1 + scale) is 30-100 at every step), which would leave the other ~3 000
 * channels of the row a handful of levels.  fmi_flux_calibrate_int8(m, 1) — bf16 mode, all tensors set — zeroes a set of statistics; from then on every
 * fmi_flux_forward / fmi_flux_denoise evaluation ALSO folds max |x[:, k]| of each block linear's input into them (the results are unchanged; a handful of
 * evaluations at timesteps across the schedule is enough: the outlier channels are the same at every step).  The next fmi_flux_quantize_int8 consumes them:
 *   A[k] = amax_x[k] / median(amax_x),  W[k] = amax_W[k] / median(amax_W)      (amax_W[k] = max_n |W[n, k]|)
 *   s[k] = min(sqrt(ra / rw), 2^10),  ra = A - 1 if A > 2 else 1,  rw = W / (1 - W) if W < 1/2 (W floored at 1/64) else 1      (continuous; ~sqrt(A / W) for a genuine outlier)
 *   — SmoothQuant with alpha = 1/2 for the channels that stand out of the median ONLY: every other channel has s = 1 exactly, so a checkpoint without outlier channels
 *   quantises as without calibration, and a short calibration cannot create outliers of its own
 *   weight codes from W[n, k] * s[k], activation rows from x[m, k] * (1 / s[k]) — both in f32 before the per-row recipe; x W^T is unchanged in exact arithmetic
 * (the 1 / s multiply rides in the AdaLN-modulate kernel or the row pass that quantises the activation; the GEMMs are untouched) and ends the recording.
 * fmi_flux_calibrate_int8(m, 0) drops the statistics.  Without a calibration fmi_flux_quantize_int8 is bit for bit the unsmoothed recipe; the e4m3 mode
 * never smooths (a floating-point grid has no use for it).  Oracle: orc_flux_set_calibration (flux_oracle.cpp, lin_blk mode 5). */
int fmi_flux_calibrate_int8(fmi_flux*, int enable);
/* fp8 and int8 modes, attention operands: 1 (default) = q and k leave the fused QKV epilogue as e4m3 with static per-block
 * scales 448 / (sqrt(128) * max|QkNorm weight|) (no element of a normalised, rotated head vector can exceed them) and
 * QK^T runs on the fp8 MFMA; P and V stay bf16.  Applies when both streams of a block take the fused epilogue (token
 * counts and offsets multiples of 16, model width a multiple of 256), else that block uses bf16 operands.  0 = always bf16.
 * 2 = the same in EVERY mode, the bf16 block linears included (opt-in; the static scales are read from the QkNorm weights at the next
 * evaluation; blocks whose q|k|v linear is bnb-quantised keep bf16 operands; not with sequence parallelism).  A reduced-precision
 * attention operand is not the reference's semantics, which is why it is never the default outside the 8-bit modes — measured on
 * FLUX.1-dev in full at the headline size it moves the result less than the bf16 path's own distance to f32. */
int fmi_flux_set_fp8_attention(fmi_flux*, int enable);
int fmi_flux_missing_count(const fmi_flux*);
const char* fmi_flux_missing_name(const fmi_flux*, int i);
/* Bytes of HBM held by the model (weights + current workspace). */
size_t fmi_flux_size_in_bytes(const fmi_flux*);

/* One denoise-model evaluation == Flux::forward (model.rs:790-833).
 *   img      (B,S,in_channels)         img_dtype  F32|BF16
 *   img_ids  (B,S,3)  f32              (row/col positions; reference builds them in
 *                                        State::new, flux/sampling.rs:134-148)
 *   txt      (B,T,joint_attention_dim) txt_dtype  F32|BF16
 *   txt_ids  (B,T,3)  f32
 *   timesteps(B) f32,  y (B,pooled_projection_dim) y_dtype F32|BF16,
 *   guidance (B) f32 or NULL (must be non-NULL iff cfg.guidance_embeds, as model.rs:814-819)
 *   pred_out (B,S,in_channels) f32
 * Only batch element 0's ids are used to build the RoPE table when ids are identical
 * across the batch (they always are in the reference, sampling.rs:147-150); set
 * `ids_per_sample`=1 to build one table per sample. */
typedef struct fmi_flux_inputs {
  const void* img;      fmi_dtype img_dtype;
  const float* img_ids;
  const void* txt;      fmi_dtype txt_dtype;
  const float* txt_ids;
  const float* timesteps;
  const void* y;        fmi_dtype y_dtype;
  const float* guidance;
  int B, S, T;
  int ids_per_sample;
} fmi_flux_inputs;

int fmi_flux_forward(fmi_flux*, const fmi_flux_inputs* in, float* pred_out, void* stream);

/* The whole denoise loop == Sampler::sample(FlowMatchEulerDiscrete)
 * (pipelines/sampling.rs:25-48) around the step closure (pipelines/flux/mod.rs:305-318):
 *   for (t_curr,t_prev) in windows(timesteps): img += forward(img, t_curr) * (t_prev-t_curr)
 * `in->img` must be F32 and is the state: `img_inout` (B,S,C) f32 is updated in place
 * (in->img is ignored; in->timesteps is ignored, `timesteps_host` (n_steps+1 f64, host)
 * drives the loop, exactly the Vec<f64> of SchedulerConfig::get_timesteps).
 * The latent stays f32 between steps. */
int fmi_flux_denoise(fmi_flux*, const fmi_flux_inputs* in, float* img_inout,
                     const double* timesteps_host, int n_steps, void* stream);

/* Per-phase device time of the last forward in ms (hipEvents; enabled by
 * fmi_flux_set_profiling(1), which also serialises phases).  Phases: see fmi_flux_phase_name.
 * With FMI_ROCTX=1 in the environment every phase is also a roctx range (roctxRangePushA / Pop from
 * librocprofiler-sdk-roctx, opened at run time), so `rocprofv3 --marker-trace --kernel-trace` groups the trace by phase. */
int fmi_flux_set_profiling(fmi_flux*, int enable);
/* Tuning / verification knob (default 1): QkNorm (model.rs:433-452) + apply_rope (:77-101) + the
 * head-major q,k and transposed v relayout run in the epilogue of the [q|k|v] projection GEMM
 * instead of as stand-alone kernels over the stored projection.  Both forms use the same arithmetic
 * and give bit-identical results; shapes whose token offsets are not multiples of 16 always take the
 * stand-alone kernels. */
int fmi_flux_set_fused_qkv_relayout(fmi_flux*, int enable);
int fmi_flux_phase_count(void);
const char* fmi_flux_phase_name(int i);
int fmi_flux_phase_ms(fmi_flux*, float* ms_out /* [phase_count] */);

/* ------------------------------------------------------------------------------------
 * VAE decoder — replaces AutoEncoderKl::decode (vaes/autoencoder_kl.rs:112-119) and
 * Decoder::forward (vaes/vae.rs:436-456)
 * ---------------------------------------------------------------------------------- */
typedef struct fmi_vae_config {
  int in_channels;           /* 3  (unused by decode) */
  int out_channels;          /* 3 */
  int block_out_channels[4]; /* {128,256,512,512} */
  int n_blocks;              /* 4 */
  int layers_per_block;      /* 2 */
  int latent_channels;       /* 16 */
  int norm_num_groups;       /* 32 */
  int mid_block_add_attention;
  int use_post_quant_conv;   /* 0 for FLUX */
  double scaling_factor;     /* 0.3611 */
  double shift_factor;       /* 0.1159 */
  int use_quant_conv;        /* 0 for FLUX (encoder side 1x1 conv, autoencoder_kl.rs:67-77) */
} fmi_vae_config;

typedef struct fmi_vae fmi_vae;
void fmi_vae_default_config(fmi_vae_config* cfg);
int fmi_vae_create(const fmi_vae_config* cfg, fmi_model_dtype dtype, fmi_vae** out);
void fmi_vae_destroy(fmi_vae*);
/* diffusers names under "decoder." / "post_quant_conv.", e.g.
 * "decoder.conv_in.weight", "decoder.mid_block.resnets.0.norm1.weight",
 * "decoder.mid_block.attentions.0.to_q.weight", "decoder.up_blocks.2.resnets.0.conv_shortcut.weight",
 * "decoder.up_blocks.0.upsamplers.0.conv.bias", "decoder.conv_norm_out.weight" (vae.rs:371-433).
 * Conv weights are (Cout,Cin,kh,kw) as stored; they are re-laid out for the implicit-GEMM
 * kernels.  host or device pointer, F32/F16/BF16. */
int fmi_vae_set_tensor(fmi_vae*, const char* name, const void* data, fmi_dtype dtype,
                       const int64_t* shape, int rank);
int fmi_vae_missing_count(const fmi_vae*);
const char* fmi_vae_missing_name(const fmi_vae*, int i);
double fmi_vae_scale_factor(const fmi_vae*); /* VAEModel::scale_factor, vaes/mod.rs:15-28 */
double fmi_vae_shift_factor(const fmi_vae*);
/* z (B,latent_channels,h,w) f32 NCHW -> image (B,out_channels,8h,8w) f32 NCHW.
 * == VAEModel::decode. */
int fmi_vae_decode(fmi_vae*, const float* z, int B, int h, int w, float* image_out, void* stream);
/* image (B,in_channels,H,W) f32 NCHW, H and W multiples of 8 -> z (B,latent_channels,H/8,W/8) f32 =
 * mean + exp(0.5*logvar) * noise.  == VAEModel::encode = Encoder::forward (vaes/vae.rs:330-349),
 * optional quant_conv, DiagonalGaussian (vae.rs:470-480).  The reference draws `noise` with an
 * unseedable randn_like; here the caller passes it (fmi_randn) or NULL for z = mean.
 * moments_out (B,2*latent_channels,H/8,W/8) f32 optional.  decode needs only the `decoder.*`
 * tensors, encode only `encoder.*` (+ `quant_conv.*`). */
int fmi_vae_encode(fmi_vae*, const float* image, int B, int H, int W, const float* noise, float* z_out, float* moments_out, void* stream);
/* AttnBlock::forward (vaes/vae.rs:95-111) of the decoder's mid block as one op: x and out (B,H,W,C) bf16 NHWC device
 * buffers, C = block_out_channels.last, H*W a multiple of 64; out = x + to_out(sdpa(q,k,v)(group_norm(x))).  Needs the
 * `decoder.*` tensors.  Used by the parity tests to check the block alone at production size. */
int fmi_vae_mid_attention(fmi_vae*, const void* x_bf16_nhwc, int B, int H, int W, void* out_bf16_nhwc, void* stream);

/* ------------------------------------------------------------------------------------
 * Text encoders — they run once per image in front of the denoise loop and
 * produce its `txt` (B,T,4096) and `y` (B,768) inputs (pipelines/flux/mod.rs:237-262).
 * Token ids are int32, row-major (B,T), in host OR device memory (copied with hipMemcpyDefault).
 * Tokenisation itself (tokenizers crate, flux/mod.rs:203-221) stays on the host side.
 * ---------------------------------------------------------------------------------- */
/* T5Config (diffusion_rs_core/src/models/t5/mod.rs:72-91); feed_forward_proj as an enum */
typedef enum fmi_t5_act { FMI_T5_RELU = 0 /* "relu", ungated */, FMI_T5_GATED_GELU = 1 /* "gated-gelu" (NewGelu) */, FMI_T5_GATED_SILU = 2 } fmi_t5_act;
typedef struct fmi_t5_config {
  int vocab_size, d_model, d_kv, d_ff, num_layers, num_heads;
  int relative_attention_num_buckets, relative_attention_max_distance;
  float layer_norm_epsilon;
  fmi_t5_act feed_forward_act;
} fmi_t5_config;
typedef struct fmi_t5 fmi_t5;
void fmi_t5_default_config(fmi_t5_config*); /* t5-v1_1-xxl encoder (FLUX.1 text_encoder_2) */
/* == T5EncoderModel::new (t5/mod.rs:614-627); d_kv must be 64, d_model/d_ff multiples of 64;
 * quantised (bnb) T5 linears are not supported (FMI_ERR_INVALID on their side tensors). */
int fmi_t5_create(const fmi_t5_config*, fmi_t5** out);
void fmi_t5_destroy(fmi_t5*);
/* tensor names as T5EncoderModel's VarBuilder reads them: shared.weight,
 * encoder.block.N.layer.0.{layer_norm.weight, SelfAttention.{q,k,v,o}.weight},
 * encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight,
 * encoder.block.N.layer.1.{layer_norm.weight, DenseReluDense.{wi_0,wi_1,wo}.weight},
 * encoder.final_layer_norm.weight */
int fmi_t5_set_tensor(fmi_t5*, const char* name, const void* data, fmi_dtype dtype, const int64_t* shape, int rank);
/* bitsandbytes-quantised Linears of the encoder blocks (`<prefix>` = "encoder.block.N.layer.0.SelfAttention.q" ...): the
 * reference builds every T5 Linear as a QuantMethod from text_encoder_2's quantization_config (t5/mod.rs:132-173,258-261 ->
 * BnbLinear, bitsandbytes/mod.rs:111-239).  Arguments as fmi_flux_set_linear_bnb4 / _int8 (host or device pointers).  The
 * codes are expanded once at load into the encoder's bf16 weights (BnbLinear::forward = dequantise + matmul, mod.rs:293-312;
 * the encoder runs once per image), bit-exact with dequantize_blockwise_bf16_* / dequantize_8bit_kernel_bf16. */
int fmi_t5_set_linear_bnb4(fmi_t5*, const char* prefix, const uint8_t* packed, const float* absmax, int blocksize, int quant_type,
                           int out_features, int in_features);
int fmi_t5_set_linear_int8(fmi_t5*, const char* prefix, const int8_t* weight, const float* scb, int out_features, int in_features);
int fmi_t5_missing_count(const fmi_t5*);
const char* fmi_t5_missing_name(fmi_t5*, int i);
size_t fmi_t5_size_in_bytes(const fmi_t5*);
/* == T5EncoderModel::forward (t5/mod.rs:629-631): out (B,T,d_model) device, FMI_BF16 or FMI_F32.
 * Synchronises the stream before returning (token-id range check). */
int fmi_t5_forward(fmi_t5*, const int32_t* input_ids, int B, int T, void* out, fmi_dtype out_dtype, void* stream);

/* ClipTextConfig (models/clip/text.rs:24-33): projection_dim is used as the hidden width */
typedef struct fmi_clip_config {
  int vocab_size, projection_dim, intermediate_size, max_position_embeddings, num_hidden_layers, num_attention_heads;
} fmi_clip_config;
typedef struct fmi_clip fmi_clip;
void fmi_clip_default_config(fmi_clip_config*); /* clip-vit-large-patch14 text tower (FLUX.1 text_encoder) */
int fmi_clip_create(const fmi_clip_config*, fmi_clip** out); /* head dim must be 64 */
void fmi_clip_destroy(fmi_clip*);
/* names under the "text_model." prefix (flux/mod.rs:105), e.g.
 * text_model.encoder.layers.N.self_attn.q_proj.{weight,bias} */
int fmi_clip_set_tensor(fmi_clip*, const char* name, const void* data, fmi_dtype dtype, const int64_t* shape, int rank);
int fmi_clip_missing_count(const fmi_clip*);
const char* fmi_clip_missing_name(fmi_clip*, int i);
size_t fmi_clip_size_in_bytes(const fmi_clip*);
/* == ClipTextTransformer::forward (clip/text.rs:303-317): pooled (B,projection_dim) = final
 * hidden state at argmax(token id) of each row, FMI_F32 or FMI_BF16; hidden_out (B,T,dim) f32
 * optional (NULL to skip).  Synchronises the stream before returning. */
int fmi_clip_forward(fmi_clip*, const int32_t* input_ids, int B, int T, void* pooled_out, fmi_dtype pooled_dtype, float* hidden_out, void* stream);

/* ------------------------------------------------------------------------------------
 * Pipeline glue on device — the tensor code of FluxPipeline::forward
 * (pipelines/flux/mod.rs:270-332) and flux/sampling.rs
 * ---------------------------------------------------------------------------------- */
/* State::new patchify: latent (B,C,h,w) f32 -> img (B,(h/2)*(w/2),C*4) f32 and
 * img_ids (B,(h/2)(w/2),3) f32  (flux/sampling.rs:131-148). */
int fmi_pack_latents(const float* latent, int B, int C, int h, int w, float* img_out,
                     float* img_ids_out, void* stream);
/* unpack (flux/sampling.rs:162-169) fused with `img/scale_factor + shift_factor`
 * (flux/mod.rs:329): img (B,(h/2)(w/2),C*4) -> z (B,C,h,w). */
int fmi_unpack_latents(const float* img, int B, int C, int h, int w, double scale_factor,
                       double shift_factor, float* z_out, void* stream);
/* ((clamp(x,-1,1)+1)*127.5) as u8, truncating (flux/mod.rs:332, cpu_backend/mod.rs:2571-2574),
 * NCHW f32 -> NCHW u8 (interleave=0) or NHWC u8 (interleave=1, what Pipeline::forward hands
 * to RgbImage::from_raw, pipelines/mod.rs:253-266). */
int fmi_postprocess_u8(const float* image, int B, int C, int H, int W, int interleave,
                       uint8_t* out, void* stream);
/* Deterministic N(0,1) latents from a counter-based Philox4x32-10 generator
 * (the reference's RNG is unseedable; this is the explicit-seed extension).
 * Elements 4q..4q+3 of sample b come from the four words of philox(counter = (q lo, q hi, s lo, s hi),
 * key = (seed lo, seed hi)), s = first_sample + b: words (0,1) and (2,3) each feed one Box-Muller pair,
 * u = f32(f32(word >> 8) + 0.5f) * 2^-24 (f32 roundings included), z = sqrt(-2 ln u1) * (cos, sin)(2 pi u2).  Replaces get_noise
 * (pipelines/flux/sampling.rs:5-14). */
int fmi_randn(float* out, int64_t n_per_sample, int B, uint64_t seed, uint64_t first_sample,
              void* stream);
/* The raw 32-bit words fmi_randn draws from, same indexing (out (B, n_per_sample) u32): integer work,
 * bit-exact against the oracle's Philox, which is pinned by Random123's known-answer vectors. */
int fmi_philox_u32(uint32_t* out, int64_t n_per_sample, int B, uint64_t seed, uint64_t first_sample,
                   void* stream);

/* Host-side schedule helpers (f64, bit-for-bit the reference formulas). */
double fmi_calculate_shift(int image_seq_len, int base_seq_len, int max_seq_len,
                           double base_shift, double max_shift); /* flux/sampling.rs:171-181 */
typedef struct fmi_scheduler_config { /* SchedulerConfig, pipelines/scheduler.rs:4-20 */
  int base_image_seq_len; double base_shift; int max_image_seq_len; double max_shift;
  double shift; int use_dynamic_shifting;
} fmi_scheduler_config;
/* out_host[num_steps+1]; mu is ignored unless use_dynamic_shifting. (scheduler.rs:28-51) */
int fmi_get_timesteps(const fmi_scheduler_config* cfg, int num_steps, double mu, double* out_host);

/* ------------------------------------------------------------------------------------
 * Operator-level entry points (seam S3) — used by the parity tests and
 * usable as CustomOp::cuda_fwd replacements.
 * ---------------------------------------------------------------------------------- */
typedef enum fmi_epilogue {
  FMI_EPI_NONE = 0,      /* y = x W^T (+bias) */
  FMI_EPI_GELU_TANH = 1, /* y = gelu_tanh(x W^T + bias)  (Mlp, model.rs:459-463) */
  FMI_EPI_SILU = 2
} fmi_epilogue;

/* y(M,N) = epi(x(M,K) · W(N,K)^T + bias(N)); x,W,y bf16, bias bf16 or NULL, f32 accumulate.
 * == UnquantLinear::forward (unquantized/mod.rs:34-77). K % 64 == 0 required. */
int fmi_linear_bf16(const void* x, const void* w, const void* bias, void* y, int M, int N,
                    int K, fmi_epilogue epi, void* stream);
/* Same with a bitsandbytes 4-bit weight: W = dequant(packed, absmax, blocksize) fused into
 * the GEMM's weight-tile load (BnbLinear::forward semantics, bitsandbytes/mod.rs:301-312,
 * minus the dense round trip).  quant_type 1=fp4, 2=nf4. K % 64 == 0, blocksize % 64 == 0. */
int fmi_linear_bnb4_bf16(const void* x, const uint8_t* packed, const float* absmax,
                         int blocksize, int quant_type, const void* bias, void* y, int M,
                         int N, int K, fmi_epilogue epi, void* stream);
/* Same with an LLM.int8 weight (BnbLinear::Int8, bitsandbytes/mod.rs:104-134, 293-300): W = weight_i8 * SCB[row] / 127
 * (dequantize_8bit, dequant.cu:205-214) expanded inside the GEMM's weight-tile stage, bit-identical to the stand-alone
 * dequant followed by fmi_linear_bf16.  weight (N,K) int8 row-major, 16-byte aligned; scb (N) f32.  K % 64 == 0. */
int fmi_linear_int8_bf16(const void* x, const int8_t* weight, const float* scb, const void* bias, void* y,
                         int M, int N, int K, fmi_epilogue epi, void* stream);
/* Row-wise dynamic e4m3 quantisation: scale[r] = max(absmax(x[r,:]), 1e-30) / 448,
 * out[r,k] = e4m3_rne(x[r,k] * (448 / max(absmax, 1e-30))).  x (rows,K) bf16, out (rows,K) u8, K % 8 == 0,
 * K <= 16384.  Used for weights (row = output channel) and activations (row = token). */
int fmi_quantize_rows_fp8(const void* x, int rows, int K, uint8_t* out, float* scale, void* stream);
/* y(M,N) = epi((xq · Wq^T) * xs[m] * ws[n] + bias), xq/xs = fmi_quantize_rows_fp8(x) computed internally,
 * Wq (N,K) e4m3 + ws (N) from fmi_quantize_rows_fp8(W); y bf16, f32 accumulate on the fp8 MFMA.
 * N > 128, K % 128 == 0. */
int fmi_linear_fp8(const void* x, const uint8_t* wq, const float* w_scale, const void* bias, void* y, int M, int N,
                   int K, fmi_epilogue epi, void* stream);
/* The int8 forms of the two (fmi_flux_quantize_int8's recipe; oracle: orc_quantize_rows_i8 / orc_linear_i8): scale[r] =
 * max(absmax, 1e-30) / 127, out[r,k] = clamp(rint(x[r,k] * (127 / max(absmax, 1e-30))), -127, 127); y = epi(float(xq · Wq^T as an exact
 * int32) * (xs[m] * ws[n]) + bias) on v_mfma_i32_32x32x32_i8.  Same shape rules. */
int fmi_quantize_rows_i8(const void* x, int rows, int K, int8_t* out, float* scale, void* stream);
int fmi_linear_i8(const void* x, const int8_t* wq, const float* w_scale, const void* bias, void* y, int M, int N,
                  int K, fmi_epilogue epi, void* stream);
/* Caller-owned workspace forms of the two (the reference's convention: the caller allocates, bitsandbytes/op.rs:204-228): `workspace`
 * = at least fmi_linear_q8_workspace_bytes(M, K) bytes of device memory, 256-byte aligned, that must stay untouched until the call's work
 * has run on `stream`.  fmi_linear_fp8 / fmi_linear_i8 above are these with the workspace taken from a grow-only block the
 * library keeps per (device, stream): every form is asynchronous on the stream — a call allocates only if it needs more than any earlier
 * call on that stream did (once, without waiting for the stream), and none synchronises the host. */
size_t fmi_linear_q8_workspace_bytes(int M, int K);
int fmi_linear_fp8_ws(const void* x, const uint8_t* wq, const float* w_scale, const void* bias, void* y, int M, int N, int K,
                      fmi_epilogue epi, void* workspace, size_t workspace_bytes, void* stream);
int fmi_linear_i8_ws(const void* x, const int8_t* wq, const float* w_scale, const void* bias, void* y, int M, int N, int K,
                     fmi_epilogue epi, void* workspace, size_t workspace_bytes, void* stream);
/* The 8-bit GEMM alone on operands the caller already quantised with fmi_quantize_rows_fp8 (kind 1) / fmi_quantize_rows_i8 (kind 2):
 * y(M,N) bf16 = epi(acc(xq · wq^T) * x_scale[m] * w_scale[n] + bias); stream-ordered, no allocation (the form a caller that keeps
 * activations quantised between layers binds; also what tools/hipblaslt_yardstick.py times against the vendor library's fp8 GEMM). */
int fmi_gemm_q8(const void* xq, const float* x_scale, const void* wq, const float* w_scale, const void* bias, void* y,
                int M, int N, int K, int kind, fmi_epilogue epi, void* stream);
/* The int8 recipe's form for POST-GELU operands (round 5; oracle: orc_quantize_rows_i8_asym / orc_linear_i8_asym).  gelu(h) >= -0.17, so a grid
 * symmetric around 0 wastes half of its codes there.  Columns [0, d0) of a row (a signed segment in front — the single blocks' linear2 reads
 * cat(attention, gelu(mlp)), d0 = 3072; d0 = 0 for none) stay symmetric, columns [d0, K) take 256 levels over their [min, max] = [lo, hi], all with
 * ONE step per row so the product stays one exact int32 sum:
 *   scale[r] = s = max(max((hi - lo) / 255, absmax(front) / 127), 1e-30);  offset[r] = lo + 128 s
 *   k <  d0: out = clamp(rint(x / s), -127, 127)            k >= d0: out = clamp(rint((x - lo) / s), 0, 255) - 128
 *   fmi_rowsum_i8:    w_sum[n] = w_scale[n] * float(sum_{k >= d0} wq[n,k])      (once per weight)
 *   fmi_gemm_i8_asym: y = epi(float(xq · wq^T) * (x_scale[m] * w_scale[n]) + x_offset[m] * w_sum[n] + bias[n])
 * x (rows,K) bf16, K % 8 == 0, K <= 16384, d0 % 16 == 0; GEMM shape rules as fmi_gemm_q8.  What fmi_flux_quantize_int8 uses in front of the double
 * blocks' MLP-out and the single blocks' linear2.  Stream-ordered, nothing allocated. */
int fmi_quantize_rows_i8_asym(const void* x, int rows, int K, int d0, int8_t* out, float* scale, float* offset, void* stream);
int fmi_rowsum_i8(const int8_t* wq, const float* w_scale, int N, int K, int d0, float* w_sum, void* stream);
/* The smoothed int8 recipe (fmi_flux_calibrate_int8) at the op level (ABI 6): fmi_quantize_rows_i8_scaled = fmi_quantize_rows_i8 (d0 < 0; `offset` unused)
 * or fmi_quantize_rows_i8_asym (d0 >= 0) on x[r, k] * col_scale[k], the product taken in f32 (col_scale: K floats, 16-byte aligned; 1 / s for an
 * activation, s for a weight).  fmi_col_absmax folds max_r |x[r, k]| of a bf16 matrix (rows, K) with row stride ld into amax_inout[k] (a running maximum:
 * zero it first; K, ld % 8 == 0) — the statistic s[k] = sqrt(amax_x[k] / amax_W[k]) is made of. */
int fmi_quantize_rows_i8_scaled(const void* x, int rows, int K, int d0, const float* col_scale, int8_t* out, float* scale, float* offset, void* stream);
int fmi_col_absmax(const void* x, int rows, int K, int ld, float* amax_inout, void* stream);
int fmi_gemm_i8_asym(const void* xq, const float* x_scale, const float* x_offset, const void* wq, const float* w_scale, const float* w_sum,
                     const void* bias, void* y, int M, int N, int K, fmi_epilogue epi, void* stream);
/* softmax(q k^T * scale) v, q,k,v,o (B,H,L,d) bf16, d == 128, non-causal; o is written
 * token-major (B,L,H*d) when `out_token_major`, else (B,H,L,d).
 * == backend::ops::sdpa fallback (ops.rs:247-262) without materialising the scores. */
int fmi_sdpa_bf16(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq,
                  int Lk, int d, float scale, int out_token_major, void* stream);
/* The attention kernels read V transposed (B,H,128,Lk rounded up to 64) with the kv index permuted inside groups of 16; the op-level
 * entry points build that image first.  fmi_sdpa_bf16 / fmi_sdpa_fp8qk keep it in a grow-only block the library holds per (device, stream)
 * — calls on one stream run in order and share it; a call allocates only when it needs more than any earlier call on that stream (once,
 * without waiting for the stream) and none synchronises the host: a host that binds ops::sdpa to them enqueues 57 calls per denoise step
 * without ever waiting —; the *_ws forms take it from the caller
 * (fmi_sdpa_workspace_bytes(B, H, Lk) bytes, 16-byte aligned, untouched until the call's work has run on `stream`). */
size_t fmi_sdpa_workspace_bytes(int B, int H, int Lk);
/* Threads (ABI 6): an entry that uses the library-held block takes one lock from the moment it picks the block until its last kernel is enqueued, so
 * host threads that share a stream cannot interleave their launches around it; the block is keyed by (device, stream), NULL and hipStreamLegacy being
 * one stream and hipStreamPerThread one stream PER calling thread.  fmi_release_scratch returns every such block to the driver: it drains the
 * devices that hold one first (the only op-level entry that waits; for teardown or memory pressure) and reports the bytes freed
 * (`freed_bytes` may be NULL).  The next op-level call allocates again. */
int fmi_release_scratch(size_t* freed_bytes);
int fmi_sdpa_bf16_ws(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int d, float scale,
                     int out_token_major, void* workspace, size_t workspace_bytes, void* stream);
/* Process-wide choice of the bf16 attention kernel (test / benchmark hook):
 *   5 (default) round 4's lock-step schedule of kernel 3 (attention_w16l.h: the wave's four 16-query blocks walk a KV tile together, one
 *     K / V^T fragment read feeds four MFMAs) — the arithmetic of 3 / 4; bit-identical to them when every tile rescales, equal to
 *     rounding otherwise (the deferred-rescale decision is taken over 64 queries instead of 32);
 *   3 one wave per SIMD on v_mfma_f32_16x16x32_bf16, the whole KV stream generated assembly (attention_w16.h);
 *   4 the same design on v_mfma_f32_32x32x16_bf16 (attention_w32.h) — 3 and 4 are bit-identical to each other;
 *   2 round 2's one-wave kernel (attention_w4.h), 1 the 8-wave ping-pong kernel, 0 the 8-wave single-barrier kernel — these
 *   three are bit-identical to each other, and equal to 3 / 4 to rounding (3 / 4 round q * scale * log2(e) to bf16 once and
 *   normalise by the sum of the bf16-rounded probabilities: rel-L2 ~3e-3 between the families, both within the oracle tolerance). */
int fmi_set_attention_kernel(int kind);
/* Same with q and k as OCP e4m3 bytes (B,H,L,128): QK^T on the fp8 MFMA, softmax / P / V in f32 / bf16 as above.
 * `scale` must include 1 / (q scale * k scale).  The fp8-mode attention of fmi_flux_* (fmi_flux_set_fp8_attention). */
int fmi_sdpa_fp8qk(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq,
                  int Lk, int d, float scale, int out_token_major, void* stream);
/* The one-wave fp8 streams carry the score factor scale * log2(e) in the MFMA's E8M0 block scale, so they serve factors 2^n,
 * n = -126 .. 0; anything else runs the 8-wave kernel (slower; counted in fmi_device_info's "fp8_attention_fallbacks").  A caller that
 * KNOWS its factor is 2^n says so as an integer: score_exp2 = n (then `scale` is not read); FMI_SDPA_NO_EXP2 = derive it from `scale`
 * (accepted when f32(scale * log2 e) is a power of two or one ulp beside one, which is where scale = 2^n / log2(e) built in f32 lands). */
#define FMI_SDPA_NO_EXP2 (1 << 30)
int fmi_sdpa_fp8qk_ws(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int d, float scale,
                      int score_exp2, int out_token_major, void* workspace, size_t workspace_bytes, void* stream);
/* Round 5: every operand of both products as OCP e4m3 — q, k bytes as fmi_sdpa_fp8qk; v is handed over as bf16 (B,H,Lk,128) and enters the
 * second product as e4m3(clamp(v * v_scale, +-448)); the probabilities exp2(s - m) are rounded to e4m3 as they are (the deferred rescale keeps them
 * <= 64), the row sums are taken over the rounded values, and 1 / v_scale leaves with the final normalisation.  Both products run on the
 * K = 128 / K = 64 fp8 MFMA at twice the bf16 rate (attention_w16l_kernel<.., true, true>).  Needs score_exp2 (or a `scale` that is a power of
 * two as above) and Lk > 64 — FMI_ERR_UNSUPPORTED otherwise: no other kernel reads this V^T.  An OP-LEVEL entry only: the model's 8-bit modes keep
 * bf16 P and V (the stream is issue-bound, the e4m3 second product buys no time: DESIGN 4.4), and fmi_flux_set_fp8_attention rejects 3.
 * Workspace: fmi_sdpa_workspace_bytes (half of it is used). */
int fmi_sdpa_fp8(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int d, float scale, int score_exp2,
                 float v_scale, int out_token_major, void* stream);
int fmi_sdpa_fp8_ws(const void* q, const void* k, const void* v, void* o, int B, int H, int Lq, int Lk, int d, float scale, int score_exp2,
                    float v_scale, int out_token_major, void* workspace, size_t workspace_bytes, void* stream);
/* LayerNorm(eps, no affine) then x*(1+scale)+shift: x (rows,D) f32 -> out bf16;
 * scale/shift f32 (D) (layer_norm helper model.rs:33-38 + ModulationOut::scale_shift :218-221).
 * scale/shift may be NULL (plain LN). */
int fmi_layernorm_mod(const float* x, const float* scale, const float* shift, void* out_bf16,
                      int rows, int D, float eps, void* stream);
/* The small f32 pieces of Flux::forward, one at a time (ABI 6; the kernels the model itself launches):
 * timestep_embedding (model.rs:104-122): t (B) f32 -> out (B, dim) f32 = [cos(1000 t f_i), sin(1000 t f_i)], f_i = exp(-ln(1e4) i / (dim/2)) in f32. */
int fmi_timestep_embedding(const float* t, int B, int dim, float* out, void* stream);
/* EmbedNd / rope (model.rs:65-102, 124-163): ids (B,T,3) and (B,S,3) f32 (either count may be 0) -> pe (B, T+S, sum(axes_dim)/2, 2) f32 = {cos, sin} of
 * pos * inv_freq, text rows first; inv_freq = 1f32 / f32(theta^(2j/dim) evaluated in f64) as model.rs:71-74.  (The reference materialises the 2x2
 * rotation [cos, -sin, sin, cos]; this table keeps its two distinct entries.) */
int fmi_rope_table(const float* txt_ids, const float* img_ids, int B, int T, int S, const int* axes_dim /*[3]*/, int theta, float* pe, void* stream);
/* QkNorm (RmsNorm over the head dim, eps 1e-6, model.rs:186-209) + apply_rope (model.rs:52-63) in front of the attention: q, k (B, L, ld >= H*128)
 * bf16 token-major (head h at columns h*128..), weights (128) bf16, pe (B, L, 64, 2) f32 from fmi_rope_table -> q_out, k_out (B, H, L, 128)
 * head-major, bf16 (out_dtype FMI_BF16: the operands the attention reads; the q|k|v GEMM's fused epilogue produces the same bits) or f32
 * (FMI_F32: the same arithmetic without the final rounding).  d must be 128; every pointer 16-byte aligned. */
int fmi_rmsnorm_rope(const void* q, const void* k, int ld, const void* q_weight, const void* k_weight, const float* pe, void* q_out, void* k_out,
                     int B, int H, int L, int d, fmi_dtype out_dtype, void* stream);
/* GroupNorm (+optional SiLU) on NHWC bf16 activations, f32 two-pass statistics
 * (nn/group_norm.rs:39-74): x (B,HW,C). */
int fmi_groupnorm_nhwc(const void* x_bf16, const float* weight, const float* bias, void* out_bf16,
                       int B, int HW, int C, int groups, float eps, int fuse_silu, void* stream);
/* 3x3/1x1 stride-1 conv, zero pad k/2, NHWC bf16, weights (Cout,kh,kw,Cin) bf16, bias bf16 (Cout)
 * or NULL; Cin % 64 == 0 (zero-pad the channels);
 * optional nearest-2x upsample folded into the input gather (Upsample::forward vae.rs:223-229)
 * and optional residual add (ResnetBlock::forward vae.rs:157-172). (in_h,in_w) is the stored
 * input size; output is (in_h*(up?2:1), in_w*(up?2:1)). */
int fmi_conv2d_nhwc(const void* x_bf16, const void* w_bf16, const void* bias_bf16,
                    const void* residual_bf16, void* out_bf16, int B, int in_h, int in_w, int Cin,
                    int Cout, int ksize, int upsample2x, void* stream);

/* The reference's own extern "C" convention, kept verbatim so ffi.rs:5-114 binds without
 * edits (CUstream -> hipStream_t).  n = number of OUTPUT elements; blocksize as stored in
 * quant_state; `code` is the 256-entry map for int8 and ignored for fp4/nf4 (the kernels
 * use the constant tree/LUT of dequant.cu:12-92). */
void dequantize_blockwise_f32_int8(const float* code, const uint8_t* A, const float* absmax, float* out, int blocksize, int n, void* stream);
void dequantize_blockwise_f32_fp4(const float* code, const uint8_t* A, const float* absmax, float* out, int blocksize, int n, void* stream);
void dequantize_blockwise_f32_nf4(const float* code, const uint8_t* A, const float* absmax, float* out, int blocksize, int n, void* stream);
void dequantize_blockwise_f16_int8(const float* code, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream);
void dequantize_blockwise_f16_fp4(const float* code, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream);
void dequantize_blockwise_f16_nf4(const float* code, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream);
void dequantize_blockwise_bf16_int8(const float* code, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream);
void dequantize_blockwise_bf16_fp4(const float* code, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream);
void dequantize_blockwise_bf16_nf4(const float* code, const uint8_t* A, const float* absmax, void* out, int blocksize, int n, void* stream);
/* LLM.int8 weight: out[i] = w[i]*SCB[i/col]/127 (dequant.cu:205-232). The reference
 * launches these on the legacy default stream; so do we (stream argument absent, as in ffi.rs). */
void dequantize_8bit_kernel_f32(const int8_t* weight, const float* scb, float* out, int row, int col, int n);
void dequantize_8bit_kernel_f16(const int8_t* weight, const float* scb, void* out, int row, int col, int n);
void dequantize_8bit_kernel_bf16(const int8_t* weight, const float* scb, void* out, int row, int col, int n);

/* ------------------------------------------------------------------------------------
 * Small device-memory helpers so a non-HIP host (Rust, ctypes) needs nothing else.
 * ---------------------------------------------------------------------------------- */
int fmi_malloc(void** dptr, size_t bytes);
int fmi_free(void* dptr);
int fmi_memcpy(void* dst, const void* src, size_t bytes, void* stream); /* hipMemcpyDefault, async */
int fmi_memset(void* dst, int value, size_t bytes, void* stream);
int fmi_stream_synchronize(void* stream);
/* Event timing on an arbitrary stream (bench.py uses these so the timed region is measured
 * on the stream the kernels are launched on). */
int fmi_event_create(void** ev);
int fmi_event_record(void* ev, void* stream);
int fmi_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int fmi_event_destroy(void* ev);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* FLUX_MI355X_H */
