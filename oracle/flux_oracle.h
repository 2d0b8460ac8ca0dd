/*
 * flux_oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * The oracle is a plain C++17 / f32 restatement of the reference's CPU path for the FLUX.1
 * denoise loop + VAE decode.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product (libflux_mi355x.so) never links it.
 *
 * Parity pin status: the op-level functions are pinned by the reference's own known-answer
 * vectors (tests/golden/reference_kats.json, transcribed from the orphaned candle tests the
 * reference vendors — SURVEY §8c).  The model-level functions (flux forward, sampler, VAE
 * decoder, bnb dequant) have NO golden vector anywhere in the reference ("parity unpinned"
 * at model level); they are pinned only by following the cited lines and by
 * independent PyTorch-CPU re-derivations committed under tests/golden/.
 */
#ifndef FLUX_ORACLE_H
#define FLUX_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- op level ---- */
void orc_set_threads(int n);
int orc_get_threads(void);
void orc_linear(const float* x, const float* w, const float* bias, int M, int N, int K, float* y);
void orc_layer_norm(const float* x, const float* alpha, const float* beta, float eps, int rows, int cols, float* out);
void orc_rms_norm_slow(const float* x, const float* alpha, float eps, int rows, int cols, float* out);
void orc_softmax_last_dim(const float* x, int rows, int cols, float* out);
void orc_group_norm(const float* x, const float* w, const float* b, int B, int C, int HW, int groups, float eps, float* out);
void orc_conv2d(const float* x, const float* w, const float* bias, int B, int Cin, int H, int W, int Cout, int kh, int kw, int pad, int stride, int dilation, float* out);
void orc_upsample_nearest2d(const float* x, int B, int C, int H, int W, int dstH, int dstW, float* out);
void orc_gelu(const float* x, int64_t n, float* out);
void orc_silu(const float* x, int64_t n, float* out);
void orc_sdpa(const float* q, const float* k, const float* v, int B, int H, int Lq, int Lk, int d, float scale, float* out);
void orc_rope_table(const float* ids, int n, int n_axes, const int* axes_dim, int theta, float* pe_out);
void orc_apply_rope(const float* x, const float* pe, int H, int L, int d, float* out);
void orc_timestep_embedding(const float* t, int B, int dim, float* out);
/* bitsandbytes — CUDA-kernel semantics (dequant.cu), quant_type 0=int8(code) 1=fp4 2=nf4,
 * out_dtype 0=f32 1=f16 2=bf16 (value rounded RNE to that type, returned widened to f32). */
void orc_dequantize_blockwise(const float* code, const uint8_t* A, const float* absmax, float* out, int blocksize, int n, int quant_type, int out_dtype);
void orc_dequantize_8bit(const int8_t* w, const float* scb, float* out, int row, int col, int n, int out_dtype);
/* nf4/fp4 quantiser used only to make synthetic test weights (bnb rule: absmax per block,
 * nearest code).  Not part of the reference (it only dequantises). */
void orc_quantize_blockwise_4bit(const float* w, int64_t n, int blocksize, int quant_type, uint8_t* packed, float* absmax);
float orc_round_bf16(float v);
float orc_round_f16(float v);

/* ---- pipeline level ---- */
double orc_calculate_shift(int image_seq_len, int base_seq_len, int max_seq_len, double base_shift, double max_shift);
void orc_get_timesteps(int num_steps, int use_dynamic_shifting, double mu, double shift, double* out);
void orc_pack_latents(const float* latent, int B, int C, int h, int w, float* img, float* img_ids);
void orc_unpack_latents(const float* img, int B, int C, int h, int w, float* out);
void orc_postprocess_u8(const float* x, int64_t n, uint8_t* out);
void orc_exact_bf16(uint16_t* out, int64_t n, uint64_t seed, float offset, float coeff);
void orc_philox_u32(uint32_t* out, int64_t n_per_sample, int B, uint64_t seed, uint64_t first_sample);

/* ---- FLUX model ---- */
typedef struct orc_flux orc_flux;
orc_flux* orc_flux_create(int in_channels, int pooled_projection_dim, int joint_attention_dim, int num_attention_heads, int num_layers, int num_single_layers, int guidance_embeds, const int* axes_dim, int theta);
void orc_flux_destroy(orc_flux*);
int orc_flux_set_tensor(orc_flux*, const char* name, const float* data, int64_t numel);
int orc_flux_set_tensor_bf16(orc_flux*, const char* name, const uint16_t* data, int64_t numel);  /* same values as bf16 bits, widened per block */
/* fp8 recipe (BASELINE configs[4]; no reference counterpart — parity unpinned, see flux_oracle.cpp) */
void orc_flux_set_fp8(orc_flux*, int on);
void orc_flux_set_fp8_attention(orc_flux*, int on); /* with set_fp8: q, k of the attention on e4m3 with static scales */
float orc_e4m3_to_f32(uint8_t code);
uint8_t orc_f32_to_e4m3(float x);
void orc_quantize_rows_fp8(const float* x, int rows, int K, uint8_t* out, float* scale);
/* int8 recipe (round 4; no reference counterpart — parity unpinned: pinned only to exact integer arithmetic, tests/test_oracle_fp8.py):
 * orc_flux_set_fp8(m, 5) = the block linears of orc_flux_set_q8_mask's bits (0 double q|k|v, 1 double attention out, 2 double MLP in,
 * 3 double MLP out, 4 single linear1, 5 single linear2; default all) on symmetric per-row int8 codes with EXACT integer sums, the others f32 */
void orc_flux_set_q8_mask(orc_flux*, int mask);
void orc_quantize_rows_i8(const float* x, int rows, int K, int8_t* out, float* scale);
void orc_linear_i8(const float* x, const float* w, const float* bias, int M, int N, int K, float* y);
/* returns 0 ok, <0 on missing tensor (name printed to stderr) */
int orc_flux_forward(orc_flux*, const float* img, const float* img_ids, const float* txt, const float* txt_ids, const float* timesteps, const float* y, const float* guidance, int B, int S, int T, float* pred);
int orc_flux_denoise(orc_flux*, float* img_inout, const float* img_ids, const float* txt, const float* txt_ids, const float* y, const float* guidance, int B, int S, int T, const double* timesteps, int n_steps);
/* block-level hooks for per-block parity tests and the CPU baseline timing */
int orc_flux_double_block(orc_flux*, int idx, float* img_inout, float* txt_inout, const float* vec, const float* pe, int B, int S, int T);
int orc_flux_single_block(orc_flux*, int idx, float* x_inout, const float* vec, const float* pe, int B, int L);

/* ---- VAE decoder ---- */
typedef struct orc_vae orc_vae;
orc_vae* orc_vae_create(const int* block_out_channels, int n_blocks, int layers_per_block, int latent_channels, int out_channels, int norm_num_groups, int mid_block_add_attention, int use_post_quant_conv);
void orc_vae_destroy(orc_vae*);
int orc_vae_set_tensor(orc_vae*, const char* name, const float* data, int64_t numel);
int orc_vae_decode(orc_vae*, const float* z, int B, int h, int w, float* out);
int orc_vae_mid_attention(orc_vae*, float* x_nchw_inout, int B, int H, int W);  /* AttnBlock of the decoder mid block alone */
/* image (B,in_channels,H,W) -> moments (B,2*latent,H/8,W/8) [optional] and z = mean + exp(0.5 logvar) * noise (noise NULL: z = mean) */
int orc_vae_encode(orc_vae*, const float* img, int B, int in_channels, int H, int W, int use_quant_conv, const float* noise, float* moments_out, float* z_out);

/* ---- text encoders (oracle/text_oracle.cpp) ---- */
typedef struct orc_t5 orc_t5;
/* act: 0 = relu, ungated (T5DenseActDense); 1 = gated-gelu (NewGelu); 2 = gated-silu */
orc_t5* orc_t5_create(int vocab_size, int d_model, int d_kv, int d_ff, int num_layers, int num_heads, int rel_buckets, int rel_max_distance, float eps, int act);
void orc_t5_destroy(orc_t5*);
int orc_t5_set_tensor(orc_t5*, const char* name, const float* data, int64_t numel);
int orc_t5_bucket(int i, int j, int num_buckets_total, int max_distance);
int orc_t5_forward(orc_t5*, const int32_t* ids, int B, int T, float* out /* (B,T,d_model) */);
typedef struct orc_clip orc_clip;
orc_clip* orc_clip_create(int vocab_size, int projection_dim, int intermediate_size, int max_position_embeddings, int num_hidden_layers, int num_attention_heads);
void orc_clip_destroy(orc_clip*);
int orc_clip_set_tensor(orc_clip*, const char* name, const float* data, int64_t numel);
int orc_clip_forward(orc_clip*, const int32_t* ids, int B, int T, float* hidden_out /* (B,T,hidden) or NULL */, float* pooled /* (B,hidden) */);

#ifdef __cplusplus
}
#endif
#endif
