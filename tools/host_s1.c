/*
 * host_s1.c — a torch-free, Python-free host that drives seam S1 of the C-ABI end to end (VERDICT r5 "missing" 3).
 *
 * Plain C99 against include/flux_mi355x.h and libflux_mi355x.so — no HIP header, no PyTorch, no Python: the device buffers come from
 * fmi_malloc / fmi_memcpy, which is all a Rust host would need besides the `extern "C"` block of INTEGRATION.md.  It does what
 * `Pipeline::forward` does for one prompt once the text encoders have run (diffusion_rs_core/src/pipelines/mod.rs:241-270,
 * pipelines/flux/mod.rs:270-332):
 *
 *   fmi_init -> fmi_flux_create -> fmi_flux_set_tensor x 780 (HOST pointers)  ==  Flux::new over a VarBuilder      (model.rs:722-787)
 *   fmi_vae_create -> fmi_vae_set_tensor x 138                                ==  AutoEncoderKl::new               (vaes/autoencoder_kl.rs)
 *   noise -> fmi_pack_latents                                                 ==  get_noise + State::new           (flux/sampling.rs)
 *   fmi_calculate_shift / fmi_get_timesteps                                   ==  SchedulerConfig::get_timesteps   (scheduler.rs:22-51)
 *   fmi_flux_denoise                                                          ==  Sampler::sample + step closure   (sampling.rs:25-48, flux/mod.rs:305-318)
 *   fmi_unpack_latents -> fmi_vae_decode -> fmi_postprocess_u8                ==  flux/mod.rs:320-332
 *
 * There are no checkpoints offline, so the "checkpoint" is the exact synthetic one of diffusion-rs_amd/synth.py (every tensor = Philox4x32-10
 * words seeded by its NAME -> byte sums -> one f32 multiply -> bf16; restated below in ~40 lines of C): the Python path (ctypes + torch
 * buffers) builds the same bits on the device, so the two hosts must produce the same image — tests/test_gpu_c2_fixture.py compares the
 * CRC-32 this program prints with the Python path's.
 *
 * Build:  gcc -std=c99 -O2 -fopenmp -Iinclude tools/host_s1.c -o tools/bin/host_s1 -Ldiffusion-rs_amd -lflux_mi355x -Wl,-rpath,$PWD/diffusion-rs_amd -lm
 * Run:    tools/bin/host_s1 [--steps 50] [--latent 128x128] [--txt 512] [--guidance 3.5] [--mode bf16|int8|fp8] [--dump-u8 file]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "flux_mi355x.h"

#define CHECK(call)                                                                             \
  do {                                                                                          \
    int rc_ = (call);                                                                           \
    if (rc_ != FMI_OK) {                                                                        \
      fprintf(stderr, "host_s1: %s -> %d: %s\n", #call, rc_, fmi_last_error());                 \
      exit(1);                                                                                  \
    }                                                                                           \
  } while (0)

/* ---- CRC-32 (IEEE 802.3, the zlib polynomial) ---- */
static uint32_t crc_table[256];
static void crc_init(void) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    crc_table[i] = c;
  }
}
static uint32_t crc32_of(const void* data, size_t n) {
  const uint8_t* p = (const uint8_t*)data;
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

/* ---- exact synthetic tensors (diffusion-rs_amd/synth.py: exact_values_np is the definition) ---- */
static void philox_block(uint64_t q, uint64_t seed, uint32_t w[4]) {
  uint32_t c0 = (uint32_t)q, c1 = (uint32_t)(q >> 32), c2 = 0, c3 = 0; /* sample 0 */
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0, c1 = n1, c2 = n2, c3 = n3;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  w[0] = c0, w[1] = c1, w[2] = c2, w[3] = c3;
}
static uint64_t exact_seed(const char* name) { return ((uint64_t)0x46495854u << 32) | crc32_of(name, strlen(name)); }
static void exact_bf16(uint16_t* out, int64_t n, const char* name, double offset, double scale) {
  const uint64_t seed = exact_seed(name);
  const float coeff = (float)(scale / sqrt(21845.0)), off = (float)offset;
  const int64_t quads = (n + 3) / 4;
#pragma omp parallel for schedule(static)
  for (int64_t q = 0; q < quads; ++q) {
    uint32_t w[4];
    philox_block((uint64_t)q, seed, w);
    for (int j = 0; j < 4 && 4 * q + j < n; ++j) {
      const int sum = (int)(w[j] & 0xFF) + (int)((w[j] >> 8) & 0xFF) + (int)((w[j] >> 16) & 0xFF) + (int)(w[j] >> 24);
      volatile float v = (float)(sum - 510) * coeff; /* rounded to f32 before the add: no fused multiply-add */
      float r = v;
      if (off != 0.f) r = r + off;
      uint32_t u;
      memcpy(&u, &r, 4);
      out[4 * q + j] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
  }
}
static int ends_with(const char* s, const char* suf) {
  const size_t a = strlen(s), b = strlen(suf);
  return a >= b && !strcmp(s + a - b, suf);
}
/* the rules of synth.exact_rule("flux") */
static void flux_rule(const char* name, double* off, double* sc) {
  *off = 0.0;
  if (ends_with(name, "norm_q.weight") || ends_with(name, "norm_k.weight") || ends_with(name, "norm_added_q.weight") || ends_with(name, "norm_added_k.weight")) {
    *off = 1.0, *sc = 0.1;
  } else if (ends_with(name, ".bias")) {
    *sc = 0.02;
  } else if (strstr(name, "norm1.linear") || strstr(name, "norm1_context.linear") || strstr(name, ".norm.linear") || !strncmp(name, "norm_out.linear", 15)) {
    *sc = 0.01;
  } else {
    *sc = 0.02;
  }
}
static void vae_rule(const char* name, int rank, const int64_t* shape, double* off, double* sc) {
  *off = 0.0;
  if (rank == 2 || rank == 4) {
    int64_t fan = 1;
    for (int i = 1; i < rank; ++i) fan *= shape[i];
    *sc = 1.0 / sqrt((double)fan);
  } else if (strstr(name, "norm") && ends_with(name, ".weight")) {
    *off = 1.0, *sc = 0.1;
  } else {
    *sc = strstr(name, "norm") ? 0.1 : 0.02;
  }
}

static uint16_t* g_buf = NULL; /* one host staging buffer, grown to the largest tensor */
static int64_t g_buf_n = 0;
static uint16_t* staging(int64_t n) {
  if (n > g_buf_n) {
    free(g_buf);
    g_buf = (uint16_t*)malloc((size_t)n * 2);
    g_buf_n = n;
    if (!g_buf) {
      fprintf(stderr, "host_s1: out of host memory\n");
      exit(1);
    }
  }
  return g_buf;
}
static int64_t g_weights = 0;

static void flux_tensor(fmi_flux* m, const char* name, int rank, int64_t d0, int64_t d1) {
  const int64_t shape[2] = {d0, d1};
  const int64_t n = rank == 2 ? d0 * d1 : d0;
  double off, sc;
  flux_rule(name, &off, &sc);
  uint16_t* b = staging(n);
  exact_bf16(b, n, name, off, sc);
  CHECK(fmi_flux_set_tensor(m, name, b, FMI_BF16, shape, rank)); /* a HOST pointer: the library copies (hipMemcpyDefault) into its fused layouts */
  g_weights += n;
}
static void flux_lin(fmi_flux* m, const char* prefix, int64_t o, int64_t i) {
  char name[160];
  snprintf(name, sizeof name, "%s.weight", prefix);
  flux_tensor(m, name, 2, o, i);
  snprintf(name, sizeof name, "%s.bias", prefix);
  flux_tensor(m, name, 1, o, 0);
}
/* the diffusers names Flux::new reads (model.rs:165-772), as synth.flux_tensor_shapes enumerates them */
static void load_flux(fmi_flux* m, const fmi_flux_config* c) {
  const int64_t hd = c->axes_dim[0] + c->axes_dim[1] + c->axes_dim[2], D = c->num_attention_heads * hd, M = 4 * D;
  char p[128], q[160];
  flux_lin(m, "x_embedder", D, c->in_channels);
  flux_lin(m, "context_embedder", D, c->joint_attention_dim);
  flux_lin(m, "time_text_embed.timestep_embedder.linear_1", D, 256);
  flux_lin(m, "time_text_embed.timestep_embedder.linear_2", D, D);
  if (c->guidance_embeds) {
    flux_lin(m, "time_text_embed.guidance_embedder.linear_1", D, 256);
    flux_lin(m, "time_text_embed.guidance_embedder.linear_2", D, D);
  }
  flux_lin(m, "time_text_embed.text_embedder.linear_1", D, c->pooled_projection_dim);
  flux_lin(m, "time_text_embed.text_embedder.linear_2", D, D);
  static const char* attn_lin[] = {"to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"};
  static const char* attn_norm[] = {"norm_q", "norm_k", "norm_added_q", "norm_added_k"};
  for (int i = 0; i < c->num_layers; ++i) {
    snprintf(p, sizeof p, "transformer_blocks.%d.", i);
    snprintf(q, sizeof q, "%snorm1.linear", p), flux_lin(m, q, 6 * D, D);
    snprintf(q, sizeof q, "%snorm1_context.linear", p), flux_lin(m, q, 6 * D, D);
    for (int k = 0; k < 8; ++k) snprintf(q, sizeof q, "%sattn.%s", p, attn_lin[k]), flux_lin(m, q, D, D);
    for (int k = 0; k < 4; ++k) snprintf(q, sizeof q, "%sattn.%s.weight", p, attn_norm[k]), flux_tensor(m, q, 1, hd, 0);
    snprintf(q, sizeof q, "%sff.net.0.proj", p), flux_lin(m, q, M, D);
    snprintf(q, sizeof q, "%sff.net.2", p), flux_lin(m, q, D, M);
    snprintf(q, sizeof q, "%sff_context.net.0.proj", p), flux_lin(m, q, M, D);
    snprintf(q, sizeof q, "%sff_context.net.2", p), flux_lin(m, q, D, M);
  }
  for (int i = 0; i < c->num_single_layers; ++i) {
    snprintf(p, sizeof p, "single_transformer_blocks.%d.", i);
    snprintf(q, sizeof q, "%snorm.linear", p), flux_lin(m, q, 3 * D, D);
    snprintf(q, sizeof q, "%sattn.to_q", p), flux_lin(m, q, D, D);
    snprintf(q, sizeof q, "%sattn.to_k", p), flux_lin(m, q, D, D);
    snprintf(q, sizeof q, "%sattn.to_v", p), flux_lin(m, q, D, D);
    snprintf(q, sizeof q, "%sattn.norm_q.weight", p), flux_tensor(m, q, 1, hd, 0);
    snprintf(q, sizeof q, "%sattn.norm_k.weight", p), flux_tensor(m, q, 1, hd, 0);
    snprintf(q, sizeof q, "%sproj_mlp", p), flux_lin(m, q, M, D);
    snprintf(q, sizeof q, "%sproj_out", p), flux_lin(m, q, D, D + M);
  }
  flux_lin(m, "norm_out.linear", 2 * D, D);
  flux_lin(m, "proj_out", c->in_channels, D);
}

static void vae_tensor(fmi_vae* v, const char* name, int rank, int64_t a, int64_t b, int64_t k) {
  const int64_t shape[4] = {a, b, k, k};
  int64_t n = a;
  if (rank >= 2) n *= b;
  if (rank == 4) n *= k * k;
  double off, sc;
  vae_rule(name, rank, shape, &off, &sc);
  uint16_t* buf = staging(n);
  exact_bf16(buf, n, name, off, sc);
  CHECK(fmi_vae_set_tensor(v, name, buf, FMI_BF16, shape, rank));
  g_weights += n;
}
static void vae_conv(fmi_vae* v, const char* p, int64_t o, int64_t i, int64_t k) {
  char n[160];
  snprintf(n, sizeof n, "%s.weight", p), vae_tensor(v, n, 4, o, i, k);
  snprintf(n, sizeof n, "%s.bias", p), vae_tensor(v, n, 1, o, 0, 0);
}
static void vae_gn(fmi_vae* v, const char* p, int64_t c) {
  char n[160];
  snprintf(n, sizeof n, "%s.weight", p), vae_tensor(v, n, 1, c, 0, 0);
  snprintf(n, sizeof n, "%s.bias", p), vae_tensor(v, n, 1, c, 0, 0);
}
static void vae_resnet(fmi_vae* v, const char* p, int64_t i, int64_t o) {
  char n[160];
  snprintf(n, sizeof n, "%s.norm1", p), vae_gn(v, n, i);
  snprintf(n, sizeof n, "%s.conv1", p), vae_conv(v, n, o, i, 3);
  snprintf(n, sizeof n, "%s.norm2", p), vae_gn(v, n, o);
  snprintf(n, sizeof n, "%s.conv2", p), vae_conv(v, n, o, o, 3);
  if (i != o) snprintf(n, sizeof n, "%s.conv_shortcut", p), vae_conv(v, n, o, i, 1);
}
/* the decoder tensors (vae.rs:371-433), as synth.vae_tensor_shapes enumerates them */
static void load_vae(fmi_vae* v, const fmi_vae_config* c) {
  char p[128], n[160];
  int64_t block_in = c->block_out_channels[c->n_blocks - 1];
  vae_conv(v, "decoder.conv_in", block_in, c->latent_channels, 3);
  vae_resnet(v, "decoder.mid_block.resnets.0", block_in, block_in);
  if (c->mid_block_add_attention) {
    static const char* lin[] = {"to_q", "to_k", "to_v", "to_out.0"};
    vae_gn(v, "decoder.mid_block.attentions.0.group_norm", block_in);
    for (int k = 0; k < 4; ++k) {
      snprintf(n, sizeof n, "decoder.mid_block.attentions.0.%s.weight", lin[k]), vae_tensor(v, n, 2, block_in, block_in, 0);
      snprintf(n, sizeof n, "decoder.mid_block.attentions.0.%s.bias", lin[k]), vae_tensor(v, n, 1, block_in, 0, 0);
    }
  }
  vae_resnet(v, "decoder.mid_block.resnets.1", block_in, block_in);
  for (int lvl = 0; lvl < c->n_blocks; ++lvl) {
    const int64_t block_out = c->block_out_channels[c->n_blocks - 1 - lvl];
    for (int i = 0; i < c->layers_per_block + 1; ++i) {
      snprintf(p, sizeof p, "decoder.up_blocks.%d.resnets.%d", lvl, i);
      vae_resnet(v, p, block_in, block_out);
      block_in = block_out;
    }
    if (lvl != 3) snprintf(p, sizeof p, "decoder.up_blocks.%d.upsamplers.0.conv", lvl), vae_conv(v, p, block_in, block_in, 3);
  }
  vae_gn(v, "decoder.conv_norm_out", c->block_out_channels[0]);
  vae_conv(v, "decoder.conv_out", c->out_channels, c->block_out_channels[0], 3);
}

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}
static float bf16_bits_to_f32(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  int steps = 50, lh = 128, lw = 128, T = 512;
  double guidance = 3.5;
  const char* mode = "bf16";
  const char* dump = NULL;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--latent") && i + 1 < argc) sscanf(argv[++i], "%dx%d", &lh, &lw);
    else if (!strcmp(argv[i], "--txt") && i + 1 < argc) T = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--guidance") && i + 1 < argc) guidance = atof(argv[++i]);
    else if (!strcmp(argv[i], "--mode") && i + 1 < argc) mode = argv[++i];
    else if (!strcmp(argv[i], "--dump-u8") && i + 1 < argc) dump = argv[++i];
    else {
      fprintf(stderr, "usage: host_s1 [--steps N] [--latent HxW] [--txt T] [--guidance G] [--mode bf16|int8|fp8] [--dump-u8 file]\n");
      return 2;
    }
  }
  if (steps < 1 || lh < 2 || lw < 2 || (lh | lw) & 1 || T < 1) {
    fprintf(stderr, "host_s1: steps >= 1, an even latent size and T >= 1 required\n");
    return 2;
  }
  crc_init();
  const double t_start = now_s();
  CHECK(fmi_init(0));
  if (fmi_abi_version() < 6) {
    fprintf(stderr, "host_s1: library ABI %d < 6\n", fmi_abi_version());
    return 1;
  }
  /* ---- Pipeline::load: the models */
  fmi_flux_config fc;
  fmi_flux_default_config(&fc, 1); /* FLUX.1-dev */
  fmi_flux* flux = NULL;
  CHECK(fmi_flux_create(&fc, FMI_MODEL_BF16, &flux));
  load_flux(flux, &fc);
  if (fmi_flux_missing_count(flux)) {
    fprintf(stderr, "host_s1: %d flux tensors missing, e.g. %s\n", fmi_flux_missing_count(flux), fmi_flux_missing_name(flux, 0));
    return 1;
  }
  fmi_vae_config vc;
  fmi_vae_default_config(&vc);
  fmi_vae* vae = NULL;
  CHECK(fmi_vae_create(&vc, FMI_MODEL_BF16, &vae));
  load_vae(vae, &vc);
  if (strcmp(mode, "int8") && strcmp(mode, "fp8") && strcmp(mode, "bf16")) {
    fprintf(stderr, "host_s1: unknown mode %s\n", mode);
    return 2;
  }
  if (!strcmp(mode, "fp8")) CHECK(fmi_flux_quantize_fp8(flux, NULL));
  const double t_loaded = now_s();

  /* ---- FluxPipeline::forward for one prompt: conditioning (what T5 / CLIP would hand over), noise, schedule */
  const int C = 16, S = (lh / 2) * (lw / 2), B = 1, H = 8 * lh, W = 8 * lw;
  const int64_t n_lat = (int64_t)C * lh * lw, n_t5 = (int64_t)T * fc.joint_attention_dim, n_clip = fc.pooled_projection_dim;
  float* lat_h = (float*)malloc((size_t)n_lat * 4);
  float* clip_h = (float*)malloc((size_t)n_clip * 4);
  uint16_t* b = staging(n_lat > n_t5 ? n_lat : n_t5);
  exact_bf16(b, n_lat, "input.c2.latent", 0.0, 1.0);
  for (int64_t i = 0; i < n_lat; ++i) lat_h[i] = bf16_bits_to_f32(b[i]);
  void *lat_d, *img_d, *ids_d, *t5_d, *clip_d, *txt_ids_d, *g_d, *z_d, *image_d, *u8_d;
  CHECK(fmi_malloc(&lat_d, (size_t)n_lat * 4));
  CHECK(fmi_malloc(&img_d, (size_t)S * 64 * 4));
  CHECK(fmi_malloc(&ids_d, (size_t)S * 3 * 4));
  CHECK(fmi_malloc(&t5_d, (size_t)n_t5 * 2));
  CHECK(fmi_malloc(&clip_d, (size_t)n_clip * 4));
  CHECK(fmi_malloc(&txt_ids_d, (size_t)T * 3 * 4));
  CHECK(fmi_malloc(&g_d, 4));
  CHECK(fmi_malloc(&z_d, (size_t)n_lat * 4));
  CHECK(fmi_malloc(&image_d, (size_t)3 * H * W * 4));
  CHECK(fmi_malloc(&u8_d, (size_t)3 * H * W));
  CHECK(fmi_memcpy(lat_d, lat_h, (size_t)n_lat * 4, NULL));
  CHECK(fmi_stream_synchronize(NULL));
  exact_bf16(b, n_t5, "input.c2.t5", 0.0, 1.0);
  CHECK(fmi_memcpy(t5_d, b, (size_t)n_t5 * 2, NULL));
  CHECK(fmi_stream_synchronize(NULL));
  exact_bf16(b, n_clip, "input.c2.clip", 0.0, 1.0);
  for (int64_t i = 0; i < n_clip; ++i) clip_h[i] = bf16_bits_to_f32(b[i]);
  CHECK(fmi_memcpy(clip_d, clip_h, (size_t)n_clip * 4, NULL));
  CHECK(fmi_memset(txt_ids_d, 0, (size_t)T * 3 * 4, NULL));
  const float g_h = (float)guidance;
  CHECK(fmi_memcpy(g_d, &g_h, 4, NULL));
  CHECK(fmi_pack_latents((const float*)lat_d, B, C, lh, lw, (float*)img_d, (float*)ids_d, NULL));
  fmi_scheduler_config sc = {256, 0.5, 4096, 1.15, 3.0, 1}; /* FLUX.1-dev's scheduler_config.json */
  double* ts = (double*)malloc(sizeof(double) * (size_t)(steps + 1));
  CHECK(fmi_get_timesteps(&sc, steps, fmi_calculate_shift(S, sc.base_image_seq_len, sc.max_image_seq_len, sc.base_shift, sc.max_shift), ts));

  /* ---- the denoise loop, decode, u8 */
  fmi_flux_inputs in;
  memset(&in, 0, sizeof in);
  in.img = img_d, in.img_dtype = FMI_F32, in.img_ids = (const float*)ids_d;
  in.txt = t5_d, in.txt_dtype = FMI_BF16, in.txt_ids = (const float*)txt_ids_d;
  in.y = clip_d, in.y_dtype = FMI_F32, in.guidance = (const float*)g_d;
  in.B = B, in.S = S, in.T = T, in.ids_per_sample = 0;
  if (!strcmp(mode, "int8")) {
    /* the int8 mode is calibrated (ABI 6): four bf16 evaluations across the schedule record the per-channel maxima of every block linear's input, then the
     * block linears are quantised with the smoothing factors folded in — what Pipeline(dtype = I8) does at its first request */
    void *t_d, *pred_d;
    CHECK(fmi_malloc(&t_d, 4));
    CHECK(fmi_malloc(&pred_d, (size_t)S * 64 * 4));
    CHECK(fmi_flux_calibrate_int8(flux, 1));
    const int pts[4] = {0, (steps - 1) / 3, 2 * (steps - 1) / 3, steps - 1};
    for (int i = 0; i < 4; ++i) {
      if (i && pts[i] == pts[i - 1]) continue;
      const float t_h = (float)ts[pts[i]];
      CHECK(fmi_memcpy(t_d, &t_h, 4, NULL));
      CHECK(fmi_stream_synchronize(NULL));
      in.timesteps = (const float*)t_d;
      CHECK(fmi_flux_forward(flux, &in, (float*)pred_d, NULL));
    }
    in.timesteps = NULL;
    CHECK(fmi_flux_quantize_int8(flux, FMI_INT8_DEFAULT_MASK, NULL));
    CHECK(fmi_free(t_d));
    CHECK(fmi_free(pred_d));
  }
  void *e0, *e1, *e2;
  CHECK(fmi_event_create(&e0));
  CHECK(fmi_event_create(&e1));
  CHECK(fmi_event_create(&e2));
  CHECK(fmi_event_record(e0, NULL));
  CHECK(fmi_flux_denoise(flux, &in, (float*)img_d, ts, steps, NULL));
  CHECK(fmi_event_record(e1, NULL));
  CHECK(fmi_unpack_latents((const float*)img_d, B, C, lh, lw, fmi_vae_scale_factor(vae), fmi_vae_shift_factor(vae), (float*)z_d, NULL));
  CHECK(fmi_vae_decode(vae, (const float*)z_d, B, lh, lw, (float*)image_d, NULL));
  CHECK(fmi_postprocess_u8((const float*)image_d, B, 3, H, W, 0, (uint8_t*)u8_d, NULL));
  CHECK(fmi_event_record(e2, NULL));
  float ms_denoise = 0.f, ms_tail = 0.f;
  CHECK(fmi_event_elapsed_ms(e0, e1, &ms_denoise));
  CHECK(fmi_event_elapsed_ms(e1, e2, &ms_tail));
  uint8_t* u8_h = (uint8_t*)malloc((size_t)3 * H * W);
  float* latf_h = (float*)malloc((size_t)S * 64 * 4);
  CHECK(fmi_memcpy(u8_h, u8_d, (size_t)3 * H * W, NULL));
  CHECK(fmi_memcpy(latf_h, img_d, (size_t)S * 64 * 4, NULL));
  CHECK(fmi_stream_synchronize(NULL));
  if (dump) {
    FILE* f = fopen(dump, "wb");
    if (!f || fwrite(u8_h, 1, (size_t)3 * H * W, f) != (size_t)3 * H * W) {
      fprintf(stderr, "host_s1: cannot write %s\n", dump);
      return 1;
    }
    fclose(f);
  }
  printf("{\"host\": \"tools/host_s1.c (C99, no torch, no Python)\", \"mode\": \"%s\", \"abi\": %d, \"build_id\": \"%s\", \"weights\": %lld, \"image\": \"%dx%d\", \"S\": %d, \"T\": %d, "
         "\"steps\": %d, \"image_crc32\": \"%08x\", \"latents_crc32\": \"%08x\", \"ms_denoise\": %.2f, \"ms_per_step\": %.3f, \"ms_unpack_vae_u8\": %.2f, "
         "\"load_s\": %.1f, \"total_s\": %.1f}\n",
         mode, fmi_abi_version(), fmi_build_id(), (long long)g_weights, W, H, S, T, steps, crc32_of(u8_h, (size_t)3 * H * W), crc32_of(latf_h, (size_t)S * 64 * 4),
         ms_denoise, ms_denoise / steps, ms_tail, t_loaded - t_start, now_s() - t_start);
  fmi_flux_destroy(flux);
  fmi_vae_destroy(vae);
  return 0;
}
