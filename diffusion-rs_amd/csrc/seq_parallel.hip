// seq_parallel.hip — the data movement of single-image sequence parallelism.
//
// The reference runs one image on one device (pipelines/mod.rs:214-217, "This will need to be updated!").  Here the tokens of ONE
// image are sharded over the N ranks of a sequence-parallel group (rank r holds txt tokens [r*Tl, (r+1)*Tl) and img tokens
// [r*Sl, (r+1)*Sl)); everything per-token runs on the local rows, and the joint attention (model.rs:540-552) is made head-local by
// two all-to-alls per block:
//     (local tokens, all H heads)  --all-to-all-->  (all L tokens, H/N heads)  --attention-->  --all-to-all-->  (local tokens, H)
// The kernels below only pack / unpack the exchange buffers; the collective itself is the caller's (RCCL all_to_all over xGMI,
// handed in as a callback through fmi_flux_set_sequence_parallel).  All of them move bf16 bits unchanged, so a sequence-parallel
// forward is bit-identical to the single-device one wherever the per-token kernels are.
//
// Layouts (B = 1):
//   local   Qh, Kh (H, Ll, 128)        Vt (H, 128, Lpl)   Ll = Tl + Sl local tokens [txt | img], Lpl = Ll rounded up to 64,
//                                                          Vt's token axis permuted inside groups of 16 (attention.hip vt_perm)
//   send 1  per destination p: [ q (Hr, Ll, 128) | k (Hr, Ll, 128) | vt (Hr, 128, Lpl) ]   heads p*Hr .. (p+1)*Hr
//   full    Qf, Kf (Hr, L, 128)        Vtf (Hr, 128, Lp)  joint order [txt of rank 0..N-1 | img of rank 0..N-1], L = N * Ll
//   O       (L, Hr*128) token-major attention output of this rank's heads
//   send 2  per destination p: (Ll, Hr*128) = p's tokens;   unpacked into columns [src*Hr*128, ...) of the local output rows
#include "common.h"

namespace fmi {
namespace {

__device__ __forceinline__ int sp_vt_perm(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
// local token t of rank p -> position in the joint sequence
__device__ __forceinline__ int sp_joint(int p, int t, int Tl, int Sl, int N) { return t < Tl ? p * Tl + t : N * Tl + p * Sl + (t - Tl); }

using chunk_t = uint4;  // 8 bf16

__global__ void sp_pack_qkv_kernel(const chunk_t* __restrict__ q, const chunk_t* __restrict__ k, const chunk_t* __restrict__ vt,
                                   chunk_t* __restrict__ send, int64_t cq, int64_t cv, int N) {
  const int64_t per = 2 * cq + cv, total = per * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i / per);
    const int64_t o = i - (int64_t)p * per;
    send[i] = o < cq ? q[(int64_t)p * cq + o] : o < 2 * cq ? k[(int64_t)p * cq + (o - cq)] : vt[(int64_t)p * cv + (o - 2 * cq)];
  }
}

// q, k: one 16-byte chunk per thread
__global__ void sp_unpack_qk_kernel(const chunk_t* __restrict__ recv, chunk_t* __restrict__ qf, chunk_t* __restrict__ kf, int Hr,
                                    int Tl, int Sl, int N, int64_t cq, int64_t cv) {
  const int Ll = Tl + Sl, L = N * Ll;
  const int64_t per = 2 * cq + cv, total = (int64_t)N * Hr * Ll * 16;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d8 = (int)(i & 15);
    int64_t r = i >> 4;
    const int t = (int)(r % Ll);
    r /= Ll;
    const int h = (int)(r % Hr), p = (int)(r / Hr);
    const int64_t src = (int64_t)p * per + ((int64_t)h * Ll + t) * 16 + d8;
    const int64_t dst = ((int64_t)h * L + sp_joint(p, t, Tl, Sl, N)) * 16 + d8;
    qf[dst] = recv[src];
    kf[dst] = recv[src + cq];
  }
}

// vt: one element per thread over the FULL padded layout (positions whose token is past L are zero)
__global__ void sp_unpack_vt_kernel(const bf16_t* __restrict__ recv, bf16_t* __restrict__ vtf, int Hr, int Tl, int Sl, int N, int Lpl,
                                    int Lp, int64_t cq, int64_t cv) {
  const int Ll = Tl + Sl, L = N * Ll, T = N * Tl;
  const int64_t per = (2 * cq + cv) * 8, total = (int64_t)Hr * 128 * Lp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int pos = (int)(i % Lp);
    const int64_t hd = i / Lp;  // h * 128 + d
    const int kv = sp_vt_perm(pos);
    bf16_t v = 0;
    if (kv < L) {
      int p, t;
      if (kv < T) p = kv / Tl, t = kv % Tl;
      else p = (kv - T) / Sl, t = Tl + (kv - T) % Sl;
      v = recv[(int64_t)p * per + 2 * cq * 8 + hd * Lpl + sp_vt_perm(t)];
    }
    vtf[i] = v;
  }
}

__global__ void sp_pack_o_kernel(const chunk_t* __restrict__ o, chunk_t* __restrict__ send, int Hr, int Tl, int Sl, int N) {
  const int Ll = Tl + Sl, rc = Hr * 16;  // chunks per row
  const int64_t total = (int64_t)N * Ll * rc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % rc);
    const int64_t r = i / rc;
    const int t = (int)(r % Ll), p = (int)(r / Ll);
    send[i] = o[(int64_t)sp_joint(p, t, Tl, Sl, N) * rc + c];
  }
}

__global__ void sp_unpack_o_kernel(const chunk_t* __restrict__ recv, AttnOut out, int Hr, int Ll, int N) {
  const int rc = Hr * 16;
  const int64_t total = (int64_t)N * Ll * rc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % rc);
    const int64_t r = i / rc;
    const int t = (int)(r % Ll), p = (int)(r / Ll);
    bf16_t* row = t < out.rows0 ? out.p0 + (int64_t)t * out.ld0 : out.p1 + (int64_t)(t - out.rows0) * out.ld1;
    reinterpret_cast<chunk_t*>(row + (int64_t)p * Hr * 128)[c] = recv[i];
  }
}

// one thread per (token, head, 8-wide d chunk)
__global__ void sp_merge_splits_kernel(const chunk_t* __restrict__ parts, const float* __restrict__ lse, int S, chunk_t* __restrict__ out, int heads, int L) {
  const int rc = heads * 16;
  const int64_t total = (int64_t)L * rc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % rc), t = (int)(i / rc), h = c >> 4;
    float ls[8], mx = -3.0e38f;
    for (int s = 0; s < S; ++s) ls[s] = lse[((int64_t)s * heads + h) * L + t], mx = fmaxf(mx, ls[s]);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, den = 0.f;
    for (int s = 0; s < S; ++s) {  // fixed order
      const float w = exp2f(ls[s] - mx);
      const chunk_t v = parts[(int64_t)s * total + i];
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += w * __uint_as_float(u[e] << 16);
        acc[2 * e + 1] += w * __uint_as_float(u[e] & 0xffff0000u);
      }
      den += w;
    }
    const float inv = 1.0f / den;
    chunk_t o;
    o.x = pack_bf16x2(acc[0] * inv, acc[1] * inv), o.y = pack_bf16x2(acc[2] * inv, acc[3] * inv);
    o.z = pack_bf16x2(acc[4] * inv, acc[5] * inv), o.w = pack_bf16x2(acc[6] * inv, acc[7] * inv);
    out[i] = o;
  }
}

int grid_for(int64_t total) { return (int)std::min<int64_t>((total + 255) / 256, 4096); }

}  // namespace

size_t sp_qkv_bytes_per_peer(int Hr, int Ll) { return ((size_t)2 * Hr * Ll * 128 + (size_t)Hr * 128 * ((Ll + 63) / 64 * 64)) * 2; }
size_t sp_o_bytes_per_peer(int Hr, int Ll) { return (size_t)Ll * Hr * 128 * 2; }

int launch_sp_pack_qkv(const bf16_t* q, const bf16_t* k, const bf16_t* vt, void* send, int H, int Tl, int Sl, int N, hipStream_t s) {
  const int Hr = H / N, Ll = Tl + Sl, Lpl = (Ll + 63) / 64 * 64;
  const int64_t cq = (int64_t)Hr * Ll * 16, cv = (int64_t)Hr * 128 * Lpl / 8;
  sp_pack_qkv_kernel<<<grid_for((2 * cq + cv) * N), 256, 0, s>>>(reinterpret_cast<const chunk_t*>(q), reinterpret_cast<const chunk_t*>(k),
                                                                 reinterpret_cast<const chunk_t*>(vt), reinterpret_cast<chunk_t*>(send), cq, cv, N);
  FMI_HIP_TRY(hipGetLastError());
  return FMI_OK;
}
int launch_sp_unpack_qkv(const void* recv, bf16_t* qf, bf16_t* kf, bf16_t* vtf, int H, int Tl, int Sl, int N, hipStream_t s) {
  const int Hr = H / N, Ll = Tl + Sl, Lpl = (Ll + 63) / 64 * 64, Lp = (N * Ll + 63) / 64 * 64;
  const int64_t cq = (int64_t)Hr * Ll * 16, cv = (int64_t)Hr * 128 * Lpl / 8;
  sp_unpack_qk_kernel<<<grid_for((int64_t)N * Hr * Ll * 16), 256, 0, s>>>(reinterpret_cast<const chunk_t*>(recv), reinterpret_cast<chunk_t*>(qf),
                                                                          reinterpret_cast<chunk_t*>(kf), Hr, Tl, Sl, N, cq, cv);
  FMI_HIP_TRY(hipGetLastError());
  sp_unpack_vt_kernel<<<grid_for((int64_t)Hr * 128 * Lp), 256, 0, s>>>(reinterpret_cast<const bf16_t*>(recv), vtf, Hr, Tl, Sl, N, Lpl, Lp, cq, cv);
  FMI_HIP_TRY(hipGetLastError());
  return FMI_OK;
}
int launch_sp_pack_o(const bf16_t* o, void* send, int H, int Tl, int Sl, int N, hipStream_t s) {
  const int Hr = H / N;
  sp_pack_o_kernel<<<grid_for((int64_t)N * (Tl + Sl) * Hr * 16), 256, 0, s>>>(reinterpret_cast<const chunk_t*>(o), reinterpret_cast<chunk_t*>(send), Hr, Tl, Sl, N);
  FMI_HIP_TRY(hipGetLastError());
  return FMI_OK;
}
// out: where the local rows go (rows [0, rows0) -> p0, the rest -> p1), full width H*128
int launch_sp_unpack_o(const void* recv, const AttnOut& out, int H, int Ll, int N, hipStream_t s) {
  const int Hr = H / N;
  sp_unpack_o_kernel<<<grid_for((int64_t)N * Ll * Hr * 16), 256, 0, s>>>(reinterpret_cast<const chunk_t*>(recv), out, Hr, Ll, N);
  FMI_HIP_TRY(hipGetLastError());
  return FMI_OK;
}

int launch_sp_merge_splits(const bf16_t* parts, const float* lse, int S, bf16_t* out, int heads, int L, hipStream_t s) {
  if (S < 2 || S > 8) return fail(FMI_ERR_INVALID, "sp_merge_splits: 2..8 parts");
  sp_merge_splits_kernel<<<grid_for((int64_t)L * heads * 16), 256, 0, s>>>(reinterpret_cast<const chunk_t*>(parts), lse, S, reinterpret_cast<chunk_t*>(out), heads, L);
  FMI_HIP_TRY(hipGetLastError());
  return FMI_OK;
}

}  // namespace fmi
