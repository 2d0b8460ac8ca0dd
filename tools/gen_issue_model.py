#!/usr/bin/env python
"""Generates build/issue_model.hip: what does ONE wave per SIMD pay for VALU / LDS instructions placed between MFMAs?
For each (MFMA shape, filler instruction, k fillers per MFMA) a kernel runs a long loop of 16 x [MFMA, k fillers] with
independent registers and reports clocks per MFMA slot (s_memtime / wall).  Answers the design questions of attention_w16
(DESIGN 4.4): is a filler free while the matrix pipe is busy, and what do the different filler kinds cost."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MFMAS = {"m16": ("v_mfma_f32_16x16x32_bf16 a[{a}:{a3}], v[0:3], v[4:7], a[{a}:{a3}]", 4, 16384 * 2), "m32": ("v_mfma_f32_32x32x16_bf16 a[{a}:{a15}], v[0:3], v[4:7], a[{a}:{a15}]", 16, 32768 * 2),
         "m16v": ("v_mfma_f32_16x16x32_bf16 v[{va}:{va3}], v[0:3], v[4:7], v[{va}:{va3}]", 4, 16384 * 2),   # accumulators in architectural VGPRs
         "m32v": ("v_mfma_f32_32x32x16_bf16 v[{va}:{va15}], v[0:3], v[4:7], v[{va}:{va15}]", 16, 32768 * 2),
         "m32a": ("v_mfma_f32_32x32x16_bf16 a[{a}:{a15}], v[0:3], a[240:243], a[{a}:{a15}]", 16, 32768 * 2),  # B operand in AGPRs too
         "none": ("", 4, 0)}
FILL = {
    "mov": "v_mov_b32 v{d}, v{s}",
    "exp": "v_exp_f32 v{d}, v{s}",
    "max3": "v_max3_f32 v{d}, v{s}, v{s1}, v{s2}",
    "max3chain": "v_max3_f32 v40, v40, v{s1}, v{s2}",
    "pkadd": "v_pk_add_f32 v[{de}:{de1}], v[{de}:{de1}], v[{se}:{se1}]",
    "cvtpk": "v_cvt_pk_bf16_f32 v{d}, v{s}, v{s1}",
    "dsread": "ds_read_b128 v[{q}:{q3}], v8 offset:{off}",
    "add": "v_add_f32 v{d}, v{s}, v{s1}",
    "fma": "v_fma_f32 v{d}, v{s}, v{s1}, v{s2}",
}
KS = [0, 1, 2, 3, 4, 6, 8]


def body(m, f, k):
    mtxt, nacc, _ = MFMAS[m]
    out = []
    cnt = 0
    if f == "dsread":  # v8 = lane * 16: a conflict-free linear read
        out += ["v_mbcnt_lo_u32_b32 v8, -1, 0", "v_mbcnt_hi_u32_b32 v8, -1, v8", "v_lshlrev_b32 v8, 4, v8"]
    for j in range(16):
        a = (j % (64 // nacc)) * nacc
        if mtxt:
            va = 120 + (j % (64 // nacc)) * nacc
            out.append(mtxt.format(a=a, a3=a + 3, a15=a + 15, va=va, va3=va + 3, va15=va + 15))
        for _ in range(k):
            d = 48 + (cnt % 32)
            s = 16 + (cnt % 16)
            de = 48 + 2 * (cnt % 16)
            se = 16 + 2 * (cnt % 8)
            q = 80 + 4 * (cnt % 8)
            out.append(FILL[f].format(d=d, s=s, s1=s + 1, s2=s + 2, de=de, de1=de + 1, se=se, se1=se + 1, q=q, q3=q + 3, off=(cnt % 16) * 1024))
            cnt += 1
        if f == "dsread" and k and j % 4 == 3:
            out.append("s_waitcnt lgkmcnt(4)")
    return out


def main():
    src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '#include <string>',
           '__global__ __launch_bounds__(256, 1) void dummy() {}']
    names = []
    for m in MFMAS:
        for f in FILL:
            if m in ("m16v", "m32v", "m32a") and f not in ("mov", "exp", "max3", "dsread"):
                continue
            for k in KS:
                if m == "none" and k == 0:
                    continue
                if k == 0 and f != "mov":
                    continue
                name = f"k_{m}_{f}_{k}"
                names.append((name, m, f, k))
                lines = body(m, f, k)
                asm = " \\\n".join('      "' + ln + '\\n\\t"' for ln in lines)
                src.append(f'''__global__ __launch_bounds__(256, 1) void {name}(int iters, float* out) {{
  __shared__ char smem[65536];
  if (iters < 0) out[0] = smem[threadIdx.x];
  for (int it = 0; it < iters; ++it) {{
    asm volatile(
{asm}
      ::: "memory", "v0","v1","v2","v3","v4","v5","v6","v7","v8","v40");
  }}
}}''')
    src.append('struct K { const char* name; void (*fn)(int, float*); int nm; int k; double flop; };')
    src.append('int main() {\n  float* out; hipMalloc((void**)&out, 1024);\n  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);\n  K ks[] = {')
    for (name, m, f, k) in names:
        src.append(f'    {{"{m} {f} {k}", {name}, {16 if m != "none" else 0}, {k}, {float(MFMAS[m][2] * 16)}}},')
    src.append('''  };
  const int iters = 4000;
  int dev = 0; hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.fn, dim3(256), dim3(256), 0, 0, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k.fn, dim3(256), dim3(256), 0, 0, iters, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double ns_slot = ms * 1e6 / iters / 16;
    printf("%-22s  %7.2f ns per slot (MFMA + %d fillers)   %7.1f TF\\n", k.name, ns_slot, k.k, k.flop * iters * 1024 / (ms * 1e-3) / 1e12);
  }
  return 0;
}''')
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    with open(os.path.join(ROOT, "build", "issue_model.hip"), "w") as fh:
        fh.write("\n".join(src) + "\n")


if __name__ == "__main__":
    main()
