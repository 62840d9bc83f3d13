#!/usr/bin/env python3
"""Generator of tools/microbench/pk_hazard.hip: does gfx950 need more wait states between a producer and a PACKED-fp32 consumer
(v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) than between the same producer and a plain VALU consumer?  (profiles/r03_sin_cliff.md
section 5: the level-2 fault disappears when the library is built without packed fp32 instructions.)

Every test is ONE inline-asm block on hard-coded registers v100..v131 (the compiler's hazard recognizer does not look inside inline
asm, so the distance between producer and consumer is exactly the k wait states written here):

    inputs -> v100.. ; settle ; [an independent MFMA in flight] ; PRODUCER ; k wait states ; CONSUMER ; settle ; read v108, v109

and the same block with 48 wait states in the middle is the expected value; results are compared bit for bit in the kernel.
Contexts: 1 / 2 / 3 waves per SIMD, and the waves beyond the first per SIMD either run the test too or a background loop
(MFMA, v_sin_f32, v_pk_fma_f32).

usage: python tools/microbench/gen_pk_hazard.py > tools/microbench/pk_hazard.hip
       hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/microbench/pk_hazard.hip -o tools/microbench/pk_hazard
"""
MF = "v_mfma_f32_16x16x32_f16"
PRODUCERS = {   # write v102, v103 (mfma: v[102:105])
    "sin": ["v_sin_f32 v102, v100", "v_sin_f32 v103, v101"],
    "exp": ["v_exp_f32 v102, v100", "v_exp_f32 v103, v101"],
    "rcp": ["v_rcp_f32 v102, v100", "v_rcp_f32 v103, v101"],
    "fma": ["v_fma_f32 v102, v100, v128, v130", "v_fma_f32 v103, v101, v129, v131"],
    "cvt": ["v_cvt_f32_f16 v102, v100", "v_cvt_f32_f16 v103, v101"],
    "pk": ["v_pk_mul_f32 v[102:103], v[100:101], v[130:131]"],
    "pkfma": ["v_pk_fma_f32 v[102:103], v[100:101], v[130:131], v[128:129]"],
    "mfma": [f"{MF} v[102:105], v[112:115], v[116:119], v[124:127]"],
    # WAR: the MFMA READS v[102:105] (B operand), the consumer overwrites v[102:103]
    "mfma_rdB": [f"{MF} v[108:111], v[112:115], v[102:105], v[124:127]"],
    # WAW: the MFMA writes v[108:111], the consumer overwrites v[108:109]
    "mfma_wr": [f"{MF} v[108:111], v[112:115], v[116:119], v[124:127]"],
}
CONSUMERS = {   # read v[102:103], write v108, v109
    "pk_mul": ["v_pk_mul_f32 v[108:109], v[102:103], v[128:129]"],
    "pk_fma": ["v_pk_fma_f32 v[108:109], v[102:103], v[128:129], v[130:131]"],
    "pk_add": ["v_pk_add_f32 v[108:109], v[102:103], v[128:129]"],
    "mul": ["v_mul_f32 v108, v102, v128", "v_mul_f32 v109, v103, v129"],
    "cvtpk": ["v_cvt_pkrtz_f16_f32 v108, v102, v103", "v_mov_b32 v109, v108"],
    "sin": ["v_sin_f32 v108, v102", "v_sin_f32 v109, v103"],
    "mfmaB": [f"{MF} v[108:111], v[112:115], v[102:105], v[124:127]"],
    # LDS instructions that read the pair straight away (the level-2 epilogue has ds_bpermute_b32 right behind v_pk_fma_f32)
    "bperm": ["ds_bpermute_b32 v108, v106, v102", "ds_bpermute_b32 v109, v106, v103", "s_waitcnt lgkmcnt(0)"],
    "bperm_hi": ["ds_bpermute_b32 v109, v106, v103", "s_waitcnt lgkmcnt(0)", "v_mov_b32 v108, v109"],
    "dswrite": ["ds_write_b64 v107, v[102:103]", "s_waitcnt lgkmcnt(0)", "ds_read_b64 v[108:109], v107", "s_waitcnt lgkmcnt(0)"],
    "gstore": ["global_store_dwordx2 v[110:111], v[102:103], off", "s_waitcnt vmcnt(0)", "global_load_dwordx2 v[108:109], v[110:111], off sc0 sc1",
               "s_waitcnt vmcnt(0)"],
    "pk_wr": ["v_pk_mul_f32 v[102:103], v[128:129], v[130:131]"],       # WAR partner of mfma_rdB
    "mov_wr": ["v_mov_b32 v102, v128", "v_mov_b32 v103, v129"],         # WAR control
    "pk_ovw": ["v_pk_mul_f32 v[108:109], v[128:129], v[130:131]"],      # WAW partner of mfma_wr
    "mul_ovw": ["v_mul_f32 v108, v128, v130", "v_mul_f32 v109, v129, v131"],   # WAW control
}
VALU_P = ["sin", "exp", "rcp", "fma", "cvt", "pk"]
VALU_C = ["pk_mul", "pk_fma", "pk_add", "mul", "cvtpk", "sin", "mfmaB"]
K_VALU = [0, 1, 2, 3, 4]
K_MFMA = [0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20]
TESTS = [(p, c, k) for p in VALU_P for c in VALU_C for k in K_VALU]
TESTS += [(p, c, k) for p in ("pk", "pkfma", "fma", "sin") for c in ("bperm", "bperm_hi", "dswrite", "gstore") for k in K_VALU]
TESTS += [("mfma", c, k) for c in ("pk_mul", "pk_fma", "mul") for k in K_MFMA]
TESTS += [("mfma_rdB", c, k) for c in ("pk_wr", "mov_wr") for k in K_MFMA]
TESTS += [("mfma_wr", c, k) for c in ("pk_ovw", "mul_ovw") for k in K_MFMA]


# ---- suite 2: the sequences of the faulty kernel (level2_16p_kernel<8,32,2> built with packed ops), fillers that are INSTRUCTIONS (the
# compiler counts any instruction between producer and consumer as a wait state; the first suite only used s_nop) and a busy matrix pipe
SDWA = "v_cvt_f32_f16_sdwa v103, v100 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
PRODUCERS2 = {
    "sdwa": ["v_cvt_f32_f16_e32 v102, v100", SDWA],
    "sdwa_only": [SDWA],
    "rcp2": ["v_rcp_f32 v102, v100", "v_rcp_f32 v103, v101"],
    "sin8_first": ["v_sin_f32 v102, v100", "v_sin_f32 v103, v101"] + [f"v_sin_f32 v{120 + (i & 3)}, v{128 + (i & 3)}" for i in range(6)],
    "sin8_last": [f"v_sin_f32 v{120 + (i & 3)}, v{128 + (i & 3)}" for i in range(6)] + ["v_sin_f32 v102, v100", "v_sin_f32 v103, v101"],
    "sigmoid": ["v_exp_f32 v102, v100", "v_exp_f32 v103, v101", "v_add_f32 v102, 1.0, v102", "v_rcp_f32 v102, v102", "v_add_f32 v103, 1.0, v103",
                "v_rcp_f32 v103, v103"],
}
FILLERS2 = {"nop": "s_nop 0", "valu": "v_mul_f32 v120, v128, v129", "pk": "v_pk_mul_f32 v[120:121], v[128:129], v[130:131]",
            "u32": "v_add_u32 v122, v128, v129"}
CONSUMERS2 = {
    "pk_add_neg": ["v_pk_add_f32 v[108:109], v[128:129], v[102:103] neg_lo:[0,1] neg_hi:[0,1]"],
    "pk_mul": ["v_pk_mul_f32 v[108:109], v[128:129], v[102:103]"],
    "mul": ["v_mul_f32 v108, v102, v128", "v_mul_f32 v109, v103, v129"],
    "cvtpk950": ["v_cvt_pk_f16_f32 v108, v102, v103", "v_mov_b32 v109, v108"],
}
TESTS2 = [(p, f, k, c) for p in PRODUCERS2 for f in FILLERS2 for k in (0, 1, 2, 3) for c in CONSUMERS2 if not (k == 0 and f != "nop")]


def block2(p, f, k, c, pre, safe):
    ins = prologue()
    for _ in range(pre):
        ins.append(f"{MF} v[124:127], v[112:115], v[116:119], v[124:127]")
    ins += PRODUCERS2[p] + [FILLERS2[f]] * k + (waits(48) if safe else []) + CONSUMERS2[c] + waits(48)
    ins += ["v_mov_b32 %0, v108", "v_mov_b32 %1, v109"]
    return "\\n\\t".join(ins)


def waits(k):
    out = []
    while k > 0:
        n = min(k, 16)
        out.append(f"s_nop {n - 1}")
        k -= n
    return out


def prologue():
    ins = ["v_mov_b32 v100, %2", "v_mov_b32 v101, %3", "v_mov_b32 v128, %4", "v_mov_b32 v129, %5", "v_mov_b32 v130, %6", "v_mov_b32 v131, %7"]
    for r in range(112, 120):
        ins.append(f"v_mov_b32 v{r}, %{8 + (r & 1)}")
    for r in (120, 121, 122, 123):
        ins.append(f"v_mov_b32 v{r}, %4")
    for r, s in ((124, 2), (125, 3), (126, 4), (127, 5)):
        ins.append(f"v_mov_b32 v{r}, %{s}")
    ins += ["v_mov_b32 v102, %8", "v_mov_b32 v103, %9", "v_mov_b32 v104, %9", "v_mov_b32 v105, %8", "v_mov_b32 v108, %2", "v_mov_b32 v109, %3",
            "v_mov_b32 v106, %10", "v_mov_b32 v107, %11", "v_mov_b32 v110, %12", "v_mov_b32 v111, %13"]
    ins += waits(32)
    return ins


def block(p, c, k, pre):
    ins = prologue()
    if pre:
        ins.append(f"{MF} v[120:123], v[112:115], v[116:119], v[120:123]")
    ins += PRODUCERS[p] + waits(k) + CONSUMERS[c] + waits(48)
    ins += ["v_mov_b32 %0, v108", "v_mov_b32 %1, v109"]
    return "\\n\\t".join(ins)


def main():
    print("// GENERATED by tools/microbench/gen_pk_hazard.py - do not edit")
    print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstring>\n#include <vector>\n#include <string>")
    clob = ", ".join(f'"v{r}"' for r in range(100, 132))
    print(f"#define CLOB {clob}")
    print(r"""
__device__ __forceinline__ float urand(unsigned& s) { s = s * 1664525u + 1013904223u; return 0.1f + 0.8f * (float)(s >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ unsigned halves(float a, float b) {
  const _Float16 x = (_Float16)a, y = (_Float16)b; unsigned short ux, uy; __builtin_memcpy(&ux, &x, 2); __builtin_memcpy(&uy, &y, 2); return ux | ((unsigned)uy << 16); }
__device__ void background(int kind, int iters, unsigned* bad) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  h8 a, b; for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (threadIdx.x & 15)); b[j] = (_Float16)0.5f; }
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0; float s = 0.001f * threadIdx.x, t = s; float2 q = {s, t};
  for (int i = 0; i < iters * 3; ++i) {
    if (kind == 1) {
      for (int j = 0; j < 4; ++j) { c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0); }
    } else if (kind == 2) {
      for (int j = 0; j < 16; ++j) { s = __builtin_amdgcn_sinf(s + 0.25f); t = __builtin_amdgcn_sinf(t + 0.125f); }
    } else {
      for (int j = 0; j < 16; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q) : "v"(q));
    }
  }
  if (c0[0] + c1[1] + c2[2] + c3[3] + s + t + q.x == 12345.678f) atomicAdd(bad, 1u);
}
#define TEST(NAME, SEQ, SAFE)                                                                                        \
  __global__ void __launch_bounds__(768) NAME(unsigned* bad, int iters, int bg, unsigned* scratch) {                                     \
    __shared__ unsigned lds_buf[768 * 2];                                                                            \
    const int wave = threadIdx.x >> 6;                                                                               \
    if (iters < 0) lds_buf[threadIdx.x] = 0;                                                                         \
    if (bg && wave >= 4) { background(bg, iters, bad); return; }                                                          \
    unsigned s = (blockIdx.x * 768u + threadIdx.x) * 2654435761u + 12345u, nbad = 0;                                \
    for (int it = 0; it < iters; ++it) {                                                                             \
      const float x0 = urand(s), x1 = urand(s), y0 = urand(s), y1 = urand(s), z0 = urand(s), z1 = urand(s);          \
      const unsigned h0 = halves(x0, y0), h1 = halves(x1, z1);                                                       \
      unsigned r0, r1, e0, e1;                                                                                       \
      const unsigned la = ((threadIdx.x + 17) & 63) * 4, lb = threadIdx.x * 8;                                       \
      unsigned long long gaddr = (unsigned long long)(scratch + (blockIdx.x * 768u + threadIdx.x) * 2);              \
      const unsigned ga = (unsigned)gaddr, gb = (unsigned)(gaddr >> 32);                                             \
      asm volatile(SEQ : "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(z0), "v"(z1), "v"(h0), "v"(h1), "v"(la), "v"(lb), "v"(ga), "v"(gb) : CLOB, "memory");  \
      asm volatile(SAFE : "=&v"(e0), "=&v"(e1) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "v"(z0), "v"(z1), "v"(h0), "v"(h1), "v"(la), "v"(lb), "v"(ga), "v"(gb) : CLOB, "memory"); \
      nbad += (r0 != e0) || (r1 != e1);                                                                              \
    }                                                                                                                \
    if (nbad) atomicAdd(bad, nbad);                                                                                  \
  }
""")
    names = []
    for pre in (0, 1):
        for p, c, k in TESTS:
            name = f"t_{p}__{c}__k{k}__pre{pre}"
            names.append((name, p, c, k, pre))
            print(f'TEST({name}, "{block(p, c, k, pre)}", "{block(p, c, 48, pre)}")')
    for pre in (0, 1, 4):
        for p, f, k, c in TESTS2:
            name = f"u_{p}__{f}{k}__{c}__pre{pre}"
            names.append((name, f"{p}+{k}{f}", c, k, pre))
            print(f'TEST({name}, "{block2(p, f, k, c, pre, False)}", "{block2(p, f, k, c, pre, True)}")')
    print("typedef void (*kern_t)(unsigned*, int, int, unsigned*);")
    print("struct Entry { const char *p, *c; int k, pre; kern_t fn; };")
    print("static const Entry kTests[] = {")
    for name, p, c, k, pre in names:
        print(f'  {{"{p}", "{c}", {k}, {pre}, {name}}},')
    print("};")
    print(r"""
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  unsigned *bad, *scratch; if (hipMalloc(&bad, 4) != hipSuccess || hipMalloc(&scratch, 256 * 768 * 8) != hipSuccess) return 1;
  const int n = sizeof(kTests) / sizeof(kTests[0]);
  // contexts: waves per SIMD 1..3; bg 0 = every wave runs the test, 1/2/3 = waves beyond the first per SIMD run MFMA / v_sin / v_pk_fma loops
  const int ctx_w[] = {1, 2, 3, 2, 3, 2, 3, 2, 3}, ctx_bg[] = {0, 0, 0, 1, 1, 2, 2, 3, 3};
  const int nctx = 9;
  printf("# producer consumer pre k : failing evaluations per context (w1 w2 w3 w2+mfma w3+mfma w2+sin w3+sin w2+pk w3+pk), %d iterations x 256 workgroups\n", iters);
  for (int i = 0; i < n; ++i) {
    const Entry& e = kTests[i];
    unsigned res[9]; unsigned long tot = 0;
    for (int c = 0; c < nctx; ++c) {
      (void)hipMemset(bad, 0, 4);
      hipLaunchKernelGGL(e.fn, dim3(256), dim3(256 * ctx_w[c]), 0, 0, bad, iters, ctx_bg[c], scratch);
      if (hipMemcpy(&res[c], bad, 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed %s\n", hipGetErrorString(hipGetLastError())); return 2; }
      tot += res[c];
    }
    printf("%-16s %-10s pre%d k%-2d :", e.p, e.c, e.pre, e.k);
    for (int c = 0; c < nctx; ++c) printf(" %u", res[c]);
    printf("%s\n", tot ? "   FAIL" : "");
  }
  return 0;
}
""")


if __name__ == "__main__":
    main()
