#!/usr/bin/env python3
"""Generator of profiles/tools/micro/mfma_shadow: how many instructions hide behind an asm v_mfma_f32_32x32x16_bf16?

Settles the contradiction VERDICT r01 names: DESIGN 4.2 ("no free issue capacity in the matrix pipe's shadow")
against MI355X_MICROARCH.md:447 ("<= 5 single-issue fillers hide per MFMA gap, one wave per SIMD").
Every variant is ONE hand-placed asm body of 32 MFMAs (the training kernel's exact form: inline asm, literal
AGPR or compiler-allocated VGPR accumulators, NCH interleaved chains) with F fillers of one kind behind each
MFMA; the kernel loops over the body and reports shader cycles per MFMA (s_memtime, wave 0 of block 0) and wall
ns per MFMA, at one and two waves per SIMD.

    python3 gen_mfma_shadow.py > /tmp/mfma_shadow.hip && hipcc --offload-arch=gfx950 -O3 /tmp/mfma_shadow.hip -o /tmp/mfma_shadow
"""
import sys

N_MFMA = 32
FILL_REGS = 8

def filler(kind, n, idx):
    """n fillers of `kind`; idx = running counter (register rotation).  %[f0]..%[f7] scratch VGPRs,
    %[la] an LDS byte address (lane * 16), %[d0]..%[d3] 4-dword LDS destinations."""
    out = []
    for q in range(n):
        r = (idx + q) % FILL_REGS
        if kind == "add":
            out.append(f"v_add_u32 %[f{r}], %[f{r}], 1")
        elif kind == "fma":
            out.append(f"v_fma_f32 %[f{r}], %[f{r}], %[f{r}], %[f{r}]")
        elif kind == "sin":
            out.append(f"v_sin_f32 %[f{r}], %[f{r}]")
        elif kind == "cvt":
            out.append(f"v_cvt_pk_bf16_f32 %[f{r}], %[f{r}], %[f{(r + 1) % FILL_REGS}]")
        elif kind == "dot2c":
            out.append(f"v_dot2c_f32_bf16 %[f{r}], %[f{(r + 1) % FILL_REGS}], %[f{(r + 2) % FILL_REGS}]")
        elif kind == "act":  # the forward activation's stream: fract fract cvt_pk_f16 sin sin, repeated
            seq = ["v_fract_f32 %[f{a}], %[f{a}]", "v_fract_f32 %[f{b}], %[f{b}]", "v_cvt_pk_f16_f32 %[f{c}], %[f{a}], %[f{b}]",
                   "v_sin_f32 %[f{d}], %[f{a}]", "v_sin_f32 %[f{e}], %[f{b}]"]
            k = (idx + q) % 5
            base = ((idx + q) // 5 * 2) % 4
            out.append(seq[k].format(a=base, b=base + 1, c=4 + base // 2, d=6, e=7))
        elif kind == "ldsr":  # ds_read_b128 into scratch (never waited for inside the body)
            out.append(f"ds_read_b128 %[d{(idx + q) % 4}], %[la] offset:{((idx + q) % 16) * 1024}")
        elif kind == "ldstr":
            out.append(f"ds_read_b64_tr_b16 %[e{(idx + q) % 4}], %[la] offset:{((idx + q) % 16) * 1024}")
        elif kind == "salu":
            out.append(f"s_add_u32 %[s0], %[s0], 1")
        else:
            raise ValueError(kind)
    return out

def body(acc, nch, kind, F, lds_a=0):
    """32 MFMAs round-robin over nch chains.  lds_a = D > 0: the A operand of MFMA m comes from a ds_read_b128 issued
    D MFMAs earlier into a ring of D+1 register sets (s_waitcnt lgkmcnt counted), as the training kernel's GEMMs do."""
    lines = []
    idx = 0
    if lds_a:
        for d in range(lds_a):
            lines.append(f"ds_read_b128 %[d{d % 4}], %[la] offset:{d * 1024}")
    for m in range(N_MFMA):
        c = m % nch
        if acc == "a":
            accs = f"a[{16 * c}:{16 * c + 15}]"
        else:
            accs = f"%[c{c}]"
        if lds_a:
            nxt = m + lds_a
            if nxt < N_MFMA:
                lines.append(f"ds_read_b128 %[d{nxt % 4}], %[la] offset:{(nxt % 16) * 1024}")
                lines.append(f"s_waitcnt lgkmcnt({lds_a})")
            else:
                lines.append(f"s_waitcnt lgkmcnt({N_MFMA - 1 - m})")
            aop = f"%[d{m % 4}]"
        else:
            aop = "%[a]"
        lines.append(f"v_mfma_f32_32x32x16_bf16 {accs}, {aop}, %[b], {accs}")
        lines += filler(kind, F, idx)
        idx += F
    if lds_a == 0 and kind in ("ldsr", "ldstr"):
        lines.append("s_waitcnt lgkmcnt(0)")
    lines.append("s_nop 11")
    return "\\n\\t".join(lines)

def kernel(name, acc, nch, kind, F, lds_a=0):
    agpr_clob = ", ".join(f'"a{i}"' for i in range(16 * nch)) if acc == "a" else ""
    outs = []
    if acc == "v":
        outs += [f'[c{c}] "+v"(c[{c}])' for c in range(nch)]
    outs += [f'[f{r}] "+v"(f[{r}])' for r in range(FILL_REGS)]
    outs += [f'[d{r}] "+v"(d[{r}])' for r in range(4)]
    outs += [f'[e{r}] "+v"(e[{r}])' for r in range(4)]
    outs += ['[s0] "+s"(s0)']
    ins = ['[a] "v"(a)', '[b] "v"(b)', '[la] "v"(la)']
    clob = f': "scc", {agpr_clob}' if agpr_clob else ': "scc"'
    zero_code = agpr_zero(nch) if acc == "a" else ""
    body_code = body(acc, nch, kind, F, lds_a)
    sum_code = agpr_sum(nch) if acc == "a" else "for (int q = 0; q < %d; ++q) for (int k = 0; k < 16; ++k) r += c[q][k];" % nch
    return f"""
__global__ void __launch_bounds__(512, 1) {name}(float* out, long long* cyc, int iters) {{
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  f32x16 c[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
  float f[{FILL_REGS}];
  for (int q = 0; q < {FILL_REGS}; ++q) f[q] = 0.37f * (threadIdx.x + 1) + q;
  u32x4 d[4]; u32x2 e[4];
  for (int q = 0; q < 4; ++q) {{ d[q] = u32x4{{0x3f803f80u + lane, 0x3f803f81u, 0x3f803f82u, 0x3f803f83u}}; e[q] = u32x2{{1u, 2u}}; }}
  u32x4 a = u32x4{{0x3f803f80u + lane, 0x3e803f00u, 0x3f003e80u, 0xbf803f80u}}, b = u32x4{{0x3f80bf80u, 0x3e80bf00u + lane, 0x3f00be80u, 0xbf803f80u}};
  unsigned la = lane * 16; int s0 = 0;
  {zero_code}
  asm volatile("s_nop 4" ::: "memory");
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {{
    asm volatile("{body_code}"
                 : {", ".join(outs)} : {", ".join(ins)} {clob});
  }}
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  {sum_code}
  for (int q = 0; q < {FILL_REGS}; ++q) r += f[q];
  for (int q = 0; q < 4; ++q) r += (float)d[q][0] + (float)e[q][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r + s0;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}}
"""

def agpr_zero(nch):
    return " ".join(f'asm volatile("v_accvgpr_write_b32 a{i}, 0" ::: "a{i}");' for i in range(16 * nch))

def agpr_sum(nch):
    return " ".join(f'{{ float x; asm volatile("v_accvgpr_read_b32 %0, a{i}" : "=v"(x)); r += x; }}' for i in range(0, 16 * nch, 5))

def main():
    variants = []
    for acc in ("a", "v"):
        for nch in (1, 2, 4):
            variants.append((acc, nch, "add", 0, 0))
    for acc in ("a", "v"):
        for nch in (2, 4):
            for kind in ("add", "fma", "sin", "cvt", "dot2c", "act", "ldsr", "ldstr", "salu"):
                for F in (1, 2, 3, 4, 5, 6, 8, 10):
                    if acc == "v" and kind in ("fma", "cvt", "salu"):
                        continue
                    variants.append((acc, nch, kind, F, 0))
    for acc in ("a", "v"):  # ONE chain with fillers between dependent MFMAs (MI355X_MICROARCH.md:443 warns of a cliff)
        for kind in ("add", "act"):
            for F in (1, 2, 3, 5, 8):
                variants.append((acc, 1, kind, F, 0))
    for acc in ("a", "v"):
        for D in (1, 2, 3):
            for F in (0, 3, 5):
                variants.append((acc, 2, "act", F, D))
    print("// GENERATED by profiles/tools/micro/gen_mfma_shadow.py -- do not edit")
    print("#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include <string.h>")
    print("typedef float f32x16 __attribute__((ext_vector_type(16)));\ntypedef unsigned int u32x4 __attribute__((ext_vector_type(4)));\ntypedef unsigned int u32x2 __attribute__((ext_vector_type(2)));")
    names = []
    for (acc, nch, kind, F, D) in variants:
        name = f"k_{acc}{nch}_{kind}{F}" + (f"_ldsA{D}" if D else "")
        names.append((name, acc, nch, kind, F, D))
        print(kernel(name, acc, nch, kind, F, D))
    print("""
typedef void (*kfn)(float*, long long*, int);
struct V { const char* name; kfn f; int nmfma; };
static void run(const V& v, int threads, float* out, long long* cyc, int iters) {
  hipFuncSetAttribute((const void*)v.f, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(v.f, dim3(256), dim3(threads), 65536, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(v.f, dim3(256), dim3(threads), 65536, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * v.nmfma;
  fflush(stdout);
  printf("%-28s waves/SIMD %d  %7.2f cyc/MFMA (wave)  %7.2f ns/MFMA (wall)  eff.clock %.2f GHz\\n", v.name, threads / 256, (double)h / n, ms * 1e6 / n,
         (double)h / (ms * 1e6));
  hipEventDestroy(e0); hipEventDestroy(e1);
}
int main(int argc, char** argv) {
  float* out; long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  const int iters = 3000;
  V vs[] = {""")
    for (name, acc, nch, kind, F, D) in names:
        print(f'    {{"{name}", {name}, {N_MFMA}}},')
    print("""  };
  for (const V& v : vs) {
    bool sel = argc <= 1;
    for (int q = 1; q < argc; ++q) sel = sel || strstr(v.name, argv[q]);
    if (!sel) continue;
    run(v, 256, out, cyc, iters);
    run(v, 512, out, cyc, iters);
  }
  return 0;
}""")

if __name__ == "__main__":
    main()
