// valu_rate.hip -- measures what one SIMD of a gfx950 CU sustains for the VALU instruction kinds
// the render kernel is made of, at 1/2/4/8 waves per SIMD.  Stand-alone: hipcc
// --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate.  Prints wave-instructions per SIMD per
// cycle-at-2.4-GHz (so 0.5 = one wave64 instruction every 2 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define R8(S0, S1, S2, S3, S4, S5, S6, S7) S0 S1 S2 S3 S4 S5 S6 S7

// 8 independent accumulators, 4 groups per trip: 32 instructions of the kind per trip
#define KERNEL_INDEP(NAME, INSN)                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float b, float c)          \
    {                                                                                             \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float vb = b + threadIdx.x * 1e-9f, vc = c;                                               \
        for (int i = 0; i < iters; ++i) {                                                         \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                       \
                asm volatile(INSN : "+v"(a0) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a1) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a2) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a3) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a4) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a5) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a6) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a7) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
            }                                                                                     \
        }                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));  \
    }

#define KERNEL_DEP(NAME, INSN)                                                                    \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float b, float c)          \
    {                                                                                             \
        float a0 = threadIdx.x;                                                                   \
        float vb = b + threadIdx.x * 1e-9f, vc = c;                                               \
        for (int i = 0; i < iters; ++i) {                                                         \
            _Pragma("unroll") for (int g = 0; g < 32; ++g) asm volatile(INSN : "+v"(a0) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc"); \
        }                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a0;                                                 \
    }

#define KERNEL_PK(NAME, INSN)                                                                     \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, float b, float c)          \
    {                                                                                             \
        v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        v2f vb = {b + threadIdx.x * 1e-9f, b}, vc = {c, c};                                       \
        for (int i = 0; i < iters; ++i) {                                                         \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                       \
                asm volatile(INSN : "+v"(a0) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a1) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a2) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a3) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a4) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a5) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a6) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
                asm volatile(INSN : "+v"(a7) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");                                 \
            }                                                                                     \
        }                                                                                         \
        v2f s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));                                \
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                                          \
    }

KERNEL_INDEP(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL_INDEP(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL_INDEP(k_add, "v_add_f32 %0, %0, %1")
KERNEL_INDEP(k_mul_vop3, "v_mul_f32_e64 %0, %0, -%1")
KERNEL_INDEP(k_rcp, "v_rcp_f32 %0, %0")
KERNEL_INDEP(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL_INDEP(k_rsq, "v_rsq_f32 %0, %0")
KERNEL_INDEP(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_INDEP(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL_INDEP(k_cmp_sgpr, "v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %2, s[20:21]")
KERNEL_INDEP(k_divscale, "v_div_scale_f32 %0, vcc, %0, %1, %2")
KERNEL_INDEP(k_divfmas, "v_div_fmas_f32 %0, %0, %1, %2")
KERNEL_INDEP(k_divfixup, "v_div_fixup_f32 %0, %0, %1, %2")
KERNEL_INDEP(k_mov, "v_mov_b32 %0, %1")
KERNEL_INDEP(k_floor, "v_floor_f32 %0, %0")
KERNEL_INDEP(k_cvt, "v_cvt_i32_f32 %0, %0")
KERNEL_INDEP(k_fma_salu, "v_fma_f32 %0, %0, %1, %2\n s_add_u32 s20, s20, 1")
KERNEL_INDEP(k_fma_2salu, "v_fma_f32 %0, %0, %1, %2\n s_add_u32 s20, s20, 1\n s_and_b32 s21, s20, 7")
KERNEL_INDEP(k_fma_nop, "v_fma_f32 %0, %0, %1, %2\n s_nop 0")

KERNEL_INDEP(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL_INDEP(k_sub, "v_sub_f32 %0, %0, %1")
KERNEL_INDEP(k_max, "v_max_f32 %0, %0, %1")
KERNEL_INDEP(k_min, "v_min_f32 %0, %0, %1")
KERNEL_INDEP(k_and, "v_and_b32 %0, %0, %1")
KERNEL_INDEP(k_addu, "v_add_u32 %0, %0, %1")
KERNEL_INDEP(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL_INDEP(k_fmamk, "v_fmamk_f32 %0, %0, 0x3f000000, %1")
KERNEL_INDEP(k_fmaak, "v_fmaak_f32 %0, %0, %1, 0x3f000000")
KERNEL_INDEP(k_mul_lit, "v_mul_f32 %0, 0x3f000001, %0")
KERNEL_INDEP(k_mul_sgpr, "v_mul_f32 %0, s20, %0")
KERNEL_INDEP(k_fma_sgpr, "v_fma_f32 %0, %0, s20, %1")
KERNEL_INDEP(k_sub_e64, "v_sub_f32_e64 %0, %0, |%1|")
KERNEL_INDEP(k_cnd_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL_INDEP(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL_INDEP(k_cmp_e64, "v_cmp_lt_f32_e64 s[20:21], %0, %1")
KERNEL_INDEP(k_cmp_class, "v_cmp_class_f32 vcc, %0, %1")
KERNEL_INDEP(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL_INDEP(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL_INDEP(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL_INDEP(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL_INDEP(k_rndne, "v_rndne_f32 %0, %0")
KERNEL_INDEP(k_exp, "v_exp_f32 %0, %0")
KERNEL_INDEP(k_log, "v_log_f32 %0, %0")
KERNEL_INDEP(k_fract, "v_fract_f32 %0, %0")
KERNEL_INDEP(k_ldexp, "v_ldexp_f32 %0, %0, %1")
KERNEL_INDEP(k_frexp, "v_frexp_mant_f32 %0, %0")
KERNEL_INDEP(k_trunc, "v_trunc_f32 %0, %0")
KERNEL_INDEP(k_fma_setreg, "v_fma_f32 %0, %0, %1, %2\n s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 3")
KERNEL_INDEP(k_fma_smov, "v_fma_f32 %0, %0, %1, %2\n s_mov_b64 s[20:21], exec")
KERNEL_INDEP(k_2fma_salu, "v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1\n s_add_u32 s20, s20, 1")
KERNEL_INDEP(k_4v_salu, "v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n s_add_u32 s20, s20, 1")
KERNEL_INDEP(k_mul_rcp, "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_rcp_f32 %0, %0")
KERNEL_INDEP(k_mul_pk, "v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1")
KERNEL_DEP(k_fma_dep, "v_fma_f32 %0, %0, %1, %2")
KERNEL_DEP(k_mul_dep, "v_mul_f32 %0, %0, %1")
KERNEL_DEP(k_rcp_dep, "v_rcp_f32 %0, %0")
KERNEL_PK(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL_PK(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
KERNEL_PK(k_pk_add, "v_pk_add_f32 %0, %0, %1")

// a loop with a divergent branch per 8 instructions: what a taken/not-taken s_cbranch costs a wave
__global__ __launch_bounds__(256) void k_branchy(float *out, int iters, float b, float c)
{
    float a0 = threadIdx.x, a1 = a0 + 1;
    float vb = b + threadIdx.x * 1e-9f, vc = c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
            if (a0 > 1e30f) {   // never true, per-lane condition: v_cmp + s_and_saveexec + s_cbranch_execz
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
            }
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(vb), "v"(vc) : "s20", "s21", "vcc", "scc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1;
}

typedef void (*kfn_t)(float *, int, float, float);
struct Case { const char *name; kfn_t fn; int valu_per_trip; };

int main()
{
    int cus = 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float *out;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<Case> cases = {
        {"v_fma_f32 indep", k_fma, 32}, {"v_mul_f32 indep", k_mul, 32}, {"v_add_f32 indep", k_add, 32},
        {"v_mul_f32 vop3 neg", k_mul_vop3, 32},
        {"v_fma_f32 dep", k_fma_dep, 32}, {"v_mul_f32 dep", k_mul_dep, 32}, {"v_rcp_f32 dep", k_rcp_dep, 32},
        {"v_pk_fma_f32", k_pk_fma, 32}, {"v_pk_mul_f32", k_pk_mul, 32}, {"v_pk_add_f32", k_pk_add, 32},
        {"v_rcp_f32", k_rcp, 32}, {"v_sqrt_f32", k_sqrt, 32}, {"v_rsq_f32", k_rsq, 32},
        {"v_cndmask vcc", k_cndmask, 32}, {"v_cmp+v_cndmask vcc", k_cmp_cnd, 64}, {"v_cmp+v_cndmask sgpr", k_cmp_sgpr, 64},
        {"v_div_scale_f32", k_divscale, 32}, {"v_div_fmas_f32", k_divfmas, 32}, {"v_div_fixup_f32", k_divfixup, 32},
        {"v_mov_b32", k_mov, 32}, {"v_floor_f32", k_floor, 32}, {"v_cvt_i32_f32", k_cvt, 32},
        {"v_fmac_f32", k_fmac, 32}, {"v_sub_f32", k_sub, 32}, {"v_max_f32", k_max, 32}, {"v_min_f32", k_min, 32},
        {"v_and_b32", k_and, 32}, {"v_add_u32", k_addu, 32}, {"v_lshlrev_b32", k_lshl, 32},
        {"v_fmamk_f32 (literal)", k_fmamk, 32}, {"v_fmaak_f32 (literal)", k_fmaak, 32}, {"v_mul_f32 literal", k_mul_lit, 32},
        {"v_sub_f32_e64 |abs|", k_sub_e64, 32}, {"v_cndmask_b32_e64 sgpr", k_cnd_sgpr, 32},
        {"v_cmp_lt_f32 vcc", k_cmp_vcc, 32}, {"v_cmp_lt_f32_e64 sgpr", k_cmp_e64, 32}, {"v_cmp_class_f32", k_cmp_class, 32},
        {"v_med3_f32", k_med3, 32}, {"v_lshl_add_u32", k_lshl_add, 32}, {"v_and_or_b32", k_and_or, 32},
        {"v_cvt_f32_i32", k_cvt_f32_i32, 32}, {"v_rndne_f32", k_rndne, 32}, {"v_exp_f32", k_exp, 32}, {"v_log_f32", k_log, 32},
        {"v_fract_f32", k_fract, 32}, {"v_ldexp_f32", k_ldexp, 32}, {"v_frexp_mant_f32", k_frexp, 32}, {"v_trunc_f32", k_trunc, 32},
        {"2 valu + 1 salu (valu count)", k_2fma_salu, 64}, {"4 valu + 1 salu (valu count)", k_4v_salu, 128},
        {"3 mul + 1 rcp (all count)", k_mul_rcp, 128}, {"mul + add dependent pair", k_mul_pk, 64},
        {"v_fma + 1 salu", k_fma_salu, 32}, {"v_fma + 2 salu", k_fma_2salu, 32}, {"v_fma + s_nop", k_fma_nop, 32},
        {"16 fma + 4 untaken branches", k_branchy, 32},
    };
    const int iters = 4096;
    printf("%-30s %8s %8s %8s %8s   (VALU wave-instructions per SIMD per 2.4 GHz cycle)\n", "kind", "1w/SIMD", "2w", "4w", "8w");
    for (auto &c : cases) {
        printf("%-30s", c.name);
        for (int wps : {1, 2, 4, 8}) {
            dim3 grid(cus * wps);
            hipLaunchKernelGGL(c.fn, grid, dim3(256), 0, 0, out, 64, 1.0000001f, 1e-9f);     // warm-up
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(c.fn, grid, dim3(256), 0, 0, out, iters, 1.0000001f, 1e-9f);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double insn_per_simd = (double)iters * c.valu_per_trip * wps;      // each SIMD holds wps waves
            double cycles = best * 1e-3 * 2.4e9;
            printf(" %8.3f", insn_per_simd / cycles);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
