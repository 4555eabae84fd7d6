// Stand-alone probe (not part of the product): what can a CDNA4 SIMD issue next to v_mfma_f32_16x16x4_f32?
// One workgroup on one CU, s_memtime around a loop of 8 independent MFMAs with N filler instructions behind each:
//   one wave per SIMD (256 threads)      -- does a filler of the SAME wave hide in the MFMA's 32 cycles?
//   two waves per SIMD (512 threads)     -- both run the same stream, or waves 4..7 run fillers only
// hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/_build/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define MF(i) "v_mfma_f32_16x16x4_f32 %" #i ", %16, %17, %" #i "\n"
// fillers: independent of each other and of the MFMAs
#define VADD_A "v_add_f32 %8, %8, %17\n"
#define VADD_B "v_add_f32 %9, %9, %17\n"
#define VADD_C "v_add_f32 %10, %10, %17\n"
#define VADD_D "v_add_f32 %11, %11, %17\n"
#define VOWN_A "v_add_f32 %8, %8, %11\n"
#define VOWN_B "v_add_f32 %9, %9, %11\n"
#define VOWN_C "v_add_f32 %10, %10, %11\n"
#define VMUL_A "v_mul_f32 %8, %8, %11\n"
#define VFMA_A "v_fma_f32 %8, %8, %11, %11\n"
#define VMAX_A "v_max_f32 %9, %9, %11\n"
#define PK_A "v_pk_add_f32 %12, %12, %14\n"
#define PK_B "v_pk_add_f32 %13, %13, %14\n"
#define SADD "s_add_u32 s40, s40, 1\n"
#define DSRD_A "ds_read_b32 %8, %18\n"
#define DSRD_B "ds_read_b32 %9, %18 offset:256\n"
#define DSRD4 "ds_read_b128 %15, %18 offset:1024\n"
#define DSWR "ds_write_b32 %18, %10 offset:4096\n"
#define VMOV "v_mov_b32 %11, %17\n"
#define NOP0 "s_nop 0\n"

#define BODY(F) MF(0) F MF(1) F MF(2) F MF(3) F MF(4) F MF(5) F MF(6) F MF(7) F "s_waitcnt lgkmcnt(0)\n"
#define FILLONLY(F) F F F F F F F F "s_waitcnt lgkmcnt(0)\n"

#define OPERANDS                                                                                                     \
    : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]), \
      "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(q0)                                  \
    : "v"(a), "v"(b), "v"(la)                                                                                        \
    : "s40", "memory"

// MODE 0: every wave runs BODY(F).  MODE 1: waves 0..3 run BODY("") (MFMA only), waves 4.. run FILLONLY(F) x 4.
#define KERNEL(NAME, F)                                                                                              \
    __global__ void __launch_bounds__(512) NAME(unsigned long long* out, int iters, int mode)                         \
    {                                                                                                                \
        __shared__ float lds[4096];                                                                                  \
        f32x4 acc[8];                                                                                                \
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};                                            \
        float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f, f0 = a, f1 = b, f2 = a, f3 = b;                      \
        f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {a, a};                                                                 \
        f32x4 q0 = {a, b, a, b};                                                                                     \
        lds[threadIdx.x] = a;                                                                                        \
        unsigned la = (threadIdx.x & 63) * 4;                                                                        \
        __syncthreads();                                                                                             \
        const int wave = threadIdx.x >> 6;                                                                           \
        unsigned long long t0 = __builtin_readcyclecounter();                                                        \
        asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t0));                                               \
        if (mode == 2 && wave < 4) {                                                                                 \
        } else if (mode == 0 || wave < 4) {                                                                          \
            if (mode == 0)                                                                                           \
                for (int it = 0; it < iters; ++it) asm volatile(BODY(F) OPERANDS);                                   \
            else                                                                                                     \
                for (int it = 0; it < iters; ++it) asm volatile(BODY("") OPERANDS);                                  \
        } else {                                                                                                     \
            for (int it = 0; it < iters * 4; ++it) asm volatile(FILLONLY(F) OPERANDS);                               \
        }                                                                                                            \
        unsigned long long t1;                                                                                       \
        asm volatile("s_nop 15\ns_nop 15\ns_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1));                           \
        float s = f0 + f1 + f2 + f3 + p0[0] + p1[0] + p2[1] + q0[0];                                                 \
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];                                                      \
        if (s == 12345.678f) out[100] = 1;                                                                           \
        if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;                                                            \
    }

KERNEL(k_none, "")
KERNEL(k_vadd1, VADD_A)
KERNEL(k_vadd2, VADD_A VADD_B)
KERNEL(k_vadd4, VADD_A VADD_B VADD_C VADD_D)
KERNEL(k_vadd6, VADD_A VADD_B VADD_C VADD_D VADD_A VADD_B)
KERNEL(k_vown1, VOWN_A)
KERNEL(k_vown2, VOWN_A VOWN_B)
KERNEL(k_vown4, VOWN_A VOWN_B VOWN_C VOWN_A)
KERNEL(k_vdep4, VOWN_A VOWN_A VOWN_A VOWN_A)
KERNEL(k_vfma2, VFMA_A VMAX_A)
KERNEL(k_vmul2, VMUL_A VMAX_A)
KERNEL(k_pk1, PK_A)
KERNEL(k_pk2, PK_A PK_B)
KERNEL(k_pk4, PK_A PK_B PK_A PK_B)
KERNEL(k_sadd1, SADD)
KERNEL(k_sadd4, SADD SADD SADD SADD)
KERNEL(k_nop4, NOP0 NOP0 NOP0 NOP0)
KERNEL(k_vmov4, VMOV VMOV VMOV VMOV)
KERNEL(k_dsrd1, DSRD_A)
KERNEL(k_dsrd2, DSRD_A DSRD_B)
KERNEL(k_dsrd4x1, DSRD4)
KERNEL(k_dswr1, DSWR)
KERNEL(k_mix, VADD_A DSRD_B VADD_C SADD)


// The step loop of wino4_mfma in miniature (per 16 MFMAs = one patch row: 4 A-fragment reads, one cluster of 10 VALU, 3 LDS writes)
// against the same filler work per 8 MFMAs (an MT = 2 kernel: every B operand feeds 2 MFMAs instead of 4), to be run with two
// waves per SIMD.  BODY16 / BODY8 are one iteration each.
#define V10 VADD_A VADD_B VADD_C VADD_D PK_A PK_B VADD_A VADD_B PK_A PK_B
#define BODY16 MF(0) DSRD4 MF(1) MF(2) V10 MF(3) DSWR MF(4) DSRD4 MF(5) MF(6) MF(7) DSWR MF(0) DSRD4 MF(1) MF(2) MF(3) DSWR MF(4) DSRD4 MF(5) MF(6) MF(7) "s_waitcnt lgkmcnt(0)\n"
#define BODY8 MF(0) DSRD4 MF(1) V10 MF(2) DSWR MF(3) DSRD4 MF(4) DSWR MF(5) DSRD4 MF(6) DSWR MF(7) DSRD4 "s_waitcnt lgkmcnt(0)\n"
#define KERNEL2(NAME, BODYX)                                                                                         \
    __global__ void __launch_bounds__(512) NAME(unsigned long long* out, int iters, int mode)                         \
    {                                                                                                                \
        __shared__ float lds[4096];                                                                                  \
        f32x4 acc[8];                                                                                                \
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};                                            \
        float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f, f0 = a, f1 = b, f2 = a, f3 = b;                      \
        f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {a, a};                                                                 \
        f32x4 q0 = {a, b, a, b};                                                                                     \
        lds[threadIdx.x] = a;                                                                                        \
        unsigned la = (threadIdx.x & 63) * 16;                                                                       \
        __syncthreads();                                                                                             \
        const int wave = threadIdx.x >> 6;                                                                           \
        unsigned long long t0, t1;                                                                                   \
        asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t0));                                               \
        for (int it = 0; it < iters; ++it) asm volatile(BODYX OPERANDS);                                             \
        asm volatile("s_nop 15\ns_nop 15\ns_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1));                           \
        float s = f0 + f1 + f2 + f3 + p0[0] + p1[0] + p2[1] + q0[0];                                                 \
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];                                                      \
        if (s == 12345.678f) out[100] = 1;                                                                           \
        if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;                                                            \
    }
KERNEL2(k_step16, BODY16)
KERNEL2(k_step8, BODY8)

// Same mix with the A fragments really flowing through LDS: each group of 4 MFMAs takes its A operands from the ds_read_b128 issued
// three groups earlier (registers v[200:215], rotating) behind the s_waitcnt hipcc would put there -- the dependence the product
// kernel has and the patterns above lack.
#define MFA(i, r) "v_mfma_f32_16x16x4_f32 %" #i ", v" #r ", %17, %" #i "\n"
#define GRP(w, r0, r1, r2, r3, ld, MID) "s_waitcnt lgkmcnt(" #w ")\n" MFA(0, r0) "ds_read_b128 v[" #ld "], %18 offset:1024\n" MFA(1, r1) MID MFA(2, r2) MFA(3, r3)
#define BODY16D GRP(2, 200, 201, 202, 203, 212:215, "") GRP(2, 204, 205, 206, 207, 200:203, V10) GRP(2, 208, 209, 210, 211, 204:207, "") GRP(2, 212, 213, 214, 215, 208:211, "")
#define KERNEL3(NAME, BODYX)                                                                                         \
    __global__ void __launch_bounds__(512) NAME(unsigned long long* out, int iters, int mode)                         \
    {                                                                                                                \
        __shared__ float lds[4096];                                                                                  \
        f32x4 acc[8];                                                                                                \
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};                                            \
        float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f, f0 = a, f1 = b, f2 = a, f3 = b;                      \
        f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {a, a};                                                                 \
        f32x4 q0 = {a, b, a, b};                                                                                     \
        lds[threadIdx.x] = a;                                                                                        \
        unsigned la = (threadIdx.x & 63) * 16;                                                                       \
        const unsigned long long* gp = out + 16 + (threadIdx.x & 63) * 2;                                            \
        __syncthreads();                                                                                             \
        const int wave = threadIdx.x >> 6;                                                                           \
        unsigned long long t0, t1;                                                                                   \
        asm volatile("ds_read_b128 v[200:203], %0\nds_read_b128 v[204:207], %0\nds_read_b128 v[208:211], %0\nds_read_b128 v[212:215], %0\ns_waitcnt lgkmcnt(0)\n" \
                     "ds_read_b128 v[200:203], %0\nds_read_b128 v[204:207], %0\nds_read_b128 v[208:211], %0" :: "v"(la) : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215"); \
        asm volatile("s_memtime %0" : "=s"(t0));                                                                      \
        for (int it = 0; it < iters; ++it)                                                                           \
            asm volatile(BODYX OPERANDS_D);                                                                          \
        asm volatile("s_nop 15\ns_nop 15\ns_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1));                           \
        float s = f0 + f1 + f2 + f3 + p0[0] + p1[0] + p2[1] + q0[0];                                                 \
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];                                                      \
        if (s == 12345.678f) out[100] = 1;                                                                           \
        if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;                                                            \
    }
#define OPERANDS_D                                                                                                   \
    : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]), \
      "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(q0)                                  \
    : "v"(a), "v"(b), "v"(la), "v"(gp)                                                                               \
    : "s40", "memory", "v199", "v198", "v197", "v196", "v195", "v194", "v193", "v192", "v191", "v190", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215"
KERNEL3(k_step16d, BODY16D)
// the product kernel's whole memory mix per 16 MFMAs: + 2 ds_read2_b32 (raw patch row), 2 ds_write_b32 + 1 ds_write_b128 (staging),
// 2 global_load_dword + 1 global_load_dwordx4 (refills), each behind its MFMA like the product's gaps
#define RAW2 "ds_read2_b32 v[198:199], %18 offset0:1 offset1:9\nds_read2_b32 v[196:197], %18 offset0:2 offset1:10\n"
#define STG "ds_write_b32 %18, %10 offset:8192\nglobal_load_dword v195, %19, off\n"
#define STG4 "ds_write_b128 %18, %15 offset:12288\nglobal_load_dwordx4 v[190:193], %19, off\n"
#define GRPR(w, r0, r1, r2, r3, ld, B_, C_, D_) "s_waitcnt lgkmcnt(" #w ")\n" MFA(0, r0) "ds_read_b128 v[" #ld "], %18 offset:1024\n" MFA(1, r1) B_ MFA(2, r2) C_ MFA(3, r3) D_
#define BODY16R GRPR(6, 200, 201, 202, 203, 212:215, RAW2, V10, STG) GRPR(6, 204, 205, 206, 207, 200:203, "", "", STG) GRPR(6, 208, 209, 210, 211, 204:207, "", "", STG4) GRPR(6, 212, 213, 214, 215, 208:211, "", "", "s_waitcnt vmcnt(3)\n")
KERNEL3(k_step16r, BODY16R)
#define STG_NG "ds_write_b32 %18, %10 offset:8192\n"
#define STG4_NG "ds_write_b128 %18, %15 offset:12288\n"
#define BODY16R_NOGL GRPR(6, 200, 201, 202, 203, 212:215, RAW2, V10, STG_NG) GRPR(6, 204, 205, 206, 207, 200:203, "", "", STG_NG) GRPR(6, 208, 209, 210, 211, 204:207, "", "", STG4_NG) GRPR(6, 212, 213, 214, 215, 208:211, "", "", "")
#define BODY16R_NOW4 GRPR(6, 200, 201, 202, 203, 212:215, RAW2, V10, STG) GRPR(6, 204, 205, 206, 207, 200:203, "", "", STG) GRPR(6, 208, 209, 210, 211, 204:207, "", "", STG) GRPR(6, 212, 213, 214, 215, 208:211, "", "", "s_waitcnt vmcnt(3)\n")
#define BODY16R_NORAW GRPR(6, 200, 201, 202, 203, 212:215, "", V10, STG) GRPR(6, 204, 205, 206, 207, 200:203, "", "", STG) GRPR(6, 208, 209, 210, 211, 204:207, "", "", STG4) GRPR(6, 212, 213, 214, 215, 208:211, "", "", "s_waitcnt vmcnt(3)\n")
#define BODY16R_NOSTG GRPR(6, 200, 201, 202, 203, 212:215, RAW2, V10, "") GRPR(6, 204, 205, 206, 207, 200:203, "", "", "") GRPR(6, 208, 209, 210, 211, 204:207, "", "", "") GRPR(6, 212, 213, 214, 215, 208:211, "", "", "")
KERNEL3(k_r_nogl, BODY16R_NOGL)
KERNEL3(k_r_now4, BODY16R_NOW4)
KERNEL3(k_r_noraw, BODY16R_NORAW)
KERNEL3(k_r_nostg, BODY16R_NOSTG)
// the same, 8 x unrolled (128 MFMAs of straight-line code per iteration, like one chunk body of the product kernel): instruction fetch
#define BODY128D BODY16D BODY16D BODY16D BODY16D BODY16D BODY16D BODY16D BODY16D
KERNEL3(k_step128d, BODY128D)

typedef void (*kfn)(unsigned long long*, int, int);
struct Ent { const char* name; kfn f; int fillers; };

int main()
{
    unsigned long long* out;
    CK(hipMalloc(&out, 8192));
    const int iters = 4000;
    Ent ents[] = {{"none", k_none, 0}, {"v_add x1", k_vadd1, 1}, {"v_add x2", k_vadd2, 2}, {"v_add x4", k_vadd4, 4}, {"v_add x6", k_vadd6, 6},
                  {"v_add own regs x1", k_vown1, 1}, {"v_add own regs x2", k_vown2, 2}, {"v_add own regs x4", k_vown4, 4}, {"v_add dependent x4", k_vdep4, 4},
                  {"v_fma+v_max", k_vfma2, 2}, {"v_mul+v_max", k_vmul2, 2},
                  {"v_pk_add x1", k_pk1, 1}, {"v_pk_add x2", k_pk2, 2}, {"v_pk_add x4", k_pk4, 4}, {"s_add x1", k_sadd1, 1}, {"s_add x4", k_sadd4, 4},
                  {"s_nop x4", k_nop4, 4}, {"v_mov x4", k_vmov4, 4}, {"ds_read_b32 x1", k_dsrd1, 1}, {"ds_read_b32 x2", k_dsrd2, 2},
                  {"ds_read_b128 x1", k_dsrd4x1, 1}, {"ds_write_b32 x1", k_dswr1, 1}, {"v_add+ds_read+v_add+s_add", k_mix, 4}};
    printf("cycles per MFMA of the SLOWEST wave (v_mfma_f32_16x16x4_f32 = 32 cycles of matrix pipe), N fillers behind each MFMA; AGPR accumulators\n");
    printf("A: one wave per SIMD.  B: two waves per SIMD, same stream (SIMD time per MFMA = B / 2).  C: waves 0-3 MFMA only, waves 4-7 fillers only:\n");
    printf("MFMA wave cycles per MFMA | filler wave cycles per filler (alone: D)\n");
    printf("%-28s %8s %8s %8s %8s | %6s %6s\n", "filler", "A", "B", "B/2", "C mfma", "C fill", "D fill");
    for (const Ent& e : ents) {
        double r[4] = {0, 0, 0, 0}, fill[2] = {0, 0};
        for (int cfg = 0; cfg < 4; ++cfg) {
            const int threads = cfg == 0 ? 256 : 512, mode = cfg == 0 ? 0 : cfg - 1;
            unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(out, 0, 64));
                hipLaunchKernelGGL(e.f, dim3(1), dim3(threads), 0, 0, out, iters, mode);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            unsigned long long m03 = 0, m47 = 0;
            for (int w = 0; w < 4; ++w) { if (h[w] > m03) m03 = h[w]; if (h[w + 4] > m47) m47 = h[w + 4]; }
            if (cfg <= 1) r[cfg] = (double)(m03 > m47 ? m03 : m47) / ((double)iters * 8);
            if (cfg == 2) r[2] = (double)m03 / ((double)iters * 8);
            if (cfg >= 2 && e.fillers) fill[cfg - 2] = (double)m47 / ((double)iters * 4 * 8 * e.fillers);
        }
        printf("%-28s %8.1f %8.1f %8.1f %8.1f | %6.1f %6.1f\n", e.name, r[0], r[1], r[1] / 2, r[2], fill[0], fill[1]);
    }
    {   // whole-step patterns
        printf("step-loop patterns (cycles per MFMA per SIMD): wino4-like, 16 MFMAs + 10 VALU + 4 ds_read_b128 + 3 ds_write per iteration, and the same filler work per 8 MFMAs (MT = 2)\n");
        struct P { const char* name; kfn f; int mf; } ps[] = {{"16 MFMAs per filler set", k_step16, 16}, {"8 MFMAs per filler set", k_step8, 8},
                                                              {"16 MFMAs, A through LDS", k_step16d, 16}, {"same, 8x unrolled", k_step128d, 128}, {"product memory mix", k_step16r, 16}, {" - global loads", k_r_nogl, 16},
                                                              {" - ds_write_b128 (b32)", k_r_now4, 16}, {" - raw ds_read2", k_r_noraw, 16}, {" - all staging", k_r_nostg, 16}};
        for (const P& q : ps)
            for (int threads = 256; threads <= 512; threads += 256) {
                unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipMemset(out, 0, 64));
                    hipLaunchKernelGGL(q.f, dim3(1), dim3(threads), 0, 0, out, iters, 0);
                    CK(hipDeviceSynchronize());
                }
                CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
                unsigned long long m = 0;
                for (int w = 0; w < 8; ++w) if (h[w] > m) m = h[w];
                const int waves = threads / 256;
                printf("%-28s %d wave(s) per SIMD: %6.1f cycles per MFMA per SIMD\n", q.name, waves, (double)m / ((double)iters * q.mf * waves));
            }
    }
    return 0;
}
