// Stand-alone probe (not part of the product): what does a vector-memory / LDS instruction cost a lone wave per SIMD when it sits between
// groups of four v_mfma_f32_16x16x4_f32, as in wino6_mfma's k-step (9 position groups, A operands and accumulators in literal AGPRs)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/vmem_probe.hip -o tools/_build/vmem_probe && tools/_build/vmem_probe
// One workgroup (4 waves, one per SIMD) per CU on NCU CUs; s_memtime around ITER k-steps; the loads hit a 36 KB buffer (L1 / L2 hot).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int I, int N, class F> __device__ __forceinline__ void steps(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); steps<I + 1, N>(f); }
}
#define A10(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", A10(1), A10(2), A10(3), A10(4), A10(5), A10(6), A10(7), A10(8), A10(9), A10(10), A10(11), \
              A10(12), A10(13), A10(14), A10(15), A10(16), A10(17), A10(18), A10(19), A10(20), A10(21), A10(22), A10(23)

// group J: 4 MFMAs, accumulators a[16 J : 16 J + 15], A operands a[144 + 4 J : +3]
template <int J> __device__ __forceinline__ void mfma4(float b)
{
    constexpr int c0 = J * 16, a0 = 144 + J * 4;
    asm volatile("v_mfma_f32_16x16x4_f32 a[%c1:%c2], a[%c9], %0, a[%c1:%c2]\n\tv_mfma_f32_16x16x4_f32 a[%c3:%c4], a[%c10], %0, a[%c3:%c4]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%c5:%c6], a[%c11], %0, a[%c5:%c6]\n\tv_mfma_f32_16x16x4_f32 a[%c7:%c8], a[%c12], %0, a[%c7:%c8]"
                 :: "v"(b), "i"(c0), "i"(c0 + 3), "i"(c0 + 4), "i"(c0 + 7), "i"(c0 + 8), "i"(c0 + 11), "i"(c0 + 12), "i"(c0 + 15),
                    "i"(a0), "i"(a0 + 1), "i"(a0 + 2), "i"(a0 + 3) : AGPRS);
}
template <int A0, bool PAD> __device__ __forceinline__ void load_a(unsigned voff, i32x4 r, unsigned soff)
{
    if constexpr (PAD) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[%c3:%c4], %0, %1, %2 offen" :: "v"(voff), "s"(r), "s"(soff), "i"(A0), "i"(A0 + 3) : AGPRS);
    else asm volatile("buffer_load_dwordx4 a[%c3:%c4], %0, %1, %2 offen" :: "v"(voff), "s"(r), "s"(soff), "i"(A0), "i"(A0 + 3) : AGPRS);
}
__device__ __forceinline__ void load_v(f32x4& d, unsigned voff, i32x4 r, unsigned soff)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(r), "s"(soff));
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%c0)" :: "i"(N)); }

// V: 0 MFMA groups only | 1 + load into the A quad the group just read (the product's order), one k-step ahead | 2 + load into an AGPR quad no MFMA reads
//    3 + load into VGPRs | 4 = 1 with s_nop 4 in front | 5 = 1 + two ds_read_b128 | 6 = load into the PREVIOUS group's A quad (one group of distance)
//    7 two ds_read_b128 only | 8 = 1 without the per-group s_waitcnt | 9 = 1 with TWO k-steps of distance (vmcnt 17)
template <int V>
__global__ void __launch_bounds__(256) probe(const float* buf, unsigned long long* out, int iters)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    unsigned long long a_ = (unsigned long long)buf;
    const i32x4 r = {(int)(unsigned)a_, (int)((unsigned)(a_ >> 32) & 0xFFFFu), 36864 * 4, 0x00020000};
    const unsigned voff = lane * 16u;
    const float* lp = lds + lane * 4 + wave * 1024;
    float b = lane * 1e-3f;
    f32x4 vd[9], l0 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0};
    steps<0, 9>([&](auto J) { vd[decltype(J)::value] = (f32x4){0, 0, 0, 0}; });
    asm volatile("" ::: AGPRS);
    // prime the queue: one k-step of requests
    if constexpr (V == 1 || V == 4 || V == 5 || V == 6 || V == 8 || V == 9) steps<0, 9>([&](auto J) { load_a<144 + 4 * decltype(J)::value, false>(voff, r, wave * 9216u + decltype(J)::value * 1024u); });
    if constexpr (V == 9) steps<0, 9>([&](auto J) { load_a<200 + 4 * decltype(J)::value, false>(voff, r, wave * 9216u + decltype(J)::value * 1024u); });
    if constexpr (V == 2) steps<0, 9>([&](auto J) { load_a<200 + 4 * decltype(J)::value, false>(voff, r, wave * 9216u + decltype(J)::value * 1024u); });
    if constexpr (V == 3) steps<0, 9>([&](auto J) { load_v(vd[decltype(J)::value], voff, r, wave * 9216u + decltype(J)::value * 1024u); });
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
        unsigned so = wave * 9216u;
        asm volatile("" : "+s"(so));
        steps<0, 9>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (V == 1 || V == 2 || V == 4 || V == 5 || V == 6) wait_vm<8>();
            if constexpr (V == 9) wait_vm<17>();
            if constexpr (V == 3) asm volatile("s_waitcnt vmcnt(8)" : "+v"(vd[j]));
            mfma4<j>(b);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (V == 1 || V == 5 || V == 8) load_a<144 + 4 * j, false>(voff, r, so + j * 1024u);
            if constexpr (V == 9) load_a<((j & 1) ? 200 : 144) + 4 * j, false>(voff, r, so + j * 1024u); // (alternating targets: still 9 requests per k-step)
            if constexpr (V == 4) load_a<144 + 4 * j, true>(voff, r, so + j * 1024u);
            if constexpr (V == 2) load_a<200 + 4 * j, false>(voff, r, so + j * 1024u);
            if constexpr (V == 3) load_v(vd[j], voff, r, so + j * 1024u);
            if constexpr (V == 6) load_a<144 + 4 * ((j + 8) % 9), false>(voff, r, so + j * 1024u);
            if constexpr (V == 5 || V == 7) {
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:512" : "=v"(l0), "=v"(l1) : "v"((unsigned)(size_t)lp & 0xFFFFu));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (V == 8) wait_vm<9>();
        if constexpr (V == 5 || V == 7) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l0), "+v"(l1)); b += l0[0] * 1e-30f + l1[1] * 1e-30f; }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = b;
    steps<0, 9>([&](auto J) { s += vd[decltype(J)::value][0]; });
    float a0v;
    asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(a0v) :: AGPRS);
    if (s + a0v == 12345.678f) out[1000] = 1;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

// ---- the product's memory pattern: chunks of two k-steps; A requests two chunks (36 requests + the patch pieces) ahead, streamed from a 576 KB
// weight image every workgroup shares (L2); three "patch" requests per chunk and wave from a stream of the workgroup's own (HBM, or a small hot buffer)
// M: 0 A requests only (image) | 1 + patch requests from HBM | 2 + patch requests from a hot 12 KB | 3 as 1 with ONE chunk of distance
//    4 as 1 plus a workgroup barrier per chunk | 5 A requests from a hot 36 KB instead of the image, patch from HBM
template <int M>
__global__ void __launch_bounds__(256) stream_probe(const float* img, const float* big, unsigned long long* out, int chunks)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long a_ = (unsigned long long)img;
    const i32x4 r = {(int)(unsigned)a_, (int)((unsigned)(a_ >> 32) & 0xFFFFu), 589824, 0x00020000};
    const size_t per_wg = (size_t)chunks * 12288;
    unsigned long long b_ = (unsigned long long)(big + (M == 2 ? 0 : (size_t)blockIdx.x * per_wg / 4));
    const i32x4 rb = {(int)(unsigned)b_, (int)((unsigned)(b_ >> 32) & 0xFFFFu), (int)(M == 2 ? 12288 : per_wg), 0x00020000};
    const unsigned voff = lane * 16u;
    float b = lane * 1e-3f;
    f32x4 pv[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    asm volatile("" ::: AGPRS);
    constexpr bool PATCH = M != 0;
    constexpr int DIST = M == 3 ? 1 : 2;                       // chunks of distance
    constexpr int TCH = 18 + (PATCH ? 3 : 0), NW = DIST * TCH - 1;
    auto a_off = [&](int ch, int ks, int j) { return (M == 5 ? 0u : (unsigned)(ch & 7) * 73728u) + (M == 5 ? 0u : (unsigned)ks * 36864u) + (unsigned)wave * 9216u + (unsigned)j * 1024u; };
    auto p_off = [&](int ch, int n) { return (M == 2 ? 0u : (unsigned)ch * 12288u) + (unsigned)wave * 3072u + (unsigned)n * 1024u; };
    for (int ch = 0; ch < DIST; ++ch) // prime
        steps<0, 2>([&](auto KS) { steps<0, 9>([&](auto J) {
            constexpr int ks = decltype(KS)::value, j = decltype(J)::value;
            load_a<144 + 4 * j + 36 * ks, false>(voff, r, a_off(ch, ks, j));
            if constexpr (PATCH) { if constexpr ((ks == 0 && (j == 5 || j == 7)) || (ks == 1 && j == 5)) load_v(pv[ks * 2 + (j == 7)], voff, rb, p_off(ch, ks * 2 + (j == 7))); }
        }); });
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int ch = 0; ch < chunks - DIST; ++ch) {
        int chn = ch + DIST;
        asm volatile("" : "+s"(chn));
        steps<0, 2>([&](auto KS) { steps<0, 9>([&](auto J) {
            constexpr int ks = decltype(KS)::value, j = decltype(J)::value;
            wait_vm<NW>();
            mfma4<j>(b);
            __builtin_amdgcn_sched_barrier(0);
            load_a<144 + 4 * j + 36 * ks, false>(voff, r, a_off(chn, ks, j));
            if constexpr (PATCH) { if constexpr ((ks == 0 && (j == 5 || j == 7)) || (ks == 1 && j == 5)) load_v(pv[ks * 2 + (j == 7)], voff, rb, p_off(chn, ks * 2 + (j == 7))); }
            __builtin_amdgcn_sched_barrier(0);
        }); });
        if constexpr (M == 4) __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = b + pv[0][0] + pv[1][0] + pv[2][0];
    float a0v;
    asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(a0v) :: AGPRS);
    if (s + a0v == 12345.678f) out[5000] = 1;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int M> static void run_stream(const char* name, const float* img, const float* big, unsigned long long* out, int ncu)
{
    const int chunks = 500;
    CK(hipFuncSetAttribute((const void*)stream_probe<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipLaunchKernelGGL(stream_probe<M>, dim3(ncu), dim3(256), 100 * 1024, 0, img, big, out, chunks);
    CK(hipDeviceSynchronize());
    unsigned long long h[1024];
    CK(hipMemcpy(h, out, sizeof(unsigned long long) * ncu * 4, hipMemcpyDeviceToHost));
    double mx = 0, sm = 0;
    for (int i = 0; i < ncu * 4; ++i) { mx = h[i] > mx ? (double)h[i] : mx; sm += (double)h[i]; }
    printf("%-72s %8.1f cycles per chunk of 72 MFMAs (mean), %8.1f (slowest wave)\n", name, sm / (ncu * 4) / (chunks - (M == 3 ? 1 : 2)), mx / (chunks - (M == 3 ? 1 : 2)));
}

template <int V> static void run(const char* name, const float* buf, unsigned long long* out, int ncu)
{
    const int iters = 2000;
    CK(hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipLaunchKernelGGL(probe<V>, dim3(ncu), dim3(256), 100 * 1024, 0, buf, out, iters); // 100 KB of LDS: one workgroup per CU
    CK(hipDeviceSynchronize());
    unsigned long long h[1024];
    CK(hipMemcpy(h, out, sizeof(unsigned long long) * ncu * 4, hipMemcpyDeviceToHost));
    double mx = 0, sm = 0;
    for (int i = 0; i < ncu * 4; ++i) { mx = h[i] > mx ? (double)h[i] : mx; sm += (double)h[i]; }
        printf("%-64s %8.1f cycles per k-step (mean), %8.1f (slowest wave)\n", name, sm / (ncu * 4) / iters, mx / iters);
}

int main(int argc, char** argv)
{
    const int ncu = argc > 1 ? atoi(argv[1]) : 1;
    float* buf; unsigned long long* out;
    CK(hipMalloc(&buf, 2 * 36864 * 4 + 4096)); CK(hipMemset(buf, 0, 2 * 36864 * 4 + 4096));
    CK(hipMalloc(&out, 8192 * 8)); CK(hipMemset(out, 0, 8192 * 8));
    printf("vmem_probe: %d workgroup(s) of 4 waves, 36 MFMAs (9 groups of 4) per k-step and wave\n", ncu);
    run<0>("0 MFMA groups only", buf, out, ncu);
    run<1>("1 + load -> A quad the group just read (vmcnt 8 per group)", buf, out, ncu);
    run<8>("8 same, one wait per k-step instead of per group", buf, out, ncu);
    run<4>("4 as 1 with s_nop 4 in front of the load", buf, out, ncu);
    run<2>("2 + load -> AGPR quad no MFMA reads", buf, out, ncu);
    run<3>("3 + load -> VGPR quad", buf, out, ncu);
    run<6>("6 + load -> the previous group's A quad", buf, out, ncu);
    run<9>("9 as 1, two k-steps of distance (alternating targets)", buf, out, ncu);
    run<7>("7 + two ds_read_b128 per group (no loads)", buf, out, ncu);
    run<5>("5 = 1 + two ds_read_b128 per group", buf, out, ncu);
    float* img; float* big;
    const size_t big_bytes = (size_t)ncu * 500 * 12288;
    CK(hipMalloc(&img, 589824 + 4096)); CK(hipMemset(img, 0, 589824 + 4096));
    CK(hipMalloc(&big, big_bytes + 4096)); CK(hipMemset(big, 0, big_bytes + 4096));
    run_stream<0>("stream 0: A requests two chunks ahead from the shared 576 KB image", img, big, out, ncu);
    run_stream<5>("stream 5: A requests from a hot 36 KB, patch requests from HBM", img, big, out, ncu);
    run_stream<2>("stream 2: 0 + three patch requests per chunk from a hot 12 KB", img, big, out, ncu);
    run_stream<1>("stream 1: 0 + three patch requests per chunk from the workgroup's HBM stream", img, big, out, ncu);
    run_stream<3>("stream 3: as 1 with ONE chunk of distance", img, big, out, ncu);
    run_stream<4>("stream 4: as 1 plus a workgroup barrier per chunk", img, big, out, ncu);
    return 0;
}
