// Stand-alone probe for the round-2 plan: issue rate of v_mfma_f32_16x16x4_f32 against
// v_mfma_f32_16x16x32_bf16 on gfx950, and the cost of splitting fp32 operands into (hi, lo) bf16 pairs in registers.
// Not part of the product.  hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/_build/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ACC = 8; // independent accumulators per wave (2 x 4 tile pairs)

__global__ void __launch_bounds__(256, 2) k_f32(float* out, int iters)
{
    f32x4 acc[ACC];
    for (int i = 0; i < ACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) // 8 fp32 MFMAs (K = 4 each) = one K = 32 block
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// SPLIT = 0: three bf16 MFMAs per K = 32 block on constant operands; SPLIT = 1: the B operand (8 fp32 per lane) is split
// into hi/lo bf16 every block (4 N-tiles per wave share nothing here: worst case, one split per MFMA triple)
template <int SPLIT>
__global__ void __launch_bounds__(256, 2) k_bf16x3(float* out, const float* in, int iters)
{
    f32x4 acc[ACC];
    for (int i = 0; i < ACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ah, al, bh, bl;
    for (int e = 0; e < 8; ++e) { ah[e] = (__bf16)(threadIdx.x * 1e-3f + e); al[e] = (__bf16)(1e-3f * e); bh[e] = (__bf16)(threadIdx.x * 2e-3f - e); bl[e] = (__bf16)(2e-3f * e); }
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = in[threadIdx.x * 8 + e];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) {
            if (SPLIT && (i & 1) == 0) { // one split per two M-tiles (MT = 2 reuse of the B operand)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = x[e] + acc[i][e & 3] * 1e-30f; // data dependence: the split cannot be hoisted
                    const __bf16 h = (__bf16)v;
                    bh[e] = h;
                    bl[e] = (__bf16)(v - (float)h);
                }
            }
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    const int blocks = 512, threads = 256, iters = 20000;
    float *out, *in;
    CK(hipMalloc(&out, blocks * threads * sizeof(float)));
    CK(hipMalloc(&in, threads * 8 * sizeof(float)));
    CK(hipMemset(in, 0, threads * 8 * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double waves = (double)blocks * threads / 64;
    const double kblocks = waves * iters * ACC; // (16x16 tile, K = 32) products
    const double flops = kblocks * 16 * 16 * 32 * 2;
    for (int which = 0; which < 3; ++which) {
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 0) k_f32<<<blocks, threads>>>(out, iters);
            else if (which == 1) k_bf16x3<0><<<blocks, threads>>>(out, in, iters);
            else k_bf16x3<1><<<blocks, threads>>>(out, in, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const char* nm[3] = {"fp32 16x16x4 (8 per K=32 block)", "bf16x3 16x16x32, constant operands", "bf16x3 16x16x32, B split in registers per 2 M-tiles"};
        printf("%-52s %8.3f ms  %7.1f useful TFLOP/s (fp32-equivalent products)\n", nm[which], ms, flops / ms * 1e-9);
    }
    return 0;
}
