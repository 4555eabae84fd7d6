// Stand-alone check + timing of wino6_mfma (csrc/wino6.hip) on ONE layer against a CPU direct convolution.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -munsafe-fp-atomics tools/w6_test.hip -o tools/_build/w6_test
//   tools/_build/w6_test [C] [H] [W] [B] [twt] [check 0/1] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <random>
#include <algorithm>
#include <cstring>
#include "../3d_object_detection_amd/csrc/wino6.hip"
using namespace ppc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// soak mode: compare an output tensor with the first launch's, bit for bit, on the device
__global__ void soak_cmp(const unsigned* __restrict__ a, const unsigned* __restrict__ ref, size_t n, unsigned long long* __restrict__ res)
{
    unsigned cnt = 0;
    unsigned long long first = ~0ull;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (a[i] != ref[i]) { ++cnt; if (i < first) first = i; }
    if (cnt) { atomicAdd(res, (unsigned long long)cnt); atomicMin(res + 1, first); }
}

int main(int argc, char** argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 32, W = argc > 3 ? atoi(argv[3]) : 32, B = argc > 4 ? atoi(argv[4]) : 2;
    const int twt = argc > 5 ? atoi(argv[5]) : 4, check = argc > 6 ? atoi(argv[6]) : 1, reps = argc > 7 ? atoi(argv[7]) : 3, dbg = argc > 8 ? atoi(argv[8]) : 0, nostat = argc > 9 ? atoi(argv[9]) : 0, nores = argc > 10 ? atoi(argv[10]) : 0;
    const int Cout = C;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    const size_t plane = (size_t)H * W, fs = (size_t)C * plane;
    std::vector<float> x(B * fs), w((size_t)Cout * C * 9), res(B * fs), sc(C), sh(C);
    if (check) { for (auto& v : x) v = nd(rng); for (auto& v : res) v = nd(rng); }
    else { // timing only: cheap pseudo-random fill (not zeros: zeros raise the clock)
        unsigned h = 12345u;
        for (size_t i = 0; i < x.size(); ++i) { h = h * 1664525u + 1013904223u; x[i] = (float)(h >> 8) * (1.f / 8388608.f) - 1.f; res[i] = x[i] * 0.5f; }
    }
    for (auto& v : w) v = nd(rng) * 0.05f;
    for (int c = 0; c < C; ++c) { sc[c] = 0.5f + 0.01f * c; sh[c] = 0.1f * nd(rng); }
    std::vector<float> pk;
    wino6_pack(w.data(), Cout, C, pk);
    float *dx, *dw, *dy, *dres, *dsc, *dsh;
    double* dst;
    CK(hipMalloc(&dx, (B * fs + 2 * W6_FRONT_PAD + 4096) * 4)); dx += W6_FRONT_PAD;
    CK(hipMalloc(&dw, pk.size() * 4)); CK(hipMalloc(&dy, B * fs * 4)); CK(hipMalloc(&dres, B * fs * 4));
    CK(hipMalloc(&dsc, C * 4)); CK(hipMalloc(&dsh, C * 4));
    const size_t stat_fs = (size_t)NREP * C * 2;
    CK(hipMalloc(&dst, B * stat_fs * 8));
    CK(hipMemcpy(dx, x.data(), B * fs * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dres, res.data(), B * fs * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsc, sc.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsh, sh.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dst, 0, B * stat_fs * 8)); CK(hipMemset(dy, 0xff, B * fs * 4));
    unsigned long long* dbgb; CK(hipMalloc(&dbgb, 96 * 8)); CK(hipMemset(dbgb, 0, 96 * 8));
    ConvP p; memset(&p, 0, sizeof(p));
    p.dbg_buf = dbgb;
    p.in = dx; p.w = dw; p.out = dy; p.res = dres; p.Cin = C; p.Hin = H; p.Win = W; p.Cout = Cout; p.Hout = H; p.Wout = W;
    p.pre = PRE_AFFINE; p.pre_scale = dsc; p.pre_shift = dsh; p.aff_fs = 0; p.stat_acc = dst; p.stat_C = C; p.stat_fs = stat_fs;
    p.dbg = dbg; if (nostat) p.stat_acc = nullptr; if (nores) p.res = nullptr; p.in_fs = fs; p.out_fs = fs; p.res_fs = fs; p.nb = B;
    Variant v = twt == 4 ? make_wino6<4>(false) : twt == 1 ? make_wino6<1>(false) : make_wino6<16>(false);
    p.rx0 = 0; p.ry0 = 0; p.rx1 = W; p.ry1 = H; p.rnbx = (W + v.pw - 1) / v.pw; p.rnby = (H + v.ph - 1) / v.ph;
    const int total = p.rnbx * p.rnby * (Cout / 64) * B;
    int g = 256; if (g > total) g = total; g = (g + 7) & ~7;
    CK(hipFuncSetAttribute((const void*)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(v.kern, dim3(g), dim3(256), v.lds, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<float> y(B * fs); std::vector<double> st(B * stat_fs);
    CK(hipMemcpy(y.data(), dy, B * fs * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(st.data(), dst, B * stat_fs * 8, hipMemcpyDeviceToHost));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(v.kern, dim3(g), dim3(256), v.lds, 0, p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double exec = 2.0 * plane * C * Cout * 2.25 * B;
    printf("nostat %d nores %d diag %d dbg %d %s C=%d %dx%d B=%d grid=%d: %.3f ms/launch, executed %.1f TFLOP/s (%.3f of 157.3), algorithmic %.1f\n", nostat, nores, PP_W6_DIAG, dbg, v.name, C, H, W, B, g, ms, exec / ms * 1e-9,
           exec / ms * 1e-9 / 157.3, exec * 4 / ms * 1e-9);
#if PP_W6_DIAG & 512
    {
        unsigned long long h[96]; CK(hipMemcpy(h, dbgb, sizeof(h), hipMemcpyDeviceToHost));
        for (int wv = 0; wv < 4; ++wv) {
            const unsigned long long* q = h + wv * 24; const double n = (double)q[7], t = (double)q[8];
            printf("  wave %d: per chunk: top %.0f  k-step0 %.0f  k-step1 %.0f  barrier %.0f | per tile: epi sender %.0f  wait+barrier %.0f  receiver %.0f | k-step0 groups:", wv,
                   q[0] / n, q[1] / n, q[2] / n, q[3] / n, q[4] / t, q[5] / t, q[6] / t);
            for (int g = 0; g < 9; ++g) printf(" %.0f", q[9 + g] / n);
            printf(" | receiver up to the stores %.0f | advance: tile change %.0f (x%.0f), else %.0f per chunk", q[18] / t, q[19] / (double)(q[20] ? q[20] : 1), (double)q[20], q[21] / n);
            printf("\n");
        }
    }
#endif
    if (const char* sk = getenv("W6_SOAK")) { // W6_SOAK=<launches>: repeat the launch and compare output + statistics with the first one's
        const int n_soak = atoi(sk);
        float* dref; double* sref; unsigned long long* dres;
        CK(hipMalloc(&dref, B * fs * 4)); CK(hipMalloc(&sref, B * stat_fs * 8)); CK(hipMalloc(&dres, 32));
        CK(hipMemset(dst, 0, B * stat_fs * 8));
        hipLaunchKernelGGL(v.kern, dim3(g), dim3(256), v.lds, 0, p);
        CK(hipMemcpy(dref, dy, B * fs * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(sref, dst, B * stat_fs * 8, hipMemcpyDeviceToDevice));
        int bad_launches = 0;
        for (int it = 0; it < n_soak; ++it) {
            CK(hipMemsetAsync(dst, 0, B * stat_fs * 8, 0));
            CK(hipMemsetAsync(dy, 0xff, B * fs * 4, 0));
            unsigned long long init[4] = {0ull, ~0ull, 0ull, ~0ull};
            CK(hipMemcpyAsync(dres, init, 32, hipMemcpyHostToDevice, 0));
            hipLaunchKernelGGL(v.kern, dim3(g), dim3(256), v.lds, 0, p);
            hipLaunchKernelGGL(soak_cmp, dim3(1024), dim3(256), 0, 0, (const unsigned*)dy, (const unsigned*)dref, B * fs, dres);
            unsigned long long h[4];
            CK(hipMemcpy(h, dres, 32, hipMemcpyDeviceToHost));
            if (h[0]) {
                ++bad_launches;
                const size_t i = (size_t)h[1];
                const size_t fr = i / fs, c = (i % fs) / plane, yy = (i % plane) / W, xx = i % W;
                printf("soak launch %d: %llu output words differ; first at frame %zu channel %zu y %zu x %zu (tile %zu,%zu; wave/M-tile %zu)\n", it, h[0], fr, c, yy, xx, yy / v.ph, xx / v.pw, (c % 64) / 16);
                if (bad_launches <= 3) { // the extent of the damage: per-channel and per-row mismatch counts of the first bad frame
                    std::vector<float> a(fs), r(fs);
                    CK(hipMemcpy(a.data(), dy + fr * fs, fs * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), dref + fr * fs, fs * 4, hipMemcpyDeviceToHost));
                    int x0 = W, x1 = -1, y0 = H, y1 = -1, c0 = C, c1 = -1; size_t nb = 0; double mx = 0;
                    for (size_t q = 0; q < fs; ++q) if (memcmp(&a[q], &r[q], 4)) {
                        const int cc = (int)(q / plane), y2 = (int)((q % plane) / W), x2 = (int)(q % W);
                        x0 = std::min(x0, x2); x1 = std::max(x1, x2); y0 = std::min(y0, y2); y1 = std::max(y1, y2); c0 = std::min(c0, cc); c1 = std::max(c1, cc); ++nb;
                        mx = std::max(mx, (double)fabsf(a[q] - r[q]));
                    }
                    printf("   frame %zu: %zu words, channels %d..%d, rows %d..%d, columns %d..%d, max |diff| %.4g\n", fr, nb, c0, c1, y0, y1, x0, x1, mx);
                    for (int yy2 = y0; yy2 <= std::min(y1, y0 + 7); ++yy2) { printf("   c %d y %d:", c0, yy2); for (int xx2 = x0; xx2 <= std::min(x1, x0 + 7); ++xx2) printf(" %.5g/%.5g", a[(size_t)c0 * plane + yy2 * W + xx2], r[(size_t)c0 * plane + yy2 * W + xx2]); printf("\n"); }
                }
            }
        }
        printf("soak: %d of %d launches differ from the first\n", bad_launches, n_soak);
        return bad_launches ? 3 : 0;
    }
    if (!check) return 0;
    double maxerr = 0; int bad = 0;
    std::vector<int> badmap(plane, 0);
    std::vector<float> xn(fs);
    for (int b = 0; b < B; ++b) {
        for (int c = 0; c < C; ++c) for (size_t i = 0; i < plane; ++i) { float t = __builtin_fmaf(x[b * fs + c * plane + i], sc[c], sh[c]); xn[c * plane + i] = t > 0 ? t : 0; }
        std::vector<double> s(Cout, 0.0), q(Cout, 0.0);
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < H; ++oy) for (int ox = 0; ox < W; ++ox) {
                double a = 0;
                for (int c = 0; c < C; ++c) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
                    const int iy = oy + ky - 1, ix = ox + kx - 1;
                    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                    a += (double)w[((size_t)co * C + c) * 9 + ky * 3 + kx] * xn[c * plane + iy * W + ix];
                }
                a += res[b * fs + co * plane + oy * W + ox];
                const float gv = y[b * fs + co * plane + oy * W + ox];
                const double e = fabs(a - gv);
                if (!(e <= 1e-3)) { if (bad < 12) printf("  bad b=%d co=%d y=%d x=%d ref=%f got=%f\n", b, co, oy, ox, a, gv); ++bad; badmap[oy * W + ox]++; }
                if (e > maxerr || e != e) maxerr = e;
                s[co] += gv; q[co] += (double)gv * gv;
            }
        double serr = 0;
        for (int co = 0; co < Cout; ++co) {
            double gs = 0, gq = 0;
            for (int r = 0; r < NREP; ++r) { gs += st[b * stat_fs + ((size_t)r * C + co) * 2]; gq += st[b * stat_fs + ((size_t)r * C + co) * 2 + 1]; }
            serr = fmax(serr, fmax(fabs(gs - s[co]) / (1 + fabs(s[co])), fabs(gq - q[co]) / (1 + fabs(q[co]))));
        }
        printf("frame %d: max |err| so far %.3e, bad %d, stats rel err %.3e\n", b, maxerr, bad, serr);
    }
    if (bad && H <= 64 && W <= 64)
        for (int oy = 0; oy < H; ++oy) { for (int ox = 0; ox < W; ++ox) printf("%c", badmap[oy * W + ox] ? (badmap[oy * W + ox] > 9 ? '#' : '0' + badmap[oy * W + ox]) : '.'); printf("\n"); }
    printf(bad ? "FAIL\n" : "PASS\n");
    return bad ? 2 : 0;
}
