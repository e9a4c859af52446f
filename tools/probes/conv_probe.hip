// One-off probe: phase timestamps inside conv_mfma_kernel (hipcc -DHH_CONV_DEBUG, includes the kernel source).
// usage: conv_probe C HW [cfg]   (3x3 stride-1 conv, C -> C channels on a B=32 x HW x HW map, residual + ReLU)
#define HH_CONV_DEBUG 1
#include "../../pytorch-human-pose_amd/csrc/conv_mfma.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 64, HW = argc > 2 ? atoi(argv[2]) : 64, cfg = argc > 3 ? atoi(argv[3]) : 1, B = 32;
    const ConvConfig &c = conv_config(cfg);
    if (conv_init() != hipSuccess) return 1;
    const size_t npx = (size_t)B * HW * HW;
    bf16_raw *in, *out, *res, *w; float *bias;
    hipMalloc(&in, npx * C * 2); hipMalloc(&out, npx * C * 2); hipMalloc(&res, npx * C * 2);
    const int ncg = (C + c.cout_t() - 1) / c.cout_t();
    const size_t wn = (size_t)ncg * (C / c.KC) * 9 * c.KC * c.cout_t();
    hipMalloc(&w, wn * 2); hipMalloc(&bias, ncg * c.cout_t() * 4);
    std::vector<unsigned short> h(npx * C);
    unsigned st = 1;
    for (auto &x : h) { st = st * 1664525u + 1013904223u; x = 0x3c00 + ((st >> 20) & 0xff); }  // small bf16 values
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(res, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::vector<unsigned short> hw(wn); for (auto &x : hw) { st = st * 1664525u + 1013904223u; x = 0x3000 + ((st >> 20) & 0xff); }
    hipMemcpy(w, hw.data(), wn * 2, hipMemcpyHostToDevice); hipMemset(bias, 0, ncg * c.cout_t() * 4);
    ConvParams p{};
    p.in = in; p.in_cs = C; p.Hin = p.Win = HW; p.w = w; p.bias = bias; p.res = res; p.res_cs = C; p.out = out; p.out_cs = C;
    p.Ho = p.Wo = p.Hob = p.Wob = HW; p.osy = p.osx = 1; p.pad_y = p.pad_x = 1; p.cin = C; p.cout_real = p.cout_store = C; p.relu = 1;
    p.B = B; p.tiles_x = (HW + c.TW - 1) / c.TW; p.tiles_y = (HW + c.th() - 1) / c.th(); p.ncg = ncg;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) conv_launch(cfg, p, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) conv_launch(cfg, p, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const unsigned tiles = (unsigned)B * p.tiles_y * p.tiles_x, nwg = ncg > 1 ? (tiles + 7) / 8 * 8 * ncg : tiles;
    const double gflop = 2.0 * npx * C * C * 9 / 1e9;
    printf("cfg %d C=%d %dx%d: %u workgroups, %.1f us per launch back to back (%.0f TFLOP/s), %.2f GFLOP, LDS %zu B\n", cfg, C, HW, HW, nwg,
           ms * 50.f, gflop / (ms * 50e-6) / 1e3, gflop, c.lds_bytes());
    std::vector<long long> d(8192 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_conv_dbg), d.data(), d.size() * 8);
    conv_launch(cfg, p, 0); conv_launch(cfg, p, 0); conv_launch(cfg, p, 0); hipDeviceSynchronize();  // stamps: the third of three back to back
    hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(g_conv_dbg), d.size() * 8);
    const char *nm[] = {"", "setup + issue chunk-0 loads + residual", "chunk-0 data arrives, LDS write, barrier", "MFMA chunk 0 (+ barrier)",
                        "middle chunks + last LDS write", "MFMA last chunk", "epilogue stores issued"};
    double sum[8] = {}; long long t_first = 0x3fffffffffffffffll, t_last = 0, s_last = 0; int cnt = 0;
    for (unsigned wg = 0; wg < nwg && wg < 8192; ++wg) {
        const long long *r = &d[wg * 8];
        if (!r[0]) continue;
        ++cnt;
        for (int i = 1; i <= 6; ++i) sum[i] += (double)(r[i] - r[i - 1]);
        if (r[0] < t_first) t_first = r[0];
        if (r[6] > t_last) t_last = r[6];
        if (r[0] > s_last) s_last = r[0];
    }
    double tot = 0;
    for (int i = 1; i <= 6; ++i) { printf("  %-44s %8.0f ticks\n", nm[i], sum[i] / cnt); tot += sum[i] / cnt; }
    printf("  workgroup life %.0f ticks (%d workgroups)\n", tot, cnt);
    // start / end times on the chip-wide 100 MHz clock (s_memtime counters are per XCD and not aligned)
    long long w0 = 0x3fffffffffffffffll, w1 = 0, ws = 0;
    for (unsigned wg = 0; wg < nwg && wg < 8192; ++wg) {
        const long long a = d[wg * 8 + 7];
        if (!a) continue;
        if (a < w0) w0 = a;
        if (a > ws) ws = a;
    }
    int hist[16] = {};
    for (unsigned wg = 0; wg < nwg && wg < 8192; ++wg) if (d[wg * 8 + 7]) hist[(int)((d[wg * 8 + 7] - w0) * 16 / (ws - w0 + 1))]++;
    printf("  last workgroup started %.2f us after the first; starts per sixteenth of that window:", (ws - w0) / 100.0);
    for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
    printf("\n  mean start (us) by dispatch order, blocks 0-63, 64-127, ...:");
    for (unsigned g0 = 0; g0 < nwg && g0 < 1024; g0 += 64) { double m = 0; for (unsigned wg = g0; wg < g0 + 64; ++wg) m += (d[wg * 8 + 7] - w0) / 100.0; printf(" %.2f", m / 64); }
    printf("\n");
    (void)w1;
    {   // inter-kernel gap: each of 12 back-to-back launches gets its own {first start, last end} slot (100 MHz chip clock)
        unsigned long long *clk; hipMalloc(&clk, 24 * 8);
        std::vector<unsigned long long> hc(24);
        for (int i = 0; i < 12; ++i) { hc[2 * i] = ~0ull; hc[2 * i + 1] = 0; }
        hipMemcpy(clk, hc.data(), 24 * 8, hipMemcpyHostToDevice);
        for (int i = 0; i < 12; ++i) { ConvParams q = p; q.clk = clk + 2 * i; conv_launch(cfg, q, 0); }
        hipDeviceSynchronize();
        hipMemcpy(hc.data(), clk, 24 * 8, hipMemcpyDeviceToHost);
        printf("  per launch: busy (first start -> last end) / gap to the next launch's first start, us:");
        for (int i = 2; i < 11; ++i) printf("  %.2f/%.2f", (hc[2 * i + 1] - hc[2 * i]) / 100.0, (hc[2 * i + 2] - hc[2 * i + 1]) / 100.0);
        printf("\n");
    }
    int wavg = 0; hipDeviceGetAttribute(&wavg, hipDeviceAttributeClockRate, 0); printf("  (device clock attribute %d kHz)\n", wavg);
    return 0;
}
