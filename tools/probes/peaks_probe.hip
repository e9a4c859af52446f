// One-off probe: launch time and phase timestamps of peaks_region_kernel on bench-like maps (B = 32, K = 17, 512 x 512).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DHH_PEAKS_DEBUG [-DPEAKS_WPS=n] peaks_probe.hip -o peaks_probe
//   ./peaks_probe [people per image]
#include "../../pytorch-human-pose_amd/csrc/decode_peaks.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 10;
    const int B = 32, K = 17, H = 512, W = 512, M = 30, hq = H / 4, wq = W / 4, hh = H / 2, wh = W / 2;
    DecodeSrc src{};
    src.mode = 0; src.B = B; src.K = K; src.H = H; src.W = W; src.E = 1;
    src.scale_h2 = src.scale_w2 = 0.5f; src.scale_h4 = src.scale_w4 = 0.25f;
    const size_t nq = (size_t)B * K * hq * wq, nh = (size_t)B * K * hh * wh;
    std::vector<float> q(nq), h(nh);
    unsigned st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)(st >> 8) / 16777216.f; };
    for (auto &x : q) x = rnd() * 0.02f;
    for (auto &x : h) x = rnd() * 0.02f;
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < P; ++p) {
            const float cx = (0.15f + 0.7f * rnd()) * wq, cy = (0.15f + 0.7f * rnd()) * hq;
            for (int k = 0; k < K; ++k) {
                if (rnd() < 0.15f) continue;
                const float x = std::min(std::max(cx + (rnd() * 0.24f - 0.12f) * wq, 2.f), wq - 3.f), y = std::min(std::max(cy + (rnd() * 0.24f - 0.12f) * hq, 2.f), hq - 3.f);
                const float amp = 0.5f + 0.5f * rnd();
                for (int dy = -8; dy <= 8; ++dy)
                    for (int dx = -8; dx <= 8; ++dx) {
                        const int yy = (int)y + dy, xx = (int)x + dx;
                        if (yy < 0 || yy >= hq || xx < 0 || xx >= wq) continue;
                        float &o = q[(((size_t)b * K + k) * hq + yy) * wq + xx];
                        o = std::max(o, amp * expf(-((xx - x) * (xx - x) + (yy - y) * (yy - y)) / 8.f));
                    }
                for (int dy = -16; dy <= 16; ++dy)
                    for (int dx = -16; dx <= 16; ++dx) {
                        const int yy = (int)(2 * y) + dy, xx = (int)(2 * x) + dx;
                        if (yy < 0 || yy >= hh || xx < 0 || xx >= wh) continue;
                        float &o = h[(((size_t)b * K + k) * hh + yy) * wh + xx];
                        o = std::max(o, amp * expf(-((xx - 2 * x - 0.5f) * (xx - 2 * x - 0.5f) + (yy - 2 * y - 0.5f) * (yy - 2 * y - 0.5f)) / 32.f));
                    }
            }
        }
    float *dq, *dh, *cm, *flush;
    unsigned long long *ck;
    hipMalloc(&dq, nq * 4); hipMalloc(&dh, nh * 4);
    hipMemcpy(dq, q.data(), nq * 4, hipMemcpyHostToDevice); hipMemcpy(dh, h.data(), nh * 4, hipMemcpyHostToDevice);
    src.hm_q = dq; src.hm_q_bs = (int64_t)K * hq * wq; src.hm_h = dh; src.hm_h_bs = (int64_t)K * hh * wh;
    const int nreg = peaks_regions(H, W);
    hipMalloc(&ck, (size_t)B * K * nreg * M * 8); hipMalloc(&cm, (size_t)B * K * hq * wq * 4);
    const size_t fl = 512u << 20;  // a 512 MiB fill between launches: the sources come from HBM, as behind a forward pass
    hipMalloc(&flush, fl);
    int *ctr; hipMalloc(&ctr, HH_PEAKS_PARTS * 4);
    unsigned short *sup; hipMalloc(&sup, (size_t)B * K * 1024 * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float tot = 0.f, warm = 0.f;
    for (int i = 0; i < 8; ++i) {
        hipMemsetAsync(flush, i, fl, 0);
        hipMemsetAsync(ctr, 0, HH_PEAKS_PARTS * 4, 0);
        hipEventRecord(e0, 0);
        launch_peaks(src, M, ck, cm, sup, 0.05f, ctr, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i >= 3) tot += ms;
    }
    for (int i = 0; i < 5; ++i) {
        hipMemsetAsync(ctr, 0, HH_PEAKS_PARTS * 4, 0);
        hipEventRecord(e0, 0);
        launch_peaks(src, M, ck, cm, sup, 0.05f, ctr, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        warm += ms;
    }
    std::vector<unsigned long long> keys((size_t)B * K * nreg * M);
    hipMemcpy(keys.data(), ck, keys.size() * 8, hipMemcpyDeviceToHost);
    size_t ncand = 0;
    for (auto kk : keys) ncand += kk != 0;
    printf("peaks kernel: %.1f us cold (after a 512 MiB fill), %.1f us back to back; %d people: %.1f candidates per map\n", tot / 5 * 1e3f, warm / 5 * 1e3f, P,
           (double)ncand / (B * K));
#ifdef HH_PEAKS_DEBUG
    std::vector<long long> d(4096 * 8);
    hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(g_peaks_dbg), d.size() * 8);
    const char *nm[] = {"", "loads + average", "ticket + barrier", "cell bounds + mask", "sub-tiles", "", "select + end barrier"};
    for (int act = 0; act < 2; ++act) {  // regions without / with remaining sub-tiles
        double sum[8] = {};
        int cnt = 0;
        for (int w = 0; w < 4096; ++w) {
            const long long *r = &d[w * 8];
            if (r[0] == 0 || r[6] <= r[0]) continue;
            if ((r[4] > r[3]) != (act == 1)) continue;
            ++cnt;
            sum[1] += r[1] - r[0]; sum[2] += r[2] - r[1]; sum[3] += r[3] - r[2];
            if (act) { sum[4] += r[4] - r[3]; sum[6] += r[6] - r[4]; } else sum[6] += r[6] - r[3];
        }
        if (!cnt) continue;
        printf("%s regions (%d sampled), s_memtime ticks per region:\n", act ? "active" : "finished-early", cnt);
        double t = 0;
        for (int i = 1; i <= 6; ++i)
            if (nm[i][0]) { printf("  %-28s %8.0f\n", nm[i], sum[i] / cnt); t += sum[i] / cnt; }
        printf("  %-28s %8.0f\n", "total", t);
    }
#endif
    return 0;
}
