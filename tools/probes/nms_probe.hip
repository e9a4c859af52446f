// One-off probe: phase timestamps inside nms_tile_topk_kernel (hipcc -DHH_NMS_DEBUG, includes the kernel source).
#define HH_NMS_DEBUG 1
#include "../../pytorch-human-pose_amd/csrc/decode_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
int main(int argc, char **argv)
{
    const float thr = argc > 1 ? (float)atof(argv[1]) : 0.f;  // tile-skipping threshold (0.05 = the bench's det_thr: most tiles leave early)
    const int B = 32, K = 17, H = 512, W = 512, M = 30;
    DecodeSrc src{};
    src.mode = 0; src.B = B; src.K = K; src.H = H; src.W = W; src.E = 1;
    src.scale_h2 = src.scale_w2 = 0.5f; src.scale_h4 = src.scale_w4 = 0.25f;
    size_t n = (size_t)B * K * (H / 2) * (W / 2);
    std::vector<float> h(n);
    unsigned st = 12345;
    for (auto &x : h) { st = st * 1664525u + 1013904223u; x = (float)(st >> 8) / 16777216.f * 0.04f; }
    for (int b = 0; b < B; ++b) for (int k = 0; k < K; ++k) for (int p = 0; p < 10; ++p) {
        st = st * 1664525u + 1013904223u; int y = 8 + (st >> 8) % 240; st = st * 1664525u + 1013904223u; int x = 8 + (st >> 8) % 240;
        for (int dy = -3; dy <= 3; ++dy) for (int dx = -3; dx <= 3; ++dx)
            h[(((size_t)b * K + k) * 256 + y + dy) * 256 + x + dx] += 0.9f * expf(-(dx * dx + dy * dy) / 4.f);
    }
    float *avg; hipMalloc(&avg, n * 4); hipMemcpy(avg, h.data(), n * 4, hipMemcpyHostToDevice);
    src.avg = avg;
    const int nt = ((H + HH_NMS_TILE - 1) / HH_NMS_TILE) * ((W + HH_NMS_TILE - 1) / HH_NMS_TILE);
    unsigned long long *ck; float *cv, *cm;
    hipMalloc(&ck, (size_t)B * K * nt * M * 8); hipMalloc(&cv, (size_t)B * K * nt * M * 4); hipMalloc(&cm, (size_t)B * K * (H / 4) * (W / 4) * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_nms_tile_topk(src, M, ck, cv, cm, thr, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) launch_nms_tile_topk(src, M, ck, cv, cm, thr, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("nms kernel %.1f us per launch, %d tiles per map\n", ms * 100.f, nt);
    std::vector<long long> d(4096 * 8);
    hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(g_nms_dbg), d.size() * 8);
    double sum[8] = {}; int cnt = 0;
    for (int w = 0; w < 4 * nt && w < 4096; ++w) {
        const long long *r = &d[w * 8];
        // (with tile skipping on, an active tile leaves behind stamp 5 and an inactive one behind stamp 1: count the tiles that got
        // at least through the compaction, phase by phase up to their last stamp)
        if (r[0] == 0 || r[5] <= r[0]) continue;
        ++cnt;
        for (int i = 1; i <= 6 && r[i] > r[i - 1]; ++i) sum[i] += (double)(r[i] - r[i - 1]);
    }
    const char *nm[] = {"", "pass A (global->hrow)", "pass B (v)", "cellmax + row pass", "col pass", "nv + compaction", "rank + zeros"};
    double tot = 0;
    for (int i = 1; i <= 6; ++i) { printf("%-24s %8.0f ticks\n", nm[i], sum[i] / cnt); tot += sum[i] / cnt; }
    printf("workgroup life %.0f ticks over %d workgroups (s_memtime ticks)\n", tot, cnt);
    return 0;
}
