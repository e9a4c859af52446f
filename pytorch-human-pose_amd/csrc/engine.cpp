// Plan builder + executor for HigherHRNet (see engine.h).  The layer graph restates
// /root/reference/src/keypoints/architectures/{hrnet.py:342-385, higher_hrnet.py:47-81} as
// a flat list of fused launches; parameter names are the reference's state-dict keys.
#include "engine.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

static thread_local std::string g_err;
LaunchProbe &hh_launch_probe()
{
    static thread_local LaunchProbe p;
    return p;
}
void hh_set_error(const std::string &msg) { g_err = msg; }
const char *hh_get_error() { return g_err.c_str(); }

PlanSwitches PlanSwitches::from_env()
{
    auto on = [](const char *name) { const char *v = getenv(name); return v && *v && strcmp(v, "0"); };
    auto is = [](const char *name, const char *val) { const char *v = getenv(name); return v && !strcmp(v, val); };
    PlanSwitches s;
    s.bb32_tile = is("HH_BB32", "tile");
    s.no_bb64 = on("HH_NO_BB64");
    s.no_bb_fp8 = on("HH_NO_BB_FP8");
    s.no_stem_fused = on("HH_NO_STEM_FUSED");
    s.no_junc_pair = on("HH_NO_JUNC_PAIR");
    s.full_join = on("HH_FULL_JOIN");
    s.no_fusion_merge = on("HH_NO_FUSION_MERGE");
    s.poison_ws = on("HH_POISON_WS");
    s.poison_lds = on("HH_POISON_LDS");
    s.no_head_fold = on("HH_NO_HEAD_FOLD");
    s.no_conv_db = on("HH_NO_CONV_DB");
    s.no_final_fuse = on("HH_NO_FINAL_FUSE");
    s.keep_waits = on("HH_KEEP_WAITS");
    s.event_system_fence = on("HH_EVENT_SYSTEM_FENCE");
    s.bb_tall = on("HH_NO_BB_TALL") ? 0 : is("HH_BB_TALL", "always") ? 2 : 1;
    if (const char *fc = getenv("HH_FAT_CUS")) { s.fat_cus = s.fat_cus64 = atoi(fc); if (const char *c2 = strchr(fc, ',')) s.fat_cus64 = atoi(c2 + 1); }
    if (const char *sk = getenv("HH_DEBUG_SKIP")) {
        static const struct { const char *name; unsigned bit; } cats[] = {{"s2big", SK_S2BIG}, {"s2", SK_S2}, {"upadd", SK_UPADD}, {"c1x1", SK_C1X1},
            {"c256", SK_C256}, {"c128", SK_C128}, {"junc", SK_JUNC}, {"bb32", SK_BB32}, {"bb64", SK_BB64}, {"stem", SK_STEM}, {"deconv", SK_DECONV},
            {"head", SK_HEAD}, {"trans0", SK_TRANS0}};
        std::string v = std::string(",") + sk + ",";
        for (const auto &c : cats)
            if (v.find(std::string(",") + c.name + ",") != std::string::npos) s.debug_skip |= c.bit;
    }
    s.fp8_trunk8 = is("HH_FP8_TRUNK", "e4m3");
    s.fp8_heads8 = is("HH_FP8_HEADS", "e4m3");
    return s;
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

static inline bf16_raw f2bf(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_raw)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_raw)(u >> 16);
}
static inline float bf2f(bf16_raw h)
{
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// ------------------------------------------------------------------------- parameters
int hh_net::add_param(const std::string &name, std::vector<int64_t> shape, bool counter)
{
    ParamSlot p;
    p.name = name;
    p.shape = std::move(shape);
    p.counter = counter;
    param_index[name] = (int)params.size();
    params.push_back(std::move(p));
    return (int)params.size() - 1;
}

namespace {
struct Builder {
    hh_net &n;
    int lane = 0;  // lane given to the ops created next
    explicit Builder(hh_net &net) : n(net) {}
    void join(int nlanes)
    {
        Op o;
        o.kind = OP_JOIN; o.nlanes = nlanes;
        n.ops.push_back(o);
    }

    // mark / waitl: a join taken apart.  mark(n) notes where every lane stands; waitl(lane, src) makes `lane` wait for lane
    // `src`'s position at the last mark only -- a fusion output starts on the branches that are already done instead of waiting
    // for the slowest one.
    void mark(int nlanes, int first = 0)  // lanes [first, nlanes) note their positions
    {
        Op o;
        o.kind = OP_MARK; o.nlanes = nlanes; o.dep_from = first;
        n.ops.push_back(o);
    }
    void waitl(int lane_, int src)
    {
        Op o;
        o.kind = OP_WAITL; o.lane = lane_; o.dep_from = src;
        n.ops.push_back(o);
    }

    void dep(int from, int to)
    {
        Op o;
        o.kind = OP_DEP; o.dep_from = from; o.lane = to;
        n.ops.push_back(o);
    }

    // ---- parameter registration in the reference's state_dict order
    void p_conv(const std::string &name, int cin, int cout, int k, bool bias = false)
    {
        n.add_param(name + ".weight", {cout, cin, k, k});
        if (bias) n.add_param(name + ".bias", {cout});
    }
    void p_bn(const std::string &name, int c)
    {
        n.add_param(name + ".weight", {c});
        n.add_param(name + ".bias", {c});
        n.add_param(name + ".running_mean", {c});
        n.add_param(name + ".running_var", {c});
        n.add_param(name + ".num_batches_tracked", {}, true);
    }

    void register_params()
    {
        const int C = n.C, K = n.K;
        const int w[4] = {C, 2 * C, 4 * C, 8 * C};
        const std::string bb = "backbone";
        p_conv(bb + ".conv1", 3, 64, 3); p_bn(bb + ".bn1", 64);
        p_conv(bb + ".conv2", 64, 64, 3); p_bn(bb + ".bn2", 64);
        const int nblocks[4] = {1, 1, 4, 3};
        for (int s = 0; s < 4; ++s) {
            const std::string sp = bb + ".stages." + std::to_string(s);
            const int nsc = s == 0 ? 1 : s + 1;
            for (int b = 0; b < nblocks[s]; ++b) {
                const std::string hp = sp + ".blocks." + std::to_string(2 * b);
                for (int i = 0; i < nsc; ++i)
                    for (int u = 0; u < 4; ++u) {
                        const std::string up = hp + ".scales_blocks." + std::to_string(i) + "." + std::to_string(u);
                        if (s == 0) {
                            const int cin = u == 0 ? 64 : 256;
                            p_conv(up + ".conv1", cin, 64, 1); p_bn(up + ".bn1", 64);
                            p_conv(up + ".conv2", 64, 64, 3); p_bn(up + ".bn2", 64);
                            p_conv(up + ".conv3", 64, 256, 1); p_bn(up + ".bn3", 256);
                            if (u == 0) { p_conv(up + ".downsample.0", 64, 256, 1); p_bn(up + ".downsample.1", 256); }
                        } else {
                            p_conv(up + ".conv1", w[i], w[i], 3); p_bn(up + ".bn1", w[i]);
                            p_conv(up + ".conv2", w[i], w[i], 3); p_bn(up + ".bn2", w[i]);
                        }
                    }
                if (s > 0) {
                    const std::string fp = sp + ".blocks." + std::to_string(2 * b + 1);
                    const bool last = s == 3 && b == nblocks[s] - 1;
                    const int nout = (last && n.kind == 0) ? 1 : nsc;  // final_stage_single_scale (hrnet.py:314-318)
                    for (int i = 0; i < nout; ++i)
                        for (int j = 0; j < nsc; ++j) {
                            const std::string lp = fp + ".scales_fusion_layers." + std::to_string(i) + "." + std::to_string(j);
                            if (j > i) { p_conv(lp + ".0", w[j], w[i], 1); p_bn(lp + ".1", w[i]); }
                            else if (j < i)
                                for (int k = 0; k < i - j; ++k) {
                                    const int co = k == i - j - 1 ? w[i] : w[j];
                                    p_conv(lp + "." + std::to_string(k) + ".0", w[j], co, 3);
                                    p_bn(lp + "." + std::to_string(k) + ".1", co);
                                }
                        }
                }
            }
            if (s < 3) {
                const std::string tp = sp + ".transition_layer.transition_blocks";
                if (s == 0) {
                    p_conv(tp + ".0.0", 256, w[0], 3); p_bn(tp + ".0.1", w[0]);
                    p_conv(tp + ".1.0", 256, w[1], 3); p_bn(tp + ".1.1", w[1]);
                } else {
                    const std::string q = tp + "." + std::to_string(nsc);
                    p_conv(q + ".0", w[nsc - 1], w[nsc], 3); p_bn(q + ".1", w[nsc]);
                }
            }
        }
        if (n.kind == 1) {  // ClassificationHead (classification/architectures/hrnet.py:7-46)
            const int outs[4] = {128, 256, 512, 1024};
            const std::string hp = "classification_head";
            for (int i = 0; i < 4; ++i) {
                const std::string up = hp + ".chann_incr_blocks." + std::to_string(i);
                const int mid = outs[i] / 4;
                p_conv(up + ".conv1", w[i], mid, 1); p_bn(up + ".bn1", mid);
                p_conv(up + ".conv2", mid, mid, 3); p_bn(up + ".bn2", mid);
                p_conv(up + ".conv3", mid, outs[i], 1); p_bn(up + ".bn3", outs[i]);
                if (w[i] != outs[i]) { p_conv(up + ".downsample.0", w[i], outs[i], 1); p_bn(up + ".downsample.1", outs[i]); }
            }
            for (int i = 0; i < 3; ++i) {
                const std::string dpn = hp + ".downsample_blocks." + std::to_string(i);
                p_conv(dpn + ".0", outs[i], outs[i + 1], 3, true); p_bn(dpn + ".1", outs[i + 1]);
            }
            p_conv(hp + ".final_conv.0", 1024, 2048, 1, true); p_bn(hp + ".final_conv.1", 2048);
            n.add_param(hp + ".classifier.weight", {n.num_classes, 2048});
            n.add_param(hp + ".classifier.bias", {n.num_classes});
            return;
        }
        p_conv("init_heatmaps_head", C, 2 * K, 1, true);
        const std::string dp = "deconv_layers.0";
        n.add_param(dp + ".deconv.0.weight", {C + 2 * K, C, 4, 4});
        p_bn(dp + ".deconv.1", C);
        for (int r = 0; r < 4; ++r) {
            const std::string rp = dp + ".resid_blocks." + std::to_string(r);
            p_conv(rp + ".conv1", C, C, 3); p_bn(rp + ".bn1", C);
            p_conv(rp + ".conv2", C, C, 3); p_bn(rp + ".bn2", C);
        }
        p_conv(dp + ".final_layer", C, K, 1, true);
    }

    // ---- plan helpers
    int T(int C, int shift, bool zero = false)
    {
        TensorDesc t;
        t.C = C; t.shift = shift; t.zero_init = zero;
        n.tensors.push_back(t);
        return (int)n.tensors.size() - 1;
    }
    int L(const std::string &conv, const std::string &bn, int cin, int cout, int ks, int stride, const std::string &bias = "")
    {
        ConvLayer l;
        l.conv = conv; l.bn = bn; l.bias = bias; l.cin = cin; l.cout = cout; l.ks = ks; l.stride = stride;
        n.layers.push_back(l);
        return (int)n.layers.size() - 1;
    }
    Op &conv(int layer, int in, int out, int relu, int res = -1)
    {
        Op o;
        o.kind = OP_CONV; o.layer = layer; o.in = in; o.out = out; o.relu = relu; o.res = res; o.lane = lane;
        n.ops.push_back(o);
        return n.ops.back();
    }
    // conv + BN named "<p>.<c>" / "<p>.<b>"
    Op &cb(const std::string &p, const char *c, const char *b, int cin, int cout, int ks, int stride, int in, int out,
           int relu, int res = -1)
    {
        return conv(L(p + "." + c, p + "." + b, cin, cout, ks, stride), in, out, relu, res);
    }
    // four BasicBlocks "<prefix>.<u>" on tensor x with scratch m (both C channels); result ends in x
    // `nscales` = number of resolution branches running beside this one (0: no siblings, e.g. the deconv head)
    void basic_blocks(const std::string &prefix, int C, int x, int m, int nscales = 0, int branch = 0)
    {
        // (A fused 128-channel block existed in rounds 2-3, 25.7 us per block against 2 x 17.7 us layer by layer when run alone; with
        // the branch lanes side by side it LOST every A/B -- round 3, one box, three alternations: forward 4.72 / 4.75 / 4.73 ms
        // without, 4.86 / 4.85 / 4.83 ms with it, profiles/r03_ab.md -- because its 150 KB workgroups own their CUs while the
        // layer-by-layer launches share CUs with the 256-channel branch.  Removed; `git log` has it.)
        for (int u = 0; u < 4; ++u) {
            const std::string up = prefix + "." + std::to_string(u);
            const bool fused_fp8 = n.dtype == 2 && bb_fp8_supported(C) && branch == 0 && !n.sw.no_bb_fp8;  // highest-resolution branch / deconv head
            if (((C == 32 || (C == 64 && !n.sw.no_bb64)) && n.dtype != 2) || fused_fp8) {  // fused kernels, ping-pong x <-> m (a tile reads its neighbours' halo: no in-place)
                Op o;
                o.kind = OP_BB;
                o.layer = L(up + ".conv1", up + ".bn1", C, C, 3, 1);
                o.layer2 = L(up + ".conv2", up + ".bn2", C, C, 3, 1);
                o.in = (u & 1) ? m : x;
                o.out = (u & 1) ? x : m;
                o.lane = lane;
                o.siblings = nscales > 1;  // other branches' blocks run on their lanes beside this one
                n.ops.push_back(o);
            } else {
                cb(up, "conv1", "bn1", C, C, 3, 1, x, m, 1);
                cb(up, "conv2", "bn2", C, C, 3, 1, m, x, 1, x);
            }
        }
    }
    void tap(const std::string &name, int tensor, int C, int coff = 0)
    {
        TapInfo t;
        t.name = name; t.tensor = tensor; t.coff = coff; t.C = C;
        n.taps.push_back(t);
        Op o;
        o.kind = OP_TAP; o.tap = (int)n.taps.size() - 1; o.lane = lane;
        n.ops.push_back(o);
    }

    void build_plan()
    {
        const int C = n.C, K = n.K;
        const int w[4] = {C, 2 * C, 4 * C, 8 * C};
        const std::string bb = "backbone";
        // stem (hrnet.py:354-358,378-384)
        // bf16: both stem convolutions in ONE kernel (stem_fused.hip), the half-resolution intermediate never leaves the CU
        // (HH_NO_STEM_FUSED=1 and the fp8 path: two launches through the tensor S1)
        const bool stem_fused = n.dtype != 2 && !n.sw.no_stem_fused;
        const int S1 = stem_fused ? -1 : T(64, 1), X = T(64, 2);
        {   // conv1 reads the fp32 NCHW images itself (stem_conv.hip): no layout pass, no padded input channels
            Op o;
            o.kind = OP_STEM; o.layer = L(bb + ".conv1", bb + ".bn1", 3, 64, 3, 2); o.out = stem_fused ? X : S1;
            n.layers[o.layer].stem = true;
            if (stem_fused) {
                o.layer2 = L(bb + ".conv2", bb + ".bn2", 64, 64, 3, 2);
                n.layers[o.layer2].stem2 = true;
            }
            n.ops.push_back(o);
        }
        if (!stem_fused) conv(L(bb + ".conv2", bb + ".bn2", 64, 64, 3, 2), S1, X, 1);

        // stage 0: four Bottlenecks on one scale (hrnet.py:29-74), then the 256->C / 256->2C transition
        // bf16: the junctions work in pairs -- units 0 and 2 do not store their 256-channel y, units 1 and 3 make it again per pixel
        // from the previous unit's t2 (two more 1x1 GEMMs) -- so conv2 alternates between two t2 tensors
        // (HH_NO_JUNC_PAIR=1: every junction stores y and the next one reads it)
        const bool jpair = n.dtype != 2 && !n.sw.no_junc_pair;
        const int t1 = T(64, 2), t2 = T(64, 2), t2b = jpair ? T(64, 2) : t2, Y = T(256, 2);
        // conv3 (+ downsample) of unit u and conv1 of unit u+1 are both 1x1: one junction kernel makes y and the next t1 in a
        // single pass over the 256-channel tensor (bottleneck_junction.hip)
        auto unit = [&](int u) { return bb + ".stages.0.blocks.0.scales_blocks.0." + std::to_string(u); };
        cb(unit(0), "conv1", "bn1", 64, 64, 1, 1, X, t1, 1);
        const int DS = n.dtype == 2 ? T(256, 2) : -1;  // fp8 path: the downsample branch of unit 0 is a tensor of its own
        int prev_conv3 = -1, ds_layer = -1;
        for (int u = 0; u < 4; ++u) {
            const std::string up = unit(u);
            const int t2u = (u & 1) ? t2b : t2, t2p = (u & 1) ? t2 : t2b;  // this unit's / the previous unit's conv2 output
            cb(up, "conv2", "bn2", 64, 64, 3, 1, t1, t2u, 1);
            if (n.dtype == 2) {  // layer by layer (hrnet.py:58-74): downsample, conv3 + residual + ReLU, next unit's conv1
                if (u == 0) cb(up, "downsample.0", "downsample.1", 64, 256, 1, 1, X, DS, 0);
                cb(up, "conv3", "bn3", 64, 256, 1, 1, t2, Y, 1, u == 0 ? DS : Y).res8 = true;
                if (u < 3) cb(unit(u + 1), "conv1", "bn1", 256, 64, 1, 1, Y, t1, 1);
                continue;
            }
            Op o;
            o.kind = OP_JUNC; o.lane = lane;
            o.layer2 = u == 0 ? L(up + ".downsample.0", up + ".downsample.1", 64, 256, 1, 1) : -1;
            o.layer = L(up + ".conv3", up + ".bn3", 64, 256, 1, 1);
            o.layer3 = u < 3 ? L(unit(u + 1) + ".conv1", unit(u + 1) + ".bn1", 256, 64, 1, 1) : -1;
            o.in = t2u; o.in2 = u == 0 ? X : -1; o.res = u == 0 ? -1 : Y;
            o.out = Y; o.out2 = u < 3 ? t1 : -1;
            if (jpair && !(u & 1)) o.out = -1;  // units 0, 2: y is not stored
            if (jpair && (u & 1)) {             // units 1, 3: the previous unit's y from its t2 (and, for unit 1, the downsample of x)
                o.layer4 = prev_conv3; o.in3 = t2p;
                if (u == 1) { o.layer2 = ds_layer; o.in2 = X; o.res = -1; }
            }
            if (u == 0) ds_layer = o.layer2;
            prev_conv3 = o.layer;
            n.ops.push_back(o);
        }
        tap("stem#0", X, 64);
        tap("stages.0.blocks.0#0", Y, 256);
        tap("stages.0.blocks.1#0", Y, 256);  // single-scale fusion = ReLU of a ReLU output
        int x[4] = {-1, -1, -1, -1}, m[4], f[4];
        for (int i = 0; i < 4; ++i) { m[i] = T(w[i], 2 + i); f[i] = T(w[i], 2 + i); }
        x[0] = T(w[0], 2); x[1] = T(w[1], 3);
        {
            const std::string tp = bb + ".stages.0.transition_layer.transition_blocks";
            join(2);
            lane = 0; cb(tp + ".0", "0", "1", 256, w[0], 3, 1, Y, x[0], 1);
            lane = 1; cb(tp + ".1", "0", "1", 256, w[1], 3, 2, Y, x[1], 1);
            lane = 0;
        }
        tap("stages.0#0", x[0], w[0]);
        tap("stages.0#1", x[1], w[1]);

        // bf16 handles fold init_heatmaps_head into the transposed conv (ConvLayer::fold_w): the buffer is [feats | 1 | zero pad] and
        // the 1x1 head leaves the chain of dependent launches (it runs beside the deconv head on lane 1)
        const bool head_fold = n.dtype != 2 && n.kind == 0 && !n.sw.no_head_fold;
        const int catC = head_fold ? round_up(C + 1, 16) : round_up(C + 2 * K, 16);
        const int CAT = T(catC, 2, true);  // [feats | init heatmaps | zero pad] = torch.cat of higher_hrnet.py:73
        n.tensors[CAT].shared_scale = true;
        if (head_fold) n.tensors[CAT].ones_channel = C;

        const int nblocks[4] = {1, 1, 4, 3};
        for (int s = 1; s < 4; ++s) {
            const int nsc = s + 1;
            const std::string sp = bb + ".stages." + std::to_string(s);
            for (int b = 0; b < nblocks[s]; ++b) {
                // HighResolutionBlock: 4 BasicBlocks per scale (hrnet.py:77-124,154-163); conv2 adds the
                // identity and writes in place (each lane reads the residual of the pixel it overwrites).
                const std::string hp = sp + ".blocks." + std::to_string(2 * b);
                for (int i = 0; i < nsc; ++i) {  // branches are independent (hrnet.py:154-163): one lane each
                    lane = i;
                    basic_blocks(hp + ".scales_blocks." + std::to_string(i), w[i], x[i], m[i], nsc, i);
                }
                lane = 0;
                // every fusion output reads every branch -- but not all of them at once: the lanes note their positions and
                // every output waits, source by source, right in front of the first launch that needs the source
                // (HH_FULL_JOIN=1: an all-to-all join here instead)
                const bool fine = !n.sw.full_join;
                // (Measured and dropped in round 4, profiles/r04_ab.md 8.3: the slowest lane waiting in front of its last block launch,
                // +0.1 ms; a chain join with one wait per lane instead of source-by-source waits, +0.12 ms -- the early starts on the
                // sources that are already done are worth more than the waits they cost; the lane above the lowest one handing it ONE
                // event for all higher lanes: neutral.)
                if (fine) mark(nsc); else join(nsc);
                for (int i = 0; i < nsc; ++i)
                    tap("stages." + std::to_string(s) + ".blocks." + std::to_string(2 * b) + "#" + std::to_string(i), x[i], w[i]);
                // FusionLayer (hrnet.py:166-229)
                const std::string fp = sp + ".blocks." + std::to_string(2 * b + 1);
                const bool last = s == 3 && b == nblocks[s] - 1 && n.kind == 0;  // HigherHRNet keeps one scale
                const int nout = last ? 1 : nsc;
                for (int i = 0; i < nout; ++i) {
                    lane = i;  // output i only writes its own tensors; it feeds branch i of the next block directly
                    const int OUT = last ? CAT : f[i];
                    int cur = x[i];
                    bool waited[4] = {false, false, false, false};
                    waited[i] = true;
                    auto need = [&](int j) { if (fine && !waited[j]) { waitl(i, j); waited[j] = true; } };
                    // OUT is the x[i] of the previous HR block, which the other lanes' fusion launches read then: before the first
                    // write to it this lane must be behind every other lane's mark (which is behind those reads)
                    auto need_all = [&]() { for (int j = 0; j < nsc; ++j) need(j); };
                    Op up;
                    up.kind = OP_UPADD; up.in = x[i]; up.out = OUT; up.C = w[i]; up.relu = (i == 0); up.lane = lane;
                    for (int j = i + 1; j < nsc; ++j) {  // low -> high: 1x1 conv + BN at low res
                        const std::string lp = fp + ".scales_fusion_layers." + std::to_string(i) + "." + std::to_string(j);
                        const int u = T(w[i], 2 + j);
                        need(j);
                        cb(lp, "0", "1", w[j], w[i], 1, 1, x[j], u, 0);
                        up.up[up.nup] = u; up.up_shift[up.nup] = j - i; ++up.nup;
                    }
                    if (up.nup) { need_all(); n.ops.push_back(up); cur = OUT; }
                    // high -> low, two or more sources (bf16): the LAST stride-2 conv of every chain reads at the resolution above the
                    // output's, so the sum of those convs is one conv over the concatenated channels -- one launch instead of i
                    // dependent ones on the lane that already has the longest chain.  Its inputs must share a pixel stride: chain
                    // intermediates are allocated with the stride of branch i-1.  (HH_NO_FUSION_MERGE=1: one launch per source)
                    if (i >= 2 && n.dtype != 2 && !n.sw.no_fusion_merge) {
                        int srcs[3] = {-1, -1, -1};
                        ConvLayer ml;
                        ml.cout = w[i]; ml.ks = 3; ml.stride = 2; ml.cin = 0;
                        int tins[3] = {-1, -1, -1};
                        for (int j = 0; j < i; ++j) {  // the chains, highest resolution first: those branches are done first
                            const std::string lp = fp + ".scales_fusion_layers." + std::to_string(i) + "." + std::to_string(j);
                            int tin = x[j];
                            if (j < i - 1) need(j);  // (branch i-1 feeds the merged conv directly: waited for in need_all below)
                            for (int k = 0; k < i - j - 1; ++k) {
                                const bool last_tmp = k == i - j - 2;
                                const int tmp = T(last_tmp ? w[i - 1] : w[j], 2 + j + k + 1);  // (stride of branch i-1 for the merged conv's inputs)
                                cb(lp + "." + std::to_string(k), "0", "1", w[j], w[j], 3, 2, tin, tmp, 1);
                                tin = tmp;
                            }
                            tins[j] = tin;
                        }
                        for (int j = i - 1; j >= 0; --j) {  // source order of the merged conv: branch i-1 itself first, then the chains
                            const std::string lp = fp + ".scales_fusion_layers." + std::to_string(i) + "." + std::to_string(j);
                            srcs[i - 1 - j] = tins[j];
                            ml.mconv.push_back(lp + "." + std::to_string(i - j - 1) + ".0");
                            ml.mbn.push_back(lp + "." + std::to_string(i - j - 1) + ".1");
                            ml.mcin.push_back(w[j]);
                            ml.cin += w[j];
                        }
                        ml.conv = ml.mconv[0]; ml.bn = ml.mbn[0];
                        n.layers.push_back(ml);
                        need_all();
                        Op &o = conv((int)n.layers.size() - 1, srcs[0], OUT, 1, cur);
                        o.in2 = srcs[1]; o.in3 = srcs[2];
                        cur = OUT;
                        continue;
                    }
                    for (int j = 0; j < i; ++j) {  // high -> low: chain of stride-2 convs, summed in the last epilogue
                        const std::string lp = fp + ".scales_fusion_layers." + std::to_string(i) + "." + std::to_string(j);
                        int tin = x[j];
                        need(j);
                        for (int k = 0; k < i - j - 1; ++k) {
                            const int tmp = T(w[j], 2 + j + k + 1);
                            cb(lp + "." + std::to_string(k), "0", "1", w[j], w[j], 3, 2, tin, tmp, 1);
                            tin = tmp;
                        }
                        need_all();
                        cb(lp + "." + std::to_string(i - j - 1), "0", "1", w[j], w[i], 3, 2, tin, OUT, j == i - 1, cur);
                        cur = OUT;
                    }
                }
                lane = 0;
                if (!last)
                    for (int i = 0; i < nout; ++i) std::swap(x[i], f[i]);
                for (int i = 0; i < nout; ++i)
                    tap("stages." + std::to_string(s) + ".blocks." + std::to_string(2 * b + 1) + "#" + std::to_string(i),
                        last ? CAT : x[i], w[i]);
            }
            if (s < 3) {  // TransitionLayer: only the new lowest branch has parameters (hrnet.py:262-283)
                const std::string q = sp + ".transition_layer.transition_blocks." + std::to_string(nsc);
                x[nsc] = T(w[nsc], 2 + nsc);
                // the new lane needs branch nsc-1's fusion output only, the others go straight on to the next stage's blocks
                if (!n.sw.full_join) { mark(nsc + 1); waitl(nsc, nsc - 1); }
                else join(nsc + 1);
                lane = nsc;  // the new branch starts on its own lane
                cb(q, "0", "1", w[nsc - 1], w[nsc], 3, 2, x[nsc - 1], x[nsc], 1);
                // write-after-read: branch nsc-1 of the next stage updates x[nsc-1] in place (conv2 of its first
                // BasicBlock) and must not start before the transition conv above has consumed it
                dep(nsc, nsc - 1);
                lane = 0;
                for (int i = 0; i <= nsc; ++i) tap("stages." + std::to_string(s) + "#" + std::to_string(i), x[i], w[i]);
            } else {
                // HigherHRNet: only lane 0 goes on (it has waited for the others source by source); the forward's closing edges
                // collect the rest.  The classification head reads all four branches on lane 0.
                if (n.kind != 0 || n.sw.full_join) join(4);
                if (n.kind == 0) tap("stages.3#0", CAT, w[0]);
            }
        }
        if (n.kind == 1) {  // ClassificationHead.forward (classification/architectures/hrnet.py:48-61)
            const int outs[4] = {128, 256, 512, 1024};
            const std::string hp = "classification_head";
            int cur = -1;
            for (int i = 0; i < 4; ++i) {
                const std::string up = hp + ".chann_incr_blocks." + std::to_string(i);
                const int mid = outs[i] / 4, sh = 2 + i;
                const int a1 = T(mid, sh), a2 = T(mid, sh), Dn = T(outs[i], sh), Yn = T(outs[i], sh);
                cb(up, "conv1", "bn1", w[i], mid, 1, 1, x[i], a1, 1);
                cb(up, "conv2", "bn2", mid, mid, 3, 1, a1, a2, 1);
                cb(up, "downsample.0", "downsample.1", w[i], outs[i], 1, 1, x[i], Dn, 0);
                cb(up, "conv3", "bn3", mid, outs[i], 1, 1, a2, Yn, 1, Dn);
                if (i == 0) { cur = Yn; continue; }
                const std::string dpn = hp + ".downsample_blocks." + std::to_string(i - 1);
                const int dw = T(outs[i], sh);
                conv(L(dpn + ".0", dpn + ".1", outs[i - 1], outs[i], 3, 2, dpn + ".0.bias"), cur, dw, 1);
                Op add;  // out = bottleneck(x_i) + downsampled, no activation
                add.kind = OP_UPADD; add.in = Yn; add.out = Yn; add.C = outs[i]; add.relu = 0;
                add.up[0] = dw; add.up_shift[0] = 0; add.nup = 1;
                n.ops.push_back(add);
                cur = Yn;
            }
            const int FC = T(2048, 5);
            conv(L(hp + ".final_conv.0", hp + ".final_conv.1", 1024, 2048, 1, 1, hp + ".final_conv.0.bias"), cur, FC, 1);
            { Op o; o.kind = OP_AVGPOOL; o.in = FC; o.C = 2048; n.ops.push_back(o); }
            { Op o; o.kind = OP_LINEAR; n.ops.push_back(o); }
            return;
        }

        // heads (higher_hrnet.py:52,70-79): 1x1 conv with bias -> fp32 NCHW result AND bf16 copy into CAT
        // fp8 handles run the three heads (this 1x1, the transposed conv, the final 1x1: 1.3 % of the FLOPs) on the bf16 kernels over
        // the bf16 representations: their e4m3 rounding lands on the outputs undamped (tools/probes/fp8_emulate.py: 7.8 -> 6.4 %
        // rms on the tags of W48).  HH_FP8_HEADS=e4m3 keeps them on the e4m3 kernels.
        const bool hi_heads = n.dtype == 2 && !n.sw.fp8_heads8;
        if (head_fold) {  // fp32 result only, off the main lane: lane 1 waits for the last fusion (lane 0) and is collected at the end
            dep(0, 1);
            lane = 1;
            Op &o = conv(L("init_heatmaps_head", "", C, 2 * K, 1, 1, "init_heatmaps_head.bias"), CAT, -1, 0);
            o.f32_out = 1;
            lane = 0;
        } else {
            Op &o = conv(L("init_heatmaps_head", "", C, 2 * K, 1, 1, "init_heatmaps_head.bias"), CAT, CAT, 0);
            o.out_coff = C; o.cout_store = catC - C; o.f32_out = 1;
            o.hi = hi_heads; n.layers[o.layer].hi = hi_heads;
        }
        // DeconvHeatmapsHead (higher_hrnet.py:7-44): ConvTranspose2d(k4,s2,p1) = 4 phase-wise 2x2 convs
        const std::string dp = "deconv_layers.0";
        const int DF = T(C, 1), DM = T(C, 1);
        {   // one launch: the four phase weight sets are packed back to back (py = -1 marks "all phases")
            const int l = L(dp + ".deconv.0", dp + ".deconv.1", head_fold ? C + 1 : C + 2 * K, C, 2, 1);
            n.layers[l].transposed = true; n.layers[l].py = -1; n.layers[l].px = -1;
            n.layers[l].acct_cin = C + 2 * K;
            if (head_fold) { n.layers[l].fold_w = "init_heatmaps_head.weight"; n.layers[l].fold_b = "init_heatmaps_head.bias"; }
            Op &o = conv(l, CAT, DF, 1);
            o.scatter = 1;
            o.hi = hi_heads; n.layers[l].hi = hi_heads;
        }
        if (hi_heads) {  // the residual blocks' first conv reads DF as e4m3
            Op q;
            q.kind = OP_QUANT; q.out = DF; q.lane = lane;
            n.ops.push_back(q);
        }
        basic_blocks(dp + ".resid_blocks", C, DF, DM);
        const int last_bb = (int)n.ops.size() - 1;
        tap("deconv#0", DF, C);
        {
            Op &o = conv(L(dp + ".final_layer", "", C, K, 1, 1, dp + ".final_layer.bias"), DF, -1, 0);
            o.f32_out = 2;
            o.hi = hi_heads; n.layers[o.layer].hi = hi_heads;
            // bf16, 32 channels: the head runs in the last block's epilogue (basicblock_fused_pc.hip, FIN) and this op is skipped --
            // decided per forward in enqueue() (not with taps on: "deconv#0" is the block output that is then never stored)
            if (n.dtype != 2 && C == 32 && K <= 32 && n.ops[last_bb].kind == OP_BB) {
                n.layers[o.layer].fin_head = true;
                n.ops[last_bb].fin = (int)n.ops.size() - 1;
            }
        }
    }
};
}  // namespace

int hh_net::build()
{
    Builder b(*this);
    b.register_params();
    b.build_plan();
    if (dtype == 2) assign_fp8_formats();
    return 0;
}

// fp8 plans: a tensor is kept as e4m3 (+ one scale) where a conv reads it as an MFMA operand, and as bf16 where it is read back
// as a residual, by a fusion sum or by an op that runs on the bf16 kernels -- so the residual trunk (block in / out, fusion
// outputs) is written in BOTH forms by its producer and never goes through e4m3 on its own way forward: every block then adds
// the error of its two convs only, instead of also re-rounding the whole trunk to 3 mantissa bits (16-19 % -> 6-9 % rms at the
// outputs of the seeded nets, tools/probes/fp8_emulate.py).  HH_FP8_TRUNK=e4m3: everything e4m3 (the round-2 plan).
void hh_net::assign_fp8_formats()
{
    for (auto &t : tensors) { t.f8 = sw.fp8_trunk8; t.b16 = false; }
    auto need8 = [&](int t) { if (t >= 0) tensors[t].f8 = true; };
    auto need16 = [&](int t) { if (t >= 0) { if (sw.fp8_trunk8) tensors[t].f8 = true; else tensors[t].b16 = true; } };
    for (const Op &op : ops) {
        switch (op.kind) {
        case OP_STEM: need8(op.out); break;  // stem_conv.hip writes e4m3 only
        case OP_CONV:
            if (op.hi) { if (op.in >= 0) tensors[op.in].b16 = true; if (op.out >= 0) tensors[op.out].b16 = true; }
            else { need8(op.in); if (op.res8) need8(op.res); else need16(op.res); }
            break;
        case OP_BB: need8(op.in); need16(op.in); break;
        case OP_UPADD: need16(op.in); for (int j = 0; j < op.nup; ++j) need16(op.up[j]); break;
        case OP_QUANT: tensors[op.out].f8 = true; tensors[op.out].b16 = true; break;
        default: break;
        }
    }
}

// Kernel-order weight image: [cout_group][cin_chunk][tap][KC/8][COUT_T][8] bf16, BN scale folded in, zero padded.
void hh_pack_weights(const float *W, const float *scale, int ks, int cin, int cout, int KC, int COUT_T, bool transposed,
                     int py, int px, std::vector<bf16_raw> &packed)
{
    const int coutp = round_up(cout, COUT_T), ncg = coutp / COUT_T, cin_pad = round_up(cin, KC);
    const int nch = cin_pad / KC, taps = ks * ks, C8 = KC / 8;
    packed.assign((size_t)ncg * nch * taps * C8 * COUT_T * 8, 0);
    size_t o = 0;
    for (int cg = 0; cg < ncg; ++cg)
        for (int ch = 0; ch < nch; ++ch)
            for (int t = 0; t < taps; ++t) {
                int ky = t / ks, kx = t % ks;
                if (transposed) {  // patch row 0/1 of phase py <-> ky of the 4x4 stride-2 transposed conv
                    ky = py == 0 ? (ky == 0 ? 3 : 1) : (ky == 0 ? 2 : 0);
                    kx = px == 0 ? (kx == 0 ? 3 : 1) : (kx == 0 ? 2 : 0);
                }
                for (int c8 = 0; c8 < C8; ++c8)
                    for (int ci_o = 0; ci_o < COUT_T; ++ci_o)
                        for (int j = 0; j < 8; ++j, ++o) {
                            const int co = cg * COUT_T + ci_o, ci = ch * KC + c8 * 8 + j;
                            if (co >= cout || ci >= cin) continue;
                            float v;
                            if (transposed) v = W[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
                            else v = W[(((size_t)co * cin + ci) * ks + ky) * ks + kx];
                            packed[o] = f2bf(v * scale[co]);
                        }
            }
}

// ------------------------------------------------------------------------- finalize
int hh_family_pick(int ks, int stride, int cin_pad, int coutp, int *KC, int *NT)
{
    if (ks == 3 && stride == 1) *KC = (cin_pad % 32 == 0) ? 32 : 16;
    else if (ks == 1) *KC = (cin_pad % 32 == 0) ? 32 : 16;
    else *KC = 16;
    *NT = (coutp % 64 == 0) ? 2 : 1;
    for (int i = 0; i < conv_num_configs(); ++i) {
        const ConvConfig &c = conv_config(i);
        if (c.KS == ks && c.S == stride && c.KC == *KC && c.NT == *NT && !c.DB) return 0;
    }
    return 1;
}

int hh_net::finalize()
{
    for (auto &p : params)
        if (!p.loaded && !p.counter) { hh_set_error("hh_finalize: parameter never loaded: " + p.name); return 1; }
    HH_CHECK_HIP(conv_init());
    HH_CHECK_HIP(bb_fused_init());
    HH_CHECK_HIP(stem_fused_init());
    HH_CHECK_HIP(bbpc_init());
    HH_CHECK_HIP(bb64_fused_init());
    HH_CHECK_HIP(junction_init());
    if (dtype == 2 && kind != 0) { hh_set_error("hh_finalize: the fp8 path covers HigherHRNet only"); return 1; }
    {
        int dev = 0;
        hipDeviceProp_t prop;
        HH_CHECK_HIP(hipGetDevice(&dev));
        HH_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        num_cus = prop.multiProcessorCount;
    }
    auto get = [&](const std::string &name) -> const std::vector<float> & { return params[param_index.at(name)].data; };
    for (auto &l : layers) {
        if (l.stem) {  // [cout tile 2][k-step 2][half 2][32][8], tap = c*9 + ky*3 + kx, BN scale folded
            const std::vector<float> &W = get(l.conv + ".weight");
            std::vector<bf16_raw> packed(64 * 32, 0);
            std::vector<float> shift(64);
            for (int co = 0; co < 64; ++co) {
                const float g = get(l.bn + ".weight")[co], bta = get(l.bn + ".bias")[co];
                const float mu = get(l.bn + ".running_mean")[co], var = get(l.bn + ".running_var")[co];
                const float sc = g / std::sqrt(var + 1e-5f);
                shift[co] = bta - mu * sc;
                for (int t = 0; t < 27; ++t) {
                    const int kk = t / 16, hh = (t % 16) / 8, j = t % 8;
                    packed[((((co / 32) * 2 + kk) * 2 + hh) * 32 + co % 32) * 8 + j] = f2bf(W[(size_t)co * 27 + t] * sc);
                }
            }
            if (l.d_w) { hipFree(l.d_w); l.d_w = nullptr; }
            if (l.d_bias) { hipFree(l.d_bias); l.d_bias = nullptr; }
            HH_CHECK_HIP(hipMalloc((void **)&l.d_w, packed.size() * 2));
            HH_CHECK_HIP(hipMalloc((void **)&l.d_bias, 64 * 4));
            HH_CHECK_HIP(hipMemcpy(l.d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
            HH_CHECK_HIP(hipMemcpy(l.d_bias, shift.data(), 64 * 4, hipMemcpyHostToDevice));
            continue;
        }
        if (dtype == 2 && !l.hi) continue;  // e4m3 weights: finalize_fp8() below
        const int cin_pad0 = round_up(l.cin, 16), coutp = round_up(l.cout, 32);
        if (l.stem2) { l.KC = 64; l.NT = 2; }  // stem_fused.hip: [tap][8][64 couts][8]
        else if (hh_family_pick(l.ks, l.stride, cin_pad0, coutp, &l.KC, &l.NT)) {
            hh_set_error("no kernel family for conv " + l.conv);
            return 1;
        }
        // the wide 3x3 layers (one wave per SIMD, 8+ chunks): 16-channel chunks on two LDS buffers (conv_mfma.hip, DB)
        l.db = !sw.no_conv_db && !l.stem2 && l.ks == 3 && l.stride == 1 && !l.transposed && l.mconv.empty() && l.cin >= 128 &&
               l.cin % 16 == 0 && coutp % 64 == 0;
        if (l.db) { l.KC = 16; l.NT = 2; }
        l.cin_pad = round_up(l.cin, l.KC);
        const int COUT_T = 32 * l.NT;
        l.ncg = coutp / COUT_T;
        if (!l.mconv.empty()) {
            // the sum of several conv + BN pairs as one conv over the concatenated input channels: every BN scale goes into its
            // conv's weights, the shifts add up
            std::vector<float> Wm((size_t)l.cout * l.cin * 9, 0.f), one(coutp, 1.f), shift(coutp, 0.f);
            int c0 = 0;
            for (size_t m = 0; m < l.mconv.size(); ++m) {
                if (l.mcin[m] % l.KC) { hh_set_error("merged fusion conv: input width not a multiple of the chunk size"); return 1; }
                const std::vector<float> &Wj = get(l.mconv[m] + ".weight");
                for (int co = 0; co < l.cout; ++co) {
                    const float g = get(l.mbn[m] + ".weight")[co], bta = get(l.mbn[m] + ".bias")[co];
                    const float mu = get(l.mbn[m] + ".running_mean")[co], var = get(l.mbn[m] + ".running_var")[co];
                    const float sc = g / std::sqrt(var + 1e-5f);
                    shift[co] += bta - mu * sc;
                    for (int ci = 0; ci < l.mcin[m]; ++ci)
                        for (int t = 0; t < 9; ++t) Wm[((size_t)co * l.cin + c0 + ci) * 9 + t] = Wj[((size_t)co * l.mcin[m] + ci) * 9 + t] * sc;
                }
                c0 += l.mcin[m];
            }
            std::vector<bf16_raw> packed;
            hh_pack_weights(Wm.data(), one.data(), 3, l.cin, l.cout, l.KC, COUT_T, false, 0, 0, packed);
            if (l.d_w) { hipFree(l.d_w); l.d_w = nullptr; }
            if (l.d_bias) { hipFree(l.d_bias); l.d_bias = nullptr; }
            HH_CHECK_HIP(hipMalloc((void **)&l.d_w, packed.size() * 2));
            HH_CHECK_HIP(hipMalloc((void **)&l.d_bias, (size_t)coutp * 4));
            HH_CHECK_HIP(hipMemcpy(l.d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
            HH_CHECK_HIP(hipMemcpy(l.d_bias, shift.data(), (size_t)coutp * 4, hipMemcpyHostToDevice));
            continue;
        }
        std::vector<float> Wfold;
        if (!l.fold_w.empty()) {  // [C + 1, cout, 4, 4]: rows 0..C-1 = Wd_feats + sum_k Wf[k, c] * Wd_hm[k], row C = sum_k bf[k] * Wd_hm[k]
            const std::vector<float> &Wd = get(l.conv + ".weight"), &Wf = get(l.fold_w), &bf = get(l.fold_b);
            const int Cf = l.cin - 1, K2 = (int)bf.size();
            Wfold.assign((size_t)l.cin * l.cout * 16, 0.f);
            for (int c = 0; c <= Cf; ++c)
                for (int co = 0; co < l.cout; ++co)
                    for (int t = 0; t < 16; ++t) {
                        double acc = c < Cf ? (double)Wd[((size_t)c * l.cout + co) * 16 + t] : 0.0;
                        for (int k = 0; k < K2; ++k)
                            acc += (double)(c < Cf ? Wf[(size_t)k * Cf + c] : bf[k]) * (double)Wd[((size_t)(Cf + k) * l.cout + co) * 16 + t];
                        Wfold[((size_t)c * l.cout + co) * 16 + t] = (float)acc;
                    }
        }
        const std::vector<float> &W = Wfold.empty() ? get(l.conv + ".weight") : Wfold;
        std::vector<float> scale(coutp, 0.f), shift(coutp, 0.f);
        for (int co = 0; co < l.cout; ++co) {
            if (!l.bn.empty()) {
                const float g = get(l.bn + ".weight")[co], bta = get(l.bn + ".bias")[co];
                const float mu = get(l.bn + ".running_mean")[co], var = get(l.bn + ".running_var")[co];
                const float sc = g / std::sqrt(var + 1e-5f);
                const float cb = l.bias.empty() ? 0.f : get(l.bias)[co];  // conv bias in front of BN (classification head)
                scale[co] = sc;
                shift[co] = bta + (cb - mu) * sc;
            } else {
                scale[co] = 1.f;
                shift[co] = l.bias.empty() ? 0.f : get(l.bias)[co];
            }
        }
        std::vector<bf16_raw> packed;
        if (l.transposed && l.py < 0) {
            for (int ph = 0; ph < 4; ++ph) {
                std::vector<bf16_raw> one;
                hh_pack_weights(W.data(), scale.data(), l.ks, l.cin, l.cout, l.KC, COUT_T, true, ph >> 1, ph & 1, one);
                l.phase_stride = one.size();
                packed.insert(packed.end(), one.begin(), one.end());
            }
        } else
            hh_pack_weights(W.data(), scale.data(), l.ks, l.cin, l.cout, l.KC, COUT_T, l.transposed, l.py, l.px, packed);
        if (l.d_w) { hipFree(l.d_w); l.d_w = nullptr; }
        if (l.d_bias) { hipFree(l.d_bias); l.d_bias = nullptr; }
        HH_CHECK_HIP(hipMalloc((void **)&l.d_w, packed.size() * 2));
        HH_CHECK_HIP(hipMalloc((void **)&l.d_bias, (size_t)coutp * 4));
        HH_CHECK_HIP(hipMemcpy(l.d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
        HH_CHECK_HIP(hipMemcpy(l.d_bias, shift.data(), (size_t)coutp * 4, hipMemcpyHostToDevice));
        if (l.fin_head) {  // B fragments of the head inside bbpc_final_kernel: lane (cout r, half h) of k half m holds cin 16m + 8h .. +7
            std::vector<bf16_raw> wf(2 * 2 * 32 * 8, 0);
            for (int m = 0; m < 2; ++m)
                for (int hh = 0; hh < 2; ++hh)
                    for (int co = 0; co < l.cout; ++co)
                        for (int j = 0; j < 8; ++j) wf[(((m * 2 + hh) * 32) + co) * 8 + j] = f2bf(W[(size_t)co * l.cin + 16 * m + 8 * hh + j] * scale[co]);
            if (l.d_wfin) { hipFree(l.d_wfin); l.d_wfin = nullptr; }
            HH_CHECK_HIP(hipMalloc((void **)&l.d_wfin, wf.size() * 2));
            HH_CHECK_HIP(hipMemcpy(l.d_wfin, wf.data(), wf.size() * 2, hipMemcpyHostToDevice));
        }
    }
    if (kind == 1) {
        const auto &fw = get("classification_head.classifier.weight");
        const auto &fb = get("classification_head.classifier.bias");
        if (d_fc_w) hipFree(d_fc_w);
        if (d_fc_b) hipFree(d_fc_b);
        HH_CHECK_HIP(hipMalloc((void **)&d_fc_w, fw.size() * 4));
        HH_CHECK_HIP(hipMalloc((void **)&d_fc_b, fb.size() * 4));
        HH_CHECK_HIP(hipMemcpy(d_fc_w, fw.data(), fw.size() * 4, hipMemcpyHostToDevice));
        HH_CHECK_HIP(hipMemcpy(d_fc_b, fb.data(), fb.size() * 4, hipMemcpyHostToDevice));
    }
    for (auto &g : graphs) hipGraphExecDestroy(g.exec);
    graphs.clear();
    finalized = true;
    if (dtype == 2 && finalize_fp8()) { finalized = false; return 1; }
    return 0;
}

// ------------------------------------------------------------------------- workspace
void hh_net::release_workspace()
{
    for (void *p : allocs) hipFree(p);
    allocs.clear();
    for (auto &t : tensors) { t.ptr = nullptr; t.ptr16 = nullptr; }
    ws_ready = false;
    for (auto &t : taps) t.copy = nullptr;
    for (auto &g : graphs) hipGraphExecDestroy(g.exec);
    graphs.clear();
    ws_bytes = 0;
    rB = rH = rW = 0;
}

int hh_net::reserve(int B, int H, int W)
{
    if (H % 32 || W % 32 || B <= 0) { hh_set_error("hh_reserve: H and W must be positive multiples of 32"); return 1; }
    const bool have = ws_ready && (!taps_enabled || taps.empty() || taps[0].copy);
    if (have && B <= rB && H <= rH && W <= rW) return 0;
    const int nB = std::max(B, rB), nH = std::max(H, rH), nW = std::max(W, rW);
    release_workspace();
    auto alloc = [&](size_t bytes, void **out) -> int {
        HH_CHECK_HIP(hipMalloc(out, bytes));
        allocs.push_back(*out);
        ws_bytes += (int64_t)bytes;
        return 0;
    };
    for (auto &t : tensors) {
        const size_t n = (size_t)nB * (nH >> t.shift) * (nW >> t.shift) * t.C;
        // HH_POISON_WS=1 (tests): recycled device memory is not zero -- fill the workspace with NaN patterns so that a kernel
        // that reads what no kernel wrote shows up in the outputs instead of depending on what the allocator hands out
        if (dtype != 2 || t.f8) {
            const size_t bytes = n * elem();
            if (alloc(bytes, (void **)&t.ptr)) return 1;
            if (sw.poison_ws) HH_CHECK_HIP(hipMemset(t.ptr, 0xFF, bytes));
            if (t.zero_init) HH_CHECK_HIP(hipMemset(t.ptr, 0, bytes));
            if (t.ones_channel >= 0 && dtype != 2) {  // bf16 1.0 = 0x3F80 in channel `ones_channel` of every pixel
                const size_t npix = (size_t)nB * (nH >> t.shift) * (nW >> t.shift);
                char *c0 = (char *)t.ptr + (size_t)t.ones_channel * 2;
                HH_CHECK_HIP(hipMemset2D(c0, (size_t)t.C * 2, 0x80, 1, npix));
                HH_CHECK_HIP(hipMemset2D(c0 + 1, (size_t)t.C * 2, 0x3F, 1, npix));
            }
        }
        if (dtype == 2 && t.b16) {
            if (alloc(n * 2, (void **)&t.ptr16)) return 1;
            if (sw.poison_ws) HH_CHECK_HIP(hipMemset(t.ptr16, 0xFF, n * 2));
            if (t.zero_init) HH_CHECK_HIP(hipMemset(t.ptr16, 0, n * 2));
        }
    }
    if (taps_enabled)
        for (auto &t : taps) {
            const TensorDesc &d = tensors[t.tensor];
            t.is16 = dtype == 2 && d.b16;
            if (alloc((size_t)nB * (nH >> d.shift) * (nW >> d.shift) * d.C * (t.is16 ? 2 : elem()), (void **)&t.copy)) return 1;
        }
    // The memsets above (zero-initialised tensors, the constant-one channel, poison patterns) run on the NULL stream and are
    // asynchronous with respect to the host; the forward that follows is enqueued on the caller's stream, which for torch is a
    // non-blocking one: without this wait its first launches could read the workspace before it is initialised (round 4 found
    // that race in the decoder's work counters; it may be what stopped one forward test once in round 2).  Once per shape growth.
    HH_CHECK_HIP(hipDeviceSynchronize());
    ws_ready = true;
    rB = nB; rH = nH; rW = nW;
    return 0;
}

// ------------------------------------------------------------------------- execution
int hh_pick_config(int ks, int stride, int KC, int NT, int Wo)
{
    int best = -1;
    for (int i = 0; i < conv_num_configs(); ++i) {
        const ConvConfig &c = conv_config(i);
        if (c.KS != ks || c.S != stride || c.KC != KC || c.NT != NT || c.DB) continue;
        if (best < 0) best = i;
        const bool want16 = Wo <= 16;
        if ((c.TW == 16) == want16) return i;
    }
    return best;
}
static int pick_config(const ConvLayer &l, int Wo)
{
    int best = -1;
    for (int i = 0; i < conv_num_configs(); ++i) {
        const ConvConfig &c = conv_config(i);
        if (c.KS != l.ks || c.S != l.stride || c.KC != l.KC || c.NT != l.NT || c.DB != l.db) continue;
        if (best < 0) best = i;
        const bool want16 = Wo <= 16;
        if ((c.TW == 16) == want16) return i;
    }
    return best;
}

int hh_net::enqueue(const float *images, int B, int H, int W, float *o1, float *o2, hipStream_t s0)
{
    // Lanes: the resolution branches of an HR block and the outputs of a fusion layer are independent, so they run
    // on separate HIP streams (forked from / joined to the caller's stream with events; the same calls become
    // DAG edges under hipGraph capture).  The small low-resolution launches then fill CUs the big ones leave idle.
    const bool multi = multi_lane && !taps_enabled && !prof_enabled && !calibrating && !sw.poison_lds;
    if (multi) {
        // The lanes take the PRIORITY of the caller's stream.  A caller on a highest-priority stream gets its whole forward
        // on high-priority queues (dependent launches follow each other faster there: forward 5.25 -> 4.98 ms at batch 32),
        // while lanes above the caller's own priority made the fork/join pattern erratic (1.8 -> 5.9 ms at batch 1).
        int prio = 0;
        HH_CHECK_HIP(hipStreamGetPriority(s0, &prio));
        if (!lane_priority_set || prio != lane_priority) {
            for (int l = 1; l < 4; ++l) {
                if (lane_streams[l]) { HH_CHECK_HIP(hipStreamSynchronize(lane_streams[l])); HH_CHECK_HIP(hipStreamDestroy(lane_streams[l])); }
                HH_CHECK_HIP(hipStreamCreateWithPriority(&lane_streams[l], hipStreamNonBlocking, prio));
            }
            lane_priority = prio;
            lane_priority_set = true;
        }
    }
    hipStream_t L[4] = {s0, multi ? lane_streams[1] : s0, multi ? lane_streams[2] : s0, multi ? lane_streams[3] : s0};
    lane_events_used = 0;
    auto next_event = [&](hipEvent_t *e) -> int {
        if (lane_events_used == lane_events.size()) {
            hipEvent_t ne;
            // The lane events order streams of ONE device: no system-scope fence (cache write-back / invalidate for the host's and other
            // devices' eyes) when one is recorded -- the caller's own synchronisation of its stream does that once, at the end.
            // Round 4: forward 4.30 -> 4.27 ms (HH_EVENT_SYSTEM_FENCE=1: plain hipEventDisableTiming events).
            HH_CHECK_HIP(hipEventCreateWithFlags(&ne, hipEventDisableTiming | (sw.event_system_fence ? 0u : (unsigned)hipEventDisableSystemFence)));
            lane_events.push_back(ne);
        }
        *e = lane_events[lane_events_used++];
        return 0;
    };
    if (prof_enabled && prof_used == 0) {  // first forward since hh_profile_enable: device-clock slots start as {~0, 0}
        if (!d_clk) {
            HH_CHECK_HIP(hipMalloc((void **)&d_clk, HH_PROF_SLOTS * 32));
            int dev = 0, khz = 0;
            HH_CHECK_HIP(hipGetDevice(&dev));
            HH_CHECK_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
            clk_khz = khz;
        }
        std::vector<unsigned long long> init(HH_PROF_SLOTS * 4, 0ull);  // {min start, max end, core cycles, 100 MHz ticks of workgroup 0}
        for (int i = 0; i < HH_PROF_SLOTS; ++i) init[4 * i] = ~0ull;
        HH_CHECK_HIP(hipMemcpy(d_clk, init.data(), init.size() * 8, hipMemcpyHostToDevice));
    }
    int lanes_open = 1;  // lanes [0, lanes_open) have work that the caller's stream must wait for at the end
    hipEvent_t mark_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // What each lane is already ordered behind, as vector clocks over the launches enqueued so far (check_plan's model, kept while
    // enqueueing): a wait whose event names nothing newer is dropped.  A wait is a barrier packet that costs its stream ~11 us even
    // if the event completed long before (tools/probes/event_cost.py) -- e.g. the closing edges for lanes that lane 0 has joined already.
    int seqn[4] = {0, 0, 0, 0}, vc[4][4] = {}, mark_snap[4][4] = {};
    auto snap = [&](int l, int *out) { for (int m = 0; m < 4; ++m) out[m] = m == l ? seqn[l] : vc[l][m]; };
    auto news = [&](int l, const int *sn) {  // does the snapshot hold anything lane l is not behind yet?  (then: l is behind it from now on)
        bool any = false;
        for (int m = 0; m < 4; ++m)
            if (m != l && sn[m] > vc[l][m]) { vc[l][m] = sn[m]; any = true; }
        return any;
    };
    int fin_done = -1;  // index of the head conv that the last fused block has already run
    for (const Op &op : ops) {
        hipStream_t s = L[op.lane];
        if (fin_done >= 0 && &op == &ops[fin_done]) continue;
        if (op.kind != OP_JOIN && op.kind != OP_MARK && op.kind != OP_WAITL && op.kind != OP_DEP) ++seqn[op.lane];  // (a launch, or nothing: over-counting only keeps a wait)
        if (sw.debug_skip) {  // measurement only: the outputs are wrong
            unsigned cat = 0;
            if (op.kind == OP_UPADD) cat = SK_UPADD;
            else if (op.kind == OP_JUNC) cat = SK_JUNC;
            else if (op.kind == OP_STEM) cat = SK_STEM;
            else if (op.kind == OP_BB) cat = layers[op.layer].cout == 32 ? SK_BB32 : layers[op.layer].cout == 64 ? SK_BB64 : 0;
            else if (op.kind == OP_CONV) {
                const ConvLayer &l = layers[op.layer];
                if (l.transposed) cat = SK_DECONV;
                else if (op.f32_out) cat = SK_HEAD;
                else if (l.stride == 2) cat = SK_S2 | (l.cin >= 128 ? SK_S2BIG : 0) | (l.cin == 256 && l.mconv.empty() && tensors[op.in].shift == 2 ? SK_TRANS0 : 0);
                else if (l.ks == 1) cat = SK_C1X1;
                else if (l.cin == 256 && l.cout == 256) cat = SK_C256;
                else if (l.cin == 128 && l.cout == 128) cat = SK_C128;
                else if (l.cin == 256 && tensors[op.in].shift == 2) cat = SK_TRANS0;
            }
            if (cat & sw.debug_skip) continue;
        }
        if (sw.poison_lds && op.kind != OP_JOIN && op.kind != OP_MARK && op.kind != OP_WAITL && op.kind != OP_DEP && op.kind != OP_TAP)
            HH_CHECK_HIP(launch_lds_poison(num_cus, s));
        switch (op.kind) {
        case OP_JOIN: {
            if (!multi) break;
            // lanes that have not run anything yet only wait (recording on a stream that has not joined a
            // capture and then waiting on that event from the capturing stream is illegal)
            hipEvent_t e[4];
            const int nrec = op.nlanes < lanes_open ? op.nlanes : lanes_open;
            for (int l = 0; l < nrec; ++l) {
                if (next_event(&e[l])) return 1;
                HH_CHECK_HIP(hipEventRecord(e[l], L[l]));
            }
            int sn[4][4];
            for (int m = 0; m < nrec; ++m) snap(m, sn[m]);
            for (int l = 0; l < op.nlanes; ++l)
                for (int m = 0; m < nrec; ++m)
                    if (m != l) { news(l, sn[m]); HH_CHECK_HIP(hipStreamWaitEvent(L[l], e[m], 0)); }
            if (op.nlanes > lanes_open) lanes_open = op.nlanes;
            break;
        }
        case OP_MARK: {
            if (!multi) break;
            const int nrec = op.nlanes < lanes_open ? op.nlanes : lanes_open;
            for (int l = 0; l < 4; ++l) mark_ev[l] = nullptr;  // a wait may only name a lane THIS mark recorded
            for (int l = op.dep_from; l < nrec; ++l) {
                if (next_event(&mark_ev[l])) return 1;
                HH_CHECK_HIP(hipEventRecord(mark_ev[l], L[l]));
                snap(l, mark_snap[l]);
            }
            if (op.nlanes > lanes_open) lanes_open = op.nlanes;
            break;
        }
        case OP_WAITL: {
            if (!multi) break;
            if (!mark_ev[op.dep_from]) { hh_set_error("plan: OP_WAITL names a lane the last OP_MARK did not record"); return 1; }
            if (!news(op.lane, mark_snap[op.dep_from]) && !sw.keep_waits) break;  // already behind it
            HH_CHECK_HIP(hipStreamWaitEvent(L[op.lane], mark_ev[op.dep_from], 0));
            break;
        }
        case OP_DEP: {
            if (!multi) break;
            int sn[4];
            snap(op.dep_from, sn);
            if (!news(op.lane, sn) && !sw.keep_waits) break;
            hipEvent_t e;
            if (next_event(&e)) return 1;
            HH_CHECK_HIP(hipEventRecord(e, L[op.dep_from]));
            HH_CHECK_HIP(hipStreamWaitEvent(L[op.lane], e, 0));
            break;
        }
        case OP_STEM: {
            const ConvLayer &l = layers[op.layer];
            if (op.layer2 >= 0) {  // both stem convolutions in one kernel
                const ConvLayer &l2 = layers[op.layer2];
                StemFusedParams q{};
                q.images = images; q.w1 = l.d_w; q.b1 = l.d_bias; q.w2 = l2.d_w; q.b2 = l2.d_bias;
                q.out = tensors[op.out].ptr; q.out_cs = tensors[op.out].C;
                q.B = B; q.H = H; q.W = W;
                if (!stem_fused_supported(q)) { hh_set_error("hh_forward: the fused stem needs H, W multiples of 4 and images below 2 GB (HH_NO_STEM_FUSED=1)"); return 1; }
                if (prof_enabled) {
                    if (prof_used == prof.size()) {
                        ProfRecord r{};
                        HH_CHECK_HIP(hipEventCreate(&r.e0));
                        HH_CHECK_HIP(hipEventCreate(&r.e1));
                        prof.push_back(r);
                    }
                    ProfRecord *pr = &prof[prof_used++];
                    pr->op = (int)(&op - ops.data());
                    pr->cfg = HH_CFG_STEM_FUSED;
                    pr->flops = 2.0 * B * (H / 2) * (W / 2) * 27.0 * 64.0 + 2.0 * B * (H / 4) * (W / 4) * 576.0 * 64.0;
                    pr->bytes = (double)B * H * W * 3 * 4 + (double)B * (H / 4) * (W / 4) * 64 * 2 + 64 * 32 * 2 + 73728;
                    pr->slot = prof_used <= HH_PROF_SLOTS ? (int)prof_used - 1 : -1;
                    if (pr->slot >= 0 && prof_clk) q.clk = d_clk + 4 * pr->slot;
                    hh_launch_probe() = LaunchProbe{pr->e0, pr->e1};
                }
                HH_CHECK_HIP(stem_fused_launch(q, num_cus, s));
                break;
            }
            StemParams p{};
            p.images = images; p.w = l.d_w; p.bias = l.d_bias;
            p.out = tensors[op.out].ptr; p.out_cs = tensors[op.out].C;
            p.B = B; p.H = H; p.W = W;
            if (dtype == 2) {
                p.out_fp8 = (unsigned char *)tensors[op.out].ptr; p.out_inv_scale = 1.f / op.s_out;
                if (calibrating) p.absmax = d_amax + (&op - ops.data());
            }
            ProfRecord *pr = nullptr;
            if (prof_enabled) {
                if (prof_used == prof.size()) {
                    ProfRecord r{};
                    HH_CHECK_HIP(hipEventCreate(&r.e0));
                    HH_CHECK_HIP(hipEventCreate(&r.e1));
                    prof.push_back(r);
                }
                pr = &prof[prof_used++];
                pr->op = (int)(&op - ops.data());
                pr->cfg = HH_CFG_STEM;
                pr->flops = 2.0 * B * (H / 2) * (W / 2) * 27.0 * 64.0;
                pr->bytes = (double)B * H * W * 3 * 4 + (double)B * (H / 2) * (W / 2) * 64 * 2 + 64 * 32 * 2;
                pr->slot = prof_used <= HH_PROF_SLOTS ? (int)prof_used - 1 : -1;
                if (pr->slot >= 0 && prof_clk) p.clk = d_clk + 4 * pr->slot;
                hh_launch_probe() = LaunchProbe{pr->e0, pr->e1};  // the launch below stamps e0 / e1 from its dispatch packet
            }
            HH_CHECK_HIP(stem_conv_launch(p, s));
            break;
        }
        case OP_UPADD: {
            if (dtype == 2) {
                if (enqueue_fp8_upadd(op, B, H, W, s)) return 1;
                break;
            }
            UpAddParams p{};
            const TensorDesc &b = tensors[op.in], &o = tensors[op.out];
            p.base = b.ptr; p.base_cs = b.C; p.base_coff = 0;
            p.nup = op.nup;
            for (int j = 0; j < op.nup; ++j) {
                p.up[j] = tensors[op.up[j]].ptr; p.up_cs[j] = tensors[op.up[j]].C; p.up_shift[j] = op.up_shift[j];
            }
            p.out = o.ptr; p.out_cs = o.C; p.out_coff = 0;
            p.B = B; p.H = H >> b.shift; p.W = W >> b.shift; p.C = op.C; p.relu = op.relu;
            HH_CHECK_HIP(launch_upadd(p, s));
            break;
        }
        case OP_TAP: {
            if (!taps_enabled) break;
            const TapInfo &t = taps[op.tap];
            const TensorDesc &d = tensors[t.tensor];
            HH_CHECK_HIP(hipMemcpyAsync(t.copy, t.is16 ? d.ptr16 : d.ptr, (size_t)B * (H >> d.shift) * (W >> d.shift) * d.C * (t.is16 ? 2 : elem()),
                                        hipMemcpyDeviceToDevice, s));
            break;
        }
        case OP_AVGPOOL: {
            const TensorDesc &ti = tensors[op.in];
            if (pool_cap < B * op.C) {
                if (d_pool) hipFree(d_pool);
                HH_CHECK_HIP(hipMalloc((void **)&d_pool, (size_t)B * op.C * 4));
                pool_cap = B * op.C;
            }
            HH_CHECK_HIP(launch_avgpool(ti.ptr, ti.C, d_pool, B, (H >> ti.shift) * (W >> ti.shift), op.C, s));
            break;
        }
        case OP_LINEAR:
            HH_CHECK_HIP(launch_linear(d_pool, d_fc_w, d_fc_b, o1, B, 2048, num_classes, s));
            break;
        case OP_JUNC: {
            const ConvLayer &l3 = layers[op.layer];
            JuncParams p{};
            const TensorDesc &ti = tensors[op.in];
            p.t2 = tensors[op.in].ptr; p.t2_cs = tensors[op.in].C;
            if (op.res >= 0) { p.res = tensors[op.res].ptr; p.res_cs = tensors[op.res].C; }
            if (op.in2 >= 0) { p.x = tensors[op.in2].ptr; p.x_cs = tensors[op.in2].C; p.wd = layers[op.layer2].d_w; p.bd = layers[op.layer2].d_bias; }
            p.w3 = l3.d_w; p.b3 = l3.d_bias;
            if (op.layer4 >= 0) {
                p.t2a = tensors[op.in3].ptr; p.t2a_cs = tensors[op.in3].C;
                p.w3a = layers[op.layer4].d_w; p.b3a = layers[op.layer4].d_bias;
            }
            if (op.layer3 >= 0) { p.w1 = layers[op.layer3].d_w; p.b1 = layers[op.layer3].d_bias; p.t1 = tensors[op.out2].ptr; p.t1_cs = tensors[op.out2].C; }
            if (op.out >= 0) { p.y = tensors[op.out].ptr; p.y_cs = tensors[op.out].C; }
            p.npix = B * (H >> ti.shift) * (W >> ti.shift);
            ProfRecord *pr = nullptr;
            if (prof_enabled) {
                if (prof_used == prof.size()) {
                    ProfRecord r{};
                    HH_CHECK_HIP(hipEventCreate(&r.e0));
                    HH_CHECK_HIP(hipEventCreate(&r.e1));
                    prof.push_back(r);
                }
                pr = &prof[prof_used++];
                pr->op = (int)(&op - ops.data());
                pr->cfg = HH_CFG_JUNCTION;
                pr->flops = 2.0 * p.npix * 64.0 * 256.0 * (1 + (op.in2 >= 0 && op.layer4 < 0) + (op.layer3 >= 0));
                // (algorithmic: what the unit needs when every junction stores y and reads the previous one -- the pair mode's savings
                // and its extra GEMMs are the implementation's business)
                pr->bytes = 2.0 * p.npix * (64 + ((op.in2 >= 0 && op.layer4 < 0) ? 64 : 256) + 256 + (op.layer3 >= 0 ? 64 : 0)) +
                            2.0 * 64 * 256 * (1 + (op.in2 >= 0 && op.layer4 < 0) + (op.layer3 >= 0));
                pr->slot = prof_used <= HH_PROF_SLOTS ? (int)prof_used - 1 : -1;
                if (pr->slot >= 0 && prof_clk) p.clk = d_clk + 4 * pr->slot;
                hh_launch_probe() = LaunchProbe{pr->e0, pr->e1};  // the launch below stamps e0 / e1 from its dispatch packet
            }
            HH_CHECK_HIP(junction_launch(p, num_cus, s));
            break;
        }
        case OP_BB: {
            if (dtype == 2) {
                ProfRecord *pr = nullptr;
                if (prof_enabled) {
                    if (prof_used == prof.size()) {
                        ProfRecord r{};
                        HH_CHECK_HIP(hipEventCreate(&r.e0));
                        HH_CHECK_HIP(hipEventCreate(&r.e1));
                        prof.push_back(r);
                    }
                    pr = &prof[prof_used++];
                    pr->op = (int)(&op - ops.data());
                    pr->slot = -1;
                }
                if (enqueue_fp8_bb(op, B, H, W, s, pr)) return 1;
                break;
            }
            const ConvLayer &l1 = layers[op.layer], &l2 = layers[op.layer2];
            const TensorDesc &ti = tensors[op.in], &to = tensors[op.out];
            BBParams p{};
            p.in = ti.ptr; p.in_cs = ti.C; p.out = to.ptr; p.out_cs = to.C;
            p.w1 = l1.d_w; p.w2 = l2.d_w; p.b1 = l1.d_bias; p.b2 = l2.d_bias;
            p.B = B; p.H = H >> ti.shift; p.W = W >> ti.shift;
            p.tall = sw.bb_tall;
            if (op.fin >= 0 && !sw.no_final_fuse && !sw.bb32_tile && !taps_enabled && o2 && !(sw.debug_skip & SK_HEAD)) {
                const ConvLayer &lf = layers[ops[op.fin].layer];
                p.fin_w = lf.d_wfin; p.fin_b = lf.d_bias; p.fin_out = o2; p.fin_K = lf.cout;
                if (lf.d_wfin && bbpc_final_supported(p)) fin_done = op.fin;
                else { p.fin_w = nullptr; p.fin_b = nullptr; p.fin_out = nullptr; p.fin_K = 0; }
            }
            ProfRecord *pr = nullptr;
            if (prof_enabled) {
                if (prof_used == prof.size()) {
                    ProfRecord r{};
                    HH_CHECK_HIP(hipEventCreate(&r.e0));
                    HH_CHECK_HIP(hipEventCreate(&r.e1));
                    prof.push_back(r);
                }
                pr = &prof[prof_used++];
                pr->op = (int)(&op - ops.data());
                const double Cb = l1.cout;
                pr->cfg = l1.cout == 64 ? HH_CFG_BB64_FUSED : HH_CFG_BB_FUSED;
                pr->slot = prof_used <= HH_PROF_SLOTS ? (int)prof_used - 1 : -1;
                if (pr->slot >= 0 && prof_clk) p.clk = d_clk + 4 * pr->slot;
                pr->flops = 2.0 * 2.0 * B * p.H * p.W * Cb * Cb * 9.0;
                pr->bytes = 2.0 * B * p.H * p.W * Cb * 2 + 2.0 * 2 * 9 * Cb * Cb;
                if (p.fin_out) {  // + the head: the block output stays on the chip, K fp32 planes leave instead
                    pr->flops += 2.0 * B * p.H * p.W * Cb * p.fin_K;
                    pr->bytes += (4.0 * p.fin_K - 2.0 * Cb) * B * p.H * p.W + 2.0 * Cb * p.fin_K;
                }
                hh_launch_probe() = LaunchProbe{pr->e0, pr->e1};  // the launch below stamps e0 / e1 from its dispatch packet
            }
            // inside an HR module with the lanes on: half the chip per fat kernel (PlanSwitches::fat_cus); alone: all of it
            auto budget = [&](int want) { return !(multi && op.siblings) ? num_cus : want > 0 ? (want < num_cus ? want : num_cus) : (num_cus + 1) / 2; };
            const int fat = budget(sw.fat_cus), fat64 = budget(sw.fat_cus64);
            if (l1.cout == 64) HH_CHECK_HIP(bb64_fused_launch(p, fat64, s));
            else if (!sw.bb32_tile && bbpc_supported(p)) HH_CHECK_HIP(bbpc_launch(p, fat, s));
            else HH_CHECK_HIP(bb_fused_launch(p, num_cus, s));
            break;
        }
        case OP_QUANT:
            if (enqueue_fp8_quant(op, B, H, W, s)) return 1;
            break;
        case OP_CONV: {
            if (dtype == 2 && !op.hi) {
                ProfRecord *pr = nullptr;
                if (prof_enabled) {
                    if (prof_used == prof.size()) {
                        ProfRecord r{};
                        HH_CHECK_HIP(hipEventCreate(&r.e0));
                        HH_CHECK_HIP(hipEventCreate(&r.e1));
                        prof.push_back(r);
                    }
                    pr = &prof[prof_used++];
                    pr->op = (int)(&op - ops.data());
                    pr->slot = -1;
                }
                if (enqueue_fp8_conv(op, B, H, W, o1, o2, s, pr)) return 1;
                break;
            }
            const ConvLayer &l = layers[op.layer];
            const TensorDesc &ti = tensors[op.in];
            ConvParams p{};
            p.Hin = H >> ti.shift; p.Win = W >> ti.shift;
            p.in = dtype == 2 ? ti.ptr16 : ti.ptr; p.in_cs = ti.C; p.in_coff = op.in_coff;  // (fp8 handle, op.hi: the bf16 representations)
            p.w = l.d_w; p.bias = l.d_bias;
            p.Ho = l.stride == 2 ? p.Hin / 2 : p.Hin;
            p.Wo = l.stride == 2 ? p.Win / 2 : p.Win;
            p.osy = p.osx = 1; p.ooy = p.oox = 0;
            p.pad_y = p.pad_x = (l.ks - 1) / 2;
            if (l.transposed) {
                p.osy = p.osx = 2; p.ooy = l.py; p.oox = l.px;
                p.pad_y = l.py == 0 ? 1 : 0; p.pad_x = l.px == 0 ? 1 : 0;
                if (l.py < 0) { p.nphase = 4; p.phase_stride = l.phase_stride; }
            }
            p.Hob = p.Ho * p.osy; p.Wob = p.Wo * p.osx;
            if (op.out >= 0) {
                const TensorDesc &to = tensors[op.out];
                p.out = dtype == 2 ? to.ptr16 : to.ptr; p.out_cs = to.C; p.out_coff = op.out_coff;
            }
            if (op.res >= 0) {
                const TensorDesc &tr = tensors[op.res];
                p.res = dtype == 2 ? tr.ptr16 : tr.ptr; p.res_cs = tr.C; p.res_coff = op.res_coff;
            }
            p.out_f32 = op.f32_out == 1 ? o1 : op.f32_out == 2 ? o2 : nullptr;
            if (op.in2 >= 0) {  // conv over the concatenated channels of two or three tensors of one shape and pixel stride
                const TensorDesc &t2 = tensors[op.in2];
                if (t2.C != ti.C || t2.shift != ti.shift || (op.in3 >= 0 && (tensors[op.in3].C != ti.C || tensors[op.in3].shift != ti.shift))) {
                    hh_set_error("merged fusion conv: inputs differ in shape or pixel stride");
                    return 1;
                }
                p.nch0 = l.mcin[0] / l.KC;
                p.nch1 = l.mcin[1] / l.KC;
                p.src_delta1 = t2.ptr - (ti.ptr + op.in_coff);
                if (op.in3 >= 0) p.src_delta2 = tensors[op.in3].ptr - (ti.ptr + op.in_coff);
            }
            p.cin = l.cin_pad;
            p.cout_real = l.cout;
            p.cout_store = op.cout_store >= 0 ? op.cout_store : round_up(l.cout, 8);
            p.relu = op.relu;
            p.B = B;
            const int cfg = pick_config(l, p.Wo);
            const ConvConfig &c = conv_config(cfg);
            p.tiles_x = (p.Wo + c.TW - 1) / c.TW;
            p.tiles_y = (p.Ho + c.th() - 1) / c.th();
            p.ncg = l.ncg;
            ProfRecord *pr = nullptr;
            if (prof_enabled) {
                if (prof_used == prof.size()) {
                    ProfRecord r{};
                    HH_CHECK_HIP(hipEventCreate(&r.e0));
                    HH_CHECK_HIP(hipEventCreate(&r.e1));
                    prof.push_back(r);
                }
                pr = &prof[prof_used++];
                pr->op = (int)(&op - ops.data());
                pr->cfg = cfg;
                pr->slot = prof_used <= HH_PROF_SLOTS ? (int)prof_used - 1 : -1;
                if (pr->slot >= 0 && prof_clk) p.clk = d_clk + 4 * pr->slot;
                const double acin = l.acct_cin ? l.acct_cin : l.cin;  // (algorithmic: the reference layer's input width)
                pr->flops = 2.0 * B * p.Ho * p.Wo * acin * l.cout * l.ks * l.ks * (p.nphase > 1 ? 4 : 1);
                {
                    const double opix = (double)B * p.Ho * p.Wo * (p.nphase > 1 ? 4 : 1);
                    pr->bytes = 2.0 * B * p.Hin * p.Win * acin + (p.out ? 2.0 * opix * l.cout : 0.0) + (p.res ? 2.0 * opix * l.cout : 0.0) +
                                (p.out_f32 ? 4.0 * opix * l.cout : 0.0) + 2.0 * acin * l.cout * l.ks * l.ks * (p.nphase > 1 ? 4 : 1);
                }
                hh_launch_probe() = LaunchProbe{pr->e0, pr->e1};  // the launch below stamps e0 / e1 from its dispatch packet
            }
            HH_CHECK_HIP(conv_launch(cfg, p, s));
            break;
        }
        }
    }
    if (multi)
        for (int l = 1; l < lanes_open; ++l) {  // close the fork: the caller's stream waits for every lane it is not behind yet
            int sn[4];
            snap(l, sn);
            if (!news(0, sn) && !sw.keep_waits) continue;
            hipEvent_t e;
            if (next_event(&e)) return 1;
            HH_CHECK_HIP(hipEventRecord(e, L[l]));
            HH_CHECK_HIP(hipStreamWaitEvent(s0, e, 0));
        }
    return 0;
}

int hh_net::forward(const float *images, int B, int H, int W, float *o1, float *o2, int use_graph, hipStream_t s)
{
    if (!finalized) { hh_set_error("hh_forward: call hh_finalize first"); return 1; }
    if (dtype == 2 && !calibrated) { hh_set_error("hh_forward: fp8 handle without activation scales, call hh_calibrate first"); return 1; }
    if (B > rB || H > rH || W > rW || !ws_ready || (taps_enabled && !taps.empty() && !taps[0].copy))
        if (reserve(B, H, W)) return 1;
    lastB = B; lastH = H; lastW = W;
    // hipGraph capture of the multi-stream fork/join segfaults inside the ROCm 7.2 runtime on this plan, so the
    // multi-lane mode always launches eagerly (at B=32 eager and graph replay time identically); graphs remain
    // available for single-lane execution (hh_set_multi_lane(net, 0)), which is what small batches want.
    // The multi-lane plan always launches eagerly.  Stream capture of its fork / join pattern segfaults inside the ROCm 7.2 runtime,
    // and the explicit graph built in round 3 (kernel nodes re-added one by one, the plan's edges as node dependencies: git log,
    // profiles/r03_ab.md) replays bit-equal but ~4 ms SLOWER per forward than the eager launches at every batch size (1.55 vs
    // 5.6 ms at batch 1): the runtime's graph executor costs ~15 us per node.  Graphs remain for single-lane execution
    // (hh_set_multi_lane(net, 0)), where eager and replay time alike.
    if (!use_graph || s == nullptr || taps_enabled || prof_enabled || sw.poison_lds || sw.debug_skip || multi_lane)
        return enqueue(images, B, H, W, o1, o2, s);
    for (auto &g : graphs)
        if (g.images == images && g.o1 == o1 && g.o2 == o2 && g.B == B && g.H == H && g.W == W) {
            HH_CHECK_HIP(hipGraphLaunch(g.exec, s));
            return 0;
        }
    hipGraph_t graph;
    HH_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue(images, B, H, W, o1, o2, s);
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc) return rc;
    HH_CHECK_HIP(e);
    GraphEntry g{images, o1, o2, B, H, W, nullptr};
    HH_CHECK_HIP(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
    hipGraphDestroy(graph);
    if (graphs.size() >= 8) {  // evict the oldest entry; it may still be executing on this stream
        HH_CHECK_HIP(hipStreamSynchronize(s));
        hipGraphExecDestroy(graphs.front().exec);
        graphs.erase(graphs.begin());
    }
    graphs.push_back(g);
    HH_CHECK_HIP(hipGraphLaunch(g.exec, s));
    return 0;
}

// Static hazard check of the multi-lane plan: replays the fork/join/dep edges of enqueue() with vector clocks (one
// component per lane) and verifies for every op that (RAW) the last writer of each tensor it reads, and (WAR/WAW) every
// earlier reader and the last writer of each tensor it writes, happen-before it.  Ops of one lane are ordered by the stream.
int hh_net::check_plan(std::string *why) const
{
    struct Stamp { int lane; int t; int op; };
    int clk[4][4] = {};  // clk[l][m] = latest event of lane m that lane l is ordered after
    int mark_clk[4][4] = {};  // the lanes' clocks at the last OP_MARK
    bool mark_ok[4] = {false, false, false, false};  // lanes the last OP_MARK recorded
    auto before = [&](const Stamp &st, int lane) { return st.lane < 0 || clk[lane][st.lane] >= st.t; };
    std::vector<Stamp> writer(tensors.size(), Stamp{-1, 0, -1});
    std::vector<std::vector<Stamp>> readers(tensors.size());
    int lanes_open = 1;
    for (size_t i = 0; i < ops.size(); ++i) {
        const Op &op = ops[i];
        if (op.kind == OP_JOIN) {
            const int nrec = op.nlanes < lanes_open ? op.nlanes : lanes_open;
            int merged[4] = {};
            for (int m = 0; m < nrec; ++m)
                for (int c = 0; c < 4; ++c) merged[c] = std::max(merged[c], clk[m][c]);
            for (int l = 0; l < op.nlanes; ++l)
                for (int c = 0; c < 4; ++c) clk[l][c] = std::max(clk[l][c], merged[c]);
            if (op.nlanes > lanes_open) lanes_open = op.nlanes;
            continue;
        }
        if (op.kind == OP_DEP) {
            for (int c = 0; c < 4; ++c) clk[op.lane][c] = std::max(clk[op.lane][c], clk[op.dep_from][c]);
            continue;
        }
        if (op.kind == OP_MARK) {
            const int nrec = op.nlanes < lanes_open ? op.nlanes : lanes_open;
            for (int m = 0; m < 4; ++m) mark_ok[m] = m >= op.dep_from && m < nrec;
            for (int m = op.dep_from; m < nrec; ++m)
                for (int c = 0; c < 4; ++c) mark_clk[m][c] = clk[m][c];
            if (op.nlanes > lanes_open) lanes_open = op.nlanes;
            continue;
        }
        if (op.kind == OP_WAITL) {
            if (!mark_ok[op.dep_from]) { if (why) *why = "op " + std::to_string(i) + " waits for a lane the last mark did not record"; return 1; }
            for (int c = 0; c < 4; ++c) clk[op.lane][c] = std::max(clk[op.lane][c], mark_clk[op.dep_from][c]);
            continue;
        }
        const int l = op.lane;
        if (l >= lanes_open) { if (why) *why = "op " + std::to_string(i) + " runs on a lane that was never forked"; return 1; }
        std::vector<int> rd, wr;
        switch (op.kind) {
        case OP_CONV: rd = {op.in, op.res, op.in2, op.in3}; wr = {op.out}; break;
        case OP_UPADD: rd = {op.in, op.up[0], op.up[1], op.up[2]}; wr = {op.out}; break;
        case OP_BB: rd = {op.in}; wr = {op.out}; break;
        case OP_JUNC: rd = {op.in, op.in2, op.res, op.in3}; wr = {op.out, op.out2}; break;
        case OP_STEM: wr = {op.out}; break;
        case OP_QUANT: rd = {op.out}; wr = {op.out}; break;
        case OP_TAP: continue;  // taps only run with the lanes switched off (enqueue: multi = ... && !taps_enabled)
        case OP_AVGPOOL: rd = {op.in}; break;
        default: break;
        }
        const int t = ++clk[l][l];
        for (int x : rd) {
            if (x < 0) continue;
            if (!before(writer[x], l)) {
                if (why) *why = "RAW: op " + std::to_string(i) + " (lane " + std::to_string(l) + ") reads tensor " + std::to_string(x) +
                                " written by op " + std::to_string(writer[x].op) + " (lane " + std::to_string(writer[x].lane) + ") without an edge";
                return 1;
            }
        }
        for (int x : wr) {
            if (x < 0) continue;
            if (!before(writer[x], l)) {
                if (why) *why = "WAW: op " + std::to_string(i) + " overwrites tensor " + std::to_string(x) + " of op " + std::to_string(writer[x].op);
                return 1;
            }
            for (const Stamp &r : readers[x])
                if (!(r.lane == l) && !before(r, l)) {
                    if (why) *why = "WAR: op " + std::to_string(i) + " (lane " + std::to_string(l) + ") overwrites tensor " + std::to_string(x) +
                                    " still read by op " + std::to_string(r.op) + " (lane " + std::to_string(r.lane) + ")";
                    return 1;
                }
        }
        for (int x : rd)
            if (x >= 0) readers[x].push_back(Stamp{l, t, (int)i});
        for (int x : wr)
            if (x >= 0) { writer[x] = Stamp{l, t, (int)i}; readers[x].clear(); }
    }
    return 0;
}

double hh_net::flops(int B, int H, int W) const
{
    double macs = 0;
    for (const Op &op : ops) {
        if (op.kind == OP_BB) {
            const TensorDesc &ti = tensors[op.in];
            const double Cb = layers[op.layer].cout;
            macs += 2.0 * (double)(H >> ti.shift) * (W >> ti.shift) * Cb * Cb * 9.0;
            continue;
        }
        if (op.kind == OP_STEM) {
            macs += (double)(H / 2) * (W / 2) * 27.0 * 64.0;
            if (op.layer2 >= 0) macs += (double)(H / 4) * (W / 4) * 576.0 * 64.0;  // conv2 rides in the same launch
            continue;
        }
        if (op.kind == OP_JUNC) {
            const TensorDesc &ti = tensors[op.in];
            macs += (double)(H >> ti.shift) * (W >> ti.shift) * 64.0 * 256.0 * (1 + (op.in2 >= 0 && op.layer4 < 0) + (op.layer3 >= 0));
            continue;
        }
        if (op.kind != OP_CONV) continue;
        const ConvLayer &l = layers[op.layer];
        const TensorDesc &ti = tensors[op.in];
        const double hin = H >> ti.shift, win = W >> ti.shift;
        const double ho = l.stride == 2 ? hin / 2 : hin, wo = l.stride == 2 ? win / 2 : win;
        // a transposed-conv phase: every input pixel meets 4 of the 16 taps per phase (16 over the 4 phases)
        macs += ho * wo * (double)(l.acct_cin ? l.acct_cin : l.cin) * l.cout * l.ks * l.ks * ((l.transposed && l.py < 0) ? 4 : 1);
    }
    if (kind == 1) macs += 2048.0 * num_classes;
    return 2.0 * macs * B;
}

hh_net::~hh_net()
{
    release_workspace();
    for (auto &r : prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    if (d_pool) hipFree(d_pool);
    if (d_clk) hipFree(d_clk);
    if (d_fc_w) hipFree(d_fc_w);
    if (d_fc_b) hipFree(d_fc_b);
    for (auto &e : lane_events) hipEventDestroy(e);
    for (int l = 1; l < 4; ++l)
        if (lane_streams[l]) hipStreamDestroy(lane_streams[l]);
    for (auto &l : layers) {
        if (l.d_w) hipFree(l.d_w);
        if (l.d_bias) hipFree(l.d_bias);
        if (l.d_mult) hipFree(l.d_mult);
        if (l.d_wfin) hipFree(l.d_wfin);
    }
    if (d_amax) hipFree(d_amax);
}

// fp32 NCHW host copy of a tap
int hh_tap_read_impl(hh_net *n, int index, float *host)
{
    if (index < 0 || index >= (int)n->taps.size() || !n->taps[index].copy) { hh_set_error("hh_tap_read: no such tap / taps disabled"); return 1; }
    const TapInfo &t = n->taps[index];
    const TensorDesc &d = n->tensors[t.tensor];
    const int B = n->lastB, h = n->lastH >> d.shift, w = n->lastW >> d.shift;
    HH_CHECK_HIP(hipDeviceSynchronize());
    if (n->dtype == 2 && !t.is16) {  // e4m3 bytes * the tensor's scale at the tap
        std::vector<unsigned char> q((size_t)B * h * w * d.C);
        HH_CHECK_HIP(hipMemcpy(q.data(), t.copy, q.size(), hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < t.C; ++c)
                for (int y = 0; y < h; ++y)
                    for (int x = 0; x < w; ++x)
                        host[(((size_t)b * t.C + c) * h + y) * w + x] = hh_e4m3_to_f32(q[(((size_t)b * h + y) * w + x) * d.C + t.coff + c]) * t.scale;
        return 0;
    }
    std::vector<bf16_raw> tmp((size_t)B * h * w * d.C);
    HH_CHECK_HIP(hipMemcpy(tmp.data(), t.copy, tmp.size() * 2, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < t.C; ++c)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x)
                    host[(((size_t)b * t.C + c) * h + y) * w + x] = bf2f(tmp[(((size_t)b * h + y) * w + x) * d.C + t.coff + c]);
    return 0;
}
