// extern "C" surface declared in include/hhrnet.h (network part).
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hhrnet.h"
#include "engine.h"

const char *hh_get_error();
int hh_tap_read_impl(hh_net *n, int index, float *host);

extern "C" {

int hh_abi_version(void) { return HH_ABI_VERSION; }
const char *hh_last_error(void) { return hh_get_error(); }

hh_net *hh_create(int num_kpts, int C, int dtype)
{
    if (dtype != HH_DTYPE_BF16 && dtype != HH_DTYPE_FP8) { hh_set_error("hh_create: dtype must be HH_DTYPE_BF16 or HH_DTYPE_FP8"); return nullptr; }
    if (num_kpts <= 0 || num_kpts > 64 || C <= 0 || C % 16) { hh_set_error("hh_create: need 0 < num_kpts <= 64 and C % 16 == 0"); return nullptr; }
    hh_net *n = new hh_net();
    n->K = num_kpts; n->C = C; n->dtype = dtype;
    n->sw = PlanSwitches::from_env();
    n->build();
    return n;
}
hh_net *hh_create_classifier(int C, int num_classes, int dtype)
{
    if (dtype != HH_DTYPE_BF16) { hh_set_error("hh_create_classifier: only HH_DTYPE_BF16 is implemented"); return nullptr; }
    if (C <= 0 || C % 16 || num_classes <= 0) { hh_set_error("hh_create_classifier: need C % 16 == 0 and num_classes > 0"); return nullptr; }
    hh_net *n = new hh_net();
    n->K = 17; n->C = C; n->dtype = dtype; n->kind = 1; n->num_classes = num_classes;
    n->sw = PlanSwitches::from_env();
    n->build();
    return n;
}
int hh_forward_classifier(hh_net *net, const float *images, int B, int H, int W, float *logits, void *stream)
{
    if (!net || net->kind != 1) { hh_set_error("hh_forward_classifier: not a classifier handle"); return 1; }
    if (!images || !logits) { hh_set_error("hh_forward_classifier: null buffer"); return 1; }
    if (B <= 0 || H <= 0 || W <= 0 || H % 32 || W % 32) { hh_set_error("hh_forward_classifier: H and W must be positive multiples of 32"); return 1; }
    return net->forward(images, B, H, W, logits, nullptr, 0, (hipStream_t)stream);
}
void hh_destroy(hh_net *net) { delete net; }

int hh_num_params(const hh_net *net) { return (int)net->params.size(); }
const char *hh_param_name(const hh_net *net, int i) { return (i >= 0 && i < (int)net->params.size()) ? net->params[i].name.c_str() : nullptr; }
int hh_param_shape(const hh_net *net, int i, int64_t shape[4])
{
    if (i < 0 || i >= (int)net->params.size()) return -1;
    const auto &s = net->params[i].shape;
    for (size_t d = 0; d < s.size(); ++d) shape[d] = s[d];
    return (int)s.size();
}

int hh_load_weights(hh_net *net, const char *name, const float *host, const int64_t *shape, int ndim)
{
    auto it = net->param_index.find(name);
    if (it == net->param_index.end()) { hh_set_error(std::string("hh_load_weights: unexpected key ") + name); return 1; }
    ParamSlot &p = net->params[it->second];
    if (p.counter) { p.loaded = true; return 0; }
    if (ndim != (int)p.shape.size()) { hh_set_error(std::string("hh_load_weights: rank mismatch for ") + name); return 1; }
    size_t n = 1;
    for (int d = 0; d < ndim; ++d) {
        if (shape[d] != p.shape[d]) { hh_set_error(std::string("hh_load_weights: size mismatch for ") + name); return 1; }
        n *= (size_t)shape[d];
    }
    p.data.assign(host, host + n);
    p.loaded = true;
    net->finalized = false;
    return 0;
}
int hh_e4m3_encode(const float *x, int64_t n, unsigned char *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = hh_f32_to_e4m3(x[i]);
    return 0;
}
int hh_e4m3_decode(const unsigned char *x, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = hh_e4m3_to_f32(x[i]);
    return 0;
}
int hh_finalize(hh_net *net) { return net->finalize(); }
int hh_calibrate(hh_net *net, const float *images, int B, int H, int W, int rounds, void *stream)
{
    if (!images || B <= 0 || H <= 0 || W <= 0 || H % 32 || W % 32) { hh_set_error("hh_calibrate: need images [B,3,H,W] with H, W multiples of 32"); return 1; }
    return net->calibrate(images, B, H, W, rounds > 0 ? rounds : 2, (hipStream_t)stream);
}
int hh_reserve(hh_net *net, int B, int H, int W) { return net->reserve(B, H, W); }
int64_t hh_workspace_bytes(const hh_net *net) { return net->ws_bytes; }

int hh_forward(hh_net *net, const float *images, int B, int H, int W, float *init_heatmaps, float *deconv_heatmaps,
               int use_graph, void *stream)
{
    if (net->kind != 0) { hh_set_error("hh_forward: classifier handle, use hh_forward_classifier"); return 1; }
    if (!images || !init_heatmaps || !deconv_heatmaps) { hh_set_error("hh_forward: null buffer"); return 1; }
    if (B <= 0 || H <= 0 || W <= 0 || H % 32 || W % 32) { hh_set_error("hh_forward: H and W must be positive multiples of 32"); return 1; }
    return net->forward(images, B, H, W, init_heatmaps, deconv_heatmaps, use_graph, (hipStream_t)stream);
}
double hh_forward_flops(const hh_net *net, int B, int H, int W) { return net->flops(B, H, W); }

int hh_set_multi_lane(hh_net *net, int enable)
{
    net->multi_lane = enable != 0;
    for (auto &g : net->graphs) hipGraphExecDestroy(g.exec);
    net->graphs.clear();
    return 0;
}
int hh_set_taps(hh_net *net, int enable) { net->taps_enabled = enable != 0; return 0; }
int hh_num_taps(const hh_net *net) { return (int)net->taps.size(); }
const char *hh_tap_name(const hh_net *net, int i) { return (i >= 0 && i < (int)net->taps.size()) ? net->taps[i].name.c_str() : nullptr; }
int hh_tap_shape(const hh_net *net, int i, int64_t shape[4])
{
    if (i < 0 || i >= (int)net->taps.size()) return 1;
    const TensorDesc &d = net->tensors[net->taps[i].tensor];
    shape[0] = net->lastB; shape[1] = net->taps[i].C; shape[2] = net->lastH >> d.shift; shape[3] = net->lastW >> d.shift;
    return 0;
}
int hh_tap_read(hh_net *net, int index, float *host_nchw) { return hh_tap_read_impl(net, index, host_nchw); }

int hh_debug_check_plan(const hh_net *net)
{
    std::string why;
    if (net->check_plan(&why)) { hh_set_error("schedule hazard: " + why); return 1; }
    return 0;
}

int hh_profile_enable(hh_net *net, int enable) { net->prof_enabled = enable != 0; net->prof_clk = enable == 2; net->prof_used = 0; return 0; }
int hh_profile_count(const hh_net *net) { return (int)net->prof_used; }
int hh_profile_get(hh_net *net, int i, int *cfg, double *flops, double *bytes, float *ms, float *kernel_ms, const char **layer)
{
    if (i < 0 || i >= (int)net->prof_used) { hh_set_error("hh_profile_get: index out of range"); return 1; }
    const ProfRecord &r = net->prof[i];
    HH_CHECK_HIP(hipEventSynchronize(r.e1));
    HH_CHECK_HIP(hipEventElapsedTime(ms, r.e0, r.e1));
    *cfg = r.cfg; *flops = r.flops; *bytes = r.bytes;
    *layer = net->layers[net->ops[r.op].layer].conv.c_str();
    *kernel_ms = -1.f;
    if (r.slot >= 0 && net->prof_clk && net->d_clk && net->clk_khz > 0) {
        unsigned long long t[2];
        HH_CHECK_HIP(hipMemcpy(t, net->d_clk + 4 * r.slot, 16, hipMemcpyDeviceToHost));
        if (t[1] > t[0]) *kernel_ms = (float)((double)(t[1] - t[0]) / net->clk_khz);
    }
    return 0;
}
int hh_profile_clock(hh_net *net, int i, double *ghz)
{
    if (i < 0 || i >= (int)net->prof_used) { hh_set_error("hh_profile_clock: index out of range"); return 1; }
    const ProfRecord &r = net->prof[i];
    *ghz = 0.0;
    if (r.slot >= 0 && net->prof_clk && net->d_clk && net->clk_khz > 0) {
        unsigned long long t[2];
        HH_CHECK_HIP(hipEventSynchronize(r.e1));
        HH_CHECK_HIP(hipMemcpy(t, net->d_clk + 4 * r.slot + 2, 16, hipMemcpyDeviceToHost));
        if (t[1]) *ghz = (double)t[0] / (double)t[1] * net->clk_khz * 1e-6;  // core cycles per wall tick x wall ticks per second
    }
    return 0;
}
int hh_conv_config(int cfg, int out[7])
{
    if (cfg >= 1000 && cfg - 1000 < conv_fp8_num_configs()) {  // fp8 instantiations are reported as 1000 + index
        const Fp8ConvConfig &c = conv_fp8_config(cfg - 1000);
        out[0] = c.KS; out[1] = c.S; out[2] = c.KC; out[3] = c.NT; out[4] = 1; out[5] = c.PT; out[6] = c.TW;
        return 0;
    }
    if (cfg < 0 || cfg >= conv_num_configs()) return 1;
    const ConvConfig &c = conv_config(cfg);
    out[0] = c.KS; out[1] = c.S; out[2] = c.KC; out[3] = c.NT; out[4] = c.WC; out[5] = c.PT; out[6] = c.TW;
    return 0;
}

int hh_conv_config_double_buffered(int cfg)
{
    if (cfg >= 1000 && cfg - 1000 < conv_fp8_num_configs()) return 0;
    if (cfg < 0 || cfg >= conv_num_configs()) return -1;
    return conv_config(cfg).DB;
}

int hh_preprocess_u8(const unsigned char *image_hwc, int h, int w, const double dst_to_src[6], float *out_nchw, int H, int W,
                     const float mean[3], const float stdv[3], void *stream)
{
    if (!image_hwc || !out_nchw || h <= 0 || w <= 0 || H <= 0 || W <= 0) { hh_set_error("hh_preprocess_u8: bad argument"); return 1; }
    HH_CHECK_HIP(launch_preprocess(image_hwc, h, w, dst_to_src, out_nchw, H, W, mean, stdv, (hipStream_t)stream));
    return 0;
}

int hh_preprocess_u8_batch(const unsigned char *images_base, const hh_image_desc *descs_dev, int n, float *out_nchw, int H, int W,
                           const float mean[3], const float stdv[3], void *stream)
{
    static_assert(sizeof(hh_image_desc) == sizeof(HHImageDesc) && sizeof(HHImageDesc) == 64, "descriptor layout");
    if (!images_base || !descs_dev || !out_nchw || n <= 0 || n > 65535 || H <= 0 || W <= 0) { hh_set_error("hh_preprocess_u8_batch: bad argument"); return 1; }
    HH_CHECK_HIP(launch_preprocess_batch(images_base, reinterpret_cast<const HHImageDesc *>(descs_dev), n, out_nchw, H, W, mean, stdv,
                                         (hipStream_t)stream));
    return 0;
}

int hh_loss_heatmaps(const float *pred, int64_t pred_bstride, const float *target, const float *mask, int B, int K, int h, int w,
                     float *loss, float *grad, int64_t grad_bstride, double *scratch, void *stream)
{
    if (!pred || !target || !mask || !loss || !scratch || B <= 0 || K <= 0 || h <= 0 || w <= 0) { hh_set_error("hh_loss_heatmaps: bad argument"); return 1; }
    if ((h * w) % 4 || pred_bstride % 4 || (grad && grad_bstride % 4)) { hh_set_error("hh_loss_heatmaps: h*w and the batch strides must be multiples of 4"); return 1; }
    HH_CHECK_HIP(launch_masked_mse(pred, pred_bstride, target, mask, B, K, h, w, loss, grad, grad_bstride, scratch, (hipStream_t)stream));
    return 0;
}

int hh_loss_ae_grouping(const float *tags, int64_t tags_bstride, const int32_t *joints, const int32_t *num_people, int B, int P, int K,
                        int h, int w, float *push_pull, float *grad, int64_t grad_bstride, float push_scale, float pull_scale,
                        double *scratch, void *stream)
{
    if (!tags || !joints || !num_people || !push_pull || !scratch || B <= 0 || P <= 0 || K <= 0 || h <= 0 || w <= 0) { hh_set_error("hh_loss_ae_grouping: bad argument"); return 1; }
    if (P > 2048) { hh_set_error("hh_loss_ae_grouping: at most 2048 people per image"); return 1; }
    HH_CHECK_HIP(launch_ae_grouping(tags, tags_bstride, joints, num_people, B, P, K, h, w, push_pull, grad, grad_bstride, push_scale,
                                    pull_scale, scratch, (hipStream_t)stream));
    return 0;
}

// ------------------------------------------------------------------ training building blocks
static int round_up_i(int a, int b) { return (a + b - 1) / b * b; }

int64_t hh_conv2d_workspace_bytes(int cin, int cout, int ks, int mode)
{
    const int ci = mode ? cout : cin, co = mode ? cin : cout;
    const int coutp = round_up_i(co, 32);
    int KC = 0, NT = 0;
    if (hh_family_pick(mode == 2 ? 2 : ks, 1, round_up_i(ci, 16), coutp, &KC, &NT)) return -1;
    const int cin_pad = round_up_i(ci, KC);
    return (int64_t)coutp * cin_pad * ks * ks * 2 + (int64_t)coutp * 4 + 512;
}

// Shape bookkeeping shared by hh_conv2d / hh_conv2d_packed / hh_pack_conv_weights_batch: the conv that actually runs for
// (cin, cout, ks, stride, mode) and the size of one packed weight set.
struct Conv2dPlan {
    int ci, co, coutp, kks, kstride, KC, NT, COUT_T, cin_pad, nsets;
    size_t wel;  // bf16 elements of ONE packed set (mode 2 has four: the output-parity phases)
};
static int conv2d_plan(int cin, int cout, int ks, int stride, int mode, Conv2dPlan *pl, const char *who)
{
    pl->ci = mode ? cout : cin; pl->co = mode ? cin : cout;  // channels of the conv that actually runs
    if (pl->ci % 16 || pl->co % 8) { hh_set_error((std::string(who) + ": input channels must be a multiple of 16 and output channels of 8").c_str()); return 1; }
    if (mode == 1 && stride != 1) { hh_set_error((std::string(who) + ": mode 1 is the data gradient of a stride-1 convolution").c_str()); return 1; }
    if (ks == 2 && stride != 1) { hh_set_error((std::string(who) + ": 2x2 kernels run at stride 1 only").c_str()); return 1; }
    if (mode == 2 && (stride != 2 || ks != 3)) { hh_set_error((std::string(who) + ": mode 2 is the data gradient of a 3x3 stride-2 convolution").c_str()); return 1; }
    pl->coutp = round_up_i(pl->co, 32);
    pl->kks = mode == 2 ? 2 : ks; pl->kstride = mode == 2 ? 1 : stride;  // kernel that actually runs
    if (hh_family_pick(pl->kks, pl->kstride, pl->ci, pl->coutp, &pl->KC, &pl->NT)) { hh_set_error((std::string(who) + ": no kernel family for this shape").c_str()); return 1; }
    pl->COUT_T = 32 * pl->NT; pl->cin_pad = round_up_i(pl->ci, pl->KC);
    pl->wel = (size_t)pl->coutp * pl->cin_pad * pl->kks * pl->kks;
    pl->nsets = mode == 2 ? 4 : 1;
    return 0;
}

// w: fp32 weights (packed into `workspace` here) or, with prepacked != NULL, weight sets packed by hh_pack_conv_weights_batch
static int conv2d_impl(const void *x, int B, int H, int W, int cin, const float *w, const bf16_raw *prepacked, int cout, int ks, int stride,
                       int mode, int pad_y, int pad_x, const float *bias, const void *res, int relu, void *y, void *workspace, void *stream,
                       const char *who)
{
    if (pad_y < 0) pad_y = (ks - 1) / 2;
    if (pad_x < 0) pad_x = (ks - 1) / 2;
    static bool inited = false;
    if (!inited) { HH_CHECK_HIP(conv_init()); inited = true; }
    if (!x || (!w && !prepacked) || !y || (!workspace && !prepacked) || B <= 0 || H <= 0 || W <= 0) { hh_set_error((std::string(who) + ": bad argument").c_str()); return 1; }
    Conv2dPlan pl;
    if (conv2d_plan(cin, cout, ks, stride, mode, &pl, who)) return 1;
    if (mode == 1) { pad_y = ks - 1 - pad_y; pad_x = ks - 1 - pad_x; }  // the adjoint correlates with the rotated kernel
    const int co = pl.co, coutp = pl.coutp, KC = pl.KC, COUT_T = pl.COUT_T;
    const size_t wel = pl.wel;
    bf16_raw *packed = (bf16_raw *)workspace;
    float *zbias = workspace ? (float *)((char *)workspace + ((wel * 2 + 255) & ~(size_t)255)) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    // the kernel reads coutp bias values: an all-zero device array serves the bias-free case (no memset per call), a bias
    // whose length is already a multiple of 32 is used in place, anything else is copied behind the packed weights
    static float *zeros_dev[64] = {};  // one per device: a process may drive several GPUs
    if (!bias && coutp <= 4096) {
        int dev = 0;
        HH_CHECK_HIP(hipGetDevice(&dev));
        float *&zeros = zeros_dev[dev & 63];
        if (!zeros) {  // (once per device; the wait: a null-stream memset is not ordered in front of launches on a non-blocking stream)
            HH_CHECK_HIP(hipMalloc((void **)&zeros, 4096 * 4));
            HH_CHECK_HIP(hipMemset(zeros, 0, 4096 * 4));
            HH_CHECK_HIP(hipDeviceSynchronize());
        }
        zbias = zeros;
    } else if (bias && co == coutp) {
        zbias = const_cast<float *>(bias);
    } else if (!workspace) {
        hh_set_error((std::string(who) + ": without a workspace the bias must be NULL or have a multiple of 32 entries").c_str());
        return 1;
    } else if (!bias) {
        HH_CHECK_HIP(hipMemsetAsync(zbias, 0, (size_t)coutp * 4, s));
    } else {
        HH_CHECK_HIP(hipMemcpyAsync(zbias, bias, (size_t)co * 4, hipMemcpyDeviceToDevice, s));
    }
    const int Ho = pl.kstride == 2 ? H / 2 : H, Wo = pl.kstride == 2 ? W / 2 : W;
    const int cfg = hh_pick_config(pl.kks, pl.kstride, KC, pl.NT, Wo);
    if (cfg < 0) { hh_set_error((std::string(who) + ": no kernel instantiation for this shape").c_str()); return 1; }
    const ConvConfig &cc = conv_config(cfg);
    ConvParams p{};
    p.in = (const bf16_raw *)x; p.in_cs = pl.ci; p.Hin = H; p.Win = W;
    p.w = prepacked ? prepacked : packed; p.bias = zbias;
    p.res = (const bf16_raw *)res; p.res_cs = co;
    p.out = (bf16_raw *)y; p.out_cs = co;
    p.Ho = Ho; p.Wo = Wo; p.Hob = Ho; p.Wob = Wo; p.osy = p.osx = 1;
    p.cin = pl.cin_pad; p.cout_real = co; p.cout_store = co; p.relu = relu;
    p.pad_y = pad_y; p.pad_x = pad_x; p.B = B;
    p.tiles_x = (Wo + cc.TW - 1) / cc.TW; p.tiles_y = (Ho + cc.th() - 1) / cc.th(); p.ncg = coutp / cc.cout_t();
    if (mode == 2) {  // four output-parity phases, each a 2x2 conv over dL/dy scattered onto the 2H x 2W grid
        p.Hob = 2 * H; p.Wob = 2 * W; p.osy = p.osx = 2; p.pad_y = p.pad_x = 0;
        for (int ph = 0; ph < 4; ++ph) {
            if (prepacked) p.w = prepacked + (size_t)ph * wel;
            else HH_CHECK_HIP(launch_pack_weights(w, cout, cin, 2, 2, KC, COUT_T, packed, wel, s, ph >> 1, ph & 1));
            p.ooy = ph >> 1; p.oox = ph & 1;
            HH_CHECK_HIP(conv_launch(cfg, p, s));
        }
        return 0;
    }
    if (!prepacked) HH_CHECK_HIP(launch_pack_weights(w, cout, cin, ks, mode, KC, COUT_T, packed, wel, s));
    HH_CHECK_HIP(conv_launch(cfg, p, s));
    return 0;
}

int hh_conv2d(const void *x, int B, int H, int W, int cin, const float *w, int cout, int ks, int stride, int mode, int pad_y, int pad_x,
              const float *bias, const void *res, int relu, void *y, void *workspace, void *stream)
{
    if (!w || !workspace) { hh_set_error("hh_conv2d: bad argument"); return 1; }
    return conv2d_impl(x, B, H, W, cin, w, nullptr, cout, ks, stride, mode, pad_y, pad_x, bias, res, relu, y, workspace, stream, "hh_conv2d");
}

int hh_conv2d_packed(const void *x, int B, int H, int W, int cin, const void *w_packed, int cout, int ks, int stride, int mode, int pad_y,
                     int pad_x, const float *bias, const void *res, int relu, void *y, void *stream)
{
    if (!w_packed) { hh_set_error("hh_conv2d_packed: bad argument"); return 1; }
    return conv2d_impl(x, B, H, W, cin, nullptr, (const bf16_raw *)w_packed, cout, ks, stride, mode, pad_y, pad_x, bias, res, relu, y, nullptr,
                       stream, "hh_conv2d_packed");
}

int64_t hh_conv2d_packed_elems(int cin, int cout, int ks, int stride, int mode)
{
    Conv2dPlan pl;
    if (conv2d_plan(cin, cout, ks, stride, mode, &pl, "hh_conv2d_packed_elems")) return -1;
    return (int64_t)(pl.wel * pl.nsets);
}

int hh_pack_conv_weights_batch(int n, const float *const *w, void *const *packed, const int32_t *shapes, void *descs_dev, void *stream)
{
    if (n < 0 || (n && (!w || !packed || !shapes || !descs_dev))) { hh_set_error("hh_pack_conv_weights_batch: bad argument"); return 1; }
    std::vector<PackDesc> d;
    d.reserve((size_t)n * 4);
    for (int i = 0; i < n; ++i) {
        const int cout = shapes[5 * i], cin = shapes[5 * i + 1], ks = shapes[5 * i + 2], stride = shapes[5 * i + 3], mode = shapes[5 * i + 4];
        Conv2dPlan pl;
        if (!w[i] || !packed[i] || conv2d_plan(cin, cout, ks, stride, mode, &pl, "hh_pack_conv_weights_batch")) {
            if (w[i] && packed[i]) return 1;
            hh_set_error("hh_pack_conv_weights_batch: null pointer in the table");
            return 1;
        }
        for (int ph = 0; ph < pl.nsets; ++ph) {
            PackDesc e{};
            e.W = w[i]; e.packed = (bf16_raw *)packed[i] + (size_t)ph * pl.wel; e.total = (long long)pl.wel;
            e.cout = cout; e.cin = cin; e.ks = pl.kks; e.mode = mode; e.KC = pl.KC; e.COUT_T = pl.COUT_T; e.py = ph >> 1; e.px = ph & 1;
            d.push_back(e);
        }
    }
    if (d.empty()) return 0;
    // (pageable source: the copy is staged before hipMemcpyAsync returns, so the vector may go)
    HH_CHECK_HIP(hipMemcpyAsync(descs_dev, d.data(), d.size() * sizeof(PackDesc), hipMemcpyHostToDevice, (hipStream_t)stream));
    HH_CHECK_HIP(launch_pack_weights_batch((const PackDesc *)descs_dev, (int)d.size(), (hipStream_t)stream));
    return 0;
}

int64_t hh_conv2d_wgrad_workspace_bytes(int B, int H, int W, int cin, int cout, int ks, int stride)
{
    const int Ho = stride == 2 ? H / 2 : H, Wo = stride == 2 ? W / 2 : W;
    // one partial-sum set [ks*ks][cout64][cin64] per worker; the reduction combines them in LDS (no staging rows since round 3)
    return (int64_t)conv_wgrad_num_workers(B, Ho, Wo, stride, cin, cout) * ks * ks * round_up_i(cout, 64) * round_up_i(cin, 64) * 4;
}

int hh_conv2d_wgrad(const void *x, const void *dy, int B, int H, int W, int cin, int cout, int ks, int stride, int pad_y, int pad_x, float *dw,
                    void *workspace, void *stream)
{
    if (pad_y < 0) pad_y = (ks - 1) / 2;
    if (pad_x < 0) pad_x = (ks - 1) / 2;
    if (!x || !dy || !dw || !workspace || B <= 0 || H <= 0 || W <= 0) { hh_set_error("hh_conv2d_wgrad: bad argument"); return 1; }
    if (cin % 8 || cout % 8) { hh_set_error("hh_conv2d_wgrad: channel counts must be multiples of 8"); return 1; }
    if (!((ks == 3 && (stride == 1 || stride == 2)) || ((ks == 1 || ks == 2) && stride == 1))) { hh_set_error("hh_conv2d_wgrad: 3x3 (stride 1 or 2), 2x2 and 1x1 (stride 1) only"); return 1; }
    WgradParams p{};
    p.x = (const bf16_raw *)x; p.dy = (const bf16_raw *)dy; p.partial = (float *)workspace;
    p.B = B; p.H = H; p.W = W; p.Ho = stride == 2 ? H / 2 : H; p.Wo = stride == 2 ? W / 2 : W; p.cin = cin; p.cout = cout;
    p.pad_y = pad_y; p.pad_x = pad_x;
    HH_CHECK_HIP(conv_wgrad_launch(p, ks, stride, dw, (hipStream_t)stream));
    return 0;
}

int hh_bn_train_forward(const void *x, int64_t P, int C, const float *gamma, const float *beta, float eps, const void *res, int relu,
                        void *y, float *mean, float *invstd, double *scratch, void *stream)
{
    if (!x || !y || !gamma || !beta || !mean || !invstd || !scratch || P <= 0 || C <= 0 || C % 8 || C > 2048) { hh_set_error("hh_bn_train_forward: bad argument (C must be a multiple of 8, <= 2048)"); return 1; }
    HH_CHECK_HIP(launch_bn_train_forward((const bf16_raw *)x, C, (size_t)P, C, gamma, beta, eps, (const bf16_raw *)res, relu, (bf16_raw *)y,
                                         mean, invstd, scratch, (hipStream_t)stream));
    return 0;
}

int hh_bn_train_backward(const void *x, const void *y, const void *dy, int64_t P, int C, const float *mean, const float *invstd,
                         const float *gamma, int relu, void *dx, void *dres, float *dgamma, float *dbeta, double *scratch, void *stream)
{
    if (!x || !y || !dy || !dx || !mean || !invstd || !gamma || !dgamma || !dbeta || !scratch || P <= 0 || C <= 0 || C % 8 || C > 2048) { hh_set_error("hh_bn_train_backward: bad argument"); return 1; }
    HH_CHECK_HIP(launch_bn_train_backward((const bf16_raw *)x, (const bf16_raw *)y, (const bf16_raw *)dy, C, (size_t)P, C, mean, invstd, gamma,
                                          nullptr, relu, (bf16_raw *)dx, (bf16_raw *)dres, dgamma, dbeta, scratch, (hipStream_t)stream));
    return 0;
}

int hh_bn_train_backward_plain(const void *x, const void *dy, int64_t P, int C, const float *mean, const float *invstd, const float *gamma,
                               const float *beta, int relu, void *dx, float *dgamma, float *dbeta, double *scratch, void *stream)
{
    if (!x || !dy || !dx || !mean || !invstd || !gamma || !beta || !dgamma || !dbeta || !scratch || P <= 0 || C <= 0 || C % 8 || C > 2048) { hh_set_error("hh_bn_train_backward_plain: bad argument"); return 1; }
    HH_CHECK_HIP(launch_bn_train_backward((const bf16_raw *)x, nullptr, (const bf16_raw *)dy, C, (size_t)P, C, mean, invstd, gamma, beta, relu,
                                          (bf16_raw *)dx, nullptr, dgamma, dbeta, scratch, (hipStream_t)stream));
    return 0;
}

int hh_fusion_sum_forward(const void *const *terms, const int *shifts, int nterms, int B, int H, int W, int C, int relu, void *out, void *stream)
{
    if (!terms || !shifts || nterms < 1 || nterms > 4 || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || shifts[0] != 0) { hh_set_error("hh_fusion_sum_forward: 1..4 terms, the first at the output resolution, C a multiple of 8"); return 1; }
    UpAddParams p{};
    p.base = (const bf16_raw *)terms[0]; p.base_cs = C;
    p.nup = nterms - 1;
    for (int j = 1; j < nterms; ++j) {
        if (shifts[j] < 0 || shifts[j] > 5 || (H >> shifts[j]) << shifts[j] != H || (W >> shifts[j]) << shifts[j] != W) { hh_set_error("hh_fusion_sum_forward: bad shift"); return 1; }
        p.up[j - 1] = (const bf16_raw *)terms[j]; p.up_cs[j - 1] = C; p.up_shift[j - 1] = shifts[j];
    }
    p.out = (bf16_raw *)out; p.out_cs = C;
    p.B = B; p.H = H; p.W = W; p.C = C; p.relu = relu;
    HH_CHECK_HIP(launch_upadd(p, (hipStream_t)stream));
    return 0;
}
int hh_fusion_sum_backward(const void *dy, const void *out, int relu, int B, int H, int W, int C, void *g, void *const *dup, const int *up_shift, int nup,
                           void *stream)
{
    if (!dy || (relu && (!out || !g)) || nup < 0 || nup > 3 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) { hh_set_error("hh_fusion_sum_backward: bad argument"); return 1; }
    HH_CHECK_HIP(launch_upadd_backward((const bf16_raw *)dy, (const bf16_raw *)out, relu, B, H, W, C, (bf16_raw *)g, (bf16_raw *const *)dup, up_shift, nup,
                                       (hipStream_t)stream));
    return 0;
}

static bool bn_dims_ok(int64_t P, int C) { return P > 0 && C > 0 && C % 8 == 0 && C <= 2048; }

int hh_bn_train_stats(const void *x, int64_t P, int C, double *sums, double *scratch, void *stream)
{
    if (!x || !sums || !scratch || !bn_dims_ok(P, C)) { hh_set_error("hh_bn_train_stats: bad argument (C must be a multiple of 8, <= 2048)"); return 1; }
    HH_CHECK_HIP(launch_bn_train_stats((const bf16_raw *)x, C, (size_t)P, C, sums, scratch, (hipStream_t)stream));
    return 0;
}

int hh_bn_train_normalize(const void *x, int64_t P, int C, const double *sums, double count, const float *gamma, const float *beta, float eps,
                          const void *res, int relu, void *y, float *mean, float *invstd, void *stream)
{
    if (!x || !y || !sums || !gamma || !beta || !mean || !invstd || !bn_dims_ok(P, C) || !(count >= (double)P)) { hh_set_error("hh_bn_train_normalize: bad argument (count = pixels of all ranks >= P)"); return 1; }
    HH_CHECK_HIP(launch_bn_train_normalize((const bf16_raw *)x, C, (size_t)P, C, sums, count, gamma, beta, eps, (const bf16_raw *)res, relu,
                                           (bf16_raw *)y, mean, invstd, (hipStream_t)stream));
    return 0;
}

int hh_bn_train_backward_stats(const void *x, const void *y, const void *dy, int64_t P, int C, const float *mean, const float *invstd, int relu,
                               double *sums, float *dgamma, float *dbeta, double *scratch, void *stream)
{
    if (!x || !y || !dy || !mean || !invstd || !sums || !dgamma || !dbeta || !scratch || !bn_dims_ok(P, C)) { hh_set_error("hh_bn_train_backward_stats: bad argument"); return 1; }
    HH_CHECK_HIP(launch_bn_train_backward_stats((const bf16_raw *)x, (const bf16_raw *)y, (const bf16_raw *)dy, C, (size_t)P, C, mean, invstd,
                                                relu, sums, dgamma, dbeta, scratch, (hipStream_t)stream));
    return 0;
}

int hh_bn_train_backward_apply(const void *x, const void *y, const void *dy, int64_t P, int C, const float *mean, const float *invstd,
                               const float *gamma, int relu, const double *sums, double count, void *dx, void *dres, double *scratch,
                               void *stream)
{
    if (!x || !y || !dy || !dx || !mean || !invstd || !gamma || !sums || !scratch || !bn_dims_ok(P, C) || !(count >= (double)P)) { hh_set_error("hh_bn_train_backward_apply: bad argument"); return 1; }
    HH_CHECK_HIP(launch_bn_train_backward_apply((const bf16_raw *)x, (const bf16_raw *)y, (const bf16_raw *)dy, C, (size_t)P, C, mean, invstd,
                                                gamma, relu, sums, count, (bf16_raw *)dx, (bf16_raw *)dres, scratch, (hipStream_t)stream));
    return 0;
}

int hh_flip_images(const float *images, float *out, int B, int C, int H, int W, void *stream)
{
    HH_CHECK_HIP(launch_flip_images(images, out, B, C, H, W, (hipStream_t)stream));
    return 0;
}

int hh_flip_merge(float *hm, int64_t hm_bstride, const float *hm_flipped, int64_t hmf_bstride, const float *tags_flipped,
                  int64_t tf_bstride, float *tags_out, int64_t to_bstride, const int32_t *perm_host, int B, int K, int h,
                  int w, void *stream)
{
    if (K > 64) { hh_set_error("hh_flip_merge: K > 64"); return 1; }
    for (int k = 0; k < K; ++k)
        if (perm_host[k] < 0 || perm_host[k] >= K) { hh_set_error("hh_flip_merge: perm is not a permutation of 0..K-1"); return 1; }
    HH_CHECK_HIP(launch_flip_merge(hm, hm_bstride, hm_flipped, hmf_bstride, tags_flipped, tf_bstride, tags_out, to_bstride,
                                   perm_host, B, K, h, w, (hipStream_t)stream));
    return 0;
}

// cv::solve(A, b, DECOMP_LU) for the 6x6 system of cv::getAffineTransform: OpenCV's own LU (matrix_decomp.cpp LUImpl: partial
// pivoting on the largest magnitude, first one wins; eliminate with alpha = A[j][i] * (-1 / A[i][i]); back-substitute), which is
// what hal::LU64f runs for matrices this small.  Every product and sum rounded on its own.
#pragma clang fp contract(off)
static int lu_solve6(double A[6][6], double b[6])
{
    const int m = 6;
    const double eps = 2.220446049250313e-16 * 100;
    for (int i = 0; i < m; ++i) {
        int k = i;
        for (int j = i + 1; j < m; ++j)
            if (std::fabs(A[j][i]) > std::fabs(A[k][i])) k = j;
        if (std::fabs(A[k][i]) < eps) return 1;
        if (k != i) {
            for (int j = i; j < m; ++j) std::swap(A[i][j], A[k][j]);
            std::swap(b[i], b[k]);
        }
        const double d = -1 / A[i][i];
        for (int j = i + 1; j < m; ++j) {
            const double alpha = A[j][i] * d;
            for (int c = i + 1; c < m; ++c) A[j][c] += alpha * A[i][c];
            b[j] += alpha * b[i];
        }
    }
    for (int i = m - 1; i >= 0; --i) {
        double sres = b[i];
        for (int k = i + 1; k < m; ++k) sres -= A[i][k] * b[k];
        b[i] = sres / A[i][i];
    }
    return 0;
}

int hh_get_affine_transform(double cx, double cy, double scale_w, double dst_w, double dst_h, int inverse, double out[6])
{
#pragma clang fp contract(off)
    // the three point pairs of get_affine_transform(center, scale, rot=0, output_size) as the reference's numpy code leaves them
    // in its float32 arrays (base/transforms/utils.py:37-53): [center, center + (0, -scale_w/2), third], [dst centre, dst centre +
    // (0, -dst_w/2), third]; third = b + (-(a-b).y, (a-b).x) in float32
    float src[3][2], dst[3][2];
    src[0][0] = (float)cx; src[0][1] = (float)cy;
    src[1][0] = (float)(cx + 0.0); src[1][1] = (float)(cy + (-scale_w / 2));
    const float dst_dir_y = (float)(-dst_w / 2);
    dst[0][0] = (float)(dst_w * 0.5); dst[0][1] = (float)(dst_h * 0.5);
    dst[1][0] = (float)(dst_w * 0.5 + 0.0); dst[1][1] = (float)(dst_h * 0.5 + (double)dst_dir_y);
    auto third = [](float p[3][2]) {
        const float dx = p[0][0] - p[1][0], dy = p[0][1] - p[1][1];
        p[2][0] = p[1][0] + (-dy);
        p[2][1] = p[1][1] + dx;
    };
    third(src); third(dst);
    float (*from)[2] = inverse ? dst : src, (*to)[2] = inverse ? src : dst;
    double A[6][6] = {}, b[6];
    for (int i = 0; i < 3; ++i) {  // cv::getAffineTransform (imgwarp.cpp): rows (x y 1 0 0 0), (0 0 0 x y 1)
        A[2 * i][0] = A[2 * i + 1][3] = from[i][0];
        A[2 * i][1] = A[2 * i + 1][4] = from[i][1];
        A[2 * i][2] = A[2 * i + 1][5] = 1;
        b[2 * i] = to[i][0];
        b[2 * i + 1] = to[i][1];
    }
    if (lu_solve6(A, b)) { hh_set_error("hh_get_affine_transform: degenerate point set (scale_w or dst_w is 0)"); return 1; }
    for (int i = 0; i < 6; ++i) out[i] = b[i];
    return 0;
}

int hh_invert_affine(const double m[6], double out[6])
{
#pragma clang fp contract(off)
    // cv::warpAffine without WARP_INVERSE_MAP (imgwarp.cpp), same order of operations
    double M[6];
    for (int i = 0; i < 6; ++i) M[i] = m[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    for (int i = 0; i < 6; ++i) out[i] = M[i];
    return 0;
}

int hh_warp_affine_u8(const unsigned char *image_hwc, int h, int w, const double dst_to_src[6], unsigned char *out_hwc, int H, int W, void *stream)
{
    if (!image_hwc || !out_hwc || h <= 0 || w <= 0 || H <= 0 || W <= 0) { hh_set_error("hh_warp_affine_u8: bad argument"); return 1; }
    HH_CHECK_HIP(launch_warp_affine_u8(image_hwc, h, w, dst_to_src, out_hwc, H, W, (hipStream_t)stream));
    return 0;
}

int hh_transform_coords(const float *xy_in, int n, double cx, double cy, double scale_w, double dst_w, double dst_h,
                        double *xy_out)
{
#pragma clang fp contract(off)
    double M[6];
    if (hh_get_affine_transform(cx, cy, scale_w, dst_w, dst_h, 1, M)) return 1;
    for (int i = 0; i < n; ++i) {  // affine_transform (base/transforms/utils.py:5-8): M @ (x, y, 1) in float64
        const double x = (double)xy_in[2 * i + 0], y = (double)xy_in[2 * i + 1];
        xy_out[2 * i + 0] = M[0] * x + M[1] * y + M[2] * 1.0;
        xy_out[2 * i + 1] = M[3] * x + M[4] * y + M[5] * 1.0;
    }
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// Kernel micro-benchmark (tools/conv_bench.py): one convolution shape, random bf16 data, `iters` back-to-back
// launches of instantiation `cfg` timed with HIP events.  If `max_diff`
// is given, the output is also compared with generic instantiation `ref_cfg` on the same data.  Not on the hot path.
extern "C" int hh_debug_conv_bench(int cfg, int B, int Hin, int Win, int cin, int cout, int with_res, int relu, int iters,
                                   float *ms_per_launch, unsigned long long *stamps16, int ref_cfg, float *max_diff)
{
    HH_CHECK_HIP(conv_init());
    if (cfg < 0 || cfg >= conv_num_configs()) { hh_set_error("bad cfg"); return 1; }
    ConvConfig c = conv_config(cfg);
    if (cin % c.KC) { hh_set_error("cin must be a multiple of the config's KC"); return 1; }
    const int coutp = (cout + c.cout_t() - 1) / c.cout_t() * c.cout_t();
    const int Ho = c.S == 2 ? Hin / 2 : Hin, Wo = c.S == 2 ? Win / 2 : Win;
    const size_t n_in = (size_t)B * Hin * Win * cin, n_out = (size_t)B * Ho * Wo * coutp;
    std::vector<bf16_raw> h_in(n_in), h_res(n_out);
    std::vector<float> W((size_t)cout * cin * c.KS * c.KS), scale(coutp, 1.f);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (bf16_raw)(0x3c00u + ((s >> 16) & 0x1ffu) + ((s >> 30) << 15)); };
    auto rndf = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 20) - 2048) / 16384.0f; };
    for (auto &v : h_in) v = rnd();
    for (auto &v : h_res) v = rnd();
    for (auto &v : W) v = rndf();
    bf16_raw *d_in, *d_out, *d_out2, *d_res, *d_zero;
    float *d_bias;
    HH_CHECK_HIP(hipMalloc((void **)&d_in, n_in * 2));
    HH_CHECK_HIP(hipMalloc((void **)&d_out, n_out * 2));
    HH_CHECK_HIP(hipMalloc((void **)&d_out2, n_out * 2));
    HH_CHECK_HIP(hipMalloc((void **)&d_res, n_out * 2));
    HH_CHECK_HIP(hipMalloc((void **)&d_bias, (size_t)coutp * 4));
    HH_CHECK_HIP(hipMalloc((void **)&d_zero, 256));
    HH_CHECK_HIP(hipMemcpy(d_in, h_in.data(), n_in * 2, hipMemcpyHostToDevice));
    HH_CHECK_HIP(hipMemcpy(d_res, h_res.data(), n_out * 2, hipMemcpyHostToDevice));
    HH_CHECK_HIP(hipMemset(d_bias, 0, (size_t)coutp * 4));
    HH_CHECK_HIP(hipMemset(d_zero, 0, 256));
    HH_CHECK_HIP(hipMemset(d_out, 0, n_out * 2));
    HH_CHECK_HIP(hipMemset(d_out2, 0, n_out * 2));
    unsigned long long *d_st = nullptr;
    HH_CHECK_HIP(hipMalloc((void **)&d_st, 16 * 8));
    HH_CHECK_HIP(hipMemset(d_st, 0, 16 * 8));
    hipStream_t st;
    HH_CHECK_HIP(hipStreamCreate(&st));
    auto run = [&](int k, bf16_raw *out, int n, float *ms_out) -> int {
        const ConvConfig cc = conv_config(k);
        std::vector<bf16_raw> packed;
        hh_pack_weights(W.data(), scale.data(), cc.KS, cin, cout, cc.KC, cc.cout_t(), false, 0, 0, packed);
        bf16_raw *d_w;
        HH_CHECK_HIP(hipMalloc((void **)&d_w, packed.size() * 2));
        HH_CHECK_HIP(hipMemcpy(d_w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
        ConvParams p{};
        p.in = d_in; p.in_cs = cin; p.Hin = Hin; p.Win = Win; p.w = d_w; p.bias = d_bias;
        p.res = with_res ? d_res : nullptr; p.res_cs = coutp;
        p.out = out; p.out_cs = coutp; p.Hob = Ho; p.Wob = Wo; p.osy = p.osx = 1;
        p.Ho = Ho; p.Wo = Wo; p.cin = cin; p.cout_real = cout; p.cout_store = coutp; p.relu = relu;
        p.pad_y = p.pad_x = (cc.KS - 1) / 2; p.B = B; p.stamps = d_st;
        p.tiles_x = (Wo + cc.TW - 1) / cc.TW; p.tiles_y = (Ho + cc.th() - 1) / cc.th(); p.ncg = coutp / cc.cout_t();
        auto launch = [&]() { return conv_launch(k, p, st); };
        hipEvent_t e0, e1;
        HH_CHECK_HIP(hipEventCreate(&e0));
        HH_CHECK_HIP(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) HH_CHECK_HIP(launch());
        HH_CHECK_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < n; ++i) HH_CHECK_HIP(launch());
        HH_CHECK_HIP(hipEventRecord(e1, st));
        HH_CHECK_HIP(hipEventSynchronize(e1));
        float ms = 0;
        HH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (ms_out) *ms_out = ms / n;
        hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d_w);
        return 0;
    };
    if (run(cfg, d_out, iters, ms_per_launch)) return 1;
    if (stamps16) HH_CHECK_HIP(hipMemcpy(stamps16, d_st, 16 * 8, hipMemcpyDeviceToHost));
    if (max_diff) {
        if (run(ref_cfg, d_out2, 1, nullptr)) return 1;
        std::vector<bf16_raw> a(n_out), b2(n_out);
        HH_CHECK_HIP(hipMemcpy(a.data(), d_out, n_out * 2, hipMemcpyDeviceToHost));
        HH_CHECK_HIP(hipMemcpy(b2.data(), d_out2, n_out * 2, hipMemcpyDeviceToHost));
        float md = 0.f;
        for (size_t i = 0; i < n_out; ++i) {
            uint32_t ua = (uint32_t)a[i] << 16, ub = (uint32_t)b2[i] << 16;
            float fa, fb;
            memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
            const float d = fa > fb ? fa - fb : fb - fa;
            if (d > md || d != d) md = d;
        }
        *max_diff = md;
    }
    hipStreamDestroy(st);
    hipFree(d_in); hipFree(d_out); hipFree(d_out2); hipFree(d_res); hipFree(d_bias); hipFree(d_st); hipFree(d_zero);
    return 0;
}

// A/B of the two fused 32-channel blocks on the same random input (H, W need not be tile multiples): the largest absolute
// difference of the outputs (both round to bf16, the accumulation orders differ: expect a few bf16 ulps) and the time per launch.
extern "C" int hh_debug_bb_compare(int B, int H, int W, int iters, float *max_diff, float *ms_classic, float *ms_pc)
{
    HH_CHECK_HIP(conv_init());
    HH_CHECK_HIP(bb_fused_init());
    HH_CHECK_HIP(bbpc_init());
    const size_t n = (size_t)B * H * W * 32;
    std::vector<bf16_raw> h_in(n), h_w(2 * 9 * 32 * 32), h_o1(n), h_o2(n);
    std::vector<float> h_b(64);
    uint32_t s = 4242u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (bf16_raw)(0x3c00u + ((s >> 16) & 0x1ffu) + ((s >> 30) << 15)); };  // +-[0.0078, 0.031)
    for (auto &v : h_in) v = (bf16_raw)(rnd() + 0x0300u);
    for (auto &v : h_w) v = rnd();
    for (auto &v : h_b) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 20) - 2048) * 1e-4f; }
    bf16_raw *d_in, *d_o1, *d_o2, *d_w;
    float *d_b;
    HH_CHECK_HIP(hipMalloc((void **)&d_in, n * 2));
    HH_CHECK_HIP(hipMalloc((void **)&d_o1, n * 2));
    HH_CHECK_HIP(hipMalloc((void **)&d_o2, n * 2));
    HH_CHECK_HIP(hipMalloc((void **)&d_w, h_w.size() * 2));
    HH_CHECK_HIP(hipMalloc((void **)&d_b, 64 * 4));
    HH_CHECK_HIP(hipMemcpy(d_in, h_in.data(), n * 2, hipMemcpyHostToDevice));
    HH_CHECK_HIP(hipMemcpy(d_w, h_w.data(), h_w.size() * 2, hipMemcpyHostToDevice));
    HH_CHECK_HIP(hipMemcpy(d_b, h_b.data(), 64 * 4, hipMemcpyHostToDevice));
    HH_CHECK_HIP(hipMemset(d_o1, 0xff, n * 2));
    HH_CHECK_HIP(hipMemset(d_o2, 0xff, n * 2));
    BBParams p{};
    p.in = d_in; p.in_cs = 32; p.out_cs = 32; p.w1 = d_w; p.w2 = d_w + 9 * 32 * 32; p.b1 = d_b; p.b2 = d_b + 32;
    p.B = B; p.H = H; p.W = W;
    unsigned long long *d_st = nullptr;
    HH_CHECK_HIP(hipMalloc((void **)&d_st, 72 * 8));
    HH_CHECK_HIP(hipMemset(d_st, 0, 72 * 8));
    p.stamps = d_st;
    int dev = 0;
    hipDeviceProp_t prop;
    HH_CHECK_HIP(hipGetDevice(&dev));
    HH_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    hipStream_t st;
    HH_CHECK_HIP(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    HH_CHECK_HIP(hipEventCreate(&e0));
    HH_CHECK_HIP(hipEventCreate(&e1));
    float ms[2] = {0, 0};
    for (int v = 0; v < 2; ++v) {
        p.out = v ? d_o2 : d_o1;
        auto go = [&]() { return v ? bbpc_launch(p, prop.multiProcessorCount, st) : bb_fused_launch(p, prop.multiProcessorCount, st); };
        for (int i = 0; i < 3; ++i) HH_CHECK_HIP(go());
        HH_CHECK_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) HH_CHECK_HIP(go());
        HH_CHECK_HIP(hipEventRecord(e1, st));
        HH_CHECK_HIP(hipEventSynchronize(e1));
        HH_CHECK_HIP(hipEventElapsedTime(&ms[v], e0, e1));
#ifdef HH_STAMP
        {
            unsigned long long c[4];
            HH_CHECK_HIP(hipMemcpy(c, d_st + 64, sizeof(c), hipMemcpyDeviceToHost));
            if (c[3] > c[1]) fprintf(stderr, "  %s: in-kernel clock of the last launch %.3f GHz (%lld core cycles in %.2f us)\n", v ? "producer/consumer" : "tile form",
                                     (double)(c[2] - c[0]) / (double)(c[3] - c[1]) * 0.1, (long long)(c[2] - c[0]), (double)(c[3] - c[1]) * 0.01);
        }
#endif
    }
    HH_CHECK_HIP(hipMemcpy(h_o1.data(), d_o1, n * 2, hipMemcpyDeviceToHost));
    HH_CHECK_HIP(hipMemcpy(h_o2.data(), d_o2, n * 2, hipMemcpyDeviceToHost));
    float md = 0.f;
    for (size_t i = 0; i < n; ++i) {
        uint32_t a = (uint32_t)h_o1[i] << 16, b = (uint32_t)h_o2[i] << 16;
        float fa, fb;
        memcpy(&fa, &a, 4); memcpy(&fb, &b, 4);
        const float d = fabsf(fa - fb);
        if (!(d <= md)) md = d;  // NaN sticks
    }
    *max_diff = md; *ms_classic = ms[0] / iters; *ms_pc = ms[1] / iters;
#ifdef HH_STAMP
    {
        unsigned long long stv[72];
        HH_CHECK_HIP(hipMemcpy(stv, d_st, sizeof(stv), hipMemcpyDeviceToHost));
        for (int w = 0; w < 8; ++w) {
            fprintf(stderr, "  wave %d (100 MHz ticks from its iteration start):", w);
            for (int i = 1; i < 8; ++i) fprintf(stderr, " %lld", stv[w * 8 + i] ? (long long)(stv[w * 8 + i] - stv[w * 8]) : -1ll);
            fprintf(stderr, "   start %+lld vs wave 0\n", (long long)(stv[w * 8] - stv[0]));
        }
    }
#endif
    hipFree(d_st);
    hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(st);
    hipFree(d_in); hipFree(d_o1); hipFree(d_o2); hipFree(d_w); hipFree(d_b);
    return 0;
}

// Fused BasicBlock micro-benchmark; with a -DHH_STAMP build also returns s_memtime stamps of workgroup 0.
extern "C" int hh_debug_bb_bench(int B, int H, int W, int iters, float *ms_per_launch, unsigned long long *stamps64)
{
    HH_CHECK_HIP(conv_init());
    HH_CHECK_HIP(bb_fused_init());
    const size_t n = (size_t)B * H * W * 32;
    std::vector<bf16_raw> h_in(n), h_w(9 * 32 * 32);
    uint32_t s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (bf16_raw)(0x3c00u + ((s >> 16) & 0x1ffu) + ((s >> 30) << 15)); };
    for (auto &v : h_in) v = rnd();
    for (auto &v : h_w) v = (bf16_raw)(rnd() - 0x0400u);
    bf16_raw *d_in, *d_out, *d_w;
    float *d_b;
    unsigned long long *d_st;
    HH_CHECK_HIP(hipMalloc((void **)&d_in, n * 2));
    HH_CHECK_HIP(hipMalloc((void **)&d_out, n * 2));
    HH_CHECK_HIP(hipMalloc((void **)&d_w, h_w.size() * 2));
    HH_CHECK_HIP(hipMalloc((void **)&d_b, 32 * 4));
    HH_CHECK_HIP(hipMalloc((void **)&d_st, 72 * 8));
    HH_CHECK_HIP(hipMemcpy(d_in, h_in.data(), n * 2, hipMemcpyHostToDevice));
    HH_CHECK_HIP(hipMemcpy(d_w, h_w.data(), h_w.size() * 2, hipMemcpyHostToDevice));
    HH_CHECK_HIP(hipMemset(d_b, 0, 32 * 4));
    HH_CHECK_HIP(hipMemset(d_st, 0, 64 * 8));
    BBParams p{};
    p.in = d_in; p.in_cs = 32; p.out = d_out; p.out_cs = 32; p.w1 = d_w; p.w2 = d_w; p.b1 = d_b; p.b2 = d_b;
    p.B = B; p.H = H; p.W = W; p.stamps = d_st;
    int dev = 0;
    hipDeviceProp_t prop;
    HH_CHECK_HIP(hipGetDevice(&dev));
    HH_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    hipStream_t st;
    HH_CHECK_HIP(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    HH_CHECK_HIP(hipEventCreate(&e0));
    HH_CHECK_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) HH_CHECK_HIP(bb_fused_launch(p, prop.multiProcessorCount, st));
    HH_CHECK_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) HH_CHECK_HIP(bb_fused_launch(p, prop.multiProcessorCount, st));
    HH_CHECK_HIP(hipEventRecord(e1, st));
    HH_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0;
    HH_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / iters;
    if (stamps64) HH_CHECK_HIP(hipMemcpy(stamps64, d_st, 64 * 8, hipMemcpyDeviceToHost));
    hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(st);
    hipFree(d_in); hipFree(d_out); hipFree(d_w); hipFree(d_b); hipFree(d_st);
    return 0;
}
