// fp8 (OCP e4m3) execution of the HigherHRNet plan -- BASELINE.json configs[4].  The layer graph is the one engine.cpp builds
// (hrnet.py:342-385, higher_hrnet.py:47-81), run layer by layer on conv_fp8.hip; this file holds what is specific to the
// low-precision path: weight quantisation (one scale per output channel, BN folded first), the per-tensor activation scales
// and their calibration, and the launch glue.  No reference precedent exists (the reference infers in fp32,
// keypoints/model.py:79-83): parity is a stated tolerance against the fp32 oracle (tests/test_gpu_parity.py).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "engine.h"

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- e4m3fn (OCP): 1 sign, 4 exponent (bias 7), 3 mantissa bits; max finite 448 (0x7e), 0x7f = NaN, no infinities
unsigned char hh_f32_to_e4m3(float f)
{
    if (f != f) return 0x7f;
    const unsigned char sign = std::signbit(f) ? 0x80 : 0x00;
    const float a = std::fabs(f);
    if (a >= 464.f) return sign | 0x7e;  // beyond the midpoint between 448 and the missing 480: saturate
    if (a < 0.015625f) {                 // below 2^-6: subnormal grid of 2^-9 (the code IS the multiple; 8 = the first normal)
        return sign | (unsigned char)std::nearbyint(a * 512.f);
    }
    int e;
    const float m = std::frexp(a, &e);  // a = m * 2^e, m in [0.5, 1)
    int ex = e - 1;                     // a = (2m) * 2^ex, 2m in [1, 2)
    int q = (int)std::nearbyint((2.f * m - 1.f) * 8.f);  // round to nearest even
    if (q == 8) { q = 0; ++ex; }
    if (ex > 8 || (ex == 8 && q == 7)) return sign | 0x7e;
    return sign | (unsigned char)(((ex + 7) << 3) | q);
}
float hh_e4m3_to_f32(unsigned char v)
{
    const int ex = (v >> 3) & 15, m = v & 7;
    float a;
    if (ex == 15 && m == 7) return NAN;
    if (ex == 0) a = (float)m * (1.f / 512.f);
    else a = std::ldexp(1.f + (float)m / 8.f, ex - 7);
    return (v & 0x80) ? -a : a;
}

int hh_fp8_family_pick(int cin, int cout, int *KC, int *NT)
{
    const int c16 = round_up(cin, 16);
    *KC = c16 % 64 == 0 ? 64 : c16 % 48 == 0 ? 48 : c16 % 32 == 0 ? 32 : 16;
    *NT = round_up(cout, 32) % 64 == 0 ? 2 : 1;
    return 0;
}
int hh_fp8_pick_config(int ks, int stride, int KC, int NT, int Wo)
{
    int best = -1;
    for (int i = 0; i < conv_fp8_num_configs(); ++i) {
        const Fp8ConvConfig &c = conv_fp8_config(i);
        if (c.KS != ks || c.S != stride || c.KC != KC || c.NT != NT) continue;
        if (best < 0) best = i;
        if ((c.TW == 16) == (Wo <= 16)) return i;
    }
    return best;
}

// Kernel-order weight image of one conv: [cout_group][cin_chunk][k-step][lane half][piece 0/1][COUT_T][16 bytes], where
// piece pc = 4*step + 2*half + i is (tap, 16-channel group) = (pc / G, pc % G), G = KC / 16; pieces past the real ones are
// zero.  w8 = e4m3(W * bn_scale[co] / w_scale[co]).
static void pack_weights_fp8(const float *W, const float *bn_scale, const float *w_scale, int ks, int cin, int cout, int KC, int COUT_T,
                             bool transposed, int py, int px, std::vector<unsigned char> &packed)
{
    const int coutp = round_up(cout, COUT_T), ncg = coutp / COUT_T, cin_pad = round_up(cin, KC), nch = cin_pad / KC;
    const int G = KC / 16, npiece = ks * ks * G, nstep = (npiece + 3) / 4;
    packed.assign((size_t)ncg * nch * nstep * 4 * COUT_T * 16, 0);
    size_t o = 0;
    for (int cg = 0; cg < ncg; ++cg)
        for (int ch = 0; ch < nch; ++ch)
            for (int pc = 0; pc < nstep * 4; ++pc)
                for (int ci_o = 0; ci_o < COUT_T; ++ci_o)
                    for (int j = 0; j < 16; ++j, ++o) {
                        if (pc >= npiece) continue;
                        const int tap = pc / G, g = pc % G;
                        int ky = tap / ks, kx = tap % ks;
                        if (transposed) {  // patch row 0/1 of phase py <-> ky of the 4x4 stride-2 transposed conv
                            ky = py == 0 ? (ky == 0 ? 3 : 1) : (ky == 0 ? 2 : 0);
                            kx = px == 0 ? (kx == 0 ? 3 : 1) : (kx == 0 ? 2 : 0);
                        }
                        const int co = cg * COUT_T + ci_o, ci = ch * KC + g * 16 + j;
                        if (co >= cout || ci >= cin) continue;
                        const float v = transposed ? W[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx]
                                                   : W[(((size_t)co * cin + ci) * ks + ky) * ks + kx];
                        packed[o] = hh_f32_to_e4m3(v * bn_scale[co] / w_scale[co]);
                    }
}

int hh_net::finalize_fp8()
{
    HH_CHECK_HIP(conv_fp8_init());
    auto get = [&](const std::string &name) -> const std::vector<float> & { return params[param_index.at(name)].data; };
    for (auto &l : layers) {
        if (l.stem || l.hi) continue;  // packed by the common path (bf16 operands)
        hh_fp8_family_pick(l.cin, l.cout, &l.KC, &l.NT);
        if (hh_fp8_pick_config(l.ks, l.stride, l.KC, l.NT, 32) < 0) { hh_set_error("no fp8 kernel for conv " + l.conv); return 1; }
        l.cin_pad = round_up(l.cin, l.KC);
        const int COUT_T = 32 * l.NT, coutp = round_up(l.cout, COUT_T);
        l.ncg = coutp / COUT_T;
        const std::vector<float> &W = get(l.conv + ".weight");
        std::vector<float> scale(coutp, 0.f), shift(coutp, 0.f);
        for (int co = 0; co < l.cout; ++co) {
            if (!l.bn.empty()) {
                const float g = get(l.bn + ".weight")[co], bta = get(l.bn + ".bias")[co];
                const float mu = get(l.bn + ".running_mean")[co], var = get(l.bn + ".running_var")[co];
                const float sc = g / std::sqrt(var + 1e-5f);
                const float cb = l.bias.empty() ? 0.f : get(l.bias)[co];
                scale[co] = sc;
                shift[co] = bta + (cb - mu) * sc;
            } else {
                scale[co] = 1.f;
                shift[co] = l.bias.empty() ? 0.f : get(l.bias)[co];
            }
        }
        // one weight scale per output channel: the largest |w * bn_scale| of the channel maps to 448
        l.w_scale.assign(coutp, 0.f);
        const size_t per_co = (size_t)l.cin * l.ks * l.ks * (l.transposed ? 4 : 1);  // transposed: 4x4 kernel = 4 phases of 2x2
        for (int co = 0; co < l.cout; ++co) {
            float m = 0.f;
            if (l.transposed) {
                for (int ci = 0; ci < l.cin; ++ci)
                    for (int t = 0; t < 16; ++t) m = std::max(m, std::fabs(W[((size_t)ci * l.cout + co) * 16 + t] * scale[co]));
            } else {
                for (size_t i = 0; i < per_co; ++i) m = std::max(m, std::fabs(W[(size_t)co * per_co + i] * scale[co]));
            }
            l.w_scale[co] = m > 0.f ? m / 448.f : 1.f;
        }
        for (int co = l.cout; co < coutp; ++co) l.w_scale[co] = 1.f;
        std::vector<unsigned char> packed;
        if (l.transposed && l.py < 0) {
            for (int ph = 0; ph < 4; ++ph) {
                std::vector<unsigned char> one;
                pack_weights_fp8(W.data(), scale.data(), l.w_scale.data(), l.ks, l.cin, l.cout, l.KC, COUT_T, true, ph >> 1, ph & 1, one);
                l.phase_stride = one.size();
                packed.insert(packed.end(), one.begin(), one.end());
            }
        } else
            pack_weights_fp8(W.data(), scale.data(), l.w_scale.data(), l.ks, l.cin, l.cout, l.KC, COUT_T, l.transposed, l.py, l.px, packed);
        if (l.d_w) { hipFree(l.d_w); l.d_w = nullptr; }
        if (l.d_bias) { hipFree(l.d_bias); l.d_bias = nullptr; }
        if (l.d_mult) { hipFree(l.d_mult); l.d_mult = nullptr; }
        HH_CHECK_HIP(hipMalloc((void **)&l.d_w, packed.size()));
        HH_CHECK_HIP(hipMalloc((void **)&l.d_bias, (size_t)coutp * 4));
        HH_CHECK_HIP(hipMalloc((void **)&l.d_mult, (size_t)coutp * 4));
        HH_CHECK_HIP(hipMemcpy(l.d_w, packed.data(), packed.size(), hipMemcpyHostToDevice));
        HH_CHECK_HIP(hipMemcpy(l.d_bias, shift.data(), (size_t)coutp * 4, hipMemcpyHostToDevice));
    }
    if (!d_amax) HH_CHECK_HIP(hipMalloc((void **)&d_amax, 2 * ops.size() * 4));
    calibrated = false;  // new weights: the activation ranges may have moved
    amax.assign(2 * ops.size(), 0.f);
    for (auto &op : ops) { op.s_in = op.s_in2 = op.s_res = op.s_out = op.s_mid = 1.f; op.s_up[0] = op.s_up[1] = op.s_up[2] = 1.f; }
    return resolve_scales();
}

// Calibration maxima -> scales.  A tensor's scale maps the largest value seen to 240 (e4m3 tops out at 448: ~1.9x headroom
// for inputs beyond the calibration batch; the format is floating point, so the position inside the range costs no
// precision).  Ops are walked in plan order (the order the lanes' dependency edges enforce), every op reading the scale of
// the last writer of its inputs.  Tensors written in channel slices by several ops share one scale.
int hh_net::resolve_scales()
{
    const float target = 240.f;
    auto scale_of = [&](float m) { return m > 0.f ? m / target : 1.f; };
    std::vector<float> shared(tensors.size(), 0.f);
    for (size_t i = 0; i < ops.size(); ++i) {
        const Op &op = ops[i];
        if (op.out >= 0 && tensors[op.out].shared_scale && (op.kind == OP_CONV || op.kind == OP_UPADD)) shared[op.out] = std::max(shared[op.out], amax[i]);
    }
    std::vector<float> cur(tensors.size(), 1.f);
    // d_mult = s_in * w_scale lives with the LAYER while s_in belongs to the OP: sound only while no layer is launched by two ops
    std::vector<int> layer_ops(layers.size(), 0);
    for (const Op &op : ops)
        if ((op.kind == OP_CONV && !op.hi) || op.kind == OP_BB) {
            if (++layer_ops[op.layer] > 1 || (op.kind == OP_BB && ++layer_ops[op.layer2] > 1)) {
                hh_set_error("fp8 plan: a conv layer is launched by more than one op (per-layer d_mult would be ambiguous)");
                return 1;
            }
        }
    for (size_t i = 0; i < ops.size(); ++i) {
        Op &op = ops[i];
        switch (op.kind) {
        case OP_STEM:
            op.s_out = scale_of(amax[i]);
            cur[op.out] = op.s_out;
            break;
        case OP_CONV: {
            if (op.hi) break;  // bf16 kernels on bf16 representations: no scales
            op.s_in = cur[op.in];
            if (op.res >= 0) op.s_res = cur[op.res];
            if (op.out >= 0) {
                op.s_out = tensors[op.out].shared_scale ? scale_of(shared[op.out]) : scale_of(amax[i]);
                cur[op.out] = op.s_out;
            }
            ConvLayer &l = layers[op.layer];
            std::vector<float> mult(l.w_scale.size(), 0.f);
            for (int co = 0; co < l.cout; ++co) mult[co] = op.s_in * l.w_scale[co];
            HH_CHECK_HIP(hipMemcpy(l.d_mult, mult.data(), mult.size() * 4, hipMemcpyHostToDevice));
            break;
        }
        case OP_BB: {  // fused BasicBlock: conv1 -> intermediate (its own scale) -> conv2 + residual
            op.s_in = cur[op.in];
            op.s_mid = scale_of(amax[ops.size() + i]);
            op.s_out = scale_of(amax[i]);
            cur[op.out] = op.s_out;
            ConvLayer &l1 = layers[op.layer], &l2 = layers[op.layer2];
            std::vector<float> m1(l1.w_scale.size(), 0.f), m2(l2.w_scale.size(), 0.f);
            for (int co = 0; co < l1.cout; ++co) { m1[co] = op.s_in * l1.w_scale[co]; m2[co] = op.s_mid * l2.w_scale[co]; }
            HH_CHECK_HIP(hipMemcpy(l1.d_mult, m1.data(), m1.size() * 4, hipMemcpyHostToDevice));
            HH_CHECK_HIP(hipMemcpy(l2.d_mult, m2.data(), m2.size() * 4, hipMemcpyHostToDevice));
            break;
        }
        case OP_QUANT:
            op.s_out = scale_of(amax[i]);
            cur[op.out] = op.s_out;
            break;
        case OP_UPADD:
            op.s_in = cur[op.in];
            for (int j = 0; j < op.nup; ++j) op.s_up[j] = cur[op.up[j]];
            op.s_out = tensors[op.out].shared_scale ? scale_of(shared[op.out]) : scale_of(amax[i]);
            cur[op.out] = op.s_out;
            break;
        case OP_TAP:
            taps[op.tap].scale = cur[taps[op.tap].tensor];
            break;
        default:
            break;
        }
    }
    for (auto &g : graphs) hipGraphExecDestroy(g.exec);  // captured launches carry the old scales by value
    graphs.clear();
    return 0;
}

// hh_calibrate: `rounds` forwards over the calibration batch.  Round 1 runs with unit scales (activations of a BN'ed net
// are O(1): well inside e4m3's range) and records every op's output maximum from the fp32 epilogue values, before they
// are quantised; the following rounds repeat that under the scales of the previous one, so the recorded ranges are those of
// the quantised net.  Maxima accumulate over calls until the weights change (hh_finalize).
int hh_net::calibrate(const float *images, int B, int H, int W, int rounds, hipStream_t s)
{
    if (dtype != 2) { hh_set_error("hh_calibrate: not an fp8 handle"); return 1; }
    if (!finalized) { hh_set_error("hh_calibrate: call hh_finalize first"); return 1; }
    if (reserve(B, H, W)) return 1;
    struct Outs {  // the forward's fp32 outputs are not wanted here: scratch, freed on every way out
        float *o1 = nullptr, *o2 = nullptr;
        ~Outs() { if (o1) hipFree(o1); if (o2) hipFree(o2); }
    } outs;
    float *&o1 = outs.o1, *&o2 = outs.o2;
    HH_CHECK_HIP(hipMalloc((void **)&o1, (size_t)B * 2 * K * (H / 4) * (W / 4) * 4));
    HH_CHECK_HIP(hipMalloc((void **)&o2, (size_t)B * K * (H / 2) * (W / 2) * 4));
    int rc = 0;
    for (int r = 0; r < rounds && !rc; ++r) {
        HH_CHECK_HIP(hipMemsetAsync(d_amax, 0, 2 * ops.size() * 4, s));
        calibrating = true;
        lastB = B; lastH = H; lastW = W;
        rc = enqueue(images, B, H, W, o1, o2, s);
        calibrating = false;
        if (rc) break;
        HH_CHECK_HIP(hipStreamSynchronize(s));
        std::vector<unsigned> bits(2 * ops.size());
        HH_CHECK_HIP(hipMemcpy(bits.data(), d_amax, bits.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < bits.size(); ++i) {
            float f;
            memcpy(&f, &bits[i], 4);
            if (f == f && std::isfinite(f)) amax[i] = std::max(amax[i], f);
        }
        rc = resolve_scales();
    }
    if (!rc) calibrated = true;
    return rc;
}

int hh_net::enqueue_fp8_conv(const Op &op, int B, int H, int W, float *o1, float *o2, hipStream_t s, ProfRecord *pr)
{
    const ConvLayer &l = layers[op.layer];
    const TensorDesc &ti = tensors[op.in];
    Fp8ConvParams p{};
    p.in = (const unsigned char *)ti.ptr; p.in_cs = ti.C; p.in_coff = op.in_coff;
    p.Hin = H >> ti.shift; p.Win = W >> ti.shift;
    p.w = (const unsigned char *)l.d_w; p.mult = l.d_mult; p.bias = l.d_bias;
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
    if (op.out >= 0) {  // the representations the tensor's readers need (assign_fp8_formats)
        const TensorDesc &to = tensors[op.out];
        if (to.f8) { p.out = (unsigned char *)to.ptr; p.out_cs = to.C; p.out_coff = op.out_coff; p.out_inv_scale = 1.f / op.s_out; }
        if (to.b16) { p.out16 = to.ptr16; p.out16_cs = to.C; p.out16_coff = op.out_coff; }
    }
    if (op.res >= 0) {
        const TensorDesc &tr = tensors[op.res];
        if (tr.b16) { p.res16 = tr.ptr16; p.res16_cs = tr.C; p.res16_coff = op.res_coff; }
        else { p.res = (const unsigned char *)tr.ptr; p.res_cs = tr.C; p.res_coff = op.res_coff; p.res_scale = op.s_res; }
    }
    p.out_f32 = op.f32_out == 1 ? o1 : op.f32_out == 2 ? o2 : nullptr;
    p.cin = l.cin_pad;
    p.cout_real = l.cout;
    p.cout_store = op.cout_store >= 0 ? op.cout_store : round_up(l.cout, 16);
    p.relu = op.relu;
    p.B = B;
    const int cfg = hh_fp8_pick_config(l.ks, l.stride, l.KC, l.NT, p.Wo);
    const Fp8ConvConfig &c = conv_fp8_config(cfg);
    p.tiles_x = (p.Wo + c.TW - 1) / c.TW;
    p.tiles_y = (p.Ho + c.th() - 1) / c.th();
    p.ncg = l.ncg;
    if (calibrating && op.out >= 0) p.absmax = d_amax + (&op - ops.data());
    if (pr) {
        pr->cfg = 1000 + cfg;
        pr->flops = 2.0 * B * p.Ho * p.Wo * (double)l.cin * l.cout * l.ks * l.ks * (p.nphase > 1 ? 4 : 1);
        const double opix = (double)B * p.Ho * p.Wo * (p.nphase > 1 ? 4 : 1);
        pr->bytes = 1.0 * B * p.Hin * p.Win * l.cin + (p.out ? opix * l.cout : 0.0) + (p.out16 ? 2.0 * opix * l.cout : 0.0) +
                    (p.res ? opix * l.cout : 0.0) + (p.res16 ? 2.0 * opix * l.cout : 0.0) +
                    (p.out_f32 ? 4.0 * opix * l.cout : 0.0) + 1.0 * l.cin * l.cout * l.ks * l.ks * (p.nphase > 1 ? 4 : 1);
        hh_launch_probe() = LaunchProbe{pr->e0, pr->e1};
    }
    HH_CHECK_HIP(conv_fp8_launch(cfg, p, s));
    return 0;
}

int hh_net::enqueue_fp8_upadd(const Op &op, int B, int H, int W, hipStream_t s)
{
    UpAddFp8Params p{};
    const TensorDesc &b = tensors[op.in], &o = tensors[op.out];
    if (b.b16) { p.base16 = b.ptr16; p.base16_cs = b.C; }
    else { p.base = (const unsigned char *)b.ptr; p.base_cs = b.C; p.base_scale = op.s_in; }
    p.nup = op.nup;
    for (int j = 0; j < op.nup; ++j) {
        const TensorDesc &u = tensors[op.up[j]];
        p.up_shift[j] = op.up_shift[j];
        if (u.b16) { p.up16[j] = u.ptr16; p.up16_cs[j] = u.C; }
        else { p.up[j] = (const unsigned char *)u.ptr; p.up_cs[j] = u.C; p.up_scale[j] = op.s_up[j]; }
    }
    if (o.f8) { p.out = (unsigned char *)o.ptr; p.out_cs = o.C; p.out_inv_scale = 1.f / op.s_out; }
    if (o.b16) { p.out16 = o.ptr16; p.out16_cs = o.C; }
    p.B = B; p.H = H >> b.shift; p.W = W >> b.shift; p.C = round_up(op.C, 16); p.relu = op.relu;
    if (calibrating) p.absmax = d_amax + (&op - ops.data());
    HH_CHECK_HIP(launch_upadd_fp8(p, s));
    return 0;
}

int hh_net::enqueue_fp8_bb(const Op &op, int B, int H, int W, hipStream_t s, ProfRecord *pr)
{
    const ConvLayer &l1 = layers[op.layer], &l2 = layers[op.layer2];
    const TensorDesc &ti = tensors[op.in], &to = tensors[op.out];
    Fp8BBParams p{};
    p.in = (const unsigned char *)ti.ptr; p.in_cs = ti.C; p.out = (unsigned char *)to.ptr; p.out_cs = to.C;
    p.w1 = (const unsigned char *)l1.d_w; p.w2 = (const unsigned char *)l2.d_w;
    p.mult1 = l1.d_mult; p.bias1 = l1.d_bias; p.mult2 = l2.d_mult; p.bias2 = l2.d_bias;
    p.mid_inv_scale = 1.f / op.s_mid; p.res_scale = op.s_in; p.out_inv_scale = 1.f / op.s_out;
    if (ti.b16) { p.res16 = ti.ptr16; p.res16_cs = ti.C; }
    if (to.b16) { p.out16 = to.ptr16; p.out16_cs = to.C; }
    p.B = B; p.H = H >> ti.shift; p.W = W >> ti.shift;
    if (calibrating) { p.amax_out = d_amax + (&op - ops.data()); p.amax_mid = d_amax + ops.size() + (&op - ops.data()); }
    if (pr) {
        const double Cb = l1.cout;
        pr->cfg = HH_CFG_BB_FP8;
        pr->flops = 2.0 * 2.0 * B * p.H * p.W * Cb * Cb * 9.0;
        pr->bytes = (2.0 + (p.res16 ? 2.0 : 0.0) + (p.out16 ? 2.0 : 0.0)) * B * p.H * p.W * Cb + 2.0 * 9 * Cb * Cb;
        hh_launch_probe() = LaunchProbe{pr->e0, pr->e1};
    }
    HH_CHECK_HIP(bb_fp8_launch(l1.cout, p, num_cus, s));
    return 0;
}

// OP_QUANT: the e4m3 representation of a tensor from its bf16 one (a bf16-kernel op wrote it, an fp8 conv reads it)
int hh_net::enqueue_fp8_quant(const Op &op, int B, int H, int W, hipStream_t s)
{
    const TensorDesc &t = tensors[op.out];
    const size_t npix = (size_t)B * (H >> t.shift) * (W >> t.shift);
    HH_CHECK_HIP(launch_quant_fp8(t.ptr16, t.C, (unsigned char *)t.ptr, t.C, npix, t.C, 1.f / op.s_out,
                                  calibrating ? d_amax + (&op - ops.data()) : nullptr, s));
    return 0;
}
