// Internal launch interface between the host engine and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <hip/hip_ext.h>

typedef uint16_t bf16_raw;  // storage type of a bf16 element in HBM

// Per-launch timing probe (bench.py's roofline line).  When the engine arms it, the NEXT forward-path kernel launch of this
// thread goes through hipExtLaunchKernelGGL with a start / stop event pair: the runtime fills those from the dispatch packet's
// own begin / end timestamps, i.e. hipEventElapsedTime(start, stop) is the kernel duration that rocprofv3's kernel trace
// reports (no marker packets, no dispatch gap inside the bracket).
struct LaunchProbe { hipEvent_t start = nullptr, stop = nullptr; };
LaunchProbe &hh_launch_probe();  // thread-local, consumed (reset) by the launch that uses it
#define HH_LAUNCH(kernel, grid, block, lds, stream, ...)                                                            \
    do {                                                                                                            \
        LaunchProbe &lp_ = hh_launch_probe();                                                                       \
        if (lp_.start) {                                                                                            \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, lp_.start, lp_.stop, 0, __VA_ARGS__);           \
            lp_ = LaunchProbe{};                                                                                    \
        } else {                                                                                                    \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                      \
        }                                                                                                           \
    } while (0)

// One convolution launch (3x3 / 1x1 / 2x2-phase-of-deconv; stride 1 or 2) on NHWC bf16.
struct ConvParams {
    const bf16_raw *in;   // [B, Hin, Win, in_cs]
    int in_cs, in_coff;   // pixel stride / first channel (elements, multiples of 8)
    int Hin, Win;
    const bf16_raw *w;    // packed [cout_group][cin_chunk][tap][KC/8][COUT_T][8]
    const float *bias;    // [ncg*COUT_T] folded BN shift (or conv bias), zero padded
    const bf16_raw *res;  // optional residual, same pixel grid as `out`
    int res_cs, res_coff;
    bf16_raw *out;        // optional bf16 NHWC output [B, Hob, Wob, out_cs]
    int out_cs, out_coff;
    float *out_f32;       // optional fp32 NCHW output [B, cout_real, Hob, Wob]
    int Hob, Wob;         // output buffer spatial dims
    int osy, ooy, osx, oox;  // output scatter: Y = oy*osy + ooy (deconv phases use 2, phase)
    int Ho, Wo;           // conv output grid (before scatter)
    int cin;              // padded input channels (multiple of KC)
    int cout_real, cout_store;
    int relu;
    int pad_y, pad_x;     // top/left zero padding
    int B, tiles_x, tiles_y, ncg;
    int nphase;           // 4: the four 2x2 phases of a 4x4 stride-2 transposed conv in ONE launch (pad / output offset follow the
    size_t phase_stride;  //    phase index, weights of phase f start at w + f * phase_stride elements); 0 or 1: plain conv
    unsigned long long *stamps;  // diagnostic build (-DHH_STAMP) only
    unsigned long long *clk;     // optional {min start, max end} of the launch in wall_clock64() ticks (profiling probe)
    // Several input tensors as ONE conv over their concatenated channels (the summed stride-2 convs of a fusion layer,
    // hrnet.py:166-229): same spatial dims and the same pixel stride as `in`; chunks [0, nch0) come from `in`, the next nch1 from
    // in + src_delta1 elements, the rest from in + src_delta2.  nch0 = 0: a single input.
    int nch0, nch1;
    ptrdiff_t src_delta1, src_delta2;
};

// Tile configuration of one kernel instantiation.
struct ConvConfig {
    int KS, S, KC, NT, WC, PT, TW;
    int DB;  // 1: the K loop runs on two LDS buffers (conv_mfma.hip), each followed by 16 bytes per thread for staging units without a destination
    int cout_t() const { return 32 * NT * WC; }
    int th() const { return (4 / WC) * PT * (32 / TW); }
    size_t lds_bytes() const {
        int PH = (th() - 1) * S + KS, PW = (TW - 1) * S + KS;
        size_t patch = ((size_t)PH * PW * (KC * 2 + 16) + 15) & ~(size_t)15;
        const size_t buf = patch + (size_t)KS * KS * KC * cout_t() * 2;
        return DB ? 2 * (buf + 256 * 16) : buf;
    }
};

// Returns the number of instantiated configs / the i-th one.
int conv_num_configs();
const ConvConfig &conv_config(int i);
// Launches config `cfg_index`; the grid is B*tiles_y*tiles_x*ncg blocks of 256 threads.
hipError_t conv_launch(int cfg_index, const ConvParams &p, hipStream_t stream);
hipError_t conv_init();  // raises the dynamic-LDS limit of every instantiation

// ---- fp8 path (conv_fp8.hip): e4m3 NHWC activations with one scale per tensor, e4m3 weights with one scale per cout
struct Fp8ConvParams {
    const unsigned char *in;  // [B, Hin, Win, in_cs] e4m3
    int in_cs, in_coff;       // pixel stride / first channel (bytes, multiples of 16)
    int Hin, Win;
    const unsigned char *w;   // packed [cout_group][cin_chunk][k-step][lane half][piece 0/1][COUT_T][16]
    const float *mult;        // [ncg*COUT_T] s_in * s_w[co] (0 for padding couts)
    const float *bias;        // [ncg*COUT_T] folded BN shift / conv bias
    const unsigned char *res; // optional residual (e4m3), same pixel grid as `out`
    int res_cs, res_coff;
    float res_scale;
    const bf16_raw *res16;    // optional residual in bf16 (real values, no scale): the residual TRUNK of an fp8 net stays bf16
    int res16_cs, res16_coff; //   (elements); takes precedence over `res`
    unsigned char *out;       // optional e4m3 NHWC output
    int out_cs, out_coff;
    float out_inv_scale;      // 1 / s_out
    bf16_raw *out16;          // optional bf16 NHWC output of the same values (the trunk's second representation: read back as a
    int out16_cs, out16_coff; //   residual / by the fusion sums; the e4m3 one feeds the next conv's MFMA); elements
    float *out_f32;           // optional fp32 NCHW output [B, cout_real, Hob, Wob]
    int Hob, Wob;
    int osy, ooy, osx, oox;
    int Ho, Wo;
    int cin;                  // padded input channels (multiple of KC)
    int cout_real, cout_store;
    int relu;
    int pad_y, pad_x;
    int B, tiles_x, tiles_y, ncg;
    int nphase;
    size_t phase_stride;      // bytes between the weight sets of two transposed-conv phases
    unsigned *absmax;         // calibration: atomicMax of the bits of max |y| over the launch (NULL = off)
};
struct Fp8ConvConfig {
    int KS, S, KC, NT, PT, TW;
    // bytes per staged pixel: KC + padding to an ODD number of 16-byte slots (bank-conflict-free ds_read_b128 across pixels)
    static constexpr int pixel_stride(int kc) { return ((kc / 16 + 1) | 1) * 16; }
    int cout_t() const { return 32 * NT; }
    int th() const { return 4 * PT * (32 / TW); }
    int nstep() const { return (KS * KS * (KC / 16) + 3) / 4; }
    size_t lds_bytes() const {
        const int PH = (th() - 1) * S + KS, PW = (TW - 1) * S + KS;
        const size_t patch = ((size_t)PH * PW * pixel_stride(KC) + 15) & ~(size_t)15;
        return patch + (size_t)nstep() * 4 * cout_t() * 16;
    }
};
int conv_fp8_num_configs();
const Fp8ConvConfig &conv_fp8_config(int i);
hipError_t conv_fp8_launch(int cfg_index, const Fp8ConvParams &p, hipStream_t stream);
hipError_t conv_fp8_init();

// fused BasicBlock on e4m3 tensors (basicblock_fused_fp8.hip): weights packed as conv_fp8 packs a 3x3 conv with KC = C, NT = 2
struct Fp8BBParams {
    const unsigned char *in; int in_cs;   // [B,H,W,in_cs bytes], channels 0..C-1
    unsigned char *out; int out_cs;       // must not alias `in`
    const unsigned char *w1, *w2;
    const float *mult1, *bias1, *mult2, *bias2;  // [64]: s_in * s_w1[co], shift1, s_mid * s_w2[co], shift2 (0 for padding couts)
    float mid_inv_scale, res_scale, out_inv_scale;
    const bf16_raw *res16; int res16_cs;  // optional: the block input in bf16 (same pixels as `in`): residual read from it instead of
                                          // from the e4m3 patch
    bf16_raw *out16; int out16_cs;        // optional: the block output in bf16 beside the e4m3 one
    int B, H, W;
    int tiles_x, tiles_y, ntiles;         // filled by the launcher
    unsigned *amax_mid, *amax_out;        // calibration (NULL = off): atomicMax of the bits of the tensors' maxima
};
bool bb_fp8_supported(int C);
hipError_t bb_fp8_launch(int C, const Fp8BBParams &p, int num_cus, hipStream_t s);
#define HH_CFG_BB_FP8 105

// Fused 32-channel BasicBlock (basicblock_fused.hip): out = relu(conv2(relu(conv1(in))) + in), BN folded.
struct BBParams {
    const bf16_raw *in; int in_cs;   // [B,H,W,in_cs], channels 0..31
    bf16_raw *out; int out_cs;       // must not alias `in` (tiles read a 2-pixel halo of their neighbours)
    const bf16_raw *w1, *w2;         // packed as conv_mfma family (KS=3,S=1,KC=32,NT=1): [tap][4][32][8]
    const float *b1, *b2;            // [32]
    int B, H, W;
    int tiles_x, tiles_y, ntiles;    // filled by bb_fused_launch
    int tall;                        // bbpc_launch: 1 = lay the batch out as one tall image when that needs fewer tiles (2: always), 0 = never
    int VH;                          // filled by bbpc_launch: rows per image in the tall layout (H + 2), or 2^30
    bf16_raw *trash;                 // filled by bb_fused_launch: dummy line for the stores of lanes outside the image
    // bbpc_launch only: the 1x1 head behind the block, run in its epilogue (fin_out != nullptr; `out` is then not written)
    const bf16_raw *fin_w;           // [k half 2][lane half 2][32 couts][8 cin] bf16, couts >= fin_K zero
    const float *fin_b;              // [32]
    float *fin_out;                  // [B, fin_K, H, W] fp32
    int fin_K;
    unsigned long long *stamps;      // diagnostic build (-DHH_STAMP) only
    unsigned long long *clk;         // optional {min start, max end} of the launch in wall_clock64() ticks
};
#define HH_PROF_SLOTS 1024   // launches per forward the device-clock probe can record
#define HH_CFG_BB_FUSED 100  // pseudo instantiation index used by the profiler
hipError_t bb_fused_init();
hipError_t bb_fused_launch(BBParams p, int num_cus, hipStream_t s);
// producer / consumer form of the same block (basicblock_fused_pc.hip): weights resident in registers, row-band fragment reuse
hipError_t bbpc_init();
bool bbpc_supported(const BBParams &p);
bool bbpc_final_supported(const BBParams &p);
hipError_t bbpc_launch(BBParams p, int num_cus, hipStream_t s);
// the same block for the 64-channel branch (basicblock_fused_c64.hip): weights packed KS=3,S=1,KC=32,NT=2 ([chunk][tap][4][64][8])
#define HH_CFG_BB64_FUSED 103
hipError_t bb64_fused_init();
hipError_t bb64_fused_launch(BBParams p, int num_cus, hipStream_t s);

// Junction of two stage-0 Bottlenecks (bottleneck_junction.hip): y = relu(W3 t2 + shift (+ Wd x | + res)), t1 = relu(W1 y + shift1)
struct JuncParams {
    const bf16_raw *t2; int t2_cs;     // [npix, 64]  conv2 output of this unit
    const bf16_raw *res; int res_cs;   // [npix, 256] previous y (units 1-3), or nullptr
    const bf16_raw *x; int x_cs;       // [npix, 64]  unit-0 input of the downsample conv, or nullptr
    const bf16_raw *t2a; int t2a_cs;   // pair mode: conv2 output of the PREVIOUS unit, whose y is made again here instead of read (else nullptr)
    const bf16_raw *w3a; const float *b3a;  // pair mode: the previous unit's conv3 (its downsample, if it is unit 0, is wd / bd with x)
    const bf16_raw *w3, *wd, *w1;      // packed as the conv_mfma family packs 1x1 KC=32 NT=2 layers; wd / w1 may be nullptr
    const float *b3, *bd, *b1;
    bf16_raw *y; int y_cs;             // [npix, 256]; nullptr: not stored (the next junction makes it again, pair mode)
    bf16_raw *t1; int t1_cs;           // [npix, 64] (only when w1)
    int npix;
    unsigned long long *clk;           // optional {min start, max end} device-clock probe
};
#define HH_CFG_JUNCTION 101  // pseudo instantiation index used by the profiler
hipError_t junction_init();
hipError_t junction_launch(const JuncParams &p, int num_cus, hipStream_t s);

// First stem conv (stem_conv.hip): fp32 NCHW images -> conv3x3 s2 p1 (3->64) + BN + ReLU -> bf16 NHWC [B,H/2,W/2,out_cs]
struct StemParams {
    const float *images;      // [B,3,H,W]
    const bf16_raw *w;        // [cout tile 2][k-step 2][half 2][32 couts][8 taps] bf16, tap = c*9 + ky*3 + kx, BN scale folded, taps 27..31 zero
    const float *bias;        // [64]
    bf16_raw *out; int out_cs;
    int B, H, W;              // H, W even
    unsigned long long *clk;  // optional device-clock probe
    unsigned char *out_fp8;   // fp8 path: e4m3 NHWC output [B,H/2,W/2,out_cs bytes] instead of `out`, q = e4m3(y * out_inv_scale)
    float out_inv_scale;
    unsigned *absmax;         // fp8 calibration: atomicMax of the bits of max |y|
};
#define HH_CFG_STEM 102  // pseudo instantiation index used by the profiler
hipError_t stem_conv_launch(const StemParams &p, hipStream_t s);
// Both stem convolutions in one kernel (stem_fused.hip): conv3x3 s2 3->64 + BN + ReLU -> conv3x3 s2 64->64 + BN + ReLU.
struct StemFusedParams {
    const float *images;      // [B,3,H,W], H and W multiples of 4
    const bf16_raw *w1;       // conv1 as StemParams::w
    const float *b1;          // [64]
    const bf16_raw *w2;       // conv2 packed [tap 9][cin/8 8][64 couts][8] (hh_pack_weights with KC = 64, COUT_T = 64), BN scale folded
    const float *b2;          // [64]
    bf16_raw *out; int out_cs;  // [B,H/4,W/4,out_cs], channels 0..63
    int B, H, W;
    unsigned long long *clk;  // optional device-clock probe
};
#define HH_CFG_STEM_FUSED 106
hipError_t stem_fused_init();
bool stem_fused_supported(const StemFusedParams &p);
hipError_t stem_fused_launch(const StemFusedParams &p, int num_cus, hipStream_t s);


// out[b,y,x,c] = act(base[b,y,x,c] + sum_j up_j[b, y>>sh_j, x>>sh_j, c]),  c in [0,C)
struct UpAddParams {
    const bf16_raw *base; int base_cs, base_coff;
    const bf16_raw *up[3]; int up_cs[3]; int up_shift[3]; int nup;
    bf16_raw *out; int out_cs, out_coff;
    int B, H, W, C, relu;
};
hipError_t launch_upadd(const UpAddParams &p, hipStream_t s);
hipError_t launch_upadd_backward(const bf16_raw *dy, const bf16_raw *out, int relu, int B, int H, int W, int C, bf16_raw *g, bf16_raw *const *dup,
                                 const int *up_shift, int nup, hipStream_t s);
// the same on e4m3 tensors: out = e4m3(act(base * base_scale + sum_j up_j * up_scale[j]) * out_inv_scale); C multiple of 16
struct UpAddFp8Params {
    const unsigned char *base; int base_cs; float base_scale;
    const unsigned char *up[3]; int up_cs[3]; int up_shift[3]; float up_scale[3]; int nup;
    // an operand given in bf16 (real values) instead of e4m3 * scale: base16 / up16[j] non-NULL take precedence
    const bf16_raw *base16; int base16_cs;
    const bf16_raw *up16[3]; int up16_cs[3];
    unsigned char *out; int out_cs;       // optional e4m3 output
    float out_inv_scale;
    bf16_raw *out16; int out16_cs;        // optional bf16 output
    int B, H, W, C, relu;
    unsigned *absmax;
};
hipError_t launch_upadd_fp8(const UpAddFp8Params &p, hipStream_t s);
// bf16 tensor -> its e4m3 representation (q = e4m3(x * inv_scale)); n16 = groups of 16 channels per pixel; absmax: calibration
hipError_t launch_quant_fp8(const bf16_raw *in, int in_cs, unsigned char *out, int out_cs, size_t npix, int C, float inv_scale, unsigned *absmax,
                            hipStream_t s);

// ClassificationHead tail: global average pool (bf16 NHWC -> fp32 [B,C]) and Linear (fp32)
hipError_t launch_lds_poison(int num_cus, hipStream_t s);  // debug: NaN patterns into every CU's LDS (misc_kernels.hip)
hipError_t launch_avgpool(const bf16_raw *in, int in_cs, float *out, int B, int HW, int C, hipStream_t s);
hipError_t launch_linear(const float *x, const float *w, const float *bias, float *y, int B, int K, int N, hipStream_t s);

struct HHImageDesc {  // one raw image of a batch (hh_image_desc of include/hhrnet.h): 64 bytes
    long long offset;  // bytes from the batch's base pointer to the image's first pixel
    int h, w;
    double inv[6];     // destination -> source affine
};
hipError_t launch_preprocess_batch(const unsigned char *base, const HHImageDesc *descs, int n, float *out, int H, int W,
                                   const float mean[3], const float stdv[3], hipStream_t s);
hipError_t launch_warp_affine_u8(const unsigned char *img, int h, int w, const double inv[6], unsigned char *out, int H, int W, hipStream_t s);
hipError_t launch_preprocess(const unsigned char *img, int h, int w, const double inv[6], float *out, int H, int W,
                             const float mean[3], const float stdv[3], hipStream_t s);
hipError_t launch_flip_images(const float *in, float *out, int B, int C, int H, int W, hipStream_t s);
struct FlipPerm { unsigned char v[64]; };
hipError_t launch_flip_merge(float *hm, int64_t hm_bs, const float *hmf, int64_t hmf_bs, const float *tf, int64_t tf_bs,
                             float *to, int64_t to_bs, const int32_t *perm_host, int B, int K, int h, int w, hipStream_t s);

// Training loss (loss_kernels.hip).  scratch: >= max(HH_LOSS_SCRATCH, 2*B) doubles of device memory.
#define HH_LOSS_SCRATCH 1024
hipError_t launch_masked_mse(const float *pred, int64_t pred_bs, const float *target, const float *mask, int B, int K, int h, int w,
                             float *loss, float *grad, int64_t grad_bs, double *scratch, hipStream_t s);
hipError_t launch_ae_grouping(const float *tags, int64_t tags_bs, const int32_t *joints, const int32_t *num_people, int B, int P, int K,
                              int h, int w, float *push_pull, float *grad, int64_t grad_bs, float push_scale, float pull_scale,
                              double *scratch, hipStream_t s);

// Training building blocks (train_ops.hip)
#define HH_BN_BLOCKS 256  // partial-sum blocks of the BatchNorm reductions; scratch = HH_BN_BLOCKS * C * 2 doubles
// one weight set of a batched packing launch (launch_pack_weights_batch)
struct PackDesc {
    const float *W;
    bf16_raw *packed;
    long long total;
    int cout, cin, ks, mode, KC, COUT_T, py, px;
};
hipError_t launch_pack_weights_batch(const PackDesc *descs_dev, int n, hipStream_t s);
hipError_t launch_pack_weights(const float *W, int cout, int cin, int ks, int mode, int KC, int COUT_T, bf16_raw *packed, size_t total,
                               hipStream_t s, int py = 0, int px = 0);
hipError_t launch_bn_train_forward(const bf16_raw *x, int cs, size_t P, int C, const float *gamma, const float *beta, float eps,
                                   const bf16_raw *res, int relu, bf16_raw *y, float *mean, float *invstd, double *scratch, hipStream_t s);
hipError_t launch_bn_train_backward(const bf16_raw *x, const bf16_raw *y, const bf16_raw *dy, int cs, size_t P, int C, const float *mean,
                                    const float *invstd, const float *gamma, const float *beta, int relu, bf16_raw *dx, bf16_raw *dres,
                                    float *dgamma, float *dbeta, double *scratch, hipStream_t s);  // y == nullptr: no residual, mask from x
hipError_t launch_bn_train_stats(const bf16_raw *x, int cs, size_t P, int C, double *sums, double *scratch, hipStream_t s);
hipError_t launch_bn_train_normalize(const bf16_raw *x, int cs, size_t P, int C, const double *sums, double count, const float *gamma,
                                     const float *beta, float eps, const bf16_raw *res, int relu, bf16_raw *y, float *mean, float *invstd,
                                     hipStream_t s);
hipError_t launch_bn_train_backward_stats(const bf16_raw *x, const bf16_raw *y, const bf16_raw *dy, int cs, size_t P, int C, const float *mean,
                                          const float *invstd, int relu, double *sums, float *dgamma, float *dbeta, double *scratch,
                                          hipStream_t s);
hipError_t launch_bn_train_backward_apply(const bf16_raw *x, const bf16_raw *y, const bf16_raw *dy, int cs, size_t P, int C, const float *mean,
                                          const float *invstd, const float *gamma, int relu, const double *sums, double count, bf16_raw *dx,
                                          bf16_raw *dres, double *scratch, hipStream_t s);

// Convolution weight gradient (conv_wgrad.hip): ks in {1,3} x stride 1, and 3x3 stride 2; channels multiples of 8
struct WgradParams {
    const bf16_raw *x;   // [B,H,W,cin]
    const bf16_raw *dy;  // [B,Ho,Wo,cout]
    float *partial;      // workspace: workers * ks*ks * roundup(cout,64) * roundup(cin,64) floats
    int B, H, W, Ho, Wo, cin, cout;
    int pad_y, pad_x;    // top / left zero padding of x
};
#define HH_WGRAD_WORKERS 128  // base count of persistent pixel-tile workers per channel block (64 .. 512, see conv_wgrad_num_workers)
int conv_wgrad_num_workers(int B, int Ho, int Wo, int stride, int cin, int cout);
hipError_t conv_wgrad_launch(const WgradParams &p, int ks, int stride, float *dw, hipStream_t s);
