// Host-side engine: builds the HigherHRNet layer plan from (num_kpts, C), owns the
// parameters under the reference's state-dict names, folds BN, packs bf16 weights for the
// MFMA kernels and executes the plan on a HIP stream.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

void hh_set_error(const std::string &msg);
void hh_pack_weights(const float *W, const float *scale, int ks, int cin, int cout, int KC, int COUT_T, bool transposed,
                     int py, int px, std::vector<unsigned short> &packed);
#define HH_CHECK_HIP(expr)                                                                             \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            hh_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                           \
            return 1;                                                                                  \
        }                                                                                              \
    } while (0)

struct ParamSlot {
    std::string name;
    std::vector<int64_t> shape;
    std::vector<float> data;
    bool loaded = false;
    bool counter = false;  // num_batches_tracked: accepted, unused
};

struct ConvLayer {
    std::string conv, bn, bias;  // state-dict prefixes ("" = absent)
    int cin = 0, cout = 0, ks = 1, stride = 1;
    bool transposed = false;
    // conv over the concatenated channels of several inputs = the SUM of several convs + BNs (the stride-2 convs a fusion layer
    // adds up): their state-dict prefixes and input widths; `conv` / `bn` / `cin` then describe the first one / the total
    std::vector<std::string> mconv, mbn;
    std::vector<int> mcin;
    bool stem2 = false; // second stem conv inside stem_fused.hip: packed [tap][cin/8][64 couts][8]
    bool stem = false;  // first conv of the net: packed for stem_conv.hip (K = 27 taps padded to 32)
    bool hi = false;    // fp8 handle: this layer runs on the bf16 kernels (the heads), packed like a bf16 handle's
    // The transposed conv with init_heatmaps_head folded in (bf16 handles): torch.cat((feats, head(feats))) followed by a linear
    // op is linear in feats, so the layer reads [feats | 1] with W_eff = Wd_feats + Wd_hm * Wf and the head's bias on the
    // constant-one channel (which is 0 outside the image like every other channel, so the borders stay exact).  fold_w / fold_b:
    // the head's state-dict keys; acct_cin: the input width the FLOP accounting keeps (the reference's C + 2K)
    std::string fold_w, fold_b;
    int acct_cin = 0;
    int py = 0, px = 0;  // phase of the transposed conv this entry implements (-1: all four, packed back to back)
    size_t phase_stride = 0;
    // chosen at finalize
    int KC = 0, NT = 0, cin_pad = 0, ncg = 0;
    int db = 0;  // 1: the double-buffered instantiation (3x3 stride 1, >= 128 input channels; PlanSwitches::no_conv_db)
    bf16_raw *d_w = nullptr;
    float *d_bias = nullptr;
    bool fin_head = false;      // the deconv head's final 1x1: also packed as the B fragments of bbpc_final_kernel ([k half][lane half][32 couts][8])
    bf16_raw *d_wfin = nullptr;
    // fp8 path: e4m3 weights with one scale per output channel; d_mult[co] = s_in * w_scale[co] is rewritten whenever the
    // activation scales change (calibration)
    std::vector<float> w_scale;
    float *d_mult = nullptr;
};

struct TensorDesc {
    int C = 0, shift = 0;  // spatial dims = (H >> shift, W >> shift)
    bf16_raw *ptr = nullptr;  // bf16 elements, or e4m3 bytes on the fp8 path (C bytes per pixel)
    // fp8 handles keep up to TWO representations of a tensor: e4m3 * scale in `ptr` (f8: some conv reads it as an MFMA operand) and
    // bf16 in `ptr16` (b16: it is read back as a residual / by a fusion sum / by a head that runs on the bf16 kernels) -- the
    // residual trunk is never re-quantised to 3 mantissa bits (engine.cpp assign_fp8_formats)
    bool f8 = true, b16 = false;
    bf16_raw *ptr16 = nullptr;
    bool zero_init = false;
    int ones_channel = -1;  // >= 0: this channel holds 1.0 in every pixel (written at reserve, never by a kernel)
    bool shared_scale = false;  // fp8: written in channel slices by several ops (torch.cat buffer): one scale for all of them
};

enum OpKind { OP_CONV, OP_UPADD, OP_TAP, OP_BB, OP_JOIN, OP_AVGPOOL, OP_LINEAR, OP_DEP, OP_JUNC, OP_STEM, OP_MARK, OP_WAITL, OP_QUANT };

struct Op {
    OpKind kind = OP_CONV;
    int layer = -1, layer2 = -1;  // layer2: second conv of a fused BasicBlock / downsample conv of a junction (-1: none)
    int layer3 = -1;              // OP_JUNC: first conv of the next Bottleneck (-1: none)
    int layer4 = -1;              // OP_JUNC, pair mode: conv3 of the PREVIOUS Bottleneck, whose output is made again from in3 (= its t2) instead of read
    int in2 = -1, out2 = -1;      // OP_JUNC: downsample input x, and the t1 output
    int in3 = -1;                 // OP_CONV over concatenated inputs: in, in2, in3 (ConvLayer::mcin channels each)
    int in = -1, in_coff = 0;
    int out = -1, out_coff = 0;
    int res = -1, res_coff = 0;
    int relu = 0;
    int f32_out = 0;  // 0 none, 1 = init_heatmaps, 2 = deconv_heatmaps
    int cout_store = -1;
    int scatter = 0;  // 1: write to (2*oy+py, 2*ox+px)
    int up[3] = {-1, -1, -1}, up_shift[3] = {0, 0, 0}, nup = 0;
    int C = 0;  // UPADD channel count / TAP channel count
    int tap = -1;
    bool res8 = false;  // fp8 handle: this conv's residual stays e4m3 (stage 0's 256-channel trunk: its 1x1 convs are HBM-bound and a bf16
                        // twin would triple their traffic; emulation: +0.3-0.5 % rms at the outputs)
    int fin = -1;  // OP_BB: index of the 1x1 head conv right behind this block that bbpc_final_kernel runs in the block's epilogue (-1: none)
    bool siblings = false;  // OP_BB: a block of an HR module with other branches beside it (its persistent grid takes half the CUs)
    bool hi = false;  // fp8 handle: OP_CONV on the bf16 kernels over the tensors' bf16 representations; OP_QUANT: tensor `out`'s bf16 -> e4m3
    int lane = 0;     // execution lane (HIP stream): resolution branches / fusion outputs run concurrently
    int nlanes = 0;   // OP_JOIN: all-to-all barrier over lanes [0, nlanes)
    int dep_from = 0; // OP_DEP: `lane` waits for everything enqueued so far on lane dep_from; OP_WAITL: `lane` waits for lane dep_from's
                      // position at the last OP_MARK (OP_MARK: every lane [0, nlanes) records its position)
    // fp8 path: tensor scales seen by this op (real value = e4m3 * scale), resolved from the calibration maxima
    float s_in = 1.f, s_in2 = 1.f, s_res = 1.f, s_out = 1.f, s_up[3] = {1.f, 1.f, 1.f};
    float s_mid = 1.f;  // fused fp8 BasicBlock: scale of the intermediate tile
    int amax_slot = -1;  // index into hh_net::d_amax of this op's output maximum
};

struct TapInfo {
    std::string name;
    int tensor, coff, C;
    bf16_raw *copy = nullptr;
    float scale = 1.f;  // fp8: scale of the tensor at the tap's position in the plan
    bool is16 = false;  // fp8: the copy holds the tensor's bf16 representation
};

struct ProfRecord {
    int op, cfg;
    double flops, bytes;  // algorithmic: 2*MACs; input + output (+ residual) + weights once, no halo re-reads
    hipEvent_t e0, e1;
    int slot;  // index into d_clk ({min start, max end} device-clock ticks written by the kernel itself), -1 = none
};

struct GraphEntry {
    const void *images;
    void *o1, *o2;
    int B, H, W;
    hipGraphExec_t exec;
};

// Plan variants behind HH_* environment switches (A/B measurements, bit-equality tests).  hh_create / hh_create_classifier read
// them ONCE (PlanSwitches::from_env) into the handle: nothing in the engine calls getenv afterwards, so a handle's plan and
// launch choices cannot change under it when a later handle is built with other switches in the same process.
struct PlanSwitches {
    bool bb32_tile = false;        // HH_BB32=tile: basicblock_fused.hip instead of the producer / consumer form
    bool no_bb64 = false;          // HH_NO_BB64=1: 64-channel BasicBlocks layer by layer
    bool no_bb_fp8 = false;        // HH_NO_BB_FP8=1: fp8 BasicBlocks layer by layer
    bool no_stem_fused = false;    // HH_NO_STEM_FUSED=1: the stem as two launches
    bool no_junc_pair = false;     // HH_NO_JUNC_PAIR=1: every stage-0 junction stores its 256-channel output
    bool full_join = false;        // HH_FULL_JOIN=1: all-to-all joins of the branch lanes instead of per-source waits
    bool no_fusion_merge = false;  // HH_NO_FUSION_MERGE=1: one launch per summed stride-2 conv of a fusion layer
    bool poison_ws = false;        // HH_POISON_WS=1 (tests): workspace filled with NaN patterns at allocation
    // Persistent workgroups of the fused 32- / 64-channel blocks INSIDE an HR module (beside the other branches' lanes).  Default 0 =
    // half the CUs each: the two fat kernels (150 KB of LDS per workgroup: a CU holds one of them and nothing else) then run side by
    // side on disjoint halves of the chip instead of taking turns on all of it, and the thin launches of the 128- / 256-channel
    // lanes find free CUs while either runs: forward 4.48 -> 4.39 ms, +1.5-2 % img/s (three alternations, profiles/r03_ab.md).
    // HH_FAT_CUS=n[,m] sets them (256 = one per CU, the round-2 plan); a quarter of the chip for the 64-channel block loses 6 %.
    int fat_cus = 0, fat_cus64 = 0;
    bool event_system_fence = false;  // HH_EVENT_SYSTEM_FENCE=1: the lane events with the default system-scope fence at record time
    bool keep_waits = false;       // HH_KEEP_WAITS=1: enqueue() issues every wait of the plan, also those it can prove redundant (A/B)
    bool no_final_fuse = false;    // HH_NO_FINAL_FUSE=1: the deconv head's final 1x1 as its own launch (round 4: it runs in the last block's epilogue)
    int bb_tall = 1;               // the fused 32-channel block tiles the batch as one tall image when that needs fewer tiles (round 4);
                                   // HH_NO_BB_TALL=1: per-image tiles (round 3), HH_BB_TALL=always: also where it needs more
    bool no_conv_db = false;       // HH_NO_CONV_DB=1: the 128- / 256-channel 3x3 convs on the single-buffer KC = 32 instantiations (round 2)
    bool no_head_fold = false;     // HH_NO_HEAD_FOLD=1: init_heatmaps_head writes its output into the concat buffer, the transposed conv reads it
    bool poison_lds = false;       // HH_POISON_LDS=1 (tests): every CU's LDS filled with NaN patterns in front of every launch
    unsigned debug_skip = 0;       // HH_DEBUG_SKIP=cat[,cat..] (measurement only, results are WRONG): launches of these categories are not
                                   // issued -- how much of the forward's wall time hangs on a kernel family (tools/probes/skip_sensitivity.sh)
    bool fp8_trunk8 = false;       // HH_FP8_TRUNK=e4m3: fp8 handles re-quantise the residual trunk to e4m3 in every block (the round-2 plan)
    bool fp8_heads8 = false;       // HH_FP8_HEADS=e4m3: the two 1x1 heads and the transposed conv of an fp8 handle on the e4m3 kernels too
    static PlanSwitches from_env();
};

enum SkipCat { SK_S2BIG = 1, SK_S2 = 2, SK_UPADD = 4, SK_C1X1 = 8, SK_C256 = 16, SK_C128 = 32, SK_JUNC = 64, SK_BB32 = 128, SK_BB64 = 256,
               SK_STEM = 512, SK_DECONV = 1024, SK_HEAD = 2048, SK_TRANS0 = 4096 };

struct hh_net {
    int K, C, dtype;
    int kind = 0;          // 0 = HigherHRNet, 1 = ClassificationHRNet (classification/architectures/hrnet.py)
    int num_classes = 0;
    float *d_pool = nullptr, *d_fc_w = nullptr, *d_fc_b = nullptr;  // classifier: pooled features [B,2048], Linear params (fp32)
    int pool_cap = 0;
    std::vector<ParamSlot> params;
    std::map<std::string, int> param_index;
    std::vector<ConvLayer> layers;
    std::vector<TensorDesc> tensors;
    std::vector<Op> ops;
    std::vector<TapInfo> taps;
    bool taps_enabled = false;
    bool finalized = false;
    int num_cus = 256;
    int rB = 0, rH = 0, rW = 0;  // reserved shape
    int lastB = 0, lastH = 0, lastW = 0;
    int64_t ws_bytes = 0;
    bool ws_ready = false;  // reserve() has allocated the workspace for (rB, rH, rW)
    std::vector<void *> allocs;
    std::vector<GraphEntry> graphs;
    // live per-launch timing (bench.py roofline): HIP events on the launch stream around every conv
    bool prof_enabled = false;
    bool prof_clk = false;  // also stamp the device clock inside the kernels (their same-address atomics lengthen the launch by ~3-5 us)
    std::vector<ProfRecord> prof;
    size_t prof_used = 0;
    unsigned long long *d_clk = nullptr;  // [HH_PROF_SLOTS][4]: {min start, max end} on the wall clock, {core cycles, wall ticks} of workgroup 0
    double clk_khz = 0;                   // hipDeviceAttributeWallClockRate
    // lanes 1..3: internal streams forked from / joined to the caller's stream (also inside hipGraph capture)
    hipStream_t lane_streams[4] = {nullptr, nullptr, nullptr, nullptr};
    int lane_priority = 0;           // the lanes always share the priority of the caller's stream (see enqueue)
    bool lane_priority_set = false;
    std::vector<hipEvent_t> lane_events;
    size_t lane_events_used = 0;
    bool multi_lane = true;
    PlanSwitches sw;         // plan variants, read from the environment ONCE in hh_create and kept for the life of the handle
    // fp8 path (dtype == HH_DTYPE_FP8)
    bool calibrated = false;      // activation scales set (hh_calibrate)
    bool calibrating = false;     // the running forward records per-op output maxima
    unsigned *d_amax = nullptr;   // [ops.size()] bits of max |y| per op (calibration forwards only)
    std::vector<float> amax;      // host copy, kept over the calibration rounds (running maximum); [2 * ops]: the second half holds the
                                  // intermediate-tile maxima of fused BasicBlocks
    int elem() const { return dtype == 2 ? 1 : 2; }  // bytes per activation element

    int build();
    int check_plan(std::string *why) const;  // static RAW/WAR/WAW check of the multi-lane schedule
    int add_param(const std::string &name, std::vector<int64_t> shape, bool counter = false);
    int finalize();
    int reserve(int B, int H, int W);
    int enqueue(const float *images, int B, int H, int W, float *o1, float *o2, hipStream_t s);
    int forward(const float *images, int B, int H, int W, float *o1, float *o2, int use_graph, hipStream_t s);
    double flops(int B, int H, int W) const;
    int finalize_fp8();
    int calibrate(const float *images, int B, int H, int W, int rounds, hipStream_t s);
    int resolve_scales();  // amax -> per-op scales (plan order), d_mult of every layer
    int enqueue_fp8_conv(const Op &op, int B, int H, int W, float *o1, float *o2, hipStream_t s, ProfRecord *pr);
    int enqueue_fp8_upadd(const Op &op, int B, int H, int W, hipStream_t s);
    int enqueue_fp8_quant(const Op &op, int B, int H, int W, hipStream_t s);
    void assign_fp8_formats();  // which representations (e4m3 / bf16) every tensor of an fp8 plan needs, from its readers
    int enqueue_fp8_bb(const Op &op, int B, int H, int W, hipStream_t s, ProfRecord *pr);
    void release_workspace();
    ~hh_net();
};

// kernel-family / instantiation choice, shared with the standalone conv op of the training path (capi.cpp)
int hh_family_pick(int ks, int stride, int cin_pad, int coutp, int *KC, int *NT);
int hh_pick_config(int ks, int stride, int KC, int NT, int Wo);

// fp8 (e4m3, OCP) helpers shared by the engine and the C-ABI
unsigned char hh_f32_to_e4m3(float f);
float hh_e4m3_to_f32(unsigned char v);
int hh_fp8_family_pick(int cin, int cout, int *KC, int *NT);
int hh_fp8_pick_config(int ks, int stride, int KC, int NT, int Wo);
