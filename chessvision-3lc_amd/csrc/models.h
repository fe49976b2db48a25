// models.h -- per-model state owned by an Engine (packed layers + activation workspace).
#pragma once
#include "engine.h"

namespace cv {

// Workspaces are elastic: a model is loaded with no activation memory beyond what the range calibration needs and
// `reserve(n)` (re)allocates every tensor for min(n, engine chunk) images the first time a batch of that size arrives --
// a single-image server (the reference's Flask endpoint) stays under 1 GB, a throughput job grows once to its chunk.
struct Engine::UNet {
    bool bilinear = false;
    bool fuse_head = true;                          // OutConv fused into up4's last conv epilogue
    int cap = 0;                                    // images the workspace currently holds
    int max_cap = 64;                               // engine chunk: images per pass
    int last_n = 0;                                 // images in the most recent chunk
    // channel plan
    int c1 = 64, c2 = 128, c3 = 256, c4 = 512, c5 = 1024;
    ConvLayer inc0, inc1, d[4][2], upT[4], u[4][2];
    unsigned up_id[4] = {0, 0, 0, 0}, outc_id = 0;  // numeric-guard ids of the non-conv producers
    // split-f16 engine, throughput batches: up3.up / up4.up (K = 256 / 128: the whole weight block fits in LDS) on the persistent
    // LDS-resident-weights kernel (pointwise.hip: convt2x2_lds) instead of the generic tile; upT[i] remains for calibration, small
    // batches and CV_CONVT_FAST=0.  Same products in the same order: bit-identical.
    struct FastUp {
        bool on = false;
        DeviceBuffer wpk, scale, shift;
        std::vector<float> h_scale, h_shift;           // per GEMM row (dy, dx, co): 2^(row exponent) | bias
        int in_exp = 1 << 20, out_exp = 1 << 20;       // exponents the device copies are folded for
    } fast_up[4];
    DeviceBuffer outc_w, outc_b;
    DeviceBuffer inc0_wpk;                          // f16-based engines: MFMA image of inc.double_conv.0 for the fused first-layer kernel
    DeviceBuffer inc0_wpk2;                         // split-f16: the same layer for the in-kernel producer of inc.double_conv.3 (conv_halo.hip: FUSE0)
    bool fused_inc0 = false;                        // first layer + input packing in one kernel (pointwise.hip: inc0_mfma)
    bool fused_inc = false;                         // first layer produced inside inc.double_conv.3's halo kernel
    // activations
    Activation in8, a_inc0, cat[4], pool[4], dmid[4], bott, umid[4], uout[4];
    std::vector<Activation*> acts;                  // every tensor above that exists in this variant
    std::map<std::string, TensorRef> taps;          // module name -> tensor produced (capacity-sized refs)
    int64_t macs = 0;
};

struct Engine::ResNet {
    int cap = 0;
    int max_cap = 16384;
    int last_n = 0;
    DeviceBuffer stem_w, stem_wpk, stem_scale, stem_shift, fc_w, fc_b;
    std::vector<float> h_stem_scale, h_stem_shift;  // row exponents of the normalised stem filters folded in (ConvLayer::h_scale)
    int stem_out_exp = 0;                           // exponent the device copies are currently folded for
    unsigned stem_id = 0, head_id = 0;
    struct Block {
        ConvLayer conv1, conv2, down;
        bool has_down = false;
        Activation mid, out, sc;                     // conv1 output, block output, shortcut (if downsampled)
        // f16r: the shortcut convolution as a dedicated split-f16 kernel between the f32 twins (pointwise.hip: shortcut1x1s2);
        // `down` (the same layer on the f32-input MFMA through the generic kernel) remains for the calibration passes and CV_SHORTCUT_FAST=0
        bool fast_sc = false;
        DeviceBuffer sc_wpk, sc_scale, sc_shift;
        std::vector<float> h_sc_scale, h_sc_shift;   // BN affine with the weight rows' exponents folded in
        int sc_in_exp = 1 << 20, sc_out_exp = 1 << 20;   // exponents the device copies are folded for
        unsigned sc_id = 0;
    } blocks[8];
    Activation stem_out, pool_out;
    // f16r: layer1 (two BasicBlocks = four 3x3 convolutions 64 -> 64 on 16 x 16 maps) as ONE launch with the image resident in LDS
    // (conv_halo.hip: CHAIN).  chain_w = the four layers' packed weight stages back to back.  CV_RESNET_CHAIN=0 switches it off.
    DeviceBuffer chain_w;
    bool chain_ok = false;
    std::vector<Activation*> acts;
    std::map<std::string, TensorRef> taps;
    int64_t macs = 0;
};

// x in [0,1] is held as x * 2^7 inside the f16-based engines (both models' inputs)
constexpr int kInputExp = -7;

}  // namespace cv
