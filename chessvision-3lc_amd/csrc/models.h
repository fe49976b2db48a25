// models.h -- per-model state owned by an Engine (packed layers + activation workspace).
#pragma once
#include "engine.h"

namespace cv {

struct Engine::UNet {
    bool bilinear = false;
    bool fuse_head = true;                          // OutConv fused into up4's last conv epilogue
    int cap = 0;                                    // images per chunk
    int last_n = 0;                                 // images in the most recent chunk
    // channel plan
    int c1 = 64, c2 = 128, c3 = 256, c4 = 512, c5 = 1024;
    ConvLayer inc0, inc1, d[4][2], upT[4], u[4][2];
    DeviceBuffer outc_w, outc_b;
    // activations
    Activation in8, a_inc0, cat[4], pool[4], dmid[4], bott, umid[4], uout[4];
    std::map<std::string, TensorRef> taps;          // module name -> tensor produced (capacity-sized refs)
    int64_t macs = 0;
};

struct Engine::ResNet {
    int cap = 0;
    int last_n = 0;
    DeviceBuffer stem_w, stem_wpk, stem_scale, stem_shift, fc_w, fc_b;
    struct Block {
        ConvLayer conv1, conv2, down;
        bool has_down = false;
        Activation mid, out, sc;                     // conv1 output, block output, shortcut (if downsampled)
    } blocks[8];
    Activation stem_out, pool_out;
    std::map<std::string, TensorRef> taps;
    int64_t macs = 0;
};

}  // namespace cv
