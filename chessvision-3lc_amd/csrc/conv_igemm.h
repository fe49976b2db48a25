// conv_igemm.h -- host interface of the implicit-GEMM convolution kernels (conv_igemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "cv_kernels.h"

namespace cv {

// Every wave owns a 64-channel slab = kConvFC 16-row MFMA fragments; the host row permutation
// (engine.cpp: conv_row_to_channel) depends on it.
constexpr int kConvFC = 4;

enum ConvCfg {            // <channel tile> x <pixel tile> of one 256-thread workgroup
    kCfg64x256 = 0,       // Cout == 64 layers (full-resolution UNet, ResNet layer1)
    kCfg64x128 = 1,       // Cout == 64, few pixels
    kCfg128x128 = 2,      // Cout >= 128, few pixels (deep UNet / ResNet stages)
    kCfg128x256 = 3,      // Cout >= 128, many pixels
    kCfg128x256w8 = 4,    // 8 waves (two per SIMD), ring depth 3: the default for Cout >= 128
    kCfg256x256w8 = 5,    // 8 waves as 4 x 2, 64ch x 128px per wave, ring 2: fewest L2->LDS bytes per MFMA (Cout % 256 == 0)
    kNumConvCfg = 6
};

hipError_t conv_igemm_prepare();                                   // raise dynamic-LDS limits (once per device)
hipError_t conv_igemm_launch(int cfg, int ns, int dt, const ConvParams& p, hipStream_t stream);   // ns = LDS ring depth 2|3
// position-major launch (ConvParams::ptab / pcount / posN / nPtPer set; table-free offsets; unsplit): rows = [output position][image]
bool conv_cfg_has_pos(int cfg);
hipError_t conv_igemm_pos_launch(int cfg, int ns, int dt, const ConvParams& p, hipStream_t stream);
// two independent layers of one tile configuration in ONE launch (conv_igemm_pair_kernel; both with ConvParams::kbase)
hipError_t conv_igemm_pair_launch(int cfg, int ns, int dt_a, int dt_b, const ConvParams& a, const ConvParams& b, hipStream_t stream);   // dt_a == dt_b, or f32 beside f16
hipError_t conv_splitk_reduce_launch(int dt, const ConvParams& p, hipStream_t stream);   // second pass of a split-K launch (ConvParams::ksplit > 1)
bool conv_cfg_has_ns(int cfg, int ns);
int conv_cfg_ct(int cfg);

// conv_halo.hip: 3x3 / stride-1 layers with the input patch (+halo) resident in LDS across the nine taps
hipError_t conv_halo_prepare();
bool conv_halo_supported(int ct, int Ho, int Wo);
bool conv_halo_can_fuse_first_layer(int ct, int dt);     // ConvParams::f0_* (the 64-channel single-halo tile, split-f16)
bool conv_halo_has_th8(int ct);                          // the 8 x 16 patch variant of the tile exists (64-channel tile)
// th: patch rows of the tile, 16 (default) or 8 (twice the workgroups: launches with few patches)
hipError_t conv_halo_launch(int ct, int dt, const ConvParams& p, int n_images, hipStream_t stream, int th = 16);
int conv_cfg_pt(int cfg);
// ConvParams::chain: four chained 3x3 convolutions 64 -> 64 over whole 16 x 16 images in one launch (f16 kernels, f16r trunk)
bool conv_halo_has_chain();
hipError_t conv_halo_chain_launch(const ConvParams& p, int n_images, hipStream_t stream);

}  // namespace cv
