// pointwise.h -- host interface of the bandwidth-bound kernels (pointwise.hip).  `dt` is a cv::DType.
#pragma once
#include <hip/hip_runtime.h>
#include "cv_kernels.h"

namespace cv {

// NCHW float32 (n, c, h, w) -> PHWC slice `dst` (dst.C >= c; channels [c, dst.C) are written as zero)
// Range scaling: a TensorRef carries `exp` (stored = real * 2^-exp); packers multiply by 2^-dst.exp, readers that leave the
// internal layout by 2^src.exp -- powers of two, so exact.  `flag` / `layer_id`: numeric guard (cv_kernels.h), nullable.
hipError_t pack_nchw_f32(int dt, const float* src, int c, const TensorRef& dst, unsigned* flag, hipStream_t s);
// (n, h, w, 3) uint8 -> PHWC slice with value/255 (core.py:215); dst.C == 8, channels 3..7 zero
hipError_t pack_hwc3_u8(int dt, const uint8_t* src, const TensorRef& dst, hipStream_t s);
// PHWC slice -> NCHW float32
hipError_t unpack_nchw_f32(int dt, const TensorRef& src, float* dst, hipStream_t s);

hipError_t maxpool2x2(int dt, const TensorRef& src, const TensorRef& dst, hipStream_t s);
hipError_t maxpool3x3s2(int dt, const TensorRef& src, const TensorRef& dst, hipStream_t s);
hipError_t upsample_bilinear2x(int dt, const TensorRef& src, const TensorRef& dst, unsigned* flag, unsigned layer_id,
                               hipStream_t s);
// max |stored value| of the slice, as float bits (>= 0x7f800000: a non-finite value is present); *out must start at 0
hipError_t absmax(int dt, const TensorRef& src, unsigned* out, hipStream_t s);

// rounding-bias calibration: out[slice][tap][c] (double, slices x k*k x src.C) = sum of the stored f16 input under tap (ky, kx) over
// the slice's images and all Ho x Wo output positions of a k x k / stride / pad (k-1)/2 convolution (f16 tensors only)
hipError_t tap_sums_f16(const TensorRef& src, int Ho, int Wo, int stride, int k, int slices, double* out, hipStream_t s);

// 1x1 conv C -> 1 (+bias): logits (n,1,h,w) float32; mask (nullable) = sigmoid(logit) > thr ? 255 : 0
hipError_t outc_1x1(int dt, const TensorRef& src, const float* w, const float* bias, float* logits,
                    uint8_t* mask, float threshold, unsigned* flag, unsigned layer_id, hipStream_t s);

// ResNet stem: conv 7x7 s2 p3 (1 -> 64, no bias) + BN affine + ReLU.  x: (n,1,64,64) f32 or (n,64,64) u8
// (u8 is scaled by /255 first, core.py:237).  w: [64][49] f32, scale/shift [64].  dst: 64 ch @ 32x32.
hipError_t stem7x7(int dt, const void* x, bool x_is_u8, int n, const float* w, const float* scale,
                   const float* shift, const TensorRef& dst, hipStream_t s);

// Fused stem for the f16 / split-f16 engines: conv 7x7 s2 p3 + BN + ReLU + max_pool2d(3,2,1) on the MFMA.
// wpk: packed filter bank [hi|lo][k-step 2][fragment 4][lane 64] x half8 (resnet.cpp: pack_stem_mfma).  dst: 64 ch @ 16x16.
// in_exp: the input plane is held as x * 2^-in_exp inside the kernel (x in [0,1] -> in_exp = -7 keeps the lo halves normal)
hipError_t stem_pool_mfma(int dt, const void* x, bool x_is_u8, int n, const void* wpk, const float* scale,
                          const float* shift, int in_exp, const TensorRef& dst, unsigned* flag, unsigned layer_id,
                          hipStream_t s);

// UNet first layer for the f16 / split-f16 engines, fused with the input packing: conv 3x3 p1 (3 -> 64) + BN + ReLU straight from
// the caller's image (x: (n,3,256,256) f32 or (n,256,256,3) u8, scaled by /255).  wpk: [hi|lo][fragment 4][lane 64] x half8
// (unet.cpp: pack_inc0_mfma), k = (ky*3 + kx)*3 + c.  dst: 64 ch @ 256x256, whole buffer (Cs == 64).
hipError_t inc0_mfma(int dt, const void* x, bool x_is_u8, int n, const void* wpk, const float* scale, const float* shift,
                     int in_exp, const TensorRef& dst, unsigned* flag, unsigned layer_id, hipStream_t s);

// global average pool + Linear(C -> 13) (+ optional softmax).  w: [13][C] f32, b: [13]
hipError_t head_avgpool_fc(int dt, const TensorRef& src, const float* w, const float* b, float* out,
                           int softmax, unsigned* flag, unsigned layer_id, hipStream_t s);
hipError_t softmax13(const float* logits, int n, float* probs, hipStream_t s);

// f16r engine: ResNet shortcut conv 1x1 / stride 2 + BN between f32 twins (x32 -> y32, both f32_only PHWC tensors, y32.C == 2 x32.C,
// C in {64, 128, 256}) at f32 grade: three split-f16 MFMA products per MAC.  wpk: [C_out/128][C_in/32][fragment 8][hi | lo][lane 64] x
// half8, MFMA row i of fragment f = channel 32 (i/4) + 4 f + i%4 of the group (resnet.cpp: pack_shortcut); scale / shift [C_out].
hipError_t shortcut1x1s2(const TensorRef& x32, const void* wpk, const float* scale, const float* shift, const TensorRef& y32,
                         unsigned* flag, unsigned layer_id, hipStream_t s, bool split = false);

// UNet up3.up / up4.up (split-f16 engine): conv_transpose2d k2 s2 + bias, x (C in {128, 256}, own buffer) -> channel slice y (C/2 channels,
// 2H x 2W) of the concatenated tensor, on the LDS-resident-weights kernel (`CONVT` form).  wpk as for shortcut1x1s2 with GEMM row
// r = (dy * 2 + dx) * C/2 + co; scale / shift [2 C] (shift = the bias per row).
hipError_t convt2x2_lds(const TensorRef& x, const void* wpk, const float* scale, const float* shift, const TensorRef& y, unsigned* flag,
                        unsigned layer_id, hipStream_t s);

// MFMA lane-map probes used by cv_selftest_mfma (D = A*B^T with A:16xK, B:16xK row-major)
hipError_t mfma_probe_f16(const half_t* a, const half_t* b, float* d, hipStream_t s);   // K = 32
hipError_t mfma_probe_f32(const float* a, const float* b, float* d, hipStream_t s);     // K = 16

}  // namespace cv
