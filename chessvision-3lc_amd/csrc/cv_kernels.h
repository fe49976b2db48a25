// cv_kernels.h -- device-side contracts shared by the HIP kernels and the host engine.
//
// Activation layout ("PHWC"): every activation tensor lives in HBM as zero-bordered NHWC,
//     elem(n, y, x, c) = base[ ((n*(H+2) + y+1)*(W+2) + x+1) * Cs + Coff + c ]
// with a 1-pixel border that is zeroed once at allocation and never written again.  The border gives
// 3x3/pad-1 convolutions and the 3x3/s2/p1 max-pool (inputs are post-ReLU, so 0 == -inf for the max)
// their padding for free: no kernel on the path ever bounds-checks a load.  Cs (channel stride) and Coff
// let a producer write into a channel slice of a wider buffer, which is how the UNet skip `torch.cat`
// (Up.forward: cat([skip, upsampled], dim=1)) disappears: encoder and up-conv write the two halves of one
// buffer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cv {

// Element types of activation / weight tensors.
//   kF32   : float, f32-input MFMA (exact f32 products)
//   kF16   : _Float16, f16 MFMA, f32 accumulate
//   kSplit : "f16x3" -- every value v is carried as hi = f16(v), lo = f16(v - hi) (22+ significant bits);
//            a group of 8 channels occupies 32 bytes = one 16-B chunk of hi's and one of lo's (order [hi,lo]
//            for even groups, [lo,hi] for odd groups, so that MFMA fragment reads alternate LDS bank halves);
//            products are formed as hi*hi + hi*lo + lo*hi on the f16 MFMA with f32 accumulate.
enum DType : int { kF32 = 0, kF16 = 1, kSplit = 2 };
struct split_t { uint32_t raw; };              // 4 bytes per channel, see kSplit
__host__ __device__ inline int dtype_size(int dt) { return dt == kF16 ? 2 : 4; }
__host__ __device__ inline int dtype_group(int dt) { return dt == kF32 ? 4 : 8; }    // channels per 16/32-B group

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Split-f16 conversion of a PAIR of f32 values in three vector instructions: hi pair = v_cvt_pk_f16_f32(v0, v1); each lo =
// f16(v - hi) as ONE mixed-precision FMA (v_fma_mixlo / mixhi_f16: fma(f16 hi, -1, f32 v) rounded to f16 -- v - hi is exact in f32,
// so the single rounding equals the convert-back / subtract / convert chain the compiler emits for the C expression, which costs
// ~3 more instructions per value; r03_tuning.md steps 20-21).  hp / lp = the packed f16 pairs (element 0 in the low half).
// NOT for code where the scheduler may place it between INDEPENDENT MFMAs (epilogues that consume the accumulators are fine): the
// compiler does not look inside inline asm when it pads MFMA hazards, and pointwise.hip: shortcut1x1s2_kernel gave run-to-run
// different results with it until the C form replaced it (round 5, r05_tuning.md).
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hp, unsigned& lp) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t h = {(_Float16)v0, (_Float16)v1};
    hp = __builtin_bit_cast(unsigned, h);
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hp), "v"(v1));
#else
    lp = 0;                                           // host pass of the single-source compile: never executed
#endif
}

// Hand-over of wave-private LDS staging between the lanes of ONE wave (epilogue transposes, pooled-row stores): the writes of every
// lane are ordered before the reads of every lane at the language level -- release / acquire fences at wavefront scope around a wave
// barrier -- instead of relying on a compiler-only barrier plus the fact that a wave's DS operations issue in order.
// Costs no instruction: at wavefront scope the fences need no s_waitcnt and the barrier is a scheduling boundary only.
__device__ __forceinline__ void wave_lds_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

struct TensorRef {            // a channel slice of a PHWC tensor
    void* base;               // address of padded element (n=0, y=-1, x=-1, c=0)
    void* base32;             // f16r engine only: the tensor's unrounded f32 twin (same geometry, float elements) or null --
                              // residual adds read it and block outputs write it, so the ResNet trunk never rounds to f16
    int N, H, W;              // interior extent
    int Cs;                   // channel stride of the underlying buffer (elements)
    int Coff;                 // first channel of the slice
    int C;                    // channels in the slice
    // host-side bookkeeping (no kernel reads these): the tensor holds real_value * 2^-exp (power-of-two range scaling of
    // the f16 / split-f16 engines, engine.h: Activation); owner = the Activation the slice belongs to
    int exp;
    void* owner;
    // a slice that spans both halves of a concatenated buffer: channels >= split (counted inside the slice) are held with
    // exponent exp + exp_delta (the consumer folds 2^exp_delta into its weights for those input channels); 0 / 0 otherwise
    int split;
    int exp_delta;
    int f32_only;             // f16r engine: the tensor exists as f32 only (`base` already points at the f32 buffer); 0 otherwise
};

// Implicit-GEMM convolution:  D[ch][pix] = sum_k Wt[ch][k] * X[pix][k]
//   k runs over (tap, input channel); a "stage" is 128 bytes of K per row (64 f16 / 32 f32 values).
struct ConvParams {
    const char* x;            // input PHWC base
    const char* w;            // packed weights: [ctTile][stage][CT rows][8 x 16 B, XOR-swizzled]
    const int* koff;          // [nStages*8] byte offset of each 16-B K chunk relative to a pixel's base
    const int* kbase;         // [nStages] or null: per-stage base when koff[s*8+c] == kbase[s] + 16*c for every stage
    const float* scale;       // [rows] epilogue scale  (BN gamma / sqrt(var + eps), or 1) x the power-of-two range factors
    const float* shift;       // [rows] epilogue shift  (BN beta - mean*scale, or conv bias) x 2^-out_exp
    const char* res;          // optional residual PHWC base (same pixel grid as the output), or null
    float res_mul;            // residual multiplier 2^(res_exp - out_exp): brings the shortcut tensor to the output's scale
    int res_f32;              // f16 kernels only: `res` is the f32 twin of the shortcut tensor (float PHWC, same rCs / rCoff)
    char* y32;                // f16 kernels only: optional second output, the unrounded f32 twin of y (same yCs / yCoff), or null
    char* y;                  // output PHWC base
    int M;                    // output pixels = N*Ho*Wo
    int Ho, Wo;               // output pixel grid per image
    int xHp, xWp;             // input padded dims (Hi+2, Wi+2)
    int stride;               // convolution stride
    int xCs;                  // input channel stride (elements)
    int xCoffBytes;           // byte offset of the input slice's first channel (conv_halo.hip; conv_igemm folds it into koff)
    int yHp, yWp;             // output padded dims
    int yCs, yCoff;           // output channel stride / first channel
    int rCs, rCoff;           // residual channel stride / first channel
    int Cout;                 // channels per output pixel (rows / 4 in shuffle mode)
    int rows;                 // real GEMM rows (multiple of 16); rows beyond are padding
    int nStages;              // K stages of 128 bytes
    int nCt;                  // channel tiles
    int relu;
    int shuffle;              // 1: rows = (dy,dx,co) of a k2 s2 transposed conv -> pixel-shuffle store
    // optional fused 1x1 head (UNet OutConv, Cout == 64 == one channel tile): logit = sum_c relu(bn(conv))[c] * head_w[c]
    // + head_b is written to head_logits[pixel] (f32, (N,1,H,W)) INSTEAD of the activation tensor; head_mask (nullable)
    // receives sigmoid(logit) > head_thr ? 255 : 0.
    const float* head_w;
    const float* head_b;
    float* head_logits;
    uint8_t* head_mask;
    float head_thr;
    // optional fused 2x2/s2 max-pool of the (post-ReLU) output (conv_halo.hip): a second, pooled PHWC tensor
    char* pool_y;             // null = no pooled output
    int pHp, pWp;             // its padded dims (Ho/2 + 2, Wo/2 + 2)
    int pCs, pCoff;           // its channel stride / first channel
    // numeric guard: a lane that stores a non-finite value (f16 range exceeded, NaN input) records the launch's layer id
    // with atomicMin; the host turns it into CV_ERR_NUMERIC naming the layer (cv_engine_numeric_status).  Null = off.
    unsigned* flag;
    unsigned layer_id;
    // optional fused producer (conv_halo.hip, 64-channel single-halo tile only): the 64-channel input of THIS layer is never read
    // from memory -- every workgroup computes the haloed patch it needs in LDS from the caller's 3-channel image with the
    // network's first conv + BN + ReLU (UNet inc.double_conv.0).  Null f0_x = off.
    const void* f0_x;         // (n,3,256,256) f32 or (n,256,256,3) u8
    const void* f0_w;         // first-layer weights: [channel block 2][fragment 2][hi|lo][lane 64] x half8, k = (ky*3 + kx)*3 + c
    const float* f0_scale;    // [64] epilogue constants of the first layer (range factors folded)
    const float* f0_shift;
    float f0_in_mul;          // 2^-in_exp applied to the image values
    int f0_u8;
    unsigned long long* stamp;   // diagnostic builds (-DCV_STAMP=1) only: per-workgroup cycle stamps, else null
    // split-K (conv_igemm.hip; small launches: single boards, the 64-square classifier batch): the K stages are dealt to `ksplit`
    // workgroups per output tile, `kper` stages each; every workgroup writes its raw f32 accumulators to
    // partial[split][pixel][prow channels] and conv_splitk_reduce_kernel sums the splits IN ORDER (deterministic) and runs the
    // epilogue (BN affine, residual, ReLU, conversion, stores).  ksplit <= 1 / partial == null: off.
    float* partial;
    int ksplit, kper;
    int prow;                 // channels per pixel row of `partial` (= padded GEMM rows)
    // CHAIN launch of conv_halo.hip (f16r ResNet-18 layer1: four 3x3 convolutions 64 -> 64 on 16 x 16 maps = two BasicBlocks in ONE
    // launch, the image resident in LDS between them): `w` holds the 36 weight stages of convolutions 0..3 back to back, `x` is the
    // f16 copy of the trunk entering the stage.  Convolution c = 0, 2: relu(bn(conv)) stays in LDS as f16.  c = 1: + ch_res0
    // (f32 twin of the input) * ch_res_mul[0], ReLU -> f32 to ch_y32_mid (the first block's output twin) and f16 into LDS.
    // c = 3: + ch_y32_mid * ch_res_mul[1], ReLU -> y32 (f32 twin) and y (f16 copy).  All tensors 18 x 18 x 64 padded planes.
    int chain;                // 0 = off | 2 = the form described above, two workgroups per CU | 1 = one workgroup per CU, 512 registers: the
                              // f32 trunk stays in registers through both blocks (ch_y32_mid is not touched)
    const float* ch_scale[4]; // per-convolution epilogue constants (range factors folded, as `scale` / `shift`)
    const float* ch_shift[4];
    const char* ch_res0;
    char* ch_y32_mid;
    float ch_res_mul[2];
    unsigned ch_layer_id[4];  // numeric-guard ids of the four convolutions
    // POSITION-MAJOR launch (conv_igemm_kernel<..., POS = true>; round 6: 3x3 layers on the 2x2 / 4x4 / 8x8 maps of ResNet-18 at
    // throughput batch sizes).  GEMM rows are ordered [output position][image] instead of [image][position], so all rows of a pixel
    // tile share ONE output position (oy, ox) and therefore one set of taps that read a real pixel: the K loop walks only those
    // stages (`ptab`), the taps that would gather the zero border -- 5 of 9 at every position of a 2x2 map -- are never fetched nor
    // multiplied.  A skipped stage would have added exact zeros, so every output keeps its K order and its bits.  Null ptab = off.
    const int* ptab;          // [Ho*Wo][nStages][2]: {weight stage index, gather base (kbase of that stage)} of the live stages, K order
    const int* pcount;        // [Ho*Wo] live stages of each position
    const int* porder;        // [Ho*Wo] positions by falling stage count: the walk starts the long K loops first (see conv_igemm_body.h)
    int posN;                 // images in the launch = rows per position
    int nPtPer;               // pixel tiles per position = ceil(posN / PT)
};

}  // namespace cv
