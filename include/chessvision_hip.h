/*
 * chessvision_hip.h -- C ABI of libchessvision_hip.so
 *
 * MI355X (gfx950 / CDNA4) implementation of the ChessVision CNN inference hot path:
 *   - UNet(3 -> 1) 256x256 board-segmentation forward
 *   - ResNet-18 (1 channel in, 13 classes) 64x64 square-classifier forward
 *
 * The reference (gudbrandtandberg/ChessVision-3LC) is pure Python and has no FFI of its own; the seam
 * this library sits behind is the reference's "callable model object" boundary:
 *     chessvision/core.py:220   logits = self.board_extractor(image_batch)      -> cv_unet_forward
 *     chessvision/core.py:241   predictions = self.classifier(batch)            -> cv_resnet18_forward
 *     chessvision/core.py:84-106, utils.py:42-86  model construction + checkpoint load
 *                                                                               -> cv_engine_create,
 *                                                                                  cv_load_unet, cv_load_resnet18
 *     chessvision/core.py:215-216 (u8 HWC /255 -> NCHW), core.py:273 + utils.py:101-112 (sigmoid,
 *     threshold -> 0/255 mask)                                                  -> cv_unet_forward_u8
 *     chessvision/core.py:236-237 (/=255), core.py:242 (softmax dim=1)          -> cv_resnet18_forward_u8
 * The ctypes binding a maintainer adds on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns CV_OK (0) or a CV_ERR_* code; cv_last_error() returns a thread-local,
 *     NUL-terminated description of the most recent failure on the calling thread.  Nothing aborts and no C++
 *     exception crosses the boundary (host allocation failures come back as CV_ERR_NOMEM).
 *   - all tensor arguments of the *_forward* functions are DEVICE pointers on the engine's device;
 *     `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous
 *     with respect to the host and ordered on `stream`.
 *   - the caller owns inputs and outputs; the engine owns its packed weights and its workspace and
 *     releases them in cv_engine_destroy.  The library never frees caller memory.
 *   - an engine serialises concurrent forward calls with an internal mutex (one workspace per engine and model).  Forwards of the
 *     SAME model may be enqueued on different streams: when the stream of a model's forward differs from that of its previous
 *     forward, the call first waits (on the host) for the previous stream, so the two never overlap on the shared workspace
 *     (rounds 1-5 left this to the caller; a soak in round 6 showed the library's own host layer getting it wrong).  Keep one
 *     stream per engine and the wait never happens.  A UNet forward and a ResNet-18 forward of one engine may run on two streams
 *     at once (separate workspaces and scratch).
 *   - a server that runs several engines side by side (one per request thread, each on its own stream) should send ONE warm-up
 *     request through each of them, one after the other, before the threads start: the HIP runtime binds a stream to a hardware
 *     queue at the stream's first use, and streams first used at the same moment end up sharing queues (measured on MI355X, round 6:
 *     four engine pairs 1850-1880 requests/s when first used together, 2340-2370 when first used in turn).  The Python host layer
 *     does this for its request slots (ChessVision._warm_slot).
 *   - engines of one process may be created, loaded, used and destroyed from different threads at the same time: loads are
 *     serialized process-wide, and memory released by cv_engine_destroy is cached for the next engine (cv_trim_memory).
 *   - plain C types only: no torch / C++ types cross this boundary.
 */
#ifndef CHESSVISION_HIP_H
#define CHESSVISION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CV_ABI_VERSION 6      /* 2: CV_PREC_F16R, CV_ERR_NUMERIC + cv_engine_numeric_status, cv_engine_set_chunk before cv_load_* only
                                 3: cv_board_homographies, cv_engine_export/import_calibration, cv_process_image (additions only)
                                 4: cv_find_contours (addition); cv_find_quadrangle follows CHAIN_APPROX_TC89_KCOS
                                 5: cv_image_result_t gains `squares` (appended; zero the struct before use, as before)
                                 6: cv_process_image_v2 (the caller states sizeof(cv_image_result_t) as IT was compiled); cv_process_image is
                                    the ABI 3 / 4 entry point again and never touches `squares` -- a binary built against the shorter struct
                                    of those versions is safe once more */

enum cv_status {
    CV_OK = 0,
    CV_ERR_INVALID = 1,   /* bad argument / shape / missing or mis-shaped parameter            */
    CV_ERR_HIP = 2,       /* a HIP runtime call failed (message carries hipGetErrorString)     */
    CV_ERR_STATE = 3,     /* model not loaded, wrong device, ...                               */
    CV_ERR_NOMEM = 4,
    CV_ERR_NUMERIC = 5    /* a layer produced a non-finite value (f16 range exceeded / NaN input): results invalid */
};

/* arithmetic type of the convolution path */
enum cv_precision {
    CV_PREC_F32 = 0,      /* f32 activations/weights, f32-input MFMA (exact f32 products, f32 accumulate) */
    CV_PREC_F16 = 1,      /* f16 activations/weights, f16 MFMA with f32 accumulate, f32 BN/bias epilogue  */
    CV_PREC_F16X3 = 2,    /* split-f16: every activation/weight carried as hi + lo f16 (>= 22 significant bits),
                             products formed as hi*hi + hi*lo + lo*hi on the f16 MFMA with f32 accumulate:
                             f32-grade results at up to 1/3 of the f16 MFMA rate (5.3x the f32 MFMA rate) */
    CV_PREC_F16R = 3      /* f16 with an f32 residual trunk (the classifier's fp16 mode, BASELINE configs[2]): f16
                             activations/weights and ONE f16 MFMA product per MAC, f32 accumulate, as CV_PREC_F16 -- but
                             the ResNet-18 stem is computed at f32 grade and every tensor of the residual trunk (pooled stem
                             output, block outputs, down-sampled shortcuts) keeps an unrounded f32 twin that the residual
                             adds read and the block epilogues write, so f16 rounding never accumulates along the
                             skip path; soft-max probabilities within 1e-3 of the fp32 reference.  The UNet (no
                             residuals) runs exactly as under CV_PREC_F16. */
};

typedef struct cv_engine cv_engine_t;     /* opaque */

/* One entry of a PyTorch state dict: name = state-dict key (reference key names, e.g.
 * "inc.double_conv.0.weight", "layer2.0.downsample.1.running_var", "fc.bias"); data = HOST pointer to
 * contiguous row-major float32.  BatchNorm "*.num_batches_tracked" entries may be omitted or passed (any shape):
 * they are ignored. */
typedef struct cv_param {
    const char*  name;
    const float* data;
    int32_t      ndim;
    int64_t      shape[4];
} cv_param_t;

/* ---- library / error ---------------------------------------------------------------------------- */
int         cv_abi_version(void);
const char* cv_last_error(void);
int         cv_device_count(int* count);

/* ---- engine lifecycle ---------------------------------------------------------------------------- */
/* device: HIP device ordinal; precision: enum cv_precision.  Replaces nn.Module.to(device)/.eval()
 * (core.py:105-106,149-150): an engine is always in inference mode. */
int cv_engine_create(int device, int precision, cv_engine_t** out);
int cv_engine_destroy(cv_engine_t* eng);

/* Device and page-locked blocks of a destroyed engine (weights, workspaces, staging) stay in a process-wide cache and are handed to
 * the next engine that asks for a block of that size, so a server that reloads models does not unmap device memory beside the
 * running forwards of its other engines (round 6: that combination ended in device faults).  The cache holds at
 * most CV_MEM_CACHE_MB (environment, default 16384) and CV_MEM_CACHE=0 switches it off.  cv_trim_memory waits for the device and
 * returns every cached block to the driver; *bytes_freed (may be NULL) receives their total size.  The reference has no counterpart:
 * torch's caching allocator plays this role there (torch.cuda.empty_cache()). */
int cv_trim_memory(size_t* bytes_freed);

/* Load + pack weights (host fp32 state dict -> device, BN folded to per-channel scale/shift applied in
 * the conv epilogue).  cv_load_unet auto-detects the transposed-conv vs bilinear variant from the
 * presence of "up1.up.weight" and validates every key/shape, failing with the offending key name.
 * Replaces model.load_state_dict(...) of utils.py:59-79. */
int cv_load_unet(cv_engine_t* eng, const cv_param_t* params, int n_params);
int cv_load_resnet18(cv_engine_t* eng, const cv_param_t* params, int n_params);

/* Largest number of images (UNet) / squares (ResNet) processed per internal pass; larger batches are
 * looped inside the forward call.  0 = keep default (64 images / 16384 squares).  Must be called before cv_load_*.
 * The activation workspace is elastic: it is sized for the largest batch seen so far (at most one chunk) and grows
 * on demand, so an engine that only ever serves single images (the reference's Flask endpoint,
 * app/computeroot/cv_endpoint.py:131-133) stays well under 1 GB while a throughput job grows once to its chunk. */
int cv_engine_set_chunk(cv_engine_t* eng, int unet_images, int resnet_squares);

/* Bytes of activation workspace currently allocated by the engine (both models). */
int cv_engine_workspace_bytes(cv_engine_t* eng, size_t* bytes);

/* Numeric guard.  The f16-based precisions hold every tensor as value * 2^-exp with a per-tensor exponent calibrated at
 * load time (2^11 head-room over the calibration maximum); a layer that nevertheless produces a non-finite value
 * (range exceeded, NaN/inf in the input) records itself on the device.  This call synchronises `stream`, returns
 * CV_ERR_NUMERIC with the name of the first such layer in cv_last_error() and re-arms the guard; CV_OK when every
 * forward since the previous check was clean.  The forward calls themselves stay asynchronous. */
int cv_engine_numeric_status(cv_engine_t* eng, void* stream);

/* The load-time range calibration as data (f16-based precisions): the per-tensor exponents of a loaded model, two int32 per tensor
 * in the engine's fixed tensor order.  export: exps == NULL returns the count only.  import: sets them (count must match; the
 * layers re-fold their epilogue constants at their next launch; *changed = 1 when any exponent differed).  Multi-GPU jobs
 * broadcast rank 0's vector so that every rank computes with IDENTICAL scalings even if a calibration pass were ever to differ
 * between ranks (chessvision/distributed.py: sync_calibration). */
int cv_engine_export_calibration(cv_engine_t* eng, const char* model, int32_t* exps, int capacity, int* count);
int cv_engine_import_calibration(cv_engine_t* eng, const char* model, const int32_t* exps, int count, int* changed);

/* ---- forward (the hot path) ----------------------------------------------------------------------
 * Reproducibility: a forward is bit-identical run to run, across engines, processes and ranks FOR THE SAME BATCH SHAPE.  The launch
 * plan depends on the batch (tile choice; launches too small to fill the chip split their K loop over several workgroups and sum
 * the f32 partials in a fixed order), so the same image inside a batch of 1 and inside a batch of 64 goes through different f32
 * summation orders: results agree to the f32 rounding of a dot product (logits within ~1e-5 for f16x3 / f32; within the storage
 * rounding, ~5e-3, for the f16 modes), arg-max and FEN agree, bits do not.  cv_process_image (batch 1 / 64 squares) and a job of
 * ChessVision.process_images (64 boards per pass) therefore agree to that tolerance, not bit for bit.  CV_SPLITK=0 in the
 * environment removes the split launches (the largest source of the variation) at the cost of single-image latency. */
/* x: (batch,3,256,256) float32 NCHW in [0,1]  ->  logits: (batch,1,256,256) float32.
 * Same tensor contract as `self.board_extractor(image_batch)` (core.py:220). */
int cv_unet_forward(cv_engine_t* eng, const float* x, int batch, float* logits, void* stream);

/* x: (n,1,64,64) float32 in [0,1]  ->  logits: (n,13) float32 (no softmax).
 * Same tensor contract as `self.classifier(batch)` (core.py:241). */
int cv_resnet18_forward(cv_engine_t* eng, const float* x, int n, float* logits, void* stream);

/* Fused variants for the batched pipeline (u8 in, pre/post-processing of core.py on device):
 * x_u8: (batch,256,256,3) uint8 HWC, channel order as given (no BGR->RGB swap, core.py:212-216);
 * logits: (batch,1,256,256) float32; mask (nullable): (batch,256,256) uint8 = 255 where
 * sigmoid(logit) > threshold else 0 (core.py:273, utils.py:101-112). */
int cv_unet_forward_u8(cv_engine_t* eng, const uint8_t* x_u8, int batch, float* logits, uint8_t* mask,
                       float threshold, void* stream);
/* squares_u8: (n,64,64) uint8 -> probs: (n,13) float32 = softmax(logits, dim=1) (core.py:236-242). */
int cv_resnet18_forward_u8(cv_engine_t* eng, const uint8_t* squares_u8, int n, float* probs, void* stream);

/* (n,13) logits -> softmax probabilities, in place allowed (core.py:242). */
int cv_softmax13(cv_engine_t* eng, const float* logits, int n, float* probs, void* stream);

/* ---- introspection (tests, bench, profiling) ----------------------------------------------------- */
/* Copy an intermediate activation of the LAST forward chunk out of the workspace as float32 NCHW
 * (n,c,h,w written to dims[4]).  `name` is the producing module's state-dict prefix, e.g.
 * "inc.double_conv.3", "down4.maxpool_conv.1.double_conv.5" (= named_modules()[52],
 * train_unet.py:210), "up1.up", "layer1.0", "maxpool".  out_capacity in floats.  Synchronises.
 * Tensors a fused launch never writes have no tap: under CV_PREC_F16R ResNet-18's layer1 runs as ONE chained launch, so
 * "layer1.0.act1" / "layer1.1.act1" do not exist and "layer1.0" is read from the block's f32 twin (CV_ERR_INVALID naming
 * CV_RESNET_CHAIN=0, the switch that runs layer1 as four launches and materialises them); the UNet's fused first two convolutions
 * (CV_FUSE_INC=0 to split them) likewise leave "inc.double_conv.2" unwritten at the throughput sizes. */
int cv_get_activation(cv_engine_t* eng, const char* model, const char* name, float* out_host,
                      size_t out_capacity, int64_t dims[4]);

/* Power-of-two exponent the named activation is currently stored with (stored = value * 2^-exponent; 0 for f32). */
int cv_get_activation_exponent(cv_engine_t* eng, const char* model, const char* name, int* exponent);

/* Algorithmic work of one forward: multiply-accumulates per image (UNet) / per square (ResNet). */
int cv_model_macs(cv_engine_t* eng, const char* model, int64_t* macs);

/* Time the conv kernels of one forward with HIP events on `stream` (used by bench.py for the
 * roofline figure): runs the forward `iters` times on resident inputs and returns the summed
 * duration (ms) of all implicit-GEMM conv launches and their count. */
int cv_profile_convs(cv_engine_t* eng, const char* model, const void* x, int batch, void* out, int iters,
                     void* stream, float* conv_ms_total, int* conv_launches, float* all_ms_total);

/* Iterate the per-launch records of the last cv_profile_convs call (index 0 .. until CV_ERR_INVALID):
 * name = producing module, ms = event-timed duration, macs = algorithmic multiply-accumulates. */
int cv_profile_entry(cv_engine_t* eng, int index, char* name, int name_cap, float* ms, double* macs,
                     int* is_conv);

/* Algorithmic HBM bytes (inputs + outputs + weights, once each, at the engine's storage width) of record `index`. */
int cv_profile_entry_bytes(cv_engine_t* eng, int index, double* bytes);

/* Which kernel instantiation ran record `index` (conv family: "conv3x3_halo_kernel<split_t,64,16x16>", ...; empty for the
 * other kernels, whose record name already is the kernel).  Lets a caller rebuild the per-kernel roofline that
 * rocprofv3 --kernel-trace reports by template name. */
int cv_profile_entry_kernel(cv_engine_t* eng, int index, char* kernel, int kernel_cap);

/* Stand-alone single-layer entry points used by the parity tests (float32 NCHW device tensors in,
 * float32 NCHW out; packing to the internal layout happens inside, on `stream`).
 *   conv:  y = act( scale[c] * conv2d(x, w, stride, pad=(k-1)/2) + shift[c] (+ residual) )
 *          w: HOST (cout,cin,k,k) float32; scale/shift: HOST (cout); residual nullable (device). */
int cv_op_conv2d(cv_engine_t* eng, const float* x, int n, int cin, int h, int w_, const float* w_host,
                 int cout, int k, int stride, const float* scale_host, const float* shift_host,
                 const float* residual, int relu, float* y, void* stream);
/*   conv-transpose k2 s2:  y(n,cout,2h,2w) = convT(x, w) + bias;  w: HOST (cin,cout,2,2) */
int cv_op_conv_transpose2x2(cv_engine_t* eng, const float* x, int n, int cin, int h, int w_,
                            const float* w_host, int cout, const float* bias_host, float* y, void* stream);
/*   conv 1x1 c -> 1 + bias: logits (n,1,h,w) float32; mask (nullable, n x h x w uint8) = sigmoid(logit) > threshold ? 255 : 0
 *   (the stand-alone OutConv kernel; the model path fuses it into the last conv's epilogue).  w: HOST (c), bias: HOST (1) */
int cv_op_outc_1x1(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, const float* w_host, const float* bias_host,
                   float threshold, float* logits, uint8_t* mask, void* stream);
int cv_op_maxpool2x2(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, float* y, void* stream);
int cv_op_maxpool3x3s2(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, float* y, void* stream);
int cv_op_upsample_bilinear2x(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, float* y,
                              void* stream);

/* ---- classical stages either side of the CNNs (SURVEY.md section 8f "next" rows) ------------------------------ */
/* Binary mask (h*w uint8, 0 / non-0) -> board quadrangle, host-side C++: contours (outer + holes, RETR_CCOMP order) compressed
 * by CHAIN_APPROX_TC89_KCOS, the reference's area / bounding-box filter when more than one contour, closed-curve
 * Douglas-Peucker at 10 % of the (float-segment) perimeter, first 4-vertex result, reference vertex rotation.  Replaces
 * ChessVision._find_quadrangle (core.py:357-411: cv2.findContours, contourArea, boundingRect, arcLength, approxPolyDP), each in
 * OpenCV's own arithmetic and order.  quad = 4 x (x, y) in mask pixels; *found = 0 when none.  Needs no GPU and no engine. */
int cv_find_quadrangle(const uint8_t* mask, int h, int w, int32_t quad[8], int* found);
/* The same for n masks (n,h,w) on n_threads host threads (0 = hardware concurrency, capped at 32);
 * quads: n x 8 int32, found: n x int32. */
int cv_find_quadrangles(const uint8_t* masks, int n, int h, int w, int32_t* quads, int32_t* found, int n_threads);
/* The contours themselves, as cv2.findContours(mask, RETR_CCOMP, method)[0] lists them (core.py:360): method 0 =
 * CHAIN_APPROX_NONE, 1 = CHAIN_APPROX_TC89_KCOS.  Points of all contours back to back in xy (x, y pairs, room for cap_points
 * points), per contour its point count and a hole flag (room for cap_contours each); *n_contours receives the count.
 * CV_ERR_INVALID when a capacity is too small (h*w contours of 4*h*w points in total always suffice). */
int cv_find_contours(const uint8_t* mask, int h, int w, int method, int32_t* xy, int64_t cap_points, int32_t* counts,
                     int32_t* holes, int64_t cap_contours, int64_t* n_contours);

/* (n,h,w,channels) uint8 -> (n,out_h,out_w,channels) uint8, INTER_AREA semantics (cv2.resize at core.py:212): exact
 * box mean with round-half-up for integer shrink factors, coverage-weighted mean otherwise.  DEVICE pointers. */
int cv_resize_area_u8(cv_engine_t* eng, const uint8_t* src, int n, int h, int w, int channels, uint8_t* dst,
                      int out_h, int out_w, void* stream);

/* Quadrangles -> the matrices of the board warp, host side, in OpenCV's order of operations (utils.extract_perspective,
 * utils.py:115-132: cv2.getPerspectiveTransform(approx, dest) with dest = (0,0), (w,0), (w,h), (0,h); cv2.warpPerspective then
 * inverts the matrix).  quads: n x 4 x (x, y) float32 source-image pixels in the reference's vertex order (TR, TL, BL, BR);
 * forward (nullable): n x 9 doubles, row-major, m[8] = 1; inverse (nullable): n x 9 doubles = cv::invert(forward) = the map from
 * board pixels to source pixels that cv_extract_squares_u8 consumes.  Degenerate quadrangles give all-zero matrices (OpenCV's
 * behaviour: every board pixel then reads source pixel (0,0)).  Needs no GPU and no engine. */
int cv_board_homographies(const float* quads, int n, int out_w, int out_h, double* forward, double* inverse);

/* Per board: perspective warp to 512x512 (bilinear, zero border) + BGR->gray + horizontal flip + split into 64 squares
 * (utils.py:131-132, core.py:298-300, 419-439), fused.  images: DEVICE (n,h,w,3) uint8 BGR; inv_host: HOST n x 9
 * doubles = inverse of the getPerspectiveTransform matrix (board pixel -> source pixel); squares: DEVICE (n*64,64,64)
 * uint8 in a8..h1 order = the input of cv_resnet18_forward_u8; boards (nullable): DEVICE (n,512,512) uint8 gray board.
 * Byte-exact against cv2's fixed-point arithmetic as restated in oracle/classical_ref.py (1/32-pixel coordinates in
 * WarpPerspectiveInvoker's block order, integer bilinear weights, round half up, 15-bit gray).  Output pointers 4-byte aligned.
 * Synchronises `stream` before returning. */
int cv_extract_squares_u8(cv_engine_t* eng, const uint8_t* images, int n, int h, int w, const double* inv_host,
                          uint8_t* squares, uint8_t* boards, void* stream);

/* The same with the n x 9 matrices already on the DEVICE (inv_dev): fully asynchronous, nothing is staged or
 * synchronised -- the form the batched pipeline uses so that consecutive jobs overlap on the stream. */
int cv_extract_squares_u8_dev(cv_engine_t* eng, const uint8_t* images, int n, int h, int w, const double* inv_dev,
                              uint8_t* squares, uint8_t* boards, void* stream);

/* (n_boards,64,13) class probabilities (HOST, classifier order a8..h1, or h1..a8 when flip != 0) -> per board the
 * arg-max labels, the FEN piece placement before and after the reference's only live rule ("no_pawns_on_ends": a pawn on
 * rank 1 or 8 becomes the most probable non-pawn class).  Replaces, for a whole job, process_position_probabilities /
 * validate_position / board_fen (core.py:309-355, 441-469).  fen, original_fen: n_boards x 72 chars (NUL-terminated);
 * labels: n_boards x 64 validated class indices (order of constants.LABEL_NAMES = "BKNPQRbknpqrf"); fixes: capacity
 * n_boards x 16 records of {board, square index, original class, corrected class}; *n_fixes = records written.
 * Needs no GPU and no engine. */
int cv_decode_positions(const float* probs, int n_boards, int flip, char* fen, char* original_fen, int8_t* labels,
                        int32_t* fixes, int32_t* n_fixes);

/* ---- one image, host to host: the native form of ChessVision.process_image (core.py:152-195) ---------------------------------- */
/* Everything a cgo / JNI / C++ host needs for the reference's per-image entry point in ONE call: INTER_AREA resize to 256x256
 * (core.py:212), UNet forward, sigmoid / threshold mask (core.py:273, utils.py:101-112), contours -> quadrangle (core.py:357-411),
 * height-only scaling (core.py:413-417), perspective warp + gray + flip (utils.py:115-132, core.py:298-300), the 64-way split
 * (core.py:419-439), classifier forward + soft-max (core.py:236-242), arg-max, pawn rule and both FENs (core.py:309-355,
 * 441-469).  Same device stages, in the same order, as the Python class runs them; the results are bit-identical to it.
 *   unet_engine: engine with a UNet loaded; classifier_engine: engine with a ResNet-18 loaded (may be the same handle).
 *   image: HOST (h, w, 3) uint8, channels as given (BGR in the reference); flip != 0: the board is seen from black's side
 *   (square names h1..a8, constants.py:120-129); fallback_quad != 0: classify through the whole-image quadrangle when the mask
 *   yields none (not a reference behaviour: random-init weights in tests and benchmarks).
 *   out: HOST, caller-owned; every pointer may be NULL except the struct itself.  found = 0: no quadrangle, board / probabilities /
 *   FEN fields are left untouched.
 * The host waits twice (an event behind the UNet, `stream` itself at the end); the mask and the probabilities are written by their
 * kernels straight into the engine's page-locked block, the board travels on an engine-owned side stream that is joined into `stream`
 * before the final wait.  Staging buffers (page-locked host memory and device memory for the image and the results) belong to
 * unet_engine and are reused across calls: one call at a time per extractor engine (serialised internally). */
typedef struct cv_image_result {
    float*   logits;          /* 256 x 256 float32: BoardExtractionResult.probabilities (raw logits, core.py:287,306) */
    uint8_t* mask;            /* 256 x 256 uint8 0 / 255 */
    float    quadrangle[8];   /* 4 x (x, y) in image pixels, reference vertex order (TR, TL, BL, BR) */
    int32_t  found;           /* 1: a quadrangle was found (or taken as fallback) and the position fields are valid */
    uint8_t* board;           /* 512 x 512 uint8: the rectified, gray, flipped board image */
    float*   probabilities;   /* 64 x 13 float32 soft-max, squares in a8..h1 order */
    int8_t*  labels;          /* 64 validated class indices (order "BKNPQRbknpqrf") */
    char     fen[72];         /* piece placement after the pawn rule */
    char     original_fen[72];
    int32_t  fixes[16 * 4];   /* {0, square index, original class, corrected class} */
    int32_t  n_fixes;
    uint8_t* squares;         /* 64 x 64 x 64 uint8 or NULL: PositionResult.squares, the board cut into its squares a8..h1 (the
                                 reference's extract_squares, utils.py:115-132: tile (r, c) = board[64 r .. 64 r + 63][64 c .. 64 c + 63]) */
} cv_image_result_t;
/* The struct grew once (ABI 5 appended `squares`) and may grow again, so the entry point takes the size of the struct AS THE CALLER
 * COMPILED IT: fields at or beyond `out_size` are neither read nor written.  Pass sizeof(cv_image_result_t).  CV_ERR_INVALID when
 * out_size does not even cover the ABI 3 fields (through n_fixes). */
int cv_process_image_v2(cv_engine_t* unet_engine, cv_engine_t* classifier_engine, const uint8_t* image, int h, int w,
                        float threshold, int flip, int fallback_quad, cv_image_result_t* out, size_t out_size, void* stream);
/* ABI 3 / 4 entry point, kept for binaries built against the struct WITHOUT `squares`: identical to cv_process_image_v2 with out_size =
 * offsetof(cv_image_result_t, squares), i.e. `squares` is never touched whatever the caller's struct holds there. */
int cv_process_image(cv_engine_t* unet_engine, cv_engine_t* classifier_engine, const uint8_t* image, int h, int w,
                     float threshold, int flip, int fallback_quad, cv_image_result_t* out, void* stream);

/* MFMA lane-map self test: computes D = A(16xK) * B(Kx16) with the kernels' fragment loaders for both
 * precisions and returns the max abs error against a host reference (used by tests; 0 expected). */
int cv_selftest_mfma(cv_engine_t* eng, float* max_err_f16, float* max_err_f32);

#ifdef __cplusplus
}
#endif
#endif /* CHESSVISION_HIP_H */
