// engine.h -- host side of libchessvision_hip.so: weight packing, workspace, layer plans.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <utility>
#include <vector>

#include "conv_igemm.h"
#include "cv_kernels.h"

namespace cv {

bool& capture_flag();                               // this thread is recording a forward pass into a hipGraph: nothing may allocate,
                                                    // copy synchronously or synchronise (such paths fail, the caller re-runs eagerly)
// Legacy-stream operations (synchronous copies / fills, whole-device synchronisation) against graph capture: the runtime refuses a
// legacy-stream operation while ANY stream of the process records a graph ("operation would make the legacy stream depend on a
// capturing blocking stream" -- seen with eight request slots, one thread capturing its second call while another uploaded the offset
// tables of its first).  Every such call of this library goes through these helpers, which hold capture_mutex() shared; a capture
// (Engine::run_graphed) takes it exclusively, by try_lock only -- a request thread never waits for somebody's model load, it runs
// eagerly once more and captures next time.
std::shared_mutex& capture_mutex();
// ... and against EACH OTHER: synchronous copies to or from pageable memory, fills and allocations of two threads were seen to end in
// GPU memory access faults when they overlapped (round 6 soak: a model load beside another engine's lazily created offset tables;
// the runtime stages such copies through shared buffers).  They are short and rare on the request path: one at a time, process-wide.
std::mutex& legacy_mutex();
// ... and graph DESTRUCTION against graph LAUNCHES of other threads: closing an instance (hipGraphExecDestroy of its captured forwards)
// while the request threads of another instance replayed theirs faulted in ~10 % of the soak runs -- and in none with CV_GRAPH=0 or with
// the instances kept alive.  Launches hold graph_mutex() shared; a graph that is no longer needed is BURIED (graph_bury) and destroyed
// by graph_drain() at a moment when no launch is in flight (exclusive try_lock: after a bury and after every capture), never waited for.
std::shared_mutex& graph_mutex();
void graph_bury(hipGraphExec_t exec, hipGraph_t graph);
void graph_drain();
hipError_t copy_to_host_sync(void* dst_pageable, const void* src_dev, size_t n, hipStream_t s);   // D2H into pageable memory + wait for `s`
hipError_t sync_memcpy(void* dst, const void* src, size_t n, hipMemcpyKind kind);
hipError_t sync_memset(void* dst, int value, size_t n);
hipError_t device_synchronize();
void set_error(const std::string& msg);
const char* get_error();

struct Status {
    int code = 0;
    bool ok() const { return code == 0; }
};
Status fail(int code, const std::string& msg);
Status hip_fail(hipError_t e, const char* what);

#define CV_HIP(expr)                                          \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return ::cv::hip_fail(_e, #expr); \
    } while (0)
#define CV_TRY(expr)                   \
    do {                               \
        ::cv::Status _s = (expr);      \
        if (!_s.ok()) return _s;       \
    } while (0)

// state dict view: name -> (host pointer, shape)
struct ParamView {
    const float* data = nullptr;
    std::vector<int64_t> shape;
    size_t numel() const { size_t n = 1; for (auto d : shape) n *= (size_t)d; return n; }
};
typedef std::map<std::string, ParamView> ParamMap;

// one device allocation, zero-filled once (PHWC borders rely on it)
// Device (and page-locked host) blocks released by a closing instance go to a process-wide, size-keyed cache instead of back to the
// driver, and the next load takes them from there (block_alloc / block_release).  Why: with request threads replaying forwards, an
// instance closed by ANOTHER thread (hipFree of its weights and workspaces while kernels of the live instances run) ended ~1 soak in
// 5 with "Memory access fault by GPU" -- never with the instances kept alive (tests/dev/slots_soak.py, profiles/r06_tuning.md
// section 8).  A serving process that reloads models therefore never unmaps device memory under running kernels; what the cache holds
// is bounded (CV_MEM_CACHE_MB, default 16384; beyond it blocks are freed, behind a device synchronise) and cv_trim_memory() hands
// everything back.  CV_MEM_CACHE=0: every release is a hipFree again.
hipError_t block_alloc(void** ptr, size_t n, size_t* cap, bool host);
void block_release(void* ptr, size_t cap, bool host);
size_t block_cache_trim();                              // frees every cached block; returns the bytes handed back
// block_release waits for the device before a block becomes somebody else's (what hipFree did implicitly).  A caller that has just
// synchronised and releases many blocks of a quiescent owner (cv_engine_destroy: ~100 buffers) holds one of these meanwhile.
struct ReleaseAlreadySynced {
    ReleaseAlreadySynced();
    ~ReleaseAlreadySynced();
    ReleaseAlreadySynced(const ReleaseAlreadySynced&) = delete;
    ReleaseAlreadySynced& operator=(const ReleaseAlreadySynced&) = delete;
};

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    size_t cap_bytes = 0;                               // size of the underlying block (>= bytes when it came out of the cache)
    // CV_GUARD_ALLOC=1|2 (debugging): the buffer is placed at the END (1) or START (2) of its own virtual-memory mapping with an
    // unmapped granule on either side, so that a kernel reading or writing past that side of a buffer faults at once instead of
    // landing in whatever allocation happens to be its neighbour (tests/dev/guard_alloc.sh).  For FAULTS only by default: when a freed
    // range's addresses are reserved again, kernels still see the previous mapping on this stack, so numbers computed under this mode
    // are only right with CV_GUARD_KEEP_VA=1, which small cases can afford (engine.cpp: DeviceBuffer::release; r06_tuning.md section 8).
    void* guard_base = nullptr;
    size_t guard_span = 0, guard_mapped = 0;
    void* guard_handle = nullptr;
    void release();
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer();
    Status alloc(size_t n, bool zero);
    Status upload(const void* host, size_t n);
    void swap(DeviceBuffer& o) {
        std::swap(ptr, o.ptr); std::swap(bytes, o.bytes); std::swap(cap_bytes, o.cap_bytes);
        std::swap(guard_base, o.guard_base); std::swap(guard_span, o.guard_span); std::swap(guard_mapped, o.guard_mapped); std::swap(guard_handle, o.guard_handle);
    }
};

// a PHWC activation buffer sized for `cap` images
//
// Range scaling (f16 / split-f16 engines): the buffer holds real_value * 2^-exp.  f16 carries 5 exponent bits, and the lo
// half of a split-f16 value goes subnormal below 2^-14, so a tensor is only f32-grade while its magnitudes sit in roughly
// [2^-3, 2^15].  `exp` is chosen per tensor at load time by a calibration pass (models: calibrate()) so that the largest
// magnitude seen lands in [16, 32): 2^11 of head-room before the f16 range ends, full lo precision down to 2^-7 of the
// maximum, absolute error 2^-25 below that.  The factors are powers of two and are folded into the f32 epilogue
// scale/shift of the producing layer and of every consumer, so they cost nothing and change no rounding of the real
// values; what still leaves the range trips the numeric guard (ConvParams::flag).  Always 0 for the f32 engine.
struct Activation {
    DeviceBuffer buf;
    // f16r engine: tensors of the ResNet trunk (block outputs, shortcuts, the pooled stem output) keep an unrounded f32 twin
    // with the same geometry and the same exponent; residual adds read it and the producing epilogue writes it next to the
    // f16 copy the next convolution consumes
    DeviceBuffer buf32;
    bool want32 = false;
    bool only32 = false;                            // ... and no f16 copy at all (shortcut tensors: nothing but the residual add reads them)
    int cap = 0, H = 0, W = 0, C = 0;
    int dt = kF16;
    int exp = 0;
    // A concatenated buffer (UNet decoder: [skip | up-sampled]) keeps a second exponent for channels >= split_c: the two halves
    // come from different producers and can differ by orders of magnitude, and one shared exponent would push the smaller
    // half (and, after the consumer's row normalisation, its weights) into the f16 subnormals.
    int split_c = 0;                                // 0 = not a concatenation
    int exp2 = 0;
    // calibration statistics of the current pass (real-valued maximum = stored maximum * 2^exp) and policy
    float seen_max = 0.f, seen_max2 = 0.f;
    bool seen_bad = false, seen_bad2 = false;
    bool fixed_exp = false;                         // inputs: exp is set by construction, not measured
    Activation* tie = nullptr;                      // same exp as `tie` (pooled copy of a tensor, concat halves)
    void shape(int h, int w, int c, int dt_) { H = h; W = w; C = c; dt = dt_; }
    Status reserve(int cap_);                       // (re)allocate for cap_ images, zero-filled; exp and shape stay
    // Transactional growth of a whole workspace: every new buffer is allocated before any old one is released, so a failed
    // hipMalloc leaves the model exactly as it was (capacity, pointers, borders) instead of half-grown with a null tensor.
    static Status reserve_all(const std::vector<Activation*>& acts, int cap_);
    Status create(int cap_, int h, int w, int c, int dt_) { shape(h, w, c, dt_); return reserve(cap_); }
    TensorRef ref(int n, int coff = 0, int c = -1) const {
        TensorRef t;
        t.base = buf.ptr; t.base32 = buf32.ptr; t.N = n; t.H = H; t.W = W; t.Cs = C; t.Coff = coff; t.C = c < 0 ? C - coff : c;
        t.owner = const_cast<Activation*>(this);
        t.exp = exp_of(coff); t.split = 0; t.exp_delta = 0; t.f32_only = 0;
        if (split_c && coff < split_c && coff + t.C > split_c) { t.split = split_c - coff; t.exp_delta = exp2 - exp; }
        return t;
    }
    // the f32 twin as a tensor of its own (input / output of layers that run in f32 inside an f16r engine)
    TensorRef ref32(int n) const {
        TensorRef t = ref(n);
        t.base = buf32.ptr; t.base32 = nullptr; t.f32_only = 1;
        return t;
    }
    int exp_of(int coff) const { return split_c && coff >= split_c ? exp2 : exp; }
    size_t bytes_per_image() const { return (size_t)(H + 2) * (W + 2) * C * dtype_size(dt); }
};

// a packed implicit-GEMM layer (conv k x k, or k2 s2 transposed conv as 1-tap GEMM + pixel shuffle)
struct ConvLayer {
    std::string name;
    int cin = 0, cinPad = 0, cout = 0, k = 1, stride = 1;
    bool shuffle = false;
    int rows = 0, rowsPad = 0, nStages = 0, nCt = 0;
    int ct = 64;                                    // channel-tile height the weights are packed for (64 | 128)
    bool halo_ok = false;                           // 3x3 s1 on a 16-divisible square grid: conv_halo.hip tiles (64 | 128 rows)
    bool halo_img8 = false;                         // ... on 8x8 feature maps: the packed-image mode of the 128-row tile
    int64_t pixels_hint = 0;                        // output pixels per launch at the engine's chunk size (tile choice)
    int kgroup = 8;                                 // input channels per K block (see engine.cpp: K ordering)
    int dt = kF16;
    DeviceBuffer w, scale, shift;
    DeviceBuffer w_small;                           // ct == 256 layers only: the same weights packed for 128-row tiles
    // Per-row epilogue constants on the host: h_scale already contains 2^(row exponent) of the weight normalisation
    // (f16 / split-f16: every weight row is stored as w * 2^-e with its largest magnitude in [0.5, 1), so small trained
    // weights keep all 22 bits).  The device copies are h_scale * 2^(in_exp - out_exp) and h_shift * 2^-out_exp for the
    // tensor exponents the layer currently runs with (set_exps).
    std::vector<float> h_scale, h_shift;
    int in_exp = 0, out_exp = 0;
    // consumers of a concatenated buffer: the un-normalised weights stay on the host so that 2^exp_delta can be folded into
    // the input channels of the second half when calibration gives the halves different exponents (set_input_split)
    bool keep_host_weights = false;
    std::vector<float> Wk0, b_scale, b_shift;
    int Kdim = 0, in_split = 0, in_delta = 0;
    Status set_input_split(int split, int delta, hipStream_t s);
    unsigned layer_id = 0xfffffffeu;                // numeric guard id (Engine::register_layer)
    // Rounding-bias correction of the f16 layers (round 6).  An f16 weight is w_hat = w + e with |e| <= 2^-12 |w|, and the e of a
    // layer are the same for every pixel of every image: through post-ReLU inputs (positive mean) they add a CONSTANT offset
    // sum_k e[r][k] * mean(x[k]) to output channel r that no spatial pooling averages away -- 70 % of the fp16 classifier's logit
    // error variance, against 30 % from the (independent, zero-mean) roundings of the activations.  At load time the mean of the
    // layer's stored input under every tap is measured on the calibration batch (Engine::measure_tap_sums), and the offset goes into
    // the f32 epilogue shift with the opposite sign... i.e. shift += scale * sum_k (w - w_hat)[r][k] * mean(x[k]).  Exact arithmetic is
    // untouched (the term is what exact weights would have added on the calibration mean), no launch changes.  CV_BIAS_CORR=0: off.
    bool want_round_err = false;                    // build_*: keep (w - w_hat) of the normalised rows until the correction is folded
    std::vector<float> h_round_err;                 // [rows][Kdim_err] in the K order of the packed rows; released after calibration
    int Kerr = 0;
    std::vector<float> h_shift_base;                // h_shift without the correction
    std::vector<double> tap_sum;                    // [Kerr] sum of the stored input under (tap, channel) of K index k, calibration batch
    double tap_count = 0;                           // output positions x images those sums cover
    int tap_in_exp = 0;                             // exponent of the input's first half while measured
    Status fold_rounding_bias();                    // h_shift = h_shift_base + correction; forces set_exps to re-upload
    // koff tables are geometry dependent: keyed by (xWp, xCs, xCoff)
    struct KoffKey { int xWp, xCs, xCoff; bool operator<(const KoffKey& o) const {
        if (xWp != o.xWp) return xWp < o.xWp; if (xCs != o.xCs) return xCs < o.xCs; return xCoff < o.xCoff; } };
    struct KoffTab { DeviceBuffer chunks, bases; bool separable = false; std::vector<int> h_bases; };
    std::map<KoffKey, std::unique_ptr<KoffTab>> koff;
    // position-major launches (ConvParams::ptab): per output position the K stages whose tap reads a real pixel; geometry dependent
    struct PosKey { int xHp, xWp, xCs, xCoff; bool operator<(const PosKey& o) const {
        if (xHp != o.xHp) return xHp < o.xHp; if (xWp != o.xWp) return xWp < o.xWp; if (xCs != o.xCs) return xCs < o.xCs; return xCoff < o.xCoff; } };
    struct PosTab { DeviceBuffer tab, count, order; double live = 1.0; };     // live = mean fraction of the stages a position walks
    std::map<PosKey, std::unique_ptr<PosTab>> pos;
    Status get_pos(const TensorRef& x, int Ho, int Wo, const int** tab, const int** count, const int** order, double* live);

    // w_oihw: (cout, cin, k, k); scale/shift: (cout)
    Status build_conv(const std::string& name_, int dt_, const float* w_oihw, int cout_, int cin_, int k_,
                      int stride_, const float* scale_, const float* shift_, int cinPad_, int64_t pixels_hint_ = 0,
                      int out_hw_ = 0);
    // w_iohw: (cin, cout, 2, 2); bias: (cout)
    Status build_convT(const std::string& name_, int dt_, const float* w_iohw, int cin_, int cout_,
                       const float* bias, int64_t pixels_hint_ = 0);
    Status get_koff(const TensorRef& x, const int** chunks, const int** bases);
    Status set_exps(int in_exp_, int out_exp_, hipStream_t s);
    int64_t macs_per_out_pixel() const { return shuffle ? (int64_t)cin * cout * 4 : (int64_t)cin * k * k * cout; }
};

struct ProfileEntry {
    std::string name;
    std::string kernel;                             // which kernel instantiation ran the launch (conv family: tile / variant tag)
    bool is_conv = false;
    double macs = 0;
    double bytes = 0;                               // algorithmic HBM bytes of the launch (inputs + outputs + weights, once each)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
};

class Engine {
  public:
    int device = 0;
    int dt = kF16;
    bool trunk32 = false;         // precision f16r: f16 MFMA convolutions, the ResNet residual trunk carried in f32 (Activation::buf32)
    std::mutex mu;
    int unet_chunk = 64;          // images per pass: every layer still launches >= 256 workgroups of 256x256 / 128x256
    int resnet_chunk = 16384;      // squares per pass

    struct UNet;
    struct ResNet;
    std::unique_ptr<UNet> unet;
    std::unique_ptr<ResNet> resnet;

    DeviceBuffer scratch;                               // small per-call parameter blocks (homographies)
    // f32 partial sums of split-K conv launches (grown on demand).  One buffer per model: the UNet pass and the ResNet-18 pass of one
    // engine may be enqueued on two streams (bench.py --overlap 1) and must not share scratch memory; ws_slot is set by the model's
    // forward (0 = UNet and the single-layer entry points, 1 = ResNet-18) under the engine mutex.
    DeviceBuffer splitk_ws[2];
    int ws_slot = 0;
    // One workspace per model: two forwards of the SAME model must not overlap on the device.  The engine mutex orders their enqueue on
    // the host only, so a forward that arrives on another stream than the model's previous one first waits (on the host) for that stream
    // -- rare by construction (a request slot's private stream beside process_images on the caller's stream, round 6: without this
    // the B = 8 forward of a batch overwrote the activations of a B = 1 forward still in flight on the slot's stream), free otherwise.
    hipStream_t last_stream[2] = {nullptr, nullptr};
    bool last_stream_set[2] = {false, false};
    Status order_forward(int model, hipStream_t s);
    // staging of cv_process_image: one page-locked host block and one device block, carved up per call (grown on demand)
    void* pipe_host = nullptr;
    size_t pipe_host_bytes = 0, pipe_host_cap = 0;
    DeviceBuffer pipe_dev;
    std::mutex pipe_mu;                                 // one cv_process_image at a time per extractor engine (the staging is shared)
    hipEvent_t pipe_event = nullptr;
    hipEvent_t pipe_event2 = nullptr;               // cv_process_image: "the rectified board exists" -> its download on pipe_side beside the classifier
    hipStream_t pipe_side = nullptr;
    hipEvent_t pipe_event3 = nullptr;               // cv_process_image: "the board is home" (side stream) -> joined into the caller's stream on the device
    DeviceBuffer area_tabs;                             // INTER_AREA tables of the last fractional resize geometry (cv_resize_area_u8)
    long long area_key = -1;
    size_t area_off[6] = {0, 0, 0, 0, 0, 0};

    // numeric guard: one device word, 0xffffffff = clean, else the lowest id of a layer that stored a non-finite value
    DeviceBuffer guard;
    std::vector<std::string> layer_names;               // id -> name; id 0 = the caller's input tensor
    unsigned register_layer(const std::string& name);
    unsigned* guard_ptr() const { return reinterpret_cast<unsigned*>(guard.ptr); }
    Status guard_init();
    Status guard_check(hipStream_t s);                  // synchronises; fails with the layer name and re-arms
    Status guard_read_async(unsigned* pinned, hipStream_t s);   // ... in two halves: enqueue the download into page-locked memory,
    Status guard_eval(unsigned v);                      // ... and judge the word after the caller's own synchronisation

    // range calibration (see Activation): while set, every producer measures its output tensor after the launch
    bool calibrating = false;
    DeviceBuffer cal_word;
    Status measure(const TensorRef& t, hipStream_t s);   // t.f32_only: read as f32
    // rounding-bias calibration pass (ConvLayer::want_round_err): while set, run_conv accumulates the tap sums of every f16 layer's input
    bool bias_measuring = false;
    DeviceBuffer bias_ws;
    Status measure_tap_sums(ConvLayer& L, const TensorRef& x, int Ho, int Wo, hipStream_t s);
    // one pass of `forward` over the calibration batch with the tap sums switched on, then every layer of `layers` folds its correction
    template <class Fwd> Status calibrate_rounding_bias(const std::vector<ConvLayer*>& layers, Fwd&& forward, hipStream_t s);
    template <class Fwd> Status calibrate(const std::vector<Activation*>& acts, Fwd&& forward, hipStream_t s, const char* what);
    size_t workspace_bytes() const;

    // hipGraph replay of small forward passes (the single-board shape: UNet B=1 = 21-33 launches, ResNet-18 B=64 = 19-27): the launch
    // sequence of a forward that fits one chunk is captured once per (model, entry, batch, caller pointers) and replayed with one
    // hipGraphLaunch.  A captured launch holds workspace / split-K / weight pointers by value, so every cached graph carries the
    // `graph_epoch` it was captured in and anything that re-allocates or re-folds such memory bumps the epoch (workspace growth,
    // split-K buffer growth, exponent changes, offset tables created on first use).  CV_GRAPH=0 switches the replay off.
    struct GraphKey {
        int model = 0, n = 0, flags = 0;                  // model 0 = UNet, 1 = ResNet-18; flags: u8 entry / softmax
        const void* x = nullptr; void* out = nullptr; void* mask = nullptr;
        unsigned thr_bits = 0;
        bool operator==(const GraphKey& o) const {
            return model == o.model && n == o.n && flags == o.flags && x == o.x && out == o.out && mask == o.mask && thr_bits == o.thr_bits;
        }
    };
    struct GraphEntry {
        GraphKey key;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;                  // null: the key was seen once (its first call ran eagerly and warmed lazy state)
        uint64_t epoch = 0, last_use = 0, hits = 0;
    };
    std::vector<GraphEntry> graphs;
    uint64_t graph_epoch = 0, graph_clock = 0;
    int graph_wasted = 0;                               // captures evicted without a single replay: callers whose pointers never repeat
    bool graphs_on = true;
    hipStream_t capture_stream = nullptr;
    void graph_invalidate() { ++graph_epoch; }
    void graph_clear();
    template <class Run> Status run_graphed(const GraphKey& key, hipStream_t s, Run&& run);

    // profiling (cv_profile_convs)
    bool profiling = false;
    std::vector<ProfileEntry> prof;

    Engine();
    ~Engine();

    struct Head { const float* w; const float* b; float* logits; uint8_t* mask; float thr; };
    // fused producer of the layer's input (ConvParams::f0_*): the network's first conv computed inside this layer's kernel
    struct Fuse0 { const void* x; bool u8; const void* w; const float* scale; const float* shift; float in_mul; double macs_per_pixel; };
    static constexpr int kNotFused = 100;           // run_conv's status code when `fuse0` was requested but the launch cannot take it
    // pool_out: also produce max_pool2d(y, 2) -- fused into the conv epilogue when the halo kernel runs the layer,
    // otherwise by the stand-alone pooling kernel right after it
    Status run_conv(ConvLayer& L, const TensorRef& x, const TensorRef& y, const TensorRef* res, bool relu,
                    hipStream_t s, const Head* head = nullptr, const TensorRef* pool_out = nullptr, const Fuse0* fuse0 = nullptr);
    // Two INDEPENDENT layers in one launch (conv_igemm_pair_kernel; round 5, single boards).  A caller that is about to run two
    // layers with no dependency between them points `defer` at a PendingConv before each run_conv: a launch that qualifies (generic
    // kernel, table-free offsets, no profiling / calibration / stand-alone pool behind it; the FIRST of the two also
    // unsplit, so that the split-K scratch has one user) is then prepared but not issued, and flush_pending issues the pair as one
    // launch when both were held with the same tile configuration and their grids TOGETHER stay under CV_PAIR_MAX_BLOCKS (512: the
    // pair is for launches that leave the chip unfilled) -- otherwise whatever was held, one after the other.
    struct PendingConv { bool held = false; int cfg = 0, ns = 0, dt = 0; long long blocks = 0; ConvParams p; std::string name; };
    PendingConv* defer = nullptr;
    bool defer_first = false;
    Status flush_pending(PendingConv& a, PendingConv& b, hipStream_t s);
    void prof_begin(const std::string& name, bool is_conv, double macs, hipStream_t s, double bytes = 0);
    void prof_end(hipStream_t s);
    Status prof_collect();
    void prof_clear();
};

bool calibration_enabled();                         // CV_CALIBRATE=0 switches the activation exponents off (tests of the guard)
bool bias_correction_enabled();                     // CV_BIAS_CORR=0 switches the rounding-bias correction of the f16 layers off
int choose_ct(int rows, int64_t pixels_hint, bool halo_ok, bool img8);
int choose_cfg(int ct, int rows, int64_t pixels, int n_stages);
int choose_ns(int cfg, int dt, int rows, int64_t pixels, int n_stages);

Status unet_load(Engine& e, const ParamMap& pm);
Status unet_forward(Engine& e, const void* x, bool x_u8, int batch, float* logits, uint8_t* mask, float thr,
                    hipStream_t s);
Status unet_activation(Engine& e, const std::string& name, TensorRef* out);
int64_t unet_macs(Engine& e);

Status resnet_load(Engine& e, const ParamMap& pm);
Status resnet_forward(Engine& e, const void* x, bool x_u8, int n, float* out, bool softmax, hipStream_t s);
Status resnet_activation(Engine& e, const std::string& name, TensorRef* out);
int64_t resnet_macs(Engine& e);

// Iterate calibration passes until every tensor exponent is stable (hysteresis: an exponent stays while the measured
// maximum is inside [8, 64) of stored units).  `forward` runs one forward over the calibration batch with e.calibrating set.
template <class Fwd>
Status Engine::calibrate(const std::vector<Activation*>& acts, Fwd&& forward, hipStream_t s, const char* what) {
    if (dt == kF32 || !calibration_enabled()) return Status();
    for (int pass = 0; pass < 10; ++pass) {
        for (Activation* a : acts) { a->seen_max = a->seen_max2 = 0.f; a->seen_bad = a->seen_bad2 = false; }
        calibrating = true;
        Status st = forward();
        calibrating = false;
        if (!st.ok()) return st;
        CV_HIP(hipStreamSynchronize(s));
        CV_HIP(sync_memset(guard.ptr, 0xff, sizeof(unsigned)));      // overflow during calibration is expected, not an error
        bool changed = false;
        auto adjust = [&](int& ex, float seen, bool bad) {
            int want = ex;
            if (bad) want = ex + 12;
            else if (seen > 0.f) {
                int e2;
                (void)std::frexp(seen, &e2);                       // seen = f * 2^e2, f in [0.5, 1)
                const int stored_e = e2 - ex;                      // stored maximum in [2^(stored_e-1), 2^stored_e)
                if (stored_e < 4 || stored_e > 6) want = e2 - 5;   // re-centre on [16, 32)
            }
            want = want < -60 ? -60 : want > 60 ? 60 : want;
            if (want != ex) { ex = want; changed = true; }
        };
        for (Activation* a : acts) {
            if (a->fixed_exp || a->tie) continue;
            adjust(a->exp, a->seen_max, a->seen_bad);
            if (a->split_c) adjust(a->exp2, a->seen_max2, a->seen_bad2);
        }
        for (Activation* a : acts)
            if (a->tie && a->exp != a->tie->exp) { a->exp = a->tie->exp; changed = true; }
        if (!changed) return Status();
    }
    return fail(1, std::string(what) + ": activation range calibration did not converge (non-finite weights or activations beyond f32?)");
}

template <class Fwd>
Status Engine::calibrate_rounding_bias(const std::vector<ConvLayer*>& layers, Fwd&& forward, hipStream_t s) {
    if (dt != kF16 || !calibration_enabled() || !bias_correction_enabled()) return Status();
    for (ConvLayer* L : layers) { L->tap_sum.clear(); L->tap_count = 0; }
    calibrating = true;                                 // the layer-by-layer schedule: every f16 layer's input exists in memory
    bias_measuring = true;
    Status st = forward();
    bias_measuring = false;
    calibrating = false;
    if (!st.ok()) return st;
    CV_HIP(hipStreamSynchronize(s));
    CV_HIP(sync_memset(guard.ptr, 0xff, sizeof(unsigned)));
    for (ConvLayer* L : layers) {
        CV_TRY(L->fold_rounding_bias());
        std::vector<float>().swap(L->h_round_err);      // 4 bytes per weight of host memory: not needed again
    }
    return Status();
}

// Replay `run` (a launch sequence on the stream it is handed; no allocation, no synchronisation once warmed) through a cached graph.
template <class Run>
Status Engine::run_graphed(const GraphKey& key, hipStream_t s, Run&& run) {
    if (!graphs_on || profiling || calibrating) return run(s);
    ++graph_clock;
    GraphEntry* hit = nullptr;
    for (auto& g : graphs)
        if (g.key == key) { hit = &g; break; }
    if (hit && hit->exec && hit->epoch == graph_epoch) {
        hit->last_use = graph_clock; ++hit->hits;
        std::shared_lock<std::shared_mutex> launching(graph_mutex());
        CV_HIP(hipGraphLaunch(hit->exec, s));
        return Status();
    }
    if (!hit) {                                         // first sight of this call shape: run eagerly (creates offset tables, grows buffers)
        if (graphs.size() >= 8) {
            size_t victim = 0;
            for (size_t i = 1; i < graphs.size(); ++i)
                if (graphs[i].last_use < graphs[victim].last_use) victim = i;
            if (graphs[victim].exec && graphs[victim].hits == 0 && ++graph_wasted >= 4) graphs_on = false;   // pointers never repeat: stop capturing
            graph_bury(graphs[victim].exec, graphs[victim].graph);
            graphs.erase(graphs.begin() + (long)victim);
        }
        GraphEntry ge;
        ge.key = key; ge.last_use = graph_clock;
        graphs.push_back(ge);
        return run(s);
    }
    // second sight (or a stale capture): record the sequence on the engine's own stream, instantiate, replay on the caller's
    graph_bury(hit->exec, hit->graph);
    hit->exec = nullptr; hit->graph = nullptr;
    if (!capture_stream && hipStreamCreateWithFlags(&capture_stream, hipStreamNonBlocking) != hipSuccess) { graphs_on = false; return run(s); }
    const uint64_t epoch0 = graph_epoch;
    std::unique_lock<std::shared_mutex> capture_lock(capture_mutex(), std::try_to_lock);
    if (!capture_lock.owns_lock()) return run(s);       // somebody is loading a model / growing a workspace: capture on a later call
    if (hipStreamBeginCapture(capture_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); graphs_on = false; return run(s); }
    capture_flag() = true;
    Status st = run(capture_stream);
    capture_flag() = false;
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(capture_stream, &graph);
    capture_lock.unlock();
    if (!st.ok() || ce != hipSuccess || !graph || graph_epoch != epoch0) {
        graph_bury(nullptr, graph);
        (void)hipGetLastError();
        hit->last_use = graph_clock;
        if (st.ok() && graph_epoch == epoch0 && ce != hipSuccess) graphs_on = false;          // the runtime cannot capture this sequence at all
        return run(s);                                  // nothing was executed by the failed capture: run it for real
    }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || !exec) {
        graph_bury(nullptr, graph);
        (void)hipGetLastError();
        graphs_on = false;
        return run(s);
    }
    hit->graph = graph; hit->exec = exec; hit->epoch = graph_epoch; hit->last_use = graph_clock; hit->hits = 0;
    graph_drain();
    std::shared_lock<std::shared_mutex> launching(graph_mutex());
    CV_HIP(hipGraphLaunch(exec, s));
    return Status();
}

}  // namespace cv
