// cv_api.cpp -- the extern "C" surface declared in include/chessvision_hip.h.
#include "../../include/chessvision_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <stdexcept>
#include <thread>

#include "engine.h"
#include "models.h"
#include "pointwise.h"

namespace cv {
void decode_positions(const float* probs, int n_boards, int flip, char* fen, char* original_fen, int8_t* labels,
                      int32_t* fixes, int32_t* n_fixes);
bool find_quadrangle(const uint8_t* mask, int h, int w, int32_t quad[8]);
long find_contours_flat(const uint8_t* mask, int h, int w, bool tc89, int32_t* xy, long cap_pts, int32_t* counts, int32_t* holes,
                        long cap_contours);
hipError_t resize_area_u8(const uint8_t* src, int n, int h, int w, int c, uint8_t* dst, int oh, int ow, hipStream_t s);
void resize_area_table(int ssize, int dsize, std::vector<int>& ofs, std::vector<int>& si, std::vector<float>& alpha);
hipError_t resize_area_u8_tab(const uint8_t* src, int n, int h, int w, int c, uint8_t* dst, int oh, int ow, const int* xofs,
                              const int* xsi, const float* xa, const int* yofs, const int* ysi, const float* ya, hipStream_t s);
hipError_t extract_squares_u8(const uint8_t* images, int n, int h, int w, const double* inv, uint8_t* squares,
                              uint8_t* boards, hipStream_t s);
hipError_t extract_squares_u8_one(const uint8_t* image, int h, int w, const double* inv_host, uint8_t* squares, uint8_t* board, hipStream_t s);
void board_homographies(const float* quads, int n, int out_w, int out_h, double* forward, double* inverse);
}  // namespace cv

using namespace cv;

struct cv_engine {
    Engine impl;
};

namespace {

struct DeviceGuard {                       // every entry point runs on the engine's device
    int prev = -1;
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int finish(const Status& s) { return s.code; }

// No C++ exception may unwind through the C boundary (ctypes / cgo callers would abort): every entry point runs inside
// this wrapper.  The out-of-memory message fits the small-string buffer, so reporting it allocates nothing.
template <class F> int guarded(const char* what, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try { set_error("out of memory"); } catch (...) {}
        return CV_ERR_NOMEM;
    } catch (const std::length_error&) {
        try { set_error("out of memory"); } catch (...) {}
        return CV_ERR_NOMEM;
    } catch (const std::exception& ex) {
        try { set_error(std::string(what) + ": " + ex.what()); } catch (...) {}
        return CV_ERR_INVALID;
    } catch (...) {
        try { set_error("unknown C++ exception"); } catch (...) {}
        return CV_ERR_INVALID;
    }
}

Status to_map(const cv_param_t* params, int n, ParamMap& pm) {
    if (n < 0 || (n > 0 && !params)) return fail(CV_ERR_INVALID, "null parameter table");
    for (int i = 0; i < n; ++i) {
        const cv_param_t& p = params[i];
        if (!p.name) return fail(CV_ERR_INVALID, "malformed cv_param_t entry " + std::to_string(i) + " (null name)");
        const size_t nl = std::strlen(p.name);
        if (nl >= 19 && std::strcmp(p.name + nl - 19, "num_batches_tracked") == 0) continue;   // a full torch state dict may carry them
                                                                                              // (int64 scalars: data / shape are not looked at)
        if (!p.data || p.ndim < 0 || p.ndim > 4) return fail(CV_ERR_INVALID, std::string("malformed cv_param_t entry '") + p.name + "'");
        ParamView v;
        v.data = p.data;
        for (int d = 0; d < p.ndim; ++d) v.shape.push_back(p.shape[d]);
        pm[p.name] = v;
    }
    return Status();
}

Status check_engine(cv_engine_t* eng) {
    if (!eng) return fail(CV_ERR_INVALID, "null engine handle");
    return Status();
}

}  // namespace

// ---- implementations (may throw: std::vector / std::map / std::string allocations) ------------------------------

static int impl_cv_device_count(int* count) {
    if (!count) return finish(fail(CV_ERR_INVALID, "null count"));
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return finish(hip_fail(e, "hipGetDeviceCount")); }
    return CV_OK;
}

static int impl_cv_engine_create(int device, int precision, cv_engine_t** out) {
    if (!out) return finish(fail(CV_ERR_INVALID, "null out pointer"));
    *out = nullptr;
    if (precision != CV_PREC_F32 && precision != CV_PREC_F16 && precision != CV_PREC_F16X3 && precision != CV_PREC_F16R)
        return finish(fail(CV_ERR_INVALID, "unknown precision"));
    static_assert((int)CV_PREC_F32 == (int)kF32 && (int)CV_PREC_F16 == (int)kF16 && (int)CV_PREC_F16X3 == (int)kSplit, "enum values must match");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return finish(fail(CV_ERR_HIP, std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0")));
    if (device < 0 || device >= count) return finish(fail(CV_ERR_INVALID, "device ordinal out of range"));
    DeviceGuard g(device);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return finish(hip_fail(e, "hipGetDeviceProperties"));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return finish(fail(CV_ERR_STATE, std::string("libchessvision_hip is built for gfx950 (MI355X); device is ") + prop.gcnArchName));
    {
        // the dynamic-LDS limits of the conv kernels are raised ONCE per device and process, under a lock: repeating hipFuncSetAttribute
        // at every engine creation while other threads launch those very kernels (request slots: replicas are loaded in the background
        // of a serving instance) is not something the runtime promises to survive
        static std::mutex prep_mu;
        static bool prepared[64] = {};
        std::lock_guard<std::mutex> lk(prep_mu);
        if (device >= 64 || !prepared[device]) {
            if ((e = conv_igemm_prepare()) != hipSuccess) return finish(hip_fail(e, "conv_igemm_prepare"));
            if ((e = conv_halo_prepare()) != hipSuccess) return finish(hip_fail(e, "conv_halo_prepare"));
            if (device < 64) prepared[device] = true;
        }
    }
    cv_engine* eng = new (std::nothrow) cv_engine();
    if (!eng) return finish(fail(CV_ERR_NOMEM, "out of host memory"));
    eng->impl.device = device;
    eng->impl.dt = precision == CV_PREC_F16R ? (int)kF16 : precision;   // CV_PREC_F32 / F16 / F16X3 values equal cv::DType
    eng->impl.trunk32 = precision == CV_PREC_F16R;
    Status gs = eng->impl.guard_init();
    if (!gs.ok()) { delete eng; return finish(gs); }
    *out = eng;
    return CV_OK;
}

static int impl_cv_engine_destroy(cv_engine_t* eng) {
    if (!eng) return CV_OK;
    {
        DeviceGuard g(eng->impl.device);                // every block of the engine goes back to the cache under ITS device
        (void)device_synchronize();
        ReleaseAlreadySynced quiescent;                 // nothing can be queued on a dying engine's buffers from here on
        eng->impl.unet.reset();
        eng->impl.resnet.reset();
        delete eng;
    }
    return CV_OK;
}

static int impl_cv_engine_set_chunk(cv_engine_t* eng, int unet_images, int resnet_squares) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    if (eng->impl.unet || eng->impl.resnet) return finish(fail(CV_ERR_STATE, "cv_engine_set_chunk must precede cv_load_*"));
    if (unet_images < 0 || resnet_squares < 0) return finish(fail(CV_ERR_INVALID, "negative chunk"));
    if (unet_images) eng->impl.unet_chunk = unet_images;
    if (resnet_squares) eng->impl.resnet_chunk = resnet_squares;
    return CV_OK;
}

// Model loads are serialised process-wide.  A load packs, uploads and calibrates through the legacy stream (hundreds of synchronous
// allocations, fills and copies, calibration forwards on stream 0); two of them side by side in one process -- a second instance created
// while another loads, or beside the background replica of a request slot -- ended in a GPU memory access fault in 50-80 % of the
// soak runs (tests/dev/slots_soak.py; forwards beside ONE load never did, in any number of runs).  Loads take seconds and happen once
// per engine: one at a time costs nothing.
static std::mutex& load_mutex() {
    static std::mutex mu;
    return mu;
}

static int impl_cv_load_unet(cv_engine_t* eng, const cv_param_t* params, int n_params) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    std::lock_guard<std::mutex> load_lk(load_mutex());
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    ParamMap pm;
    s = to_map(params, n_params, pm);
    if (s.ok()) s = unet_load(eng->impl, pm);
    return finish(s);
}

static int impl_cv_load_resnet18(cv_engine_t* eng, const cv_param_t* params, int n_params) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    std::lock_guard<std::mutex> load_lk(load_mutex());
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    ParamMap pm;
    s = to_map(params, n_params, pm);
    if (s.ok()) s = resnet_load(eng->impl, pm);
    return finish(s);
}

static int impl_cv_unet_forward(cv_engine_t* eng, const float* x, int batch, float* logits, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    return finish(unet_forward(eng->impl, x, false, batch, logits, nullptr, 0.5f, (hipStream_t)stream));
}

static int impl_cv_unet_forward_u8(cv_engine_t* eng, const uint8_t* x_u8, int batch, float* logits, uint8_t* mask,
                       float threshold, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!(threshold >= 0.f && threshold <= 1.f)) return finish(fail(CV_ERR_INVALID, "threshold must be between 0 and 1"));
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    return finish(unet_forward(eng->impl, x_u8, true, batch, logits, mask, threshold, (hipStream_t)stream));
}

static int impl_cv_resnet18_forward(cv_engine_t* eng, const float* x, int n, float* logits, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    return finish(resnet_forward(eng->impl, x, false, n, logits, false, (hipStream_t)stream));
}

static int impl_cv_resnet18_forward_u8(cv_engine_t* eng, const uint8_t* squares_u8, int n, float* probs, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    return finish(resnet_forward(eng->impl, squares_u8, true, n, probs, true, (hipStream_t)stream));
}

static int impl_cv_softmax13(cv_engine_t* eng, const float* logits, int n, float* probs, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (n < 0 || (n > 0 && (!logits || !probs))) return finish(fail(CV_ERR_INVALID, "cv_softmax13: null tensor"));
    if (n == 0) return CV_OK;
    DeviceGuard g(eng->impl.device);
    hipError_t e = softmax13(logits, n, probs, (hipStream_t)stream);
    if (e != hipSuccess) return finish(hip_fail(e, "softmax13"));
    return CV_OK;
}

static int impl_cv_get_activation(cv_engine_t* eng, const char* model, const char* name, float* out_host, size_t out_capacity,
                      int64_t dims[4]) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!model || !name || !dims) return finish(fail(CV_ERR_INVALID, "null argument"));
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    TensorRef t;
    if (std::strcmp(model, "unet") == 0) s = unet_activation(eng->impl, name, &t);
    else if (std::strcmp(model, "resnet18") == 0) s = resnet_activation(eng->impl, name, &t);
    else s = fail(CV_ERR_INVALID, "model must be 'unet' or 'resnet18'");
    if (!s.ok()) return finish(s);
    dims[0] = t.N; dims[1] = t.C; dims[2] = t.H; dims[3] = t.W;
    const size_t numel = (size_t)t.N * t.C * t.H * t.W;
    if (!out_host) return CV_OK;                          // shape query
    if (out_capacity < numel) return finish(fail(CV_ERR_INVALID, "output buffer too small"));
    DeviceBuffer tmp;
    s = tmp.alloc(numel * sizeof(float), false);
    if (!s.ok()) return finish(s);
    hipError_t e = device_synchronize();
    if (e == hipSuccess) e = unpack_nchw_f32(t.f32_only ? (int)kF32 : eng->impl.dt, t, (float*)tmp.ptr, nullptr);
    if (e == hipSuccess) e = sync_memcpy(out_host, tmp.ptr, numel * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return finish(hip_fail(e, "cv_get_activation"));
    return CV_OK;
}

static int impl_cv_model_macs(cv_engine_t* eng, const char* model, int64_t* macs) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!model || !macs) return finish(fail(CV_ERR_INVALID, "null argument"));
    if (std::strcmp(model, "unet") == 0) *macs = unet_macs(eng->impl);
    else if (std::strcmp(model, "resnet18") == 0) *macs = resnet_macs(eng->impl);
    else return finish(fail(CV_ERR_INVALID, "model must be 'unet' or 'resnet18'"));
    if (*macs == 0) return finish(fail(CV_ERR_STATE, "model not loaded"));
    return CV_OK;
}

static int impl_cv_profile_convs(cv_engine_t* eng, const char* model, const void* x, int batch, void* out, int iters,
                     void* stream, float* conv_ms_total, int* conv_launches, float* all_ms_total) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!model || iters <= 0) return finish(fail(CV_ERR_INVALID, "bad profile arguments"));
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    Engine& e = eng->impl;
    e.prof_clear();
    e.profiling = true;
    const bool is_unet = std::strcmp(model, "unet") == 0;
    for (int i = 0; i < iters && s.ok(); ++i) {
        if (is_unet) s = unet_forward(e, x, false, batch, (float*)out, nullptr, 0.5f, (hipStream_t)stream);
        else s = resnet_forward(e, x, false, batch, (float*)out, false, (hipStream_t)stream);
    }
    e.profiling = false;
    if (s.ok()) s = e.prof_collect();
    if (!s.ok()) { e.prof_clear(); return finish(s); }
    float conv = 0.f, all = 0.f;
    int n = 0;
    for (auto& pe : e.prof) { all += pe.ms; if (pe.is_conv) { conv += pe.ms; ++n; } }
    if (conv_ms_total) *conv_ms_total = conv;
    if (conv_launches) *conv_launches = n;
    if (all_ms_total) *all_ms_total = all;
    return CV_OK;
}

static int impl_cv_profile_entry(cv_engine_t* eng, int index, char* name, int name_cap, float* ms, double* macs, int* is_conv) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    if (index < 0 || index >= (int)eng->impl.prof.size()) return finish(fail(CV_ERR_INVALID, "profile index out of range"));
    const ProfileEntry& pe = eng->impl.prof[index];
    if (name && name_cap > 0) { std::strncpy(name, pe.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (ms) *ms = pe.ms;
    if (macs) *macs = pe.macs;
    if (is_conv) *is_conv = pe.is_conv ? 1 : 0;
    return CV_OK;
}

// ---- single-layer entry points (parity tests) -------------------------------------------------------
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static int impl_cv_op_conv2d(cv_engine_t* eng, const float* x, int n, int cin, int h, int w_, const float* w_host, int cout,
                 int k, int stride, const float* scale_host, const float* shift_host, const float* residual,
                 int relu, float* y, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!x || !w_host || !y || n <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w_ <= 0 || stride <= 0)
        return finish(fail(CV_ERR_INVALID, "cv_op_conv2d: bad argument"));
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    Engine& e = eng->impl;
    hipStream_t st = (hipStream_t)stream;
    const int pad = (k - 1) / 2;
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w_ + 2 * pad - k) / stride + 1;
    const int cinPad = round_up(cin, 8);
    std::vector<float> ones(cout, 1.f), zeros(cout, 0.f);
    ConvLayer L;
    e.ws_slot = 0;
    s = L.build_conv("op_conv2d", e.dt, w_host, cout, cin, k, stride, scale_host ? scale_host : ones.data(),
                     shift_host ? shift_host : zeros.data(), cinPad, (int64_t)n * ho * wo, (ho == wo) ? ho : 0);
    if (!s.ok()) return finish(s);
    Activation ax, ay, ar;
    if ((s = ax.create(n, h, w_, cinPad, e.dt)).ok() && (s = ay.create(n, ho, wo, cout, e.dt)).ok()) {
        hipError_t err = pack_nchw_f32(e.dt, x, cin, ax.ref(n), e.guard_ptr(), st);
        TensorRef rr;
        if (err == hipSuccess && residual) {
            s = ar.create(n, ho, wo, cout, e.dt);
            if (s.ok()) { err = pack_nchw_f32(e.dt, residual, cout, ar.ref(n), e.guard_ptr(), st); rr = ar.ref(n); }
        }
        if (s.ok() && err == hipSuccess) s = e.run_conv(L, ax.ref(n), ay.ref(n), residual ? &rr : nullptr, relu != 0, st);
        if (s.ok() && err == hipSuccess) err = unpack_nchw_f32(e.dt, ay.ref(n), y, st);
        if (s.ok() && err == hipSuccess) err = hipStreamSynchronize(st);
        if (s.ok() && err != hipSuccess) s = hip_fail(err, "cv_op_conv2d");
    }
    return finish(s);
}

static int impl_cv_op_conv_transpose2x2(cv_engine_t* eng, const float* x, int n, int cin, int h, int w_, const float* w_host,
                            int cout, const float* bias_host, float* y, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!x || !w_host || !y || !bias_host || n <= 0) return finish(fail(CV_ERR_INVALID, "cv_op_conv_transpose2x2: bad argument"));
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    Engine& e = eng->impl;
    hipStream_t st = (hipStream_t)stream;
    ConvLayer L;
    e.ws_slot = 0;
    s = L.build_convT("op_convT", e.dt, w_host, cin, cout, bias_host, (int64_t)n * h * w_);
    if (!s.ok()) return finish(s);
    Activation ax, ay;
    if ((s = ax.create(n, h, w_, cin, e.dt)).ok() && (s = ay.create(n, 2 * h, 2 * w_, cout, e.dt)).ok()) {
        hipError_t err = pack_nchw_f32(e.dt, x, cin, ax.ref(n), e.guard_ptr(), st);
        if (err == hipSuccess) s = e.run_conv(L, ax.ref(n), ay.ref(n), nullptr, false, st);
        if (s.ok() && err == hipSuccess) err = unpack_nchw_f32(e.dt, ay.ref(n), y, st);
        if (s.ok() && err == hipSuccess) err = hipStreamSynchronize(st);
        if (s.ok() && err != hipSuccess) s = hip_fail(err, "cv_op_conv_transpose2x2");
    }
    return finish(s);
}

typedef hipError_t (*pool_fn)(int, const TensorRef&, const TensorRef&, hipStream_t);
static hipError_t upsample_plain(int dt, const TensorRef& a, const TensorRef& b, hipStream_t s) {
    return upsample_bilinear2x(dt, a, b, nullptr, 0u, s);
}
static int pool_like(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, int ho, int wo, float* y,
                     void* stream, pool_fn fn, const char* what) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!x || !y || n <= 0 || c <= 0 || h <= 0 || w_ <= 0 || ho <= 0 || wo <= 0) return finish(fail(CV_ERR_INVALID, std::string(what) + ": bad argument"));
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    Engine& e = eng->impl;
    hipStream_t st = (hipStream_t)stream;
    const int cp = round_up(c, 8);
    Activation ax, ay;
    if ((s = ax.create(n, h, w_, cp, e.dt)).ok() && (s = ay.create(n, ho, wo, cp, e.dt)).ok()) {
        hipError_t err = pack_nchw_f32(e.dt, x, c, ax.ref(n), e.guard_ptr(), st);
        if (err == hipSuccess) err = fn(e.dt, ax.ref(n), ay.ref(n), st);
        if (err == hipSuccess) err = unpack_nchw_f32(e.dt, ay.ref(n, 0, c), y, st);
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        if (err != hipSuccess) s = hip_fail(err, what);
    }
    return finish(s);
}

// conv 1x1 C -> 1 + bias (+ sigmoid / threshold mask): the stand-alone OutConv kernel, as a single-layer entry point
static int impl_cv_op_outc_1x1(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, const float* w_host, const float* bias_host,
                               float threshold, float* logits, uint8_t* mask, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!x || !w_host || !bias_host || !logits || n <= 0 || c <= 0 || h <= 0 || w_ <= 0)
        return finish(fail(CV_ERR_INVALID, "cv_op_outc_1x1: bad argument"));
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    Engine& e = eng->impl;
    hipStream_t st = (hipStream_t)stream;
    const int cp = round_up(c, 8);
    const int lpp = cp / dtype_group(e.dt);
    if (lpp > 64 || (lpp & (lpp - 1))) return finish(fail(CV_ERR_INVALID, "cv_op_outc_1x1: channel count must give a power-of-two lane group <= 64"));
    std::vector<float> wpad(cp, 0.f);
    std::copy(w_host, w_host + c, wpad.begin());
    Activation ax;
    DeviceBuffer dw, db;
    if ((s = ax.create(n, h, w_, cp, e.dt)).ok() && (s = dw.upload(wpad.data(), cp * sizeof(float))).ok() &&
        (s = db.upload(bias_host, sizeof(float))).ok()) {
        hipError_t err = pack_nchw_f32(e.dt, x, c, ax.ref(n), e.guard_ptr(), st);
        if (err == hipSuccess)
            err = outc_1x1(e.dt, ax.ref(n), (const float*)dw.ptr, (const float*)db.ptr, logits, mask, threshold, e.guard_ptr(),
                           e.register_layer("op_outc_1x1"), st);
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        if (err != hipSuccess) s = hip_fail(err, "cv_op_outc_1x1");
    }
    return finish(s);
}

static int impl_cv_op_maxpool2x2(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, float* y, void* stream) {
    return pool_like(eng, x, n, c, h, w_, h / 2, w_ / 2, y, stream, maxpool2x2, "cv_op_maxpool2x2");
}
static int impl_cv_op_maxpool3x3s2(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, float* y, void* stream) {
    return pool_like(eng, x, n, c, h, w_, (h + 2 - 3) / 2 + 1, (w_ + 2 - 3) / 2 + 1, y, stream, maxpool3x3s2, "cv_op_maxpool3x3s2");
}
static int impl_cv_op_upsample_bilinear2x(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, float* y, void* stream) {
    return pool_like(eng, x, n, c, h, w_, 2 * h, 2 * w_, y, stream, upsample_plain, "cv_op_upsample_bilinear2x");
}

// ---- classical stages either side of the CNNs -----------------------------------------
static int impl_cv_find_quadrangle(const uint8_t* mask, int h, int w, int32_t quad[8], int* found) {
    if (!mask || !quad || !found || h <= 0 || w <= 0) return finish(fail(CV_ERR_INVALID, "cv_find_quadrangle: bad argument"));
    *found = find_quadrangle(mask, h, w, quad) ? 1 : 0;
    return CV_OK;
}

static int impl_cv_find_contours(const uint8_t* mask, int h, int w, int method, int32_t* xy, int64_t cap_points, int32_t* counts,
                                 int32_t* holes, int64_t cap_contours, int64_t* n_contours) {
    if (!mask || !xy || !counts || !holes || !n_contours || h <= 0 || w <= 0 || cap_points < 0 || cap_contours < 0 ||
        (method != 0 && method != 1))
        return finish(fail(CV_ERR_INVALID, "cv_find_contours: bad argument"));
    const long n = find_contours_flat(mask, h, w, method == 1, xy, (long)cap_points, counts, holes, (long)cap_contours);
    if (n < 0) return finish(fail(CV_ERR_INVALID, "cv_find_contours: output capacity too small"));
    *n_contours = n;
    return CV_OK;
}

static int impl_cv_find_quadrangles(const uint8_t* masks, int n, int h, int w, int32_t* quads, int32_t* found, int n_threads) {
    if (!masks || !quads || !found || n < 0 || h <= 0 || w <= 0) return finish(fail(CV_ERR_INVALID, "cv_find_quadrangles: bad argument"));
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    // a mask takes ~50 us since round 4 (run-based labelling): starting a thread costs about as much, so every thread gets at least
    // eight masks (64 masks: 8 threads; a single board: none)
    nt = std::max(1, std::min(nt, std::min((n + 7) / 8, 32)));
    auto work = [&](int t) {
        for (int i = t; i < n; i += nt)
            found[i] = find_quadrangle(masks + (size_t)i * h * w, h, w, quads + (size_t)i * 8) ? 1 : 0;
    };
    if (nt == 1) { work(0); return CV_OK; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
    return CV_OK;
}

// INTER_AREA on `st` with the engine's cached tables for fractional shrinks (caller holds the engine mutex)
static Status resize_on_stream(Engine& en, const uint8_t* src, int n, int h, int w_, int channels, uint8_t* dst, int out_h, int out_w, hipStream_t st) {
    hipError_t e;
    const bool integer = h % out_h == 0 && w_ % out_w == 0;
    if (!integer && out_h <= h && out_w <= w_ && channels <= 4) {
        // fractional shrink: OpenCV's float32 table form; the two tables are built once per geometry and stay on the device
        const long long key = (((long long)h * 65536 + w_) * 65536 + out_h) * 65536 + out_w;
        if (en.area_key != key) {
            if (capture_flag()) return fail(CV_ERR_STATE, "resize tables rebuilt during graph capture");
            std::vector<int> xo, xs, yo, ys;
            std::vector<float> xa, ya;
            resize_area_table(w_, out_w, xo, xs, xa);
            resize_area_table(h, out_h, yo, ys, ya);
            CV_HIP(hipStreamSynchronize(st));                          // a launch in flight may still read the previous tables
            std::vector<char> blob;
            auto put = [&](const void* p, size_t nbytes) { const size_t at = blob.size(); blob.resize(at + ((nbytes + 15) & ~(size_t)15)); std::memcpy(blob.data() + at, p, nbytes); return at; };
            const size_t o0 = put(xo.data(), xo.size() * 4), o1 = put(xs.data(), xs.size() * 4), o2 = put(xa.data(), xa.size() * 4);
            const size_t o3 = put(yo.data(), yo.size() * 4), o4 = put(ys.data(), ys.size() * 4), o5 = put(ya.data(), ya.size() * 4);
            CV_TRY(en.area_tabs.upload(blob.data(), blob.size()));
            const size_t off[6] = {o0, o1, o2, o3, o4, o5};
            for (int i = 0; i < 6; ++i) en.area_off[i] = off[i];
            en.area_key = key;
        }
        const char* base = (const char*)en.area_tabs.ptr;
        e = resize_area_u8_tab(src, n, h, w_, channels, dst, out_h, out_w, (const int*)(base + en.area_off[0]), (const int*)(base + en.area_off[1]),
                               (const float*)(base + en.area_off[2]), (const int*)(base + en.area_off[3]), (const int*)(base + en.area_off[4]),
                               (const float*)(base + en.area_off[5]), st);
    } else {
        e = resize_area_u8(src, n, h, w_, channels, dst, out_h, out_w, st);
    }
    if (e != hipSuccess) return hip_fail(e, "resize_area_u8");
    return Status();
}

static int impl_cv_resize_area_u8(cv_engine_t* eng, const uint8_t* src, int n, int h, int w_, int channels, uint8_t* dst, int out_h,
                      int out_w, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!src || !dst || n <= 0 || h <= 0 || w_ <= 0 || channels <= 0 || out_h <= 0 || out_w <= 0)
        return finish(fail(CV_ERR_INVALID, "cv_resize_area_u8: bad argument"));
    DeviceGuard g(eng->impl.device);
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    return finish(resize_on_stream(eng->impl, src, n, h, w_, channels, dst, out_h, out_w, (hipStream_t)stream));
}

static int impl_cv_extract_squares_u8(cv_engine_t* eng, const uint8_t* images, int n, int h, int w_, const double* inv_host,
                          uint8_t* squares, uint8_t* boards, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!images || !inv_host || !squares || n <= 0 || h <= 0 || w_ <= 0)
        return finish(fail(CV_ERR_INVALID, "cv_extract_squares_u8: bad argument"));
    if ((size_t)h * (size_t)w_ > ((size_t)1 << 30)) return finish(fail(CV_ERR_INVALID, "cv_extract_squares_u8: image too large (32-bit byte offsets)"));
    if (((uintptr_t)squares & 3) || ((uintptr_t)boards & 3)) return finish(fail(CV_ERR_INVALID, "cv_extract_squares_u8: outputs must be 4-byte aligned"));
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    Engine& e = eng->impl;
    const size_t bytes = (size_t)n * 9 * sizeof(double);
    if (e.scratch.bytes < bytes) { s = e.scratch.alloc(bytes, false); if (!s.ok()) return finish(s); }
    hipStream_t st = (hipStream_t)stream;
    hipError_t err = hipMemcpyAsync(e.scratch.ptr, inv_host, bytes, hipMemcpyHostToDevice, st);
    if (err == hipSuccess) err = extract_squares_u8(images, n, h, w_, (const double*)e.scratch.ptr, squares, boards, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);       // inv_host may be reused by the caller; scratch by the next call
    if (err != hipSuccess) return finish(hip_fail(err, "extract_squares_u8"));
    return CV_OK;
}

// range calibration as data: the per-tensor exponents of a model, in the fixed order of its activation list (two per tensor: the
// tensor's exponent and the second half's exponent of a concatenated buffer)
static int impl_cv_engine_calibration(cv_engine_t* eng, const char* model, int32_t* exps, int capacity, int* count, int import) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!model || !count) return finish(fail(CV_ERR_INVALID, "null argument"));
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    Engine& e = eng->impl;
    std::vector<Activation*>* acts = nullptr;
    if (std::strcmp(model, "unet") == 0 && e.unet) acts = &e.unet->acts;
    else if (std::strcmp(model, "resnet18") == 0 && e.resnet) acts = &e.resnet->acts;
    else return finish(fail(CV_ERR_STATE, "model must be a loaded 'unet' or 'resnet18'"));
    const int n = 2 * (int)acts->size();
    if (!import) {
        *count = n;
        if (!exps) return CV_OK;                                  // size query
        if (capacity < n) return finish(fail(CV_ERR_INVALID, "calibration buffer too small"));
        for (size_t i = 0; i < acts->size(); ++i) { exps[2 * i] = (*acts)[i]->exp; exps[2 * i + 1] = (*acts)[i]->exp2; }
        return CV_OK;
    }
    if (!exps || capacity != n) return finish(fail(CV_ERR_INVALID, "calibration vector does not match this model (" + std::to_string(n) + " entries expected)"));
    if (e.dt == kF32) return CV_OK;                               // the f32 engine stores real values: nothing to set
    // validate the WHOLE vector before the first write: a rejected import leaves the engine exactly as it was
    for (int i = 0; i < n; ++i)
        if (exps[i] < -60 || exps[i] > 60) return finish(fail(CV_ERR_INVALID, "exponent out of range (entry " + std::to_string(i) + ")"));
    bool changed = false;
    for (size_t i = 0; i < acts->size(); ++i) {
        Activation* a = (*acts)[i];
        if (a->fixed_exp) continue;
        if (a->exp != exps[2 * i] || (a->split_c && a->exp2 != exps[2 * i + 1])) changed = true;
        a->exp = exps[2 * i];
        if (a->split_c) a->exp2 = exps[2 * i + 1];
    }
    if (changed) {                                                // the layers re-fold their epilogue constants at their next launch
        hipError_t he = device_synchronize();
        if (he != hipSuccess) return finish(hip_fail(he, "cv_engine_import_calibration"));
        e.graph_invalidate();
        e.graph_clear();
    }
    *count = changed ? 1 : 0;
    return CV_OK;
}

static int impl_cv_board_homographies(const float* quads, int n, int out_w, int out_h, double* forward, double* inverse) {
    if (n < 0 || out_w <= 0 || out_h <= 0 || (n > 0 && (!quads || (!forward && !inverse))))
        return finish(fail(CV_ERR_INVALID, "cv_board_homographies: bad argument"));
    board_homographies(quads, n, out_w, out_h, forward, inverse);
    return CV_OK;
}

// ---- one image, host to host ---------------------------------------------------------------------------------------------
static Status resize_on_stream(Engine& en, const uint8_t* src, int n, int h, int w_, int channels, uint8_t* dst, int out_h, int out_w, hipStream_t st);

static int impl_cv_process_image(cv_engine_t* ue, cv_engine_t* ce, const uint8_t* image, int h, int w_, float threshold, int flip,
                                 int fallback_quad, cv_image_result_t* out, size_t out_size, void* stream) {
    // the caller's struct may be the shorter one of an earlier ABI: nothing at or beyond out_size is touched
    if (out && out_size < offsetof(cv_image_result_t, n_fixes) + sizeof(int32_t))
        return finish(fail(CV_ERR_INVALID, "cv_process_image: result struct smaller than the ABI 3 layout"));
    uint8_t* const out_squares = (out && out_size >= offsetof(cv_image_result_t, squares) + sizeof(uint8_t*)) ? out->squares : nullptr;
    Status s = check_engine(ue);
    if (s.ok()) s = check_engine(ce);
    if (!s.ok()) return finish(s);
    if (!image || !out || h <= 0 || w_ <= 0 || (size_t)h * w_ > ((size_t)1 << 28)) return finish(fail(CV_ERR_INVALID, "cv_process_image: bad argument"));
    if (!(threshold >= 0.f && threshold <= 1.f)) return finish(fail(CV_ERR_INVALID, "threshold must be between 0 and 1"));
    if (ue->impl.device != ce->impl.device) return finish(fail(CV_ERR_STATE, "cv_process_image: both engines must live on one device"));
    Engine& U = ue->impl;
    Engine& C = ce->impl;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard g(U.device);
    out->found = 0;
    out->n_fixes = 0;
    const size_t img_b = (size_t)h * w_ * 3, small_b = 256 * 256 * 3, lg_b = 65536 * sizeof(float), mk_b = 65536, sq_b = 64 * 4096,
                 bd_b = 512 * 512, pr_b = 64 * 13 * sizeof(float), inv_b = 9 * sizeof(double);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // host block: image | mask | logits | board | probs | inv          device block: image | small | logits | mask | squares | board | probs | inv
    const size_t h_off[6] = {0, up(img_b), up(img_b) + up(mk_b), up(img_b) + up(mk_b) + up(lg_b), up(img_b) + up(mk_b) + up(lg_b) + up(bd_b),
                             up(img_b) + up(mk_b) + up(lg_b) + up(bd_b) + up(pr_b)};
    const size_t h_guard = h_off[5] + up(inv_b);              // two guard words (extractor, classifier)
    const size_t h_need = h_guard + 256;
    size_t d_off[8];
    std::lock_guard<std::mutex> pipe_lock(U.pipe_mu);   // one request at a time per extractor engine: the staging blocks are shared
    {
        const size_t sizes[8] = {img_b, small_b, lg_b, mk_b, sq_b, bd_b, pr_b, inv_b};
        size_t at = 0;
        for (int i = 0; i < 8; ++i) { d_off[i] = at; at += up(sizes[i]); }
        std::unique_lock<std::mutex> lk(U.mu);
        if (U.pipe_host_bytes < h_need) {
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) return finish(hip_fail(e, "cv_process_image"));
            if (U.pipe_host) block_release(U.pipe_host, U.pipe_host_cap, true);
            U.pipe_host = nullptr; U.pipe_host_bytes = 0; U.pipe_host_cap = 0;
            e = block_alloc(&U.pipe_host, h_need, &U.pipe_host_cap, true);
            if (e != hipSuccess) { U.pipe_host = nullptr; return finish(fail(CV_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e))); }
            U.pipe_host_bytes = h_need;
        }
        if (U.pipe_dev.bytes < at) {
            hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) return finish(hip_fail(e, "cv_process_image"));
            U.graph_invalidate();
            s = U.pipe_dev.alloc(at, false);
            if (!s.ok()) return finish(s);
        }
    }
    char* hb = (char*)U.pipe_host;
    char* db = (char*)U.pipe_dev.ptr;
    uint8_t* d_img = (uint8_t*)(db + d_off[0]); uint8_t* d_small = (uint8_t*)(db + d_off[1]); float* d_lg = (float*)(db + d_off[2]);
    uint8_t* d_mk = (uint8_t*)(db + d_off[3]); uint8_t* d_sq = (uint8_t*)(db + d_off[4]); uint8_t* d_bd = (uint8_t*)(db + d_off[5]);
    float* d_pr = (float*)(db + d_off[6]);
    uint8_t* h_img = (uint8_t*)(hb + h_off[0]); uint8_t* h_mk = (uint8_t*)(hb + h_off[1]); float* h_lg = (float*)(hb + h_off[2]);
    uint8_t* h_bd = (uint8_t*)(hb + h_off[3]); float* h_pr = (float*)(hb + h_off[4]); double* h_inv = (double*)(hb + h_off[5]);
    unsigned* h_gd = (unsigned*)(hb + h_guard);

    // The mask (64 KB) and the probabilities (3.3 KB) are written by their kernels straight into the page-locked block, which the device
    // sees under its host address: over PCIe inside the kernel instead of a copy operation behind it (-18 us per call for the mask, same-box
    // A/B in profiles/r05_tuning.md section 14).  CV_PI_HOSTMASK=0 / CV_PI_HOSTOUT=0 restore the device buffers + copies.
    static const bool host_mask = [] { const char* v = std::getenv("CV_PI_HOSTMASK"); return !(v && v[0] == '0'); }();
    std::memcpy(h_img, image, img_b);
    hipError_t e = hipMemcpyAsync(d_img, h_img, img_b, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return finish(hip_fail(e, "cv_process_image: upload"));
    {
        std::lock_guard<std::mutex> lk(U.mu);
        s = resize_on_stream(U, d_img, 1, h, w_, 3, d_small, 256, 256, st);
        if (s.ok()) s = unet_forward(U, d_small, true, 1, d_lg, host_mask ? h_mk : d_mk, threshold, st);
    }
    if (!s.ok()) return finish(s);
    e = host_mask ? hipSuccess : hipMemcpyAsync(h_mk, d_mk, mk_b, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && !U.pipe_event) e = hipEventCreateWithFlags(&U.pipe_event, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(U.pipe_event, st);
    if (e == hipSuccess) e = hipMemcpyAsync(h_lg, d_lg, lg_b, hipMemcpyDeviceToHost, st);     // travels while the host finds the quadrangle
    if (e == hipSuccess) e = hipEventSynchronize(U.pipe_event);
    if (e != hipSuccess) return finish(hip_fail(e, "cv_process_image: mask"));
    if (out->mask) std::memcpy(out->mask, h_mk, mk_b);
    int32_t quad[8];
    bool side_busy = false, side_joined = false;             // a download is in flight on U.pipe_side: every exit below waits for it
    bool found = find_quadrangle(h_mk, 256, 256, quad);
    if (!found && fallback_quad) {
        const int32_t whole[8] = {255, 0, 0, 0, 0, 255, 255, 255};            // TR, TL, BL, BR of the mask
        std::memcpy(quad, whole, sizeof(quad));
        found = true;
    }
    if (found) {
        // _scale_quadrangle (core.py:413-417): np.array(approx * (orig_h / 256.0), dtype=float32) -- a double product rounded to float
        float corners[8];
        const double factor = (double)h / 256.0;
        for (int i = 0; i < 8; ++i) { corners[i] = (float)((double)quad[i] * factor); out->quadrangle[i] = corners[i]; }
        board_homographies(corners, 1, 512, 512, nullptr, h_inv);
        e = extract_squares_u8_one(d_img, h, w_, h_inv, d_sq, d_bd, st);   // the matrix travels in the kernel arguments
        // the rectified board (256 KB) goes home on a side stream while the classifier runs: behind the classifier on `st` it was 16 us
        // at the end of every call
        static const bool side_on = [] { const char* v = std::getenv("CV_PI_SIDE"); return !(v && v[0] == '0'); }();
        static const bool host_out = [] { const char* v = std::getenv("CV_PI_HOSTOUT"); return !(v && v[0] == '0'); }();
        if (side_on) {
            if (e == hipSuccess && !U.pipe_side) e = hipStreamCreateWithFlags(&U.pipe_side, hipStreamNonBlocking);
            if (e == hipSuccess && !U.pipe_event2) e = hipEventCreateWithFlags(&U.pipe_event2, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(U.pipe_event2, st);
            if (e == hipSuccess) e = hipStreamWaitEvent(U.pipe_side, U.pipe_event2, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(h_bd, d_bd, bd_b, hipMemcpyDeviceToHost, U.pipe_side);
            side_busy = e == hipSuccess;
            if (e == hipSuccess && !U.pipe_event3) e = hipEventCreateWithFlags(&U.pipe_event3, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(U.pipe_event3, U.pipe_side);
        }
        if (e != hipSuccess) { if (side_busy) (void)hipStreamSynchronize(U.pipe_side); return finish(hip_fail(e, "cv_process_image: warp")); }
        {
            std::lock_guard<std::mutex> lk(C.mu);
            // the 64 x 13 probabilities are written by the head kernel straight into the page-locked block (device-visible under
            // its host address): 3.3 KB over PCIe inside the kernel instead of one more copy operation at the end of the stream
            s = resnet_forward(C, d_sq, true, 64, host_out ? h_pr : d_pr, true, st);
        }
        if (!s.ok()) { if (side_busy) (void)hipStreamSynchronize(U.pipe_side); return finish(s); }
        e = host_out ? hipSuccess : hipMemcpyAsync(h_pr, d_pr, pr_b, hipMemcpyDeviceToHost, st);
        // the side stream joins `st` on the DEVICE (the download finished long before the classifier): one host synchronisation per call
        if (e == hipSuccess && side_busy) { e = hipStreamWaitEvent(st, U.pipe_event3, 0); side_joined = e == hipSuccess; }
        if (e == hipSuccess && !side_busy) e = hipMemcpyAsync(h_bd, d_bd, bd_b, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) { if (side_busy) (void)hipStreamSynchronize(U.pipe_side); return finish(hip_fail(e, "cv_process_image: download")); }
    }
    {   // the numeric guards ride with the downloads into page-locked memory; ONE synchronisation, then both words are judged
        std::lock_guard<std::mutex> lk(U.mu);
        s = U.guard_read_async(h_gd, st);
    }
    h_gd[1] = 0xffffffffu;
    if (s.ok() && found && &C != &U) {
        std::lock_guard<std::mutex> lk(C.mu);
        s = C.guard_read_async(h_gd + 1, st);
    }
    if (!s.ok()) { if (side_busy) (void)hipStreamSynchronize(U.pipe_side); return finish(s); }
    e = hipStreamSynchronize(st);                           // everything above has landed afterwards
    if (side_busy && !side_joined) { const hipError_t e2 = hipStreamSynchronize(U.pipe_side); if (e == hipSuccess) e = e2; }
    if (e != hipSuccess) return finish(hip_fail(e, "cv_process_image: synchronise"));
    {
        std::lock_guard<std::mutex> lk(U.mu);
        s = U.guard_eval(h_gd[0]);
    }
    if (s.ok() && found && &C != &U) {
        std::lock_guard<std::mutex> lk(C.mu);
        s = C.guard_eval(h_gd[1]);
    }
    if (!s.ok()) return finish(s);
    if (out->logits) std::memcpy(out->logits, h_lg, lg_b);
    if (!found) return CV_OK;
    if (out->board) std::memcpy(out->board, h_bd, bd_b);
    if (out_squares)                                         // extract_squares (core.py:419-439) on the host copy: 4096 rows of 64 bytes
        for (int r = 0; r < 8; ++r)
            for (int c = 0; c < 8; ++c)
                for (int y = 0; y < 64; ++y)
                    std::memcpy(out_squares + ((size_t)(r * 8 + c) * 64 + y) * 64, h_bd + (size_t)(r * 64 + y) * 512 + c * 64, 64);
    if (out->probabilities) std::memcpy(out->probabilities, h_pr, pr_b);
    int8_t labels[64];
    decode_positions(h_pr, 1, flip, out->fen, out->original_fen, labels, out->fixes, &out->n_fixes);
    if (out->labels) std::memcpy(out->labels, labels, sizeof(labels));
    out->found = 1;
    return CV_OK;
}

static int impl_cv_profile_entry_bytes(cv_engine_t* eng, int index, double* bytes) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    if (!bytes || index < 0 || index >= (int)eng->impl.prof.size()) return finish(fail(CV_ERR_INVALID, "profile index out of range"));
    *bytes = eng->impl.prof[index].bytes;
    return CV_OK;
}

static int impl_cv_profile_entry_kernel(cv_engine_t* eng, int index, char* kernel, int kernel_cap) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    if (!kernel || kernel_cap <= 0 || index < 0 || index >= (int)eng->impl.prof.size()) return finish(fail(CV_ERR_INVALID, "profile index out of range"));
    const std::string& k = eng->impl.prof[index].kernel;
    const size_t n = std::min(k.size(), (size_t)kernel_cap - 1);
    std::memcpy(kernel, k.data(), n);
    kernel[n] = 0;
    return CV_OK;
}

static int impl_cv_engine_numeric_status(cv_engine_t* eng, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    DeviceGuard g(eng->impl.device);
    return finish(eng->impl.guard_check((hipStream_t)stream));
}

static int impl_cv_engine_workspace_bytes(cv_engine_t* eng, size_t* bytes) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!bytes) return finish(fail(CV_ERR_INVALID, "null argument"));
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    *bytes = eng->impl.workspace_bytes();
    return CV_OK;
}

static int impl_cv_get_activation_exponent(cv_engine_t* eng, const char* model, const char* name, int* exponent) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!model || !name || !exponent) return finish(fail(CV_ERR_INVALID, "null argument"));
    std::lock_guard<std::mutex> lk(eng->impl.mu);
    TensorRef t;
    if (std::strcmp(model, "unet") == 0) s = unet_activation(eng->impl, name, &t);
    else if (std::strcmp(model, "resnet18") == 0) s = resnet_activation(eng->impl, name, &t);
    else s = fail(CV_ERR_INVALID, "model must be 'unet' or 'resnet18'");
    if (!s.ok()) return finish(s);
    *exponent = t.exp;
    return CV_OK;
}

static int impl_cv_decode_positions(const float* probs, int n_boards, int flip, char* fen, char* original_fen, int8_t* labels,
                                    int32_t* fixes, int32_t* n_fixes) {
    if (n_boards < 0 || (n_boards > 0 && (!probs || !fen || !original_fen || !labels || !fixes || !n_fixes)))
        return finish(fail(CV_ERR_INVALID, "cv_decode_positions: bad argument"));
    decode_positions(probs, n_boards, flip, fen, original_fen, labels, fixes, n_fixes);
    return CV_OK;
}

static int impl_cv_extract_squares_u8_dev(cv_engine_t* eng, const uint8_t* images, int n, int h, int w_, const double* inv_dev,
                                          uint8_t* squares, uint8_t* boards, void* stream) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    if (!images || !inv_dev || !squares || n <= 0 || h <= 0 || w_ <= 0)
        return finish(fail(CV_ERR_INVALID, "cv_extract_squares_u8_dev: bad argument"));
    if ((size_t)h * (size_t)w_ > ((size_t)1 << 30)) return finish(fail(CV_ERR_INVALID, "cv_extract_squares_u8_dev: image too large (32-bit byte offsets)"));
    if (((uintptr_t)squares & 3) || ((uintptr_t)boards & 3) || ((uintptr_t)inv_dev & 7))
        return finish(fail(CV_ERR_INVALID, "cv_extract_squares_u8_dev: outputs must be 4-byte aligned, matrices 8-byte aligned"));
    DeviceGuard g(eng->impl.device);
    hipError_t err = extract_squares_u8(images, n, h, w_, inv_dev, squares, boards, (hipStream_t)stream);
    if (err != hipSuccess) return finish(hip_fail(err, "extract_squares_u8"));
    return CV_OK;
}

static int impl_cv_selftest_mfma(cv_engine_t* eng, float* max_err_f16, float* max_err_f32) {
    Status s = check_engine(eng);
    if (!s.ok()) return finish(s);
    DeviceGuard g(eng->impl.device);
    // asymmetric integer-valued operands: exact in f16 and f32, so the expected error is exactly 0
    _Float16 a16[16 * 32], b16[16 * 32];
    float a32[16 * 16], b32[16 * 16], ref16[256], ref32[256], got[256];
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 32; ++k) { a16[i * 32 + k] = (_Float16)(float)((i * 3 + k * 5) % 7 - 3); b16[i * 32 + k] = (_Float16)(float)((i * 5 + k * 2 + 1) % 9 - 4); }
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 16; ++k) { a32[i * 16 + k] = (float)((i * 3 + k * 5) % 7 - 3); b32[i * 16 + k] = (float)((i * 5 + k * 2 + 1) % 9 - 4); }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float s16 = 0, s32 = 0;
            for (int k = 0; k < 32; ++k) s16 += (float)a16[i * 32 + k] * (float)b16[j * 32 + k];
            for (int k = 0; k < 16; ++k) s32 += a32[i * 16 + k] * b32[j * 16 + k];
            ref16[i * 16 + j] = s16; ref32[i * 16 + j] = s32;
        }
    DeviceBuffer da, db, dd;
    hipError_t e = hipSuccess;
    float err16 = 0.f, err32 = 0.f;
    if (!(s = da.upload(a16, sizeof(a16))).ok() || !(s = db.upload(b16, sizeof(b16))).ok() || !(s = dd.alloc(sizeof(got), true)).ok()) return finish(s);
    e = mfma_probe_f16((const half_t*)da.ptr, (const half_t*)db.ptr, (float*)dd.ptr, nullptr);
    if (e == hipSuccess) e = sync_memcpy(got, dd.ptr, sizeof(got), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return finish(hip_fail(e, "mfma_probe_f16"));
    for (int i = 0; i < 256; ++i) err16 = std::max(err16, std::fabs(got[i] - ref16[i]));
    if (!(s = da.upload(a32, sizeof(a32))).ok() || !(s = db.upload(b32, sizeof(b32))).ok()) return finish(s);
    e = mfma_probe_f32((const float*)da.ptr, (const float*)db.ptr, (float*)dd.ptr, nullptr);
    if (e == hipSuccess) e = sync_memcpy(got, dd.ptr, sizeof(got), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return finish(hip_fail(e, "mfma_probe_f32"));
    for (int i = 0; i < 256; ++i) err32 = std::max(err32, std::fabs(got[i] - ref32[i]));
    if (max_err_f16) *max_err_f16 = err16;
    if (max_err_f32) *max_err_f32 = err32;
    return CV_OK;
}


// ---- the exported C surface: every entry point is exception-tight -----------------------------------------------
extern "C" {

int cv_abi_version(void) { return CV_ABI_VERSION; }
const char* cv_last_error(void) { return get_error(); }

int cv_device_count(int* count) {
    return guarded("cv_device_count", [&]() -> int { return impl_cv_device_count(count); });
}
int cv_trim_memory(size_t* bytes_freed) {
    return guarded("cv_trim_memory", [&]() -> int {
        const size_t n = block_cache_trim();
        if (bytes_freed) *bytes_freed = n;
        return CV_OK;
    });
}

int cv_engine_create(int device, int precision, cv_engine_t** out) {
    return guarded("cv_engine_create", [&]() -> int { return impl_cv_engine_create(device, precision, out); });
}

int cv_engine_destroy(cv_engine_t* eng) {
    return guarded("cv_engine_destroy", [&]() -> int { return impl_cv_engine_destroy(eng); });
}

int cv_engine_set_chunk(cv_engine_t* eng, int unet_images, int resnet_squares) {
    return guarded("cv_engine_set_chunk", [&]() -> int { return impl_cv_engine_set_chunk(eng, unet_images, resnet_squares); });
}

int cv_load_unet(cv_engine_t* eng, const cv_param_t* params, int n_params) {
    return guarded("cv_load_unet", [&]() -> int { return impl_cv_load_unet(eng, params, n_params); });
}

int cv_load_resnet18(cv_engine_t* eng, const cv_param_t* params, int n_params) {
    return guarded("cv_load_resnet18", [&]() -> int { return impl_cv_load_resnet18(eng, params, n_params); });
}

int cv_unet_forward(cv_engine_t* eng, const float* x, int batch, float* logits, void* stream) {
    return guarded("cv_unet_forward", [&]() -> int { return impl_cv_unet_forward(eng, x, batch, logits, stream); });
}

int cv_unet_forward_u8(cv_engine_t* eng, const uint8_t* x_u8, int batch, float* logits, uint8_t* mask,
                       float threshold, void* stream) {
    return guarded("cv_unet_forward_u8", [&]() -> int { return impl_cv_unet_forward_u8(eng, x_u8, batch, logits, mask, threshold, stream); });
}

int cv_resnet18_forward(cv_engine_t* eng, const float* x, int n, float* logits, void* stream) {
    return guarded("cv_resnet18_forward", [&]() -> int { return impl_cv_resnet18_forward(eng, x, n, logits, stream); });
}

int cv_resnet18_forward_u8(cv_engine_t* eng, const uint8_t* squares_u8, int n, float* probs, void* stream) {
    return guarded("cv_resnet18_forward_u8", [&]() -> int { return impl_cv_resnet18_forward_u8(eng, squares_u8, n, probs, stream); });
}

int cv_softmax13(cv_engine_t* eng, const float* logits, int n, float* probs, void* stream) {
    return guarded("cv_softmax13", [&]() -> int { return impl_cv_softmax13(eng, logits, n, probs, stream); });
}

int cv_get_activation(cv_engine_t* eng, const char* model, const char* name, float* out_host, size_t out_capacity,
                      int64_t dims[4]) {
    return guarded("cv_get_activation", [&]() -> int { return impl_cv_get_activation(eng, model, name, out_host, out_capacity, dims); });
}

int cv_model_macs(cv_engine_t* eng, const char* model, int64_t* macs) {
    return guarded("cv_model_macs", [&]() -> int { return impl_cv_model_macs(eng, model, macs); });
}

int cv_profile_convs(cv_engine_t* eng, const char* model, const void* x, int batch, void* out, int iters,
                     void* stream, float* conv_ms_total, int* conv_launches, float* all_ms_total) {
    return guarded("cv_profile_convs", [&]() -> int { return impl_cv_profile_convs(eng, model, x, batch, out, iters, stream, conv_ms_total, conv_launches, all_ms_total); });
}

int cv_profile_entry(cv_engine_t* eng, int index, char* name, int name_cap, float* ms, double* macs, int* is_conv) {
    return guarded("cv_profile_entry", [&]() -> int { return impl_cv_profile_entry(eng, index, name, name_cap, ms, macs, is_conv); });
}

int cv_op_conv2d(cv_engine_t* eng, const float* x, int n, int cin, int h, int w_, const float* w_host, int cout,
                 int k, int stride, const float* scale_host, const float* shift_host, const float* residual,
                 int relu, float* y, void* stream) {
    return guarded("cv_op_conv2d", [&]() -> int { return impl_cv_op_conv2d(eng, x, n, cin, h, w_, w_host, cout, k, stride, scale_host, shift_host, residual, relu, y, stream); });
}

int cv_op_conv_transpose2x2(cv_engine_t* eng, const float* x, int n, int cin, int h, int w_, const float* w_host,
                            int cout, const float* bias_host, float* y, void* stream) {
    return guarded("cv_op_conv_transpose2x2", [&]() -> int { return impl_cv_op_conv_transpose2x2(eng, x, n, cin, h, w_, w_host, cout, bias_host, y, stream); });
}

int cv_op_maxpool2x2(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, float* y, void* stream) {
    return guarded("cv_op_maxpool2x2", [&]() -> int { return impl_cv_op_maxpool2x2(eng, x, n, c, h, w_, y, stream); });
}

int cv_op_maxpool3x3s2(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, float* y, void* stream) {
    return guarded("cv_op_maxpool3x3s2", [&]() -> int { return impl_cv_op_maxpool3x3s2(eng, x, n, c, h, w_, y, stream); });
}

int cv_op_upsample_bilinear2x(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, float* y, void* stream) {
    return guarded("cv_op_upsample_bilinear2x", [&]() -> int { return impl_cv_op_upsample_bilinear2x(eng, x, n, c, h, w_, y, stream); });
}

int cv_find_quadrangle(const uint8_t* mask, int h, int w, int32_t quad[8], int* found) {
    return guarded("cv_find_quadrangle", [&]() -> int { return impl_cv_find_quadrangle(mask, h, w, quad, found); });
}

int cv_find_contours(const uint8_t* mask, int h, int w, int method, int32_t* xy, int64_t cap_points, int32_t* counts,
                     int32_t* holes, int64_t cap_contours, int64_t* n_contours) {
    return guarded("cv_find_contours", [&]() -> int {
        return impl_cv_find_contours(mask, h, w, method, xy, cap_points, counts, holes, cap_contours, n_contours);
    });
}

int cv_find_quadrangles(const uint8_t* masks, int n, int h, int w, int32_t* quads, int32_t* found, int n_threads) {
    return guarded("cv_find_quadrangles", [&]() -> int { return impl_cv_find_quadrangles(masks, n, h, w, quads, found, n_threads); });
}

int cv_resize_area_u8(cv_engine_t* eng, const uint8_t* src, int n, int h, int w_, int channels, uint8_t* dst, int out_h,
                      int out_w, void* stream) {
    return guarded("cv_resize_area_u8", [&]() -> int { return impl_cv_resize_area_u8(eng, src, n, h, w_, channels, dst, out_h, out_w, stream); });
}

int cv_extract_squares_u8(cv_engine_t* eng, const uint8_t* images, int n, int h, int w_, const double* inv_host,
                          uint8_t* squares, uint8_t* boards, void* stream) {
    return guarded("cv_extract_squares_u8", [&]() -> int { return impl_cv_extract_squares_u8(eng, images, n, h, w_, inv_host, squares, boards, stream); });
}

int cv_engine_export_calibration(cv_engine_t* eng, const char* model, int32_t* exps, int capacity, int* count) {
    return guarded("cv_engine_export_calibration", [&]() -> int { return impl_cv_engine_calibration(eng, model, exps, capacity, count, 0); });
}

int cv_engine_import_calibration(cv_engine_t* eng, const char* model, const int32_t* exps, int count, int* changed) {
    return guarded("cv_engine_import_calibration", [&]() -> int {
        int ch = 0;
        const int rc = impl_cv_engine_calibration(eng, model, const_cast<int32_t*>(exps), count, &ch, 1);
        if (changed) *changed = ch;
        return rc;
    });
}

int cv_process_image_v2(cv_engine_t* unet_engine, cv_engine_t* classifier_engine, const uint8_t* image, int h, int w, float threshold,
                        int flip, int fallback_quad, cv_image_result_t* out, size_t out_size, void* stream) {
    return guarded("cv_process_image_v2", [&]() -> int {
        return impl_cv_process_image(unet_engine, classifier_engine, image, h, w, threshold, flip, fallback_quad, out, out_size, stream);
    });
}

int cv_process_image(cv_engine_t* unet_engine, cv_engine_t* classifier_engine, const uint8_t* image, int h, int w, float threshold,
                     int flip, int fallback_quad, cv_image_result_t* out, void* stream) {
    return guarded("cv_process_image", [&]() -> int {      // the struct of ABI 3 / 4: `squares` did not exist
        return impl_cv_process_image(unet_engine, classifier_engine, image, h, w, threshold, flip, fallback_quad, out,
                                     offsetof(cv_image_result_t, squares), stream);
    });
}

int cv_board_homographies(const float* quads, int n, int out_w, int out_h, double* forward, double* inverse) {
    return guarded("cv_board_homographies", [&]() -> int { return impl_cv_board_homographies(quads, n, out_w, out_h, forward, inverse); });
}

int cv_selftest_mfma(cv_engine_t* eng, float* max_err_f16, float* max_err_f32) {
    return guarded("cv_selftest_mfma", [&]() -> int { return impl_cv_selftest_mfma(eng, max_err_f16, max_err_f32); });
}

int cv_op_outc_1x1(cv_engine_t* eng, const float* x, int n, int c, int h, int w_, const float* w_host, const float* bias_host,
                   float threshold, float* logits, uint8_t* mask, void* stream) {
    return guarded("cv_op_outc_1x1", [&]() -> int {
        return impl_cv_op_outc_1x1(eng, x, n, c, h, w_, w_host, bias_host, threshold, logits, mask, stream);
    });
}

int cv_profile_entry_bytes(cv_engine_t* eng, int index, double* bytes) {
    return guarded("cv_profile_entry_bytes", [&]() -> int { return impl_cv_profile_entry_bytes(eng, index, bytes); });
}

int cv_profile_entry_kernel(cv_engine_t* eng, int index, char* kernel, int kernel_cap) {
    return guarded("cv_profile_entry_kernel", [&]() -> int { return impl_cv_profile_entry_kernel(eng, index, kernel, kernel_cap); });
}

int cv_engine_numeric_status(cv_engine_t* eng, void* stream) {
    return guarded("cv_engine_numeric_status", [&]() -> int { return impl_cv_engine_numeric_status(eng, stream); });
}

int cv_engine_workspace_bytes(cv_engine_t* eng, size_t* bytes) {
    return guarded("cv_engine_workspace_bytes", [&]() -> int { return impl_cv_engine_workspace_bytes(eng, bytes); });
}

int cv_get_activation_exponent(cv_engine_t* eng, const char* model, const char* name, int* exponent) {
    return guarded("cv_get_activation_exponent", [&]() -> int { return impl_cv_get_activation_exponent(eng, model, name, exponent); });
}

int cv_decode_positions(const float* probs, int n_boards, int flip, char* fen, char* original_fen, int8_t* labels,
                        int32_t* fixes, int32_t* n_fixes) {
    return guarded("cv_decode_positions", [&]() -> int {
        return impl_cv_decode_positions(probs, n_boards, flip, fen, original_fen, labels, fixes, n_fixes);
    });
}

int cv_extract_squares_u8_dev(cv_engine_t* eng, const uint8_t* images, int n, int h, int w_, const double* inv_dev,
                              uint8_t* squares, uint8_t* boards, void* stream) {
    return guarded("cv_extract_squares_u8_dev", [&]() -> int {
        return impl_cv_extract_squares_u8_dev(eng, images, n, h, w_, inv_dev, squares, boards, stream);
    });
}

}  // extern "C"
