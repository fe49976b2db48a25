// engine.h -- host side of libchessvision_hip.so: weight packing, workspace, layer plans.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "conv_igemm.h"
#include "cv_kernels.h"

namespace cv {

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
struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer();
    Status alloc(size_t n, bool zero);
    Status upload(const void* host, size_t n);
};

// a PHWC activation buffer sized for `cap` images
struct Activation {
    DeviceBuffer buf;
    int cap = 0, H = 0, W = 0, C = 0;
    int dt = kF16;
    Status create(int cap_, int h, int w, int c, int dt_);
    TensorRef ref(int n, int coff = 0, int c = -1) const {
        TensorRef t;
        t.base = buf.ptr; t.N = n; t.H = H; t.W = W; t.Cs = C; t.Coff = coff; t.C = c < 0 ? C - coff : c;
        return t;
    }
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
    int64_t pixels_hint = 0;                        // output pixels per launch at the engine's chunk size (tile choice)
    int kgroup = 8;                                 // input channels per K block (see engine.cpp: K ordering)
    int dt = kF16;
    DeviceBuffer w, scale, shift;
    DeviceBuffer w_small;                           // ct == 256 layers only: the same weights packed for 128-row tiles
    // koff tables are geometry dependent: keyed by (xWp, xCs, xCoff)
    struct KoffKey { int xWp, xCs, xCoff; bool operator<(const KoffKey& o) const {
        if (xWp != o.xWp) return xWp < o.xWp; if (xCs != o.xCs) return xCs < o.xCs; return xCoff < o.xCoff; } };
    struct KoffTab { DeviceBuffer chunks, bases; bool separable = false; };
    std::map<KoffKey, std::unique_ptr<KoffTab>> koff;

    // w_oihw: (cout, cin, k, k); scale/shift: (cout)
    Status build_conv(const std::string& name_, int dt_, const float* w_oihw, int cout_, int cin_, int k_,
                      int stride_, const float* scale_, const float* shift_, int cinPad_, int64_t pixels_hint_ = 0,
                      int out_hw_ = 0);
    // w_iohw: (cin, cout, 2, 2); bias: (cout)
    Status build_convT(const std::string& name_, int dt_, const float* w_iohw, int cin_, int cout_,
                       const float* bias, int64_t pixels_hint_ = 0);
    Status get_koff(const TensorRef& x, const int** chunks, const int** bases);
    int64_t macs_per_out_pixel() const { return shuffle ? (int64_t)cin * cout * 4 : (int64_t)cin * k * k * cout; }
};

struct ProfileEntry {
    std::string name;
    bool is_conv = false;
    double macs = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
};

class Engine {
  public:
    int device = 0;
    int dt = kF16;
    std::mutex mu;
    int unet_chunk = 64;          // images per pass: every layer still launches >= 256 workgroups of 256x256 / 128x256
    int resnet_chunk = 16384;      // squares per pass

    struct UNet;
    struct ResNet;
    std::unique_ptr<UNet> unet;
    std::unique_ptr<ResNet> resnet;

    DeviceBuffer scratch;                               // small per-call parameter blocks (homographies)

    // profiling (cv_profile_convs)
    bool profiling = false;
    std::vector<ProfileEntry> prof;

    Engine();
    ~Engine();

    struct Head { const float* w; const float* b; float* logits; uint8_t* mask; float thr; };
    // pool_out: also produce max_pool2d(y, 2) -- fused into the conv epilogue when the halo kernel runs the layer,
    // otherwise by the stand-alone pooling kernel right after it
    Status run_conv(ConvLayer& L, const TensorRef& x, const TensorRef& y, const TensorRef* res, bool relu,
                    hipStream_t s, const Head* head = nullptr, const TensorRef* pool_out = nullptr);
    void prof_begin(const std::string& name, bool is_conv, double macs, hipStream_t s);
    void prof_end(hipStream_t s);
    Status prof_collect();
    void prof_clear();
};

int choose_ct(int rows, int64_t pixels_hint, bool halo_ok);
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

}  // namespace cv
