/* consumer.c -- a plain C program against include/chessvision_hip.h: what a cgo / JNI / C++ host of the reference's hot path
 * would do, with no Python in the call path.  Built by the tests with gcc (tests/test_c_abi_consumer.py):
 *
 *   consumer host                      host-only entry points (no GPU): ABI version, mask -> quadrangle, probabilities -> FEN
 *   consumer gpu <blob> <squares.bin>  load a ResNet-18 state dict from a flat blob, classify squares on device 0 through
 *                                      cv_resnet18_forward_u8, print the 13 probabilities of every square
 *
 * blob format (written by the test): int32 n_params, then per parameter: int32 name_len, name bytes, int32 ndim,
 * int64 shape[4], float32 data.  squares.bin: int32 n, then n x 64 x 64 uint8.
 * Device memory comes from the HIP runtime's C API (the reference-side host owns its buffers; the library never frees them). */
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "chessvision_hip.h"

#ifdef WITH_GPU
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#endif

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != CV_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, cv_last_error()); return 1; } \
    } while (0)

static int host_mode(void) {
    printf("abi %d\n", cv_abi_version());
    /* a 256x256 mask with one bright convex quadrilateral */
    static uint8_t mask[256 * 256];
    for (int y = 60; y < 200; ++y)
        for (int x = 50 + (y - 60) / 10; x < 210 - (y - 60) / 14; ++x) mask[y * 256 + x] = 255;
    int32_t quad[8];
    int found = 0;
    CHECK(cv_find_quadrangle(mask, 256, 256, quad, &found));
    printf("quad %d", found);
    for (int i = 0; i < 8; ++i) printf(" %d", quad[i]);
    printf("\n");
    /* the start position as one-hot probabilities, a8..h1; then a pawn forced onto a8 with a rook as runner-up */
    const char* order = "BKNPQRbknpqrf";
    const char* board = "rnbqkbnrppppppppffffffffffffffffffffffffffffffffPPPPPPPPRNBQKBNR";
    static float probs[2 * 64 * 13];
    for (int b = 0; b < 2; ++b)
        for (int s = 0; s < 64; ++s) probs[(b * 64 + s) * 13 + (int)(strchr(order, board[s]) - order)] = 1.0f;
    float* a8 = probs + 64 * 13;
    memset(a8, 0, 13 * sizeof(float));
    a8[9] = 0.6f; a8[11] = 0.3f; a8[12] = 0.1f;          /* 'p' 0.6, 'r' 0.3, empty 0.1 */
    char fen[2 * 72], orig[2 * 72];
    int8_t labels[2 * 64];
    int32_t fixes[2 * 16 * 4], n_fixes = 0;
    CHECK(cv_decode_positions(probs, 2, 0, fen, orig, labels, fixes, &n_fixes));
    printf("fen0 %s\nfen1 %s\norig1 %s\nfixes %d", fen, fen + 72, orig + 72, (int)n_fixes);
    for (int i = 0; i < n_fixes * 4; ++i) printf(" %d", (int)fixes[i]);
    printf("\n");
    /* quadrangle (scaled to a 512-row photo) -> the warp's matrices, OpenCV's order of operations: forward maps the corners onto
       (0,0), (512,0), (512,512), (0,512); inverse * forward = identity */
    float corners[8];
    for (int i = 0; i < 8; ++i) corners[i] = (float)quad[i] * 2.0f;
    double fwd[9], inv[9];
    CHECK(cv_board_homographies(corners, 1, 512, 512, fwd, inv));
    double worst = 0.0;
    const double want[8] = {0, 0, 512, 0, 512, 512, 0, 512};
    for (int i = 0; i < 4; ++i) {
        const double x = corners[2 * i], y = corners[2 * i + 1];
        const double w = fwd[6] * x + fwd[7] * y + fwd[8];
        const double u = (fwd[0] * x + fwd[1] * y + fwd[2]) / w, v = (fwd[3] * x + fwd[4] * y + fwd[5]) / w;
        const double du = u - want[2 * i], dv = v - want[2 * i + 1];
        if (du * du > worst) worst = du * du;
        if (dv * dv > worst) worst = dv * dv;
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += inv[r * 3 + k] * fwd[k * 3 + c];
            const double d = acc - (r == c ? 1.0 : 0.0);
            if (d * d > worst) worst = d * d;
        }
    printf("homography_err2 %.3g\n", worst);
    /* error path: status + message, never an abort */
    int rc = cv_find_quadrangle(NULL, 256, 256, quad, &found);
    printf("null_mask rc=%d msg=%s\n", rc, cv_last_error());
    return 0;
}

#ifdef WITH_GPU
/* a state dict from a flat file: int32 count, then per entry int32 name length, name, int32 ndim, 4 x int64 shape, float32 data */
static cv_param_t* read_blob(const char* blob_path, int32_t* n_out) {
    FILE* f = fopen(blob_path, "rb");
    if (!f) { perror(blob_path); return NULL; }
    int32_t n_params = 0;
    if (fread(&n_params, 4, 1, f) != 1) return NULL;
    cv_param_t* table = (cv_param_t*)calloc((size_t)n_params, sizeof(cv_param_t));
    for (int i = 0; i < n_params; ++i) {
        int32_t len = 0, ndim = 0;
        if (fread(&len, 4, 1, f) != 1) return NULL;
        char* name = (char*)calloc((size_t)len + 1, 1);
        if (fread(name, 1, (size_t)len, f) != (size_t)len || fread(&ndim, 4, 1, f) != 1) return NULL;
        if (fread(table[i].shape, 8, 4, f) != 4) return NULL;
        size_t numel = 1;
        for (int d = 0; d < ndim; ++d) numel *= (size_t)table[i].shape[d];
        float* data = (float*)malloc(numel * sizeof(float));
        if (fread(data, sizeof(float), numel, f) != numel) return NULL;
        table[i].name = name; table[i].data = data; table[i].ndim = ndim;
    }
    fclose(f);
    *n_out = n_params;
    return table;
}

/* the reference's per-image entry point from plain C: one photo (int32 h, int32 w, h*w*3 bytes) -> FEN, one call */
static int image_mode(const char* unet_blob, const char* resnet_blob, const char* image_path) {
    int32_t nu = 0, nr = 0;
    cv_param_t* tu = read_blob(unet_blob, &nu);
    cv_param_t* tr = read_blob(resnet_blob, &nr);
    if (!tu || !tr) return 1;
    FILE* f = fopen(image_path, "rb");
    if (!f) { perror(image_path); return 1; }
    int32_t hw[2];
    if (fread(hw, 4, 2, f) != 2) return 1;
    uint8_t* image = (uint8_t*)malloc((size_t)hw[0] * hw[1] * 3);
    if (fread(image, 3, (size_t)hw[0] * hw[1], f) != (size_t)hw[0] * hw[1]) return 1;
    fclose(f);
    cv_engine_t* eng = NULL;
    CHECK(cv_engine_create(0, CV_PREC_F16X3, &eng));
    CHECK(cv_load_unet(eng, tu, nu));
    CHECK(cv_load_resnet18(eng, tr, nr));
    static float probs[64 * 13], logits[256 * 256];
    static uint8_t mask[256 * 256], board[512 * 512], crops[64 * 64 * 64];
    cv_image_result_t res;
    memset(&res, 0, sizeof(res));
    res.logits = logits; res.mask = mask; res.board = board; res.probabilities = probs; res.squares = crops;   /* squares: ABI 5 */
    for (int rep = 0; rep < 3; ++rep)                                         /* eager, graph capture, graph replay */
        CHECK(cv_process_image_v2(eng, eng, image, hw[0], hw[1], 0.5f, 0, 1, &res, sizeof(res), NULL));   /* ABI 6: the struct size travels */
    printf("pi_found %d\npi_fen %s\npi_orig %s\npi_quad", (int)res.found, res.fen, res.original_fen);
    for (int i = 0; i < 8; ++i) printf(" %.9g", res.quadrangle[i]);
    unsigned long long msum = 0, bsum = 0;
    for (int i = 0; i < 256 * 256; ++i) msum += mask[i];
    for (int i = 0; i < 512 * 512; ++i) bsum += (unsigned long long)board[i] * (unsigned)(i % 251 + 1);
    unsigned long long csum = 0;
    for (int i = 0; i < 64 * 64 * 64; ++i) csum += (unsigned long long)crops[i] * (unsigned)(i % 253 + 1);
    printf("\npi_mask_sum %llu\npi_board_checksum %llu\npi_squares_checksum %llu\npi_probs", msum, bsum, csum);
    for (int i = 0; i < 64 * 13; ++i) printf(" %.9g", probs[i]);
    printf("\n");
    /* A binary built against the ABI 3 / 4 header: its struct ends after n_fixes, and whatever follows it in memory is NOT a `squares`
     * pointer.  The old entry point must neither read nor write there (ABI 5 read it unconditionally and would have copied
     * 256 KB to a garbage address). */
    struct { unsigned char body[offsetof(cv_image_result_t, squares)]; unsigned char after[64]; } old;
    memset(&old, 0, sizeof(old));
    memset(old.after, 0xAB, sizeof(old.after));
    cv_image_result_t* legacy = (cv_image_result_t*)(void*)&old;
    static float probs2[64 * 13];
    legacy->logits = logits; legacy->mask = mask; legacy->board = board; legacy->probabilities = probs2;
    CHECK(cv_process_image(eng, eng, image, hw[0], hw[1], 0.5f, 0, 1, legacy, NULL));
    int intact = 1;
    for (size_t i = 0; i < sizeof(old.after); ++i) intact &= old.after[i] == 0xAB;
    printf("pi_legacy_ok %d found %d same_fen %d same_probs %d\n", intact, (int)legacy->found, strcmp(legacy->fen, res.fen) == 0,
           memcmp(probs, probs2, sizeof(probs)) == 0);
    int rc_small = cv_process_image_v2(eng, eng, image, hw[0], hw[1], 0.5f, 0, 1, &res, 16, NULL);
    printf("pi_short_struct rc=%d\n", rc_small);
    int rc = cv_process_image_v2(eng, eng, NULL, 512, 512, 0.5f, 0, 0, &res, sizeof(res), NULL);
    printf("pi_null_image rc=%d msg=%s\n", rc, cv_last_error());
    CHECK(cv_engine_destroy(eng));
    return 0;
}

static int gpu_mode(const char* blob_path, const char* squares_path) {
    int32_t n_params = 0;
    cv_param_t* table = read_blob(blob_path, &n_params);
    if (!table) return 1;
    FILE* f = fopen(squares_path, "rb");
    if (!f) { perror(squares_path); return 1; }
    int32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) return 1;
    uint8_t* squares = (uint8_t*)malloc((size_t)n * 4096);
    if (fread(squares, 4096, (size_t)n, f) != (size_t)n) return 1;
    fclose(f);

    cv_engine_t* eng = NULL;
    CHECK(cv_engine_create(0, CV_PREC_F16X3, &eng));
    CHECK(cv_load_resnet18(eng, table, n_params));
    uint8_t* d_sq = NULL;
    float* d_probs = NULL;
    if (hipMalloc((void**)&d_sq, (size_t)n * 4096) != hipSuccess || hipMalloc((void**)&d_probs, (size_t)n * 13 * sizeof(float)) != hipSuccess) return 1;
    if (hipMemcpy(d_sq, squares, (size_t)n * 4096, hipMemcpyHostToDevice) != hipSuccess) return 1;
    CHECK(cv_resnet18_forward_u8(eng, d_sq, n, d_probs, NULL));               /* NULL = the default stream */
    CHECK(cv_engine_numeric_status(eng, NULL));                               /* synchronises; names a layer on overflow */
    float* probs = (float*)malloc((size_t)n * 13 * sizeof(float));
    if (hipMemcpy(probs, d_probs, (size_t)n * 13 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int i = 0; i < n; ++i) {
        printf("probs");
        for (int c = 0; c < 13; ++c) printf(" %.9g", probs[i * 13 + c]);
        printf("\n");
    }
    size_t ws = 0;
    CHECK(cv_engine_workspace_bytes(eng, &ws));
    printf("workspace %zu\n", ws);
    int rc = cv_unet_forward(eng, NULL, 1, NULL, NULL);                       /* model not loaded: a status, not a crash */
    printf("unet_not_loaded rc=%d msg=%s\n", rc, cv_last_error());
    (void)hipFree(d_sq); (void)hipFree(d_probs);
    CHECK(cv_engine_destroy(eng));
    size_t trimmed = 0, again = 1;                                            /* the destroyed engine's blocks were cached for the next one */
    CHECK(cv_trim_memory(&trimmed));
    CHECK(cv_trim_memory(&again));
    printf("trimmed %zu then %zu\n", trimmed, again);
    return 0;
}
#endif

int main(int argc, char** argv) {
    if (argc >= 2 && strcmp(argv[1], "host") == 0) return host_mode();
#ifdef WITH_GPU
    if (argc >= 4 && strcmp(argv[1], "gpu") == 0) return gpu_mode(argv[2], argv[3]);
    if (argc >= 5 && strcmp(argv[1], "image") == 0) return image_mode(argv[2], argv[3], argv[4]);
#endif
    fprintf(stderr, "usage: consumer host | consumer gpu <state.blob> <squares.bin> | consumer image <unet.blob> <resnet.blob> <image.bin>\n");
    return 2;
}
