// C-ABI plumbing shared by all entry points: version, error string, device queries.
#include "pp_common.h"

#include <cstring>
#include <mutex>

namespace pp {

static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

// ---- explicit process-wide options (dev A/B switches of kernel selection). The library never reads the environment.
struct Option { const char* name; int value; };
static Option g_options[] = {
    {"panel", 1},              // 0: every GEMM / convolution on the 128 x 128 kernel (pp_gemm.hip)
    {"conv_halo", 1},          // 0: first tower convolution (bf16) on the implicit-GEMM kernel instead of pp_conv_halo.hip
    {"psplit_nst", 0},         // 2 / 3: force the two- / three-stage form of pp_panel_split.hip (0: by shape)
    {"panel_linear_mink", 0},  // > 0: shortest K of a bf16 Linear layer that takes the wide-tile kernel (0: built-in thresholds)
    {"psplit_bf16_conv", 0},   // 1: bf16 convolutions through pp_panel_split.hip instead of pp_panel_gemm.hip
    {"psplit_tap_inner", 1},          // K walk of the gathered convolutions of pp_panel_split.hip channel-block-major (taps inner): 1 = 3x3 convolutions, 2 = deconvolutions too, 0 = tap-major
    {"psplit_conv_weight_major", 1},  // 3x3 convolution tiles weight-set-major (one weight set per XCD at a time); only together with taps inner
    {"attn_dma", 1},           // 0: split-fp16 attention of 432-token sequences with the register-staged kernel of round 2
    {"conv_pool_split", 1},    // 0: split-fp16 first tower stage as conv + pooling launches instead of pooling in the conv epilogue
    {"decode_wgs_per_cu", 3},  // most workgroups per CU the decode kernel sizes its band buffer for (5 .. 1): more than 3 measured slower at bs 64 (4.25 workgroups per CU are balanced by the dispatcher, not by residency; smaller buffers mean more bands)
    {"linear_dma", 1},         // large split-fp16 Linear layers (pp_gemm): 1 = the twelve-wave 192 x 192 kernels (pp_linear_dma.hip), 0 = the wide-tile kernel
    {"linear_loop", 1},        // one-tile twelve-wave Linear kernel: 1 = one workgroup per CU walks a column of tiles, the next tile's first stages requested under this tile's epilogue; 0 = a workgroup per tile
    {"psplit_deconv_weight_major", 0},  // dev A/B: deconvolution tiles phase(weight set)-major per XCD instead of row panel -> phase (measured: DESIGN.md 4)
    {"ffn_pair", 1},           // twelve-wave feed-forward launch: 1 = hidden chunks in PAIRS that share every streamed x k-block (x streamed 6 instead of 12 times per launch; even chunk counts only), 0 = one chunk at a time
    {"psplit_tail", 1},        // split-fp16 Linear layers on the wide-tile kernel: 0 = no second launch on 128 x 192 tiles for the rows of a ragged last round
    {"wino_order", 8},         // pp_conv3x3_winograd_maxpool_relu: column tiles per 32-workgroup super tile (0: column tiles fastest over an XCD's run: 1.19 GB fetched per launch at bs 64; 8 = 4 row blocks x 8 column tiles: 0.72 GB, same launch time)
    {"ksplit_channels", 1},    // pp_conv3x3_splitk_slices: 0 = never the four channel-range slices of the split-fp16 wide-tile kernel (whole-tap slices only)
    {"qkv_attn_deep", 1},      // pp_qkv_attention_split (unfolded form) of a launch of at most two workgroups per CU: 1 = ring of four stages, one workgroup per CU; 0 = the two-stage kernel
    {"skinny_tile", 0},        // pp_skinny_linear: 10 RT + CT forces the tile shape (0: by the cost rule in pp_skinny.hip)
    {"qkv_attn_qsplit", 1},    // pp_qkv_attention_split, deep-ring form: two workgroups per (sequence, head) while 2 x sequences x heads <= 0.8 x CUs
    {"skinny_xcd_order", 1},   // pp_skinny_linear: tiles ordered so that an XCD touches 1 / xr of the rows and 1 / xc of the weights (0: row-major)
    {"ksplit9_below", 1024},   // pp_conv3x3_splitk_slices: tower stages with fewer output rows than this take nine K-slices (one tap each) instead of three
};

// launch tallies (pp_launch_count / pp_reset_launch_counts): name -> launches since the last reset
struct LaunchCount {
    char name[48];
    long long n;
};
static LaunchCount g_launches[96];
static int g_n_launches = 0;
static std::mutex g_launch_mutex;  // (ctypes releases the GIL: two host threads may launch at once; the tally is a few string compares under a lock)

void count_launch(const char* file_or_tag) {
    std::lock_guard<std::mutex> lock(g_launch_mutex);
    const char* base = file_or_tag;
    for (const char* c = file_or_tag; *c; ++c)
        if (*c == '/') base = c + 1;
    for (int i = 0; i < g_n_launches; ++i)
        if (std::strcmp(g_launches[i].name, base) == 0) {
            ++g_launches[i].n;
            return;
        }
    if (g_n_launches < (int)(sizeof(g_launches) / sizeof(g_launches[0]))) {
        std::snprintf(g_launches[g_n_launches].name, sizeof(g_launches[0].name), "%s", base);
        g_launches[g_n_launches++].n = 1;
    }
}

int option(const char* name) {
    for (const Option& o : g_options)
        if (std::strcmp(o.name, name) == 0) return o.value;
    return 0;
}

// One wavefront that brackets a stretch of wall time with the shader-clock counter (s_memtime) and the constant 100 MHz
// counter (s_memrealtime): cycles / ticks * 100 = the average shader clock in MHz while it ran. It ends when *stop becomes
// non-zero (a host-visible word the caller sets with a plain store) or after `ticks` of the 100 MHz counter.
__global__ void clock_probe_kernel(unsigned long long* out, const unsigned long long* stop, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    for (long long guard = 0; guard < (1ll << 21) && r1 - r0 < ticks; ++guard) {
        if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        __builtin_amdgcn_s_sleep(127);  // ~127 x 64 cycles asleep per poll: the wave issues next to nothing
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store(out + 0, c1 - c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(out + 1, r1 - r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace pp

extern "C" {

int pp_set_option(const char* name, int value) {
    using namespace pp;
    PP_REQUIRE(name, PP_ERR_INVALID_ARG, "pp_set_option: NULL name");
    for (Option& o : g_options)
        if (std::strcmp(o.name, name) == 0) {
            o.value = value;
            return PP_OK;
        }
    set_error("pp_set_option: unknown option '%s'", name);
    return PP_ERR_INVALID_ARG;
}

int pp_get_option(const char* name, int* value) {
    using namespace pp;
    PP_REQUIRE(name && value, PP_ERR_INVALID_ARG, "pp_get_option: NULL argument");
    for (const Option& o : g_options)
        if (std::strcmp(o.name, name) == 0) {
            *value = o.value;
            return PP_OK;
        }
    set_error("pp_get_option: unknown option '%s'", name);
    return PP_ERR_INVALID_ARG;
}

long long pp_launch_count(const char* kernel) {
    using namespace pp;
    if (!kernel) {
        set_error("pp_launch_count: NULL name");
        return 0;  // (a count: unknown and NULL names count 0; the message says why)
    }
    std::lock_guard<std::mutex> lock(g_launch_mutex);
    for (int i = 0; i < g_n_launches; ++i)
        if (std::strcmp(g_launches[i].name, kernel) == 0) return g_launches[i].n;
    return 0;
}

int pp_reset_launch_counts(void) {
    std::lock_guard<std::mutex> lock(pp::g_launch_mutex);
    pp::g_n_launches = 0;
    return PP_OK;
}

long long pp_workspace_bytes(int buffer, int index, const pp_plan_shape* sh) {
    using namespace pp;
    if (!sh || sh->n_img <= 0 || sh->n_tokens <= 0 || sh->embed <= 0 || sh->ffn <= 0 || sh->n_keypoints <= 0 || sh->feat_h <= 0 || sh->feat_w <= 0 ||
        (sh->prec != PP_PREC_BF16 && sh->prec != PP_PREC_F32 && sh->prec != PP_PREC_F16X3)) {
        set_error("pp_workspace_bytes: bad plan shape");
        return PP_ERR_INVALID_ARG;
    }
    const long long esz = sh->prec == PP_PREC_BF16 ? 2 : 4;  // operand element: bf16, or fp32 / split fp16 (4 bytes each)
    const long long M = (long long)sh->n_img * sh->n_tokens, E = sh->embed;
    auto tower_hw = [&](int j, long long& h, long long& w, long long& ph, long long& pw) {  // pooling schedule (4,3), (2,2), (2,2): probmap_head.py:264
        h = sh->feat_h;
        w = sh->feat_w;
        for (int i = 0;; ++i) {
            ph = i == 0 ? 4 : 2;
            pw = i == 0 ? 3 : 2;
            if (i == j) return;
            h /= ph;
            w /= pw;
        }
    };
    switch (buffer) {
        case PP_WS_PATCHES: return M * sh->patch_k * esz;
        case PP_WS_X: return M * E * 4;                     // residual stream, fp32
        case PP_WS_H: case PP_WS_FEAT: case PP_WS_ATT: case PP_WS_LN2: return M * E * esz;
        case PP_WS_QKV: return M * 3 * E * esz;
        case PP_WS_FFN: return M * (long long)sh->ffn * esz;
        case PP_WS_LOGITS: return (long long)sh->n_img * sh->n_keypoints * sh->heat_h * sh->heat_w * 4;
        case PP_WS_DECONV: {                                 // output of deconvolution `index`: (n_img, 2^(i+1) feat_h, 2^(i+1) feat_w, channels)
            if (index < 0 || index >= 8 || sh->deconv_channels <= 0) break;
            const long long up = 2ll << index;
            return (long long)sh->n_img * sh->feat_h * up * sh->feat_w * up * sh->deconv_channels * esz;
        }
        case PP_WS_TOWER: case PP_WS_TOWER_PARTIAL: case PP_WS_TOWER_POOLED: {
            if (index < 0 || index > 2) break;
            long long h, w, ph, pw;
            tower_hw(index, h, w, ph, pw);
            if (buffer == PP_WS_TOWER) return 4ll * sh->n_img * h * w * E * esz;
            if (buffer == PP_WS_TOWER_PARTIAL)   // K-slices as pp_conv3x3_splitk_slices says for this stage's rows, fp32
                return (long long)pp_conv3x3_splitk_slices(sh->prec, sh->n_img, (int)h, (int)w, sh->embed, sh->embed, 4) * 4 * sh->n_img * h * w * E * 4;
            return 4ll * sh->n_img * (h / ph) * (w / pw) * E * esz;
        }
        case PP_WS_LN_STATS: return E % 96 == 0 ? M * (E / 96) * 8 : 0;
        case PP_WS_WINOGRAD: {
            const long long n = pp_winograd_scratch_bytes(sh->n_img, sh->feat_h, sh->feat_w, sh->embed);
            return n > 0 ? n : 0;  // 0: the Winograd form does not apply to this shape (no scratch needed)
        }
        default: break;
    }
    set_error("pp_workspace_bytes: unknown buffer %d (index %d)", buffer, index);
    return PP_ERR_INVALID_ARG;
}

int pp_conv3x3_splitk_slices(int prec, int B, int H, int W, int Cin, int Cout, int groups) {
    const long long rows = (long long)B * H * W;
    // split-fp16, enough rows for whole 256 x 192 tiles: four channel quarters on the wide-tile kernel when that fills the chip
    // (4 x 4 tower stage at bs 64 with flip test: 8 row tiles x 2 column tiles x 4 towers x 4 slices = 256 workgroups of 27 stages)
    if (prec == PP_PREC_F16X3 && pp::option("panel") && pp::option("ksplit_channels") && pp::option("psplit_tap_inner") >= 1 && Cin % 128 == 0 &&
        Cout % 192 == 0 && ((rows + 255) / 256) * (Cout / 192) * groups * 4 >= 192)
        return 4;
    return rows < pp::option("ksplit9_below") ? 9 : 3;
}

int pp_abi_version(void) { return PP_ABI_VERSION; }

const char* pp_last_error(void) { return pp::g_last_error; }

const char* pp_status_string(int status) {
    switch (status) {
        case PP_OK: return "PP_OK";
        case PP_ERR_INVALID_ARG: return "PP_ERR_INVALID_ARG";
        case PP_ERR_UNSUPPORTED: return "PP_ERR_UNSUPPORTED";
        case PP_ERR_HIP: return "PP_ERR_HIP";
        case PP_ERR_WORKSPACE: return "PP_ERR_WORKSPACE";
        default: return "PP_ERR_UNKNOWN";
    }
}

int pp_clock_probe(unsigned long long* out_cycles_ticks, const unsigned long long* stop_flag, unsigned int max_microseconds, void* stream) {
    using namespace pp;
    PP_REQUIRE(out_cycles_ticks, PP_ERR_INVALID_ARG, "pp_clock_probe: NULL output");
    PP_REQUIRE(max_microseconds <= 5000000u, PP_ERR_INVALID_ARG, "pp_clock_probe: at most 5 s");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_cycles_ticks, stop_flag,
                       (unsigned long long)max_microseconds * 100ull);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

int pp_stream_create(void** stream_out) {
    PP_REQUIRE(stream_out != nullptr, PP_ERR_INVALID_ARG, "pp_stream_create: NULL argument");
    hipStream_t s = nullptr;
    PP_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = reinterpret_cast<void*>(s);
    return PP_OK;
}

int pp_stream_destroy(void* stream) {
    if (stream == nullptr) return PP_OK;
    PP_HIP_CHECK(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)));
    return PP_OK;
}

int pp_device_cu_count(void) {
    int dev = 0;
    PP_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    return prop.multiProcessorCount;
}

}  // extern "C"
