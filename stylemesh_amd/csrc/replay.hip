// R1: the launch table of a step ("step program"). The Python host sequences a training step as ~70 (one UV level) to
// ~150 (four) calls into this library; for a small step - one 256 x 341 level: 1.1 ms of GPU work - the interpreter and
// ctypes cost of those calls (0.65 - 1.1 ms) is what bounds the step (512 views/s through the CLI
// against 863 in the bench loop). A step's calls are the same from step to step except for a handful of scalars (the
// optimizer's bias corrections, the lengths of a new view's active lists), so the host records them ONCE - function id +
// the argument values as 64-bit words - and afterwards replays the whole table with ONE call, patching the few words
// that change. hipGraph replay of the same step was measured in rounds 2 / 3 (no gain: a view change re-captures; graph
// launch of ~70 nodes costs the GPU front end what the eager launches cost) - this keeps eager launches and removes only
// the host interpreter from between them.
//
// Type safety: every entry point gets a thunk instantiated from its DECLARATION (include/stylemesh_hip.h), which
// converts the 64-bit words back to the parameter types; nothing is called through a cast function pointer. The
// reference has no counterpart (its step is Python over ATen, model/model.py:178-327).
#include <cstring>
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

template <typename T>
inline T word_to(uint64_t v) {
    if constexpr (std::is_pointer_v<T>) {
        return reinterpret_cast<T>(static_cast<uintptr_t>(v));
    } else if constexpr (std::is_same_v<T, float>) {
        float f;
        const uint32_t lo = static_cast<uint32_t>(v);
        std::memcpy(&f, &lo, sizeof(f));
        return f;
    } else if constexpr (std::is_same_v<T, double>) {
        double d;
        std::memcpy(&d, &v, sizeof(d));
        return d;
    } else {
        static_assert(std::is_integral_v<T>, "entry points take pointers, integers, floats and doubles only");
        return static_cast<T>(v);
    }
}

template <typename... A, size_t... I>
inline int invoke(int (*fn)(A...), const uint64_t* a, std::index_sequence<I...>) {
    return fn(word_to<A>(a[I])...);
}

template <auto Fn>
struct Thunk;
template <typename... A, int (*Fn)(A...)>
struct Thunk<Fn> {
    static constexpr int n_args = (int)sizeof...(A);
    static int call(const uint64_t* a) { return invoke(Fn, a, std::index_sequence_for<A...>{}); }
};

struct Entry {
    const char* name;
    int (*call)(const uint64_t*);
    int n_args;
};
#define SM_ENTRY(f) Entry{#f, &Thunk<&f>::call, Thunk<&f>::n_args}
// every entry point whose last parameter is the stream
const Entry ENTRIES[] = {
    SM_ENTRY(sm_tex_sample_fwd), SM_ENTRY(sm_tex_sample_fwd_grouped), SM_ENTRY(sm_tex_sample_bwd),
    SM_ENTRY(sm_tex_touch_flags), SM_ENTRY(sm_tex_scatter_planned), SM_ENTRY(sm_adam_fused),
    SM_ENTRY(sm_adam_hyper_step), SM_ENTRY(sm_step_begin), SM_ENTRY(sm_flags_or), SM_ENTRY(sm_clamp_sumsq),
    SM_ENTRY(sm_conv3x3), SM_ENTRY(sm_conv3x3_grouped),
    SM_ENTRY(sm_conv3x3_grouped_split2), SM_ENTRY(sm_fmap_amax), SM_ENTRY(sm_conv3x3_dgrad_c3),
    SM_ENTRY(sm_maxpool2x2_fwd), SM_ENTRY(sm_maxpool2x2_bwd_relu), SM_ENTRY(sm_conv3x3_dgrad_c3_grouped),
    SM_ENTRY(sm_maxpool2x2_fwd_grouped), SM_ENTRY(sm_maxpool2x2_bwd_relu_grouped), SM_ENTRY(sm_conv3x3_dgrad_c3_tiles),
    SM_ENTRY(sm_maxpool2x2_fwd_tiles), SM_ENTRY(sm_maxpool2x2_bwd_relu_tiles), SM_ENTRY(sm_maxpool2x2_fwd_codes_tiles),
    SM_ENTRY(sm_gram_masked), SM_ENTRY(sm_gram_masked_split), SM_ENTRY(sm_gram_masked_split_acc),
    SM_ENTRY(sm_gram_masked_split2_grouped), SM_ENTRY(sm_style_loss), SM_ENTRY(sm_gram_backward),
    SM_ENTRY(sm_gram_backward_split), SM_ENTRY(sm_style_loss_grouped), SM_ENTRY(sm_gram_backward_split2_grouped),
    SM_ENTRY(sm_mse_masked), SM_ENTRY(sm_copy_floats), SM_ENTRY(sm_zero_floats),
};
constexpr int N_ENTRIES = (int)(sizeof(ENTRIES) / sizeof(ENTRIES[0]));

}  // namespace

extern "C" {

int sm_copy_floats(float* dst, const float* src, size_t n, void* stream) {
    if (n == 0) return 0;
    return (int)hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
}

int sm_zero_floats(float* dst, size_t n, void* stream) {
    if (n == 0) return 0;
    return (int)hipMemsetAsync(dst, 0, n * sizeof(float), (hipStream_t)stream);
}

int sm_call_id(const char* name) {
    if (name == nullptr) return -1;
    for (int i = 0; i < N_ENTRIES; ++i)
        if (std::strcmp(ENTRIES[i].name, name) == 0) return i;
    return -1;
}

int sm_call_n_args(int id) { return id >= 0 && id < N_ENTRIES ? ENTRIES[id].n_args : -1; }

int sm_call_replay(const sm_call* calls, int n, void* stream, int* failed_index) {
    if (calls == nullptr || n < 0) return (int)hipErrorInvalidValue;
    for (int i = 0; i < n; ++i) {
        const sm_call& c = calls[i];
        if (c.skip) continue;
        if (c.fn < 0 || c.fn >= N_ENTRIES || c.n_args != ENTRIES[c.fn].n_args || c.n_args > SM_CALL_MAX_ARGS) {
            if (failed_index) *failed_index = i;
            return (int)hipErrorInvalidValue;
        }
        uint64_t a[SM_CALL_MAX_ARGS];
        std::memcpy(a, c.args, sizeof(uint64_t) * (size_t)c.n_args);
        a[c.n_args - 1] = static_cast<uint64_t>(reinterpret_cast<uintptr_t>(stream));   // the caller's stream, whatever was recorded
        if (const int rc = ENTRIES[c.fn].call(a)) {
            if (failed_index) *failed_index = i;
            return rc;
        }
    }
    return 0;
}

}  // extern "C"
