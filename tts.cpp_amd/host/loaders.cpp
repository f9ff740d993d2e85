#include <algorithm>
#include <cstdlib>
// loaders.cpp — architecture registry + runner_from_file (mirrors /root/reference/src/models/loaders.cpp:13-95)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "common.h"
#include "gguf.h"

static std::unordered_map<std::string, std::reference_wrapper<const tts_model_loader>> & registry() {
    static std::unordered_map<std::string, std::reference_wrapper<const tts_model_loader>> r;
    return r;
}

tts_model_loader::tts_model_loader(const char * arch, bool is_test) : arch{arch}, is_test{is_test} {
    registry().emplace(arch, std::ref(*this));
}

tts_generation_runner::tts_generation_runner(const tts_model_loader & loader) : loader{std::ref(loader)} {}
tts_generation_runner::~tts_generation_runner() = default;

tts_load_options & tts_thread_load_options() {
    static thread_local tts_load_options o;
    return o;
}
int tts_load_device() {
    const tts_load_options & o = tts_thread_load_options();
    if (o.device >= 0) return o.device;
    if (const char * d = getenv("TTS_HIP_DEVICE")) return atoi(d);
    return 0;
}
uint32_t tts_load_max_seqs() {
    const tts_load_options & o = tts_thread_load_options();
    if (o.max_seqs > 0) return (uint32_t) o.max_seqs;
    if (const char * ms = getenv("TTS_HIP_MAX_SEQS")) return (uint32_t) std::max(1, atoi(ms));
    return 1;
}

std::vector<std::string_view> tts_generation_runner::list_voices() {
    TTS_ABORT("The architecture '%s' does not support #list_voices.\n", loader.get().arch);
}
void tts_generation_runner::update_conditional_prompt(const char *, const char *) {
    TTS_ABORT("The architecture '%s' does not support update_conditional_prompt.\n", loader.get().arch);
}

void tts_generation_runner::generate_batch(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs,
                                           const generation_configuration & config) {
    batch_store_.assign(sentences.size(), {});
    outputs.assign(sentences.size(), tts_response{});
    for (size_t i = 0; i < sentences.size(); i++) {
        tts_response r;
        generate(sentences[i].c_str(), r, config);
        batch_store_[i].assign(r.data, r.data + r.n_outputs);
        outputs[i].data = batch_store_[i].data();
        outputs[i].n_outputs = r.n_outputs;
    }
}

// ---- "test:dummy": weightless plumbing backend (src/models/dummy/model.cpp:6-19) ----------------------
namespace {
struct dummy_loader_t final : tts_model_loader {
    dummy_loader_t() : tts_model_loader{"dummy", true} {}
    std::unique_ptr<tts_generation_runner> from_file(gguf_file *, int, bool, const generation_configuration &) const override;
};
const dummy_loader_t dummy_loader;

struct dummy_runner final : tts_generation_runner {
    std::vector<float> out;
    dummy_runner() : tts_generation_runner{dummy_loader} {}
    void assign_weight(const char *, const gguf_tensor_view &) override { TTS_ABORT("Assumed loader.is_test\n"); }
    void prepare_post_load() override { TTS_ABORT("Assumed loader.is_test\n"); }
    void generate(const char * sentence, tts_response & output, const generation_configuration &) override {
        // one second of an amplitude-modulated sine per input character, pitch keyed on the character
        constexpr size_t SR = 44100;
        sampling_rate = SR;
        const size_t n = strlen(sentence);
        out.assign(n * SR, 0.0f);
        for (size_t i = 0; i < n; i++) {
            const float wavelength = static_cast<float>(SR / M_PI / 2) / (200 + sentence[i]);
            float *     seg = out.data() + i * SR;
            for (size_t j = 0; j < SR; j++) seg[j] = sin(j * static_cast<float>(M_PI / SR)) * sin(j / wavelength);
        }
        output.data = out.data();
        output.n_outputs = out.size();
    }
};
std::unique_ptr<tts_generation_runner> dummy_loader_t::from_file(gguf_file *, int, bool, const generation_configuration &) const {
    return std::make_unique<dummy_runner>();
}
}  // namespace

void parler_register();
void orpheus_register();
void dia_register();
void kokoro_register();
[[maybe_unused]] static const bool loaders_registered = [] { parler_register(); orpheus_register(); dia_register(); kokoro_register(); return true; }();

std::unique_ptr<tts_generation_runner> runner_from_file(const char * fname, int n_threads,
                                                        const generation_configuration & config, bool cpu_only) {
    std::string_view sv{fname};
    if (sv.rfind("test:", 0) == 0) {
        const auto found = registry().find(std::string(sv.substr(5)));
        if (found == registry().end() || !found->second.get().is_test) TTS_ABORT("Unknown test model/backend %s\n", fname);
        return found->second.get().from_file(nullptr, 0, false, config);
    }
    std::string err;
    std::shared_ptr<gguf_file> meta = gguf_file::open(fname, err);
    if (!meta) TTS_ABORT("gguf_init_from_file failed for file %s: %s\n", fname, err.c_str());
    const gguf_value * arch = meta->get("general.architecture");
    if (!arch) TTS_ABORT("%s has no general.architecture key\n", fname);
    const auto found = registry().find(arch->s);
    if (found == registry().end()) TTS_ABORT("Unknown architecture %s\n", arch->s.c_str());
    const tts_model_loader & loader = found->second.get();
    std::unique_ptr<tts_generation_runner> runner = loader.from_file(meta.get(), n_threads, cpu_only, config);
    for (const gguf_tensor_view & t : meta->tensors) {
        if (!t.data || !*t.name) continue;
        runner->assign_weight(t.name, t);
    }
    runner->prepare_post_load();
    runner->buf = std::move(meta);
    return runner;
}
