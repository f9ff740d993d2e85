#include <algorithm>
#include <chrono>
#include <thread>
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

// continuous batching, the defaults: a runner without a session
void tts_generation_runner::stream_begin(const generation_configuration &) { TTS_ABORT("stream_begin: this runner has no continuous batching (stream_capacity() == 0)\n"); }
void tts_generation_runner::stream_submit(size_t, const std::string &) { TTS_ABORT("stream_submit: this runner has no continuous batching\n"); }
void tts_generation_runner::stream_step(std::vector<stream_result> &) { TTS_ABORT("stream_step: this runner has no continuous batching\n"); }
void tts_generation_runner::stream_end() {}

// any number of sentences through one session: rows freed by utterances that finish are refilled from the list at the next look-in point
void tts_generation_runner::generate_stream(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs,
                                            const generation_configuration & config) {
    if (stream_capacity() == 0) {   // no session: batches of at most batch_capacity(), one after the other
        batch_store_.assign(sentences.size(), {});
        outputs.assign(sentences.size(), tts_response{});
        const size_t cap = std::max<size_t>(1, std::min<size_t>(sentences.size(), batch_capacity()));
        for (size_t off = 0; off < sentences.size(); off += cap) {
            std::vector<std::string> part(sentences.begin() + off, sentences.begin() + std::min(sentences.size(), off + cap));
            std::vector<tts_response> out;
            generate_batch(part, out, config);
            for (size_t i = 0; i < part.size(); i++) {
                batch_store_[off + i].assign(out[i].data, out[i].data + out[i].n_outputs);
                outputs[off + i].data = batch_store_[off + i].data();
                outputs[off + i].n_outputs = out[i].n_outputs;
            }
        }
        return;
    }
    batch_store_.assign(sentences.size(), {});
    outputs.assign(sentences.size(), tts_response{});
    stream_begin(config);
    size_t next = 0;
    std::vector<stream_result> fin;
    while (next < sentences.size() || stream_live() > 0) {
        while (next < sentences.size() && stream_free() > 0) { stream_submit(next, sentences[next]); next++; }
        stream_step(fin);
        for (auto & f : fin) {
            batch_store_[f.ticket].assign(f.audio.data, f.audio.data + f.audio.n_outputs);
            outputs[f.ticket].data = batch_store_[f.ticket].data();
            outputs[f.ticket].n_outputs = f.audio.n_outputs;
        }
    }
    stream_end();
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
    // A session for the plumbing tests of the continuous batching (device_pool::process_stream, generate_stream) without a device: four rows,
    // an utterance of n characters "generates" for n look-in intervals and then yields generate()'s audio.
    struct row { size_t ticket; std::string text; size_t left; };
    std::vector<row>                st_rows;
    std::vector<std::vector<float>> st_audio;
    bool                            st_on = false;
    uint32_t stream_capacity() const override { return 4; }
    void     stream_begin(const generation_configuration & config) override {
        // test hook of the plumbing backend: a session that cannot be opened (a real runner refuses a cross-attention mismatch or a host-only sampler here)
        if (config.voice == "test:stream_begin-fails") TTS_ABORT("stream_begin: refused (test:stream_begin-fails)\n");
        st_rows.clear(); st_on = true;
    }
    uint32_t stream_free() const override { return st_on ? 4 - (uint32_t) st_rows.size() : 0; }
    uint32_t stream_live() const override { return (uint32_t) st_rows.size(); }
    void     stream_submit(size_t ticket, const std::string & sentence) override {
        if (!st_on || st_rows.size() >= 4) TTS_ABORT("stream_submit: no free row\n");
        st_rows.push_back(row{ticket, sentence, std::max<size_t>(1, sentence.size())});
    }
    void stream_step(std::vector<stream_result> & finished) override {
        finished.clear();
        st_audio.clear();
        std::this_thread::sleep_for(std::chrono::milliseconds(10));   // an interval takes a while: requests do arrive during a generation
        std::vector<row> keep;
        for (auto & r : st_rows) {
            if (--r.left > 0) { keep.push_back(r); continue; }
            tts_response resp;
            generate(r.text.c_str(), resp, generation_configuration{});
            st_audio.emplace_back(resp.data, resp.data + resp.n_outputs);
            stream_result f;
            f.ticket = r.ticket;
            finished.push_back(f);
        }
        for (size_t i = 0; i < finished.size(); i++) { finished[i].audio.data = st_audio[i].data(); finished[i].audio.n_outputs = st_audio[i].size(); }
        st_rows.swap(keep);
    }
    void stream_end() override { st_rows.clear(); st_on = false; }
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
