// tts_c.cpp — C wrapper of the C++ runner API (include/tts_c.h) for language bindings and tests.
#include "../../include/tts_c.h"

#include <cstring>
#include <stdexcept>
#include <string>

#include "common.h"
#include "gguf.h"
#include "dia_runner.h"
#include "kokoro_runner.h"
#include "orpheus_runner.h"
#include "parler_runner.h"
#include "sampler.h"
#include "tokenizer.h"

#include <atomic>
extern std::atomic<bool> g_tts_throw_on_abort;
static thread_local std::string g_c_err;

// memcpy with a zero count still requires valid pointers (an empty vector's data() may be null)
static void copy_u32(uint32_t * dst, const uint32_t * src, size_t n) {
    if (dst && src && n) memcpy(dst, src, n * 4);
}

static generation_configuration to_cfg(const tts_c_config * c) {
    if (!c) return generation_configuration{};
    generation_configuration g{c->voice ? c->voice : "", c->top_k, c->temperature, c->repetition_penalty, c->use_cross_attn != 0,
                               "", c->max_tokens, c->top_p, c->sample != 0};
    g.seed = c->seed;
    return g;
}

extern "C" {

const char * tts_c_last_error(void) { return g_c_err.c_str(); }

void tts_c_default_config(tts_c_config * c) {
    const generation_configuration g{};
    c->voice = nullptr; c->top_k = g.top_k; c->temperature = g.temperature; c->repetition_penalty = g.repetition_penalty;
    c->use_cross_attn = g.use_cross_attn; c->max_tokens = g.max_tokens; c->top_p = g.top_p; c->sample = g.sample; c->seed = 0;
}

tts_c_runner * tts_c_runner_from_file(const char * path, int n_threads, const tts_c_config * cfg, int cpu_only) {
    g_tts_throw_on_abort = true;
    // the load options are for THIS load only: whatever happens, the next load of the thread starts from the defaults again (a share_with
    // pointer left behind would make a later, unrelated load reach into a runner that may be gone)
    struct reset_options { ~reset_options() { tts_thread_load_options() = tts_load_options{}; } } reset;
    try {
        return (tts_c_runner *) runner_from_file(path, n_threads, to_cfg(cfg), cpu_only != 0).release();
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return nullptr;
    }
}

int tts_c_generate(tts_c_runner * r, const char * text, const tts_c_config * cfg, const float ** data, size_t * n_outputs) {
    g_tts_throw_on_abort = true;
    try {
        tts_response resp;
        ((tts_generation_runner *) r)->generate(text, resp, to_cfg(cfg));
        *data = resp.data;
        *n_outputs = resp.n_outputs;
        return 0;
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}

int tts_c_generate_batch(tts_c_runner * r, const char * const * texts, int n, const tts_c_config * cfg, const float ** data, size_t * n_outputs) {
    g_tts_throw_on_abort = true;
    try {
        auto * p = (tts_generation_runner *) r;
        std::vector<std::string> s(texts, texts + n);
        std::vector<tts_response> out;
        p->generate_batch(s, out, to_cfg(cfg));
        for (int i = 0; i < n; i++) { data[i] = out[(size_t) i].data; n_outputs[i] = out[(size_t) i].n_outputs; }
        return 0;
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}

int tts_c_generate_stream(tts_c_runner * r, const char * const * texts, int n, const tts_c_config * cfg, const float ** data, size_t * n_outputs) {
    g_tts_throw_on_abort = true;
    try {
        auto * p = (tts_generation_runner *) r;
        std::vector<std::string> s(texts, texts + n);
        std::vector<tts_response> out;
        p->generate_stream(s, out, to_cfg(cfg));
        for (int i = 0; i < n; i++) { data[i] = out[(size_t) i].data; n_outputs[i] = out[(size_t) i].n_outputs; }
        return 0;
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}

int tts_c_update_conditional_prompt(tts_c_runner * r, const char * text_encoder_path, const char * prompt) {
    g_tts_throw_on_abort = true;
    try {
        ((tts_generation_runner *) r)->update_conditional_prompt(text_encoder_path, prompt);
        return 0;
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}

void tts_c_set_load_options(int device, int max_seqs, int declare_only) {
    tts_load_options & o = tts_thread_load_options();
    o = tts_load_options{};
    o.device = device;
    o.max_seqs = max_seqs;
    o.declare_only = declare_only != 0;
}
void tts_c_set_load_options_ex(int device, int max_seqs, int declare_only, tts_c_runner * share_with) {
    tts_c_set_load_options(device, max_seqs, declare_only);
    tts_thread_load_options().share_with = (const tts_generation_runner *) share_with;
}
void * tts_c_runner_device_context(tts_c_runner * r) { return r ? ((tts_generation_runner *) r)->device_context() : nullptr; }
int tts_c_runner_tokenize(tts_c_runner * r, const char * text, uint32_t * out, int cap) {
    g_tts_throw_on_abort = true;
    try {
        auto * p = dynamic_cast<parler_runner *>((tts_generation_runner *) r);
        if (!p) { g_c_err = "tts_c_runner_tokenize: only the Parler runner exposes its tokenizer"; return -1; }
        std::vector<uint32_t> ids;
        p->tokenizer->tokenize(text, ids);
        ids.push_back(p->tokenizer->eos_token);
        const int n = (int) ids.size();
        copy_u32(out, ids.data(), (size_t) (n < cap ? n : cap));
        return n;
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}
float        tts_c_sampling_rate(tts_c_runner * r) { return ((tts_generation_runner *) r)->sampling_rate; }
const char * tts_c_arch(tts_c_runner * r) { return ((tts_generation_runner *) r)->loader.get().arch; }
void         tts_c_free(tts_c_runner * r) { delete (tts_generation_runner *) r; }

int tts_c_last_tokens(tts_c_runner * r, int which, uint32_t * out, int cap) {
    static const std::vector<uint32_t> none;
    const std::vector<uint32_t> * vp = nullptr;
    if (auto * p = dynamic_cast<parler_runner *>((tts_generation_runner *) r))
        vp = which == 0 ? &p->last_prompt_tokens : (which == 2 ? &p->last_conditional_tokens : &p->last_output_tokens);
    else if (auto * o = dynamic_cast<orpheus_runner *>((tts_generation_runner *) r))
        vp = which == 0 ? &o->last_prompt_tokens : (which == 1 ? &o->last_output_tokens : &none);
    else if (auto * d = dynamic_cast<dia_runner *>((tts_generation_runner *) r))
        vp = which == 0 ? &d->last_prompt_tokens : (which == 1 ? &d->last_output_tokens : &none);
    else if (auto * k = dynamic_cast<kokoro_runner *>((tts_generation_runner *) r))
        vp = which == 0 ? &k->last_prompt_tokens : &none;
    if (!vp) { g_c_err = "runner keeps no token record"; return -1; }
    const std::vector<uint32_t> & v = *vp;
    const int n = (int) v.size();
    copy_u32(out, v.data(), (size_t) (n < cap ? n : cap));
    return n;
}

int tts_c_tokenize(const char * gguf_path, const char * text, uint32_t * out, int cap) {
    g_tts_throw_on_abort = true;
    try {
        std::string err;
        auto f = gguf_file::open(gguf_path, err);
        if (!f) { g_c_err = err; return -1; }
        std::vector<uint32_t> ids;
        if (f->get("tokenizer.ggml.merges")) {   // byte-pair vocabulary (Orpheus): the bare token ids, no framing
            std::unique_ptr<bpe_tokenizer> t(bpe_tokenizer_from_gguf(*f));
            t->tokenize(text, ids);
        } else {
            std::unique_ptr<unigram_tokenizer> t(unigram_tokenizer_from_gguf(*f));
            t->tokenize(text, ids);
            ids.push_back(t->eos_token);  // batch_from_sentence appends EOS (model.cpp:478)
        }
        const int n = (int) ids.size();
        copy_u32(out, ids.data(), (size_t) (n < cap ? n : cap));
        return n;
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}

int tts_c_sampler_sample(const tts_c_sampler_cfg * c, const int32_t * last_ids, const uint32_t * counts, float * logits,
                         const float * uniforms, uint32_t * out) {
    try {
    sampler s;
    s.n_output_heads = c->n_output_heads; s.vocab_size = c->vocab_size; s.top_k = c->top_k; s.temperature = c->temperature;
    s.top_p = c->top_p; s.repetition_penalty = c->repetition_penalty; s.do_sample = c->do_sample != 0; s.seed = c->seed;
    s.reset();
    if (last_ids && c->repetition_penalty != 1.0f) {
        s.last_token_ids.assign(last_ids, last_ids + c->n_output_heads);
        s.repetition_counts.assign(counts, counts + c->n_output_heads);
    }
    std::vector<uint32_t> o;
    if (uniforms) s.sample_with_uniforms(logits, uniforms, o);
    else s.sample(logits, o);
    copy_u32(out, o.data(), o.size());
    return (int) o.size();
    } catch (const std::exception & e) {   // no exception may cross the C boundary
        g_c_err = e.what();
        return -1;
    }
}

int tts_c_gguf_summary(const char * path, uint64_t * n_tensors, uint64_t * n_kv, uint64_t * data_offset, char * arch, int arch_cap) {
    try {
    std::string err;
    auto f = gguf_file::open(path, err);
    if (!f) { g_c_err = err; return -1; }
    *n_tensors = f->tensors.size();
    *n_kv = f->kv.size();
    *data_offset = f->data_offset;
    if (auto a = f->get("general.architecture")) snprintf(arch, (size_t) arch_cap, "%s", a->s.c_str());
    else if (arch_cap) arch[0] = 0;
    return 0;
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}

int tts_c_gguf_tensor(const char * path, int index, char * name, int name_cap, int * type, int64_t ne[4], uint64_t * checksum) {
    try {
    std::string err;
    auto f = gguf_file::open(path, err);
    if (!f) { g_c_err = err; return -1; }
    if (index < 0 || index >= (int) f->tensors.size()) { g_c_err = "tensor index out of range"; return -1; }
    const gguf_tensor_view & t = f->tensors[(size_t) index];
    snprintf(name, (size_t) name_cap, "%s", t.name);
    *type = t.type;
    for (int d = 0; d < 4; d++) ne[d] = t.ne[d];
    uint64_t h = 1469598103934665603ull;  // FNV-1a over the tensor bytes
    const uint8_t * p = (const uint8_t *) t.data;
    for (size_t i = 0; i < t.nbytes; i++) { h ^= p[i]; h *= 1099511628211ull; }
    *checksum = h;
    return 0;
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}

}  // extern "C"

// ---- Dia host logic (host/dia_runner.h) -----------------------------------------------------------------------
extern "C" int tts_c_dia_tokenize(const char * sentence, uint32_t max_ctx, uint32_t * out) {
    g_tts_throw_on_abort = true;
    try {
        dia_hparams hp;
        hp.max_encoder_context_length = max_ctx;
        std::vector<uint32_t> t;
        const uint32_t n = dia_tokenize_sentence(hp, sentence, t);
        copy_u32(out, t.data(), (size_t) max_ctx);
        return (int) n;
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}

extern "C" int tts_c_dia_check_stopping(uint32_t * ids, uint32_t eos, uint32_t pad, uint32_t max_delay, uint32_t current_position,
                                        uint32_t max_generation_size, int * delay_steps) {
    dia_hparams hp;
    hp.eos_token_id = eos; hp.pad_token_id = pad; hp.max_delay = max_delay;
    std::vector<uint32_t> v(ids, ids + hp.delay_pattern.size());
    const bool stop = dia_check_stopping(hp, v, current_position, max_generation_size, *delay_steps);
    copy_u32(ids, v.data(), v.size());
    return stop ? 1 : 0;
}

extern "C" int64_t tts_c_dia_adjust_output_tokens(const uint32_t * tokens, uint64_t n_ids, uint32_t audio_vocab, uint32_t max_delay, uint32_t * filtered) {
    dia_hparams hp;
    hp.audio_vocab_size = audio_vocab; hp.max_delay = max_delay;
    std::vector<uint32_t> in(tokens, tokens + n_ids), out;
    dia_adjust_output_tokens(hp, in, out);
    copy_u32(filtered, out.data(), out.size());
    return (int64_t) out.size();
}

// ---- Kokoro host logic (host/kokoro_runner.h) -------------------------------------------------------------------
std::vector<std::vector<uint32_t>> kokoro_clause_chunks(const kokoro_hparams & hp, const single_pass_tokenizer & tok, const std::string & phonemes);

extern "C" int tts_c_single_pass_tokenize(const char * const * vocab, int n_vocab, const char * text, uint32_t * out, int cap) {
    single_pass_tokenizer t(std::vector<std::string>(vocab, vocab + n_vocab));
    std::vector<uint32_t> ids;
    t.tokenize(text, ids);
    copy_u32(out, ids.data(), (size_t) std::min<int>((int) ids.size(), cap));
    return (int) ids.size();
}

extern "C" int tts_c_kokoro_chunks(const char * const * vocab, int n_vocab, const char * phonemes, uint32_t max_ctx, uint32_t space_token_id, uint32_t * out, int cap) {
    single_pass_tokenizer t(std::vector<std::string>(vocab, vocab + n_vocab));
    kokoro_hparams hp;
    hp.max_context_length = max_ctx;
    hp.space_token_id = space_token_id;
    std::vector<uint32_t> flat;
    for (const auto & ch : kokoro_clause_chunks(hp, t, phonemes)) {
        flat.push_back((uint32_t) ch.size());
        flat.insert(flat.end(), ch.begin(), ch.end());
    }
    copy_u32(out, flat.data(), (size_t) std::min<int>((int) flat.size(), cap));
    return (int) flat.size();
}

extern "C" uint32_t tts_c_minstd0_jump(uint32_t state, uint64_t k) { return minstd0_jump(state, k); }
extern "C" uint32_t tts_c_minstd0_uniform(uint32_t state, uint64_t n, float * out, uint32_t threads) {
    minstd0_draw_uniform(state, (size_t) n, out, threads);
    return minstd0_jump(state, n);
}

// ---- quantize tool (host/quantize.h) -------------------------------------------------------------------------
#include "quantize.h"

static quantization_params to_qp(const tts_c_quantization_params * p) {
    quantization_params q;
    q.n_threads = p->n_threads ? p->n_threads : 1;
    q.quantize_type = p->quantize_type;
    q.quantize_output_heads = p->quantize_output_heads != 0;
    q.quantize_text_embeddings = p->quantize_text_embeddings != 0;
    q.quantize_cross_attn_kv = p->quantize_cross_attn_kv != 0;
    q.convert_dac_to_f16 = p->convert_dac_to_f16 != 0;
    q.convert_non_quantizable_to_f16 = p->convert_non_quantizable_to_f16 != 0;
    return q;
}

extern "C" int tts_c_quantize_gguf(const char * ifile, const char * ofile, const tts_c_quantization_params * params) {
    g_tts_throw_on_abort = true;
    try {
        quantize_gguf(ifile, ofile, to_qp(params));
        return 0;
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}

extern "C" int tts_c_quantize_decision(const char * arch, const char * tensor_name, int n_dims, const tts_c_quantization_params * params) {
    g_tts_throw_on_abort = true;
    try {
        return quantize_decision(arch, tensor_name, n_dims, to_qp(params));
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}

extern "C" int64_t tts_c_quantize_rows(int type, const float * src, void * dst, int64_t n_per_row, int64_t nrows, uint32_t n_threads) {
    g_tts_throw_on_abort = true;
    try {
        return (int64_t) quantize_rows(type, src, dst, n_per_row, nrows, n_threads ? n_threads : 1);
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return -1;
    }
}

// ---- device pool (host/device_pool.h) ------------------------------------------------------------------------
#include "device_pool.h"

struct tts_c_pool {
    std::unique_ptr<device_pool> pool;
    std::map<int, std::shared_ptr<pool_task>> held;  // tasks whose audio the caller still reads
    std::mutex mutex;
};

static thread_local std::string g_pool_text_encoder;
void tts_c_pool_set_text_encoder(const char * path) { g_pool_text_encoder = path ? path : ""; }
static thread_local bool g_pool_continuous = false;
void tts_c_pool_set_continuous(int on) { g_pool_continuous = on != 0; }
static thread_local int g_pool_yield_ms = 2000;
void tts_c_pool_set_continuous_yield_ms(int ms) { g_pool_yield_ms = ms < 0 ? 0 : ms; }
uint64_t tts_c_pool_admitted_in_flight(tts_c_pool * p) { return p->pool->stats().admitted_in_flight; }

int tts_c_pool_conditional_prompt(tts_c_pool * p, const char * prompt) {
    const int id = p->pool->submit_conditional_prompt("default", prompt);
    if (id < 0) g_c_err = "pool is terminated";
    return id;
}

tts_c_pool * tts_c_pool_create(const char * model_path, int n_workers, const int * devices, int n_devices, int max_batch,
                               int batch_window_ms, const tts_c_config * load_cfg) {
    try {
        pool_options o;
        o.text_encoder_path = g_pool_text_encoder;
        o.n_workers = n_workers;
        for (int i = 0; devices && i < n_devices; i++) o.devices.push_back(devices[i]);
        o.max_batch = max_batch;
        o.batch_window_ms = batch_window_ms;
        o.continuous = g_pool_continuous;
        o.continuous_yield_ms = g_pool_yield_ms;
        auto p = std::make_unique<tts_c_pool>();
        p->pool = std::make_unique<device_pool>(std::map<std::string, std::string>{{"default", model_path}}, to_cfg(load_cfg), o);
        if (!p->pool->ok()) { g_c_err = p->pool->error(); return nullptr; }
        return p.release();
    } catch (const std::exception & e) {
        g_c_err = e.what();
        return nullptr;
    }
}

int tts_c_pool_submit(tts_c_pool * p, const char * text, const tts_c_config * cfg) {
    const int id = p->pool->submit("default", text, to_cfg(cfg));
    if (id < 0) g_c_err = "pool is terminated";
    return id;
}

int tts_c_pool_wait(tts_c_pool * p, int id, int timeout_ms, const float ** data, size_t * n_outputs, int * batch_size, int * worker) {
    std::shared_ptr<pool_task> t = p->pool->wait(id, timeout_ms);
    if (!t) { g_c_err = "task not finished (timeout or pool terminated)"; return -1; }
    {
        std::lock_guard<std::mutex> lock(p->mutex);
        p->held[id] = t;
    }
    if (data) *data = t->audio.data();
    if (n_outputs) *n_outputs = t->audio.size();
    if (batch_size) *batch_size = t->batch_size;
    if (worker) *worker = t->worker;
    if (!t->success) { g_c_err = t->message.empty() ? "empty response" : t->message; return 1; }
    return 0;
}

void tts_c_pool_release(tts_c_pool * p, int id) {
    {
        std::lock_guard<std::mutex> lock(p->mutex);
        p->held.erase(id);
    }
    p->pool->release(id);
}

void tts_c_pool_stats(tts_c_pool * p, uint64_t * tasks, uint64_t * batches, uint64_t * largest_batch, uint64_t * timed_out) {
    const pool_stats s = p->pool->stats();
    if (tasks) *tasks = s.tasks;
    if (batches) *batches = s.batches;
    if (largest_batch) *largest_batch = s.largest_batch;
    if (timed_out) *timed_out = s.timed_out;
}

void tts_c_pool_load_stats(tts_c_pool * p, int * weight_broadcasts, int * shared_arena_loads) {
    if (weight_broadcasts) *weight_broadcasts = p->pool->weight_broadcasts();
    if (shared_arena_loads) *shared_arena_loads = p->pool->shared_arena_loads();
}
void tts_c_pool_free(tts_c_pool * p) { delete p; }
