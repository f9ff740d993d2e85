#include "orpheus_runner.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "gguf.h"

static void hip_check(int rc, const char * what) {
    if (rc != 0) TTS_ABORT("%s failed: %s\n", what, tts_hip_last_error());
}

// model.cpp:7: the voices are not in the model configuration
static constexpr std::array<const char *, 7> orpheus_voices{"zoe", "zac", "jess", "leo", "mia", "julia", "leah"};

orpheus_model_loader::orpheus_model_loader() : tts_model_loader{"orpheus"} {}
const orpheus_model_loader orpheus_loader{};
void orpheus_register() {}

// orpheus_model::prep_constants / prep_layers (model.cpp:62-120) + snac_model::prep_constants / prep_layers
// (snac_model.cpp:3-48): same keys, same defaults, same required keys.
static orpheus_hparams read_hparams(const gguf_file & m) {
    orpheus_hparams hp;
    m.get_u32({"orpheus.vocab_size"}, hp.vocab_size);
    m.get_u32({"orpheus.attn_heads"}, hp.n_attn_heads);
    m.get_u32({"orpheus.kv_attn_heads"}, hp.n_kv_attn_heads);
    m.get_u32({"orpheus.head_dim"}, hp.head_size);
    m.get_u32({"orpheus.stopping_token_id"}, hp.stopping_token_id);
    m.get_u32({"tokenizer.ggml.eos_token_id"}, hp.eos_token_id);
    m.get_u32({"tokenizer.ggml.bos_token_id"}, hp.bos_token_id);
    m.get_u32({"orpheus.hidden_size"}, hp.hidden_size);
    m.get_u32({"orpheus.kv_hidden_size"}, hp.kv_hidden_size);
    if (!m.get_u32({"orpheus.layers"}, hp.n_layers)) TTS_ABORT("the 'orpheus.layers' must be specified in the GGUF file.\n");
    // extensions (absent in the reference's files)
    m.get_u32({"orpheus.max_context_length"}, hp.max_context_length);
    m.get_u32({"orpheus.max_generation_size"}, hp.max_generation_size);
    m.get_u32({"orpheus.audio_token_offset"}, hp.audio_token_offset);
    m.get_u32({"orpheus.audio_token_stride"}, hp.audio_token_stride);
    auto u32_array = [&](const char * key, std::vector<uint32_t> & out) {
        if (const gguf_value * v = m.get(key)) {
            if (v->arr_data && v->arr_n && (v->elem_type == GGUF_U32 || v->elem_type == GGUF_I32)) out.assign((const uint32_t *) v->arr_data, (const uint32_t *) v->arr_data + v->arr_n);
        }
    };
    u32_array("orpheus.prepended_tokens", hp.prepended_tokens);
    u32_array("orpheus.appended_tokens", hp.appended_tokens);

    m.get_u32({"snac.audio_token_channels"}, hp.snac_heads);
    m.get_u32({"snac.up_sampling_factor"}, hp.snac_up);
    m.get_u32({"snac.max_generation_size"}, hp.snac_max_generation);
    // the reference fixes 4 layers (snac_model.h:12) and aborts on a missing key; like the DAC loader, the count here is
    // how many consecutive stride keys the file holds, so small synthetic codecs load too
    uint32_t n = 0, up = 1;
    for (uint32_t i = 0; i < TTS_HIP_MAX_DAC_BLOCKS; i++) {
        const std::string sk = "snac.snac_layer_stride_" + std::to_string(i), pk = "snac.snac_layer_padding_" + std::to_string(i),
                          gk = "snac.snac_layer_grouping_" + std::to_string(i);
        if (!m.get_u32({sk.c_str()}, hp.snac_stride[i])) {
            if (i == 0) TTS_ABORT("key %s must be specified in gguf file inorder to initialize the SNAC audio decoder.\n", sk.c_str());
            break;
        }
        if (!m.get_u32({pk.c_str()}, hp.snac_padding[i])) TTS_ABORT("key %s must be specified in gguf file inorder to initialize the SNAC audio decoder.\n", pk.c_str());
        if (!m.get_u32({gk.c_str()}, hp.snac_groups[i])) TTS_ABORT("key %s must be specified in gguf file inorder to initialize the SNAC audio decoder.\n", gk.c_str());
        up *= hp.snac_stride[i];
        n++;
    }
    hp.snac_layers = n;
    if (up != hp.snac_up) hp.snac_up = up;
    if (hp.snac_heads != 3) TTS_ABORT("SNAC with %u token channels is unsupported (3: 4/2/1 repeats, snac_model.h:17)\n", hp.snac_heads);
    return hp;
}

std::unique_ptr<tts_generation_runner> orpheus_model_loader::from_file(gguf_file * meta, int, bool, const generation_configuration &) const {
    const orpheus_hparams hp = read_hparams(*meta);
    const int device = tts_load_device();
    return std::make_unique<orpheus_runner>(hp, bpe_tokenizer_from_gguf(*meta), device);
}

orpheus_runner::orpheus_runner(const orpheus_hparams & hp_, bpe_tokenizer * tok, int device)
    : tts_generation_runner{orpheus_loader}, hp(hp_), tokenizer(tok) {
    tts_hip_orpheus_desc d{};
    d.struct_size = sizeof(d);
    d.hidden_size = hp.hidden_size; d.n_layers = hp.n_layers; d.n_attn_heads = hp.n_attn_heads; d.n_kv_heads = hp.n_kv_attn_heads;
    d.head_dim = hp.head_size; d.vocab_size = hp.vocab_size;
    d.n_ctx = hp.max_context_length + hp.max_generation_size;   // orpheus_kv_cache_init, model.cpp:176-177
    max_seqs = std::min<uint32_t>(std::max<uint32_t>(1, tts_load_max_seqs()), 64);
    d.max_seqs = max_seqs;
    lm = tts_hip_orpheus_create(device, &d);
    if (!lm) TTS_ABORT("tts_hip_orpheus_create failed: %s\n", tts_hip_last_error());
    tts_hip_snac_desc s{};
    s.struct_size = sizeof(s);
    s.n_blocks = hp.snac_layers;
    for (uint32_t i = 0; i < hp.snac_layers; i++) { s.stride[i] = hp.snac_stride[i]; s.padding[i] = hp.snac_padding[i]; s.groups[i] = hp.snac_groups[i]; }
    s.n_codebooks = hp.snac_heads;
    for (uint32_t i = 0; i < 3; i++) s.repeats[i] = hp.snac_repeats[i];
    s.max_frames = hp.snac_max_generation;
    snac = tts_hip_snac_create(device, &s);
    if (!snac) {
        // the destructor does not run for a constructor that throws (TTS_ABORT under g_tts_throw_on_abort): release the decoder context
        tts_hip_destroy(lm);
        lm = nullptr;
        TTS_ABORT("tts_hip_snac_create failed: %s\n", tts_hip_last_error());
    }
    sampling_rate = 24000.0f;           // model.h:112
    supports_voices = true;
    smp.n_output_heads = 1;             // model.h:113-115
    smp.vocab_size = hp.vocab_size;
    smp.eos_token_id = hp.eos_token_id;
}

orpheus_runner::~orpheus_runner() {
    tts_hip_destroy(lm);
    tts_hip_destroy(snac);
}

void orpheus_runner::assign_weight(const char * name, const gguf_tensor_view & t) {
    // model.cpp:430-438: "snac." goes to the codec, "orpheus." to the decoder (the shim routes by the same prefixes)
    if (!strncmp(name, "snac.", 5)) hip_check(tts_hip_upload(snac, name, t.type, t.n_dims, t.ne, t.data), name);
    else if (!strncmp(name, "orpheus.", 8)) hip_check(tts_hip_upload(lm, name, t.type, t.n_dims, t.ne, t.data), name);
    else fprintf(stdout, "Warning: function %s encountered an unhandled tensor named '%s'.\n", __func__, name);
}

void orpheus_runner::prepare_post_load() {
    hip_check(tts_hip_finalize(lm, nullptr), "tts_hip_finalize(orpheus)");
    hip_check(tts_hip_finalize(snac, nullptr), "tts_hip_finalize(snac)");
    logits.resize(hp.vocab_size);
}

std::vector<std::string_view> orpheus_runner::list_voices() {
    return std::vector<std::string_view>(orpheus_voices.begin(), orpheus_voices.end());
}

std::vector<uint32_t> orpheus_runner::batch_from_sentence(const std::string & sentence, const std::string & voice) const {
    std::vector<uint32_t> tokens(hp.prepended_tokens);
    tokenizer->tokenize(voice.empty() ? sentence : voice + ": " + sentence, tokens);
    tokens.insert(tokens.end(), hp.appended_tokens.begin(), hp.appended_tokens.end());
    return tokens;
}

std::vector<std::vector<uint32_t>> orpheus_runner::prepare_output_tokens(const std::vector<uint32_t> & out) const {
    std::vector<std::vector<uint32_t>> levels(hp.audio_heads);
    const size_t chunks = out.size() / 7;
    for (size_t i = 0; i < chunks; i++)
        for (size_t ii = 0; ii < 7; ii++)
            levels[hp.heads[ii]].push_back(out[i * 7 + ii] - hp.audio_token_offset - (uint32_t) (ii % 7) * hp.audio_token_stride);
    return levels;
}

void orpheus_runner::generate(const char * sentence, tts_response & output, const generation_configuration & config) {
    smp.temperature = config.temperature;
    smp.repetition_penalty = config.repetition_penalty;
    smp.do_sample = config.sample;
    smp.top_k = (uint32_t) config.top_k;
    smp.top_p = config.top_p;
    smp.seed = config.seed;
    smp.n_calls = 0;
    if (!config.voice.empty() && std::find(orpheus_voices.begin(), orpheus_voices.end(), config.voice) == orpheus_voices.end())
        TTS_ABORT("Voice '%s' is not a valid voice for Orpheus.\n", config.voice.c_str());
    const std::vector<uint32_t> prompt = batch_from_sentence(sentence, config.voice);
    last_prompt_tokens = prompt;
    if (prompt.size() > hp.max_context_length)
        TTS_ABORT("The prompt was too large for the default context window. Try splitting up or shortenning the prompt.\n");
    smp.reset();
    output.data = nullptr;
    output.n_outputs = 0;

    // generate_from_batch (model.cpp:378-392)
    std::vector<uint32_t> & out = last_output_tokens;
    out.clear();
    if (!config.sample) {
        out.resize(hp.max_generation_size);
        uint32_t n = 0;
        hip_check(tts_hip_orpheus_generate_greedy(lm, prompt.data(), (uint32_t) prompt.size(), hp.max_generation_size, hp.stopping_token_id, out.data(), &n),
                  "tts_hip_orpheus_generate_greedy");
        out.resize(n);
    } else if (!getenv("TTS_HOST_LOOP") && config.top_p > 0.0f && config.top_k >= 1 && config.top_k <= 64 && (uint32_t) config.top_k < hp.vocab_size) {
        // top_k in 1..64 (the default generation_configuration: top_k 50, top_p 1): sampler::sample runs on the device, two kernels pick the
        // top_k candidates out of the 156 940 logits (top_p < 1: a third accumulates the full-vocabulary softmax total in index order first,
        // round 5); the U[0,1) draws are made here, one generator per call as sampler.cpp:47-48
        std::vector<float> u(hp.max_generation_size);
        for (auto & v : u) smp.draw_uniforms(&v);
        tts_hip_sampling sp{(uint32_t) config.top_k, config.top_p, config.temperature, config.repetition_penalty};
        out.resize(hp.max_generation_size);
        uint32_t n = 0;
        hip_check(tts_hip_orpheus_generate_sampled(lm, prompt.data(), (uint32_t) prompt.size(), hp.max_generation_size, hp.stopping_token_id, &sp, u.data(), out.data(), &n),
                  "tts_hip_orpheus_generate_sampled");
        out.resize(n);
    } else {
        // a top_k the device sampler does not take (0 = off, or > 64: the reference sorts the whole vocabulary): logits come back,
        // sampler::sample runs here
        std::vector<uint32_t> batch = prompt;
        uint32_t pos = 0;
        while ((out.empty() || out.back() != hp.stopping_token_id) && out.size() < hp.max_generation_size) {
            hip_check(tts_hip_orpheus_decode(lm, batch.data(), (uint32_t) batch.size(), pos, logits.data(), nullptr), "tts_hip_orpheus_decode");
            pos += (uint32_t) batch.size();
            smp.sample(logits.data(), out);
            batch.assign(1, out.back());
        }
    }
    decode_audio(out, pcm);
    output.data = pcm.empty() ? nullptr : pcm.data();
    output.n_outputs = pcm.size();
}

// the 7-ids-per-frame stream -> SNAC levels -> audio (model.cpp:358-376, snac_runner::run)
void orpheus_runner::decode_audio(const std::vector<uint32_t> & out, std::vector<float> & audio) {
    audio.clear();
    if (out.size() >= hp.max_generation_size)
        fprintf(stdout, "Warning: generation hit its max default length. The generated audio may not contain the entire prompt.\n");
    const std::vector<std::vector<uint32_t>> levels = prepare_output_tokens(out);
    const uint32_t T = (uint32_t) levels[2].size();   // finest level: 4 ids per 7-id chunk (snac_runner::run :181)
    if (T == 0) return;
    std::vector<uint32_t> codes;
    for (auto & l : levels) codes.insert(codes.end(), l.begin(), l.end());
    // snac_runner::set_inputs (:177): noise_steps_sum * T standard normals from the never-reseeded engine
    size_t noise_len = 0, up = 1;
    for (uint32_t i = 0; i < hp.snac_layers; i++) { up *= hp.snac_stride[i]; noise_len += up * (size_t) T; }
    std::vector<float> noise;
    if (!getenv("TTS_SNAC_NO_NOISE")) {
        noise.resize(noise_len);
        for (auto & v : noise) v = noise_dist(noise_engine);
    }
    audio.assign((size_t) T * hp.snac_up, 0.0f);
    hip_check(tts_hip_snac_decode(snac, codes.data(), T, noise.empty() ? nullptr : noise.data(), audio.data()), "tts_hip_snac_decode");
}

// n utterances in lock-step on the device: one cache slot and one row of the forward per utterance (the reference: one worker and one model copy per
// concurrent request, examples/server/server.cpp:225-321).  Every utterance gets the ids a generate() call of its own would produce — the sampler is
// reset per utterance exactly as generate() does, so with config.sample every utterance draws the same uniform sequence its own call would —
// and the codec then runs utterance by utterance in order (the noise engine is never reseeded: the draws follow the order of sequential calls).
void orpheus_runner::generate_batch(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs, const generation_configuration & config) {
    const uint32_t n = (uint32_t) sentences.size();
    const bool dev_sample = config.sample && !getenv("TTS_HOST_LOOP") && config.top_p > 0.0f && config.top_k >= 1 && config.top_k <= 64 && (uint32_t) config.top_k < hp.vocab_size;
    if (n <= 1 || max_seqs <= 1 || (config.sample && !dev_sample)) { tts_generation_runner::generate_batch(sentences, outputs, config); return; }
    if (n > max_seqs) TTS_ABORT("generate_batch: %u utterances but the runner was loaded with max_seqs=%u (TTS_HIP_MAX_SEQS)\n", n, max_seqs);
    if (!config.voice.empty() && std::find(orpheus_voices.begin(), orpheus_voices.end(), config.voice) == orpheus_voices.end())
        TTS_ABORT("Voice '%s' is not a valid voice for Orpheus.\n", config.voice.c_str());
    std::vector<uint32_t> prompts, lens(n);
    for (uint32_t u = 0; u < n; u++) {
        const std::vector<uint32_t> p = batch_from_sentence(sentences[u], config.voice);
        if (p.size() > hp.max_context_length) TTS_ABORT("The prompt was too large for the default context window. Try splitting up or shortenning the prompt.\n");
        lens[u] = (uint32_t) p.size();
        prompts.insert(prompts.end(), p.begin(), p.end());
        if (u + 1 == n) last_prompt_tokens = p;
    }
    const uint32_t M = hp.max_generation_size;
    std::vector<uint32_t> toks((size_t) n * M), cnt(n);
    std::vector<float> uni;
    tts_hip_sampling sp{(uint32_t) config.top_k, config.top_p, config.temperature, config.repetition_penalty};
    if (dev_sample) {
        uni.resize((size_t) n * M);
        for (uint32_t u = 0; u < n; u++) {   // generate() per utterance: same parameters, n_calls = 0, reset, then one draw per sampler call
            smp.temperature = config.temperature; smp.repetition_penalty = config.repetition_penalty; smp.do_sample = true;
            smp.top_k = (uint32_t) config.top_k; smp.top_p = config.top_p; smp.seed = config.seed; smp.n_calls = 0;
            smp.reset();
            for (uint32_t k = 0; k < M; k++) smp.draw_uniforms(&uni[(size_t) u * M + k]);
        }
    }
    hip_check(tts_hip_orpheus_generate_batch(lm, n, prompts.data(), lens.data(), M, hp.stopping_token_id, dev_sample ? &sp : nullptr, dev_sample ? uni.data() : nullptr, toks.data(),
                                             cnt.data()), "tts_hip_orpheus_generate_batch");
    outputs.assign(n, tts_response{});
    batch_store_.assign(n, {});
    last_batch_tokens.assign(n, {});
    for (uint32_t u = 0; u < n; u++) {
        last_batch_tokens[u].assign(toks.begin() + (size_t) u * M, toks.begin() + (size_t) u * M + cnt[u]);
        decode_audio(last_batch_tokens[u], batch_store_[u]);
        outputs[u].data = batch_store_[u].empty() ? nullptr : batch_store_[u].data();
        outputs[u].n_outputs = batch_store_[u].size();
    }
    last_output_tokens = last_batch_tokens[n - 1];
}
