#include "dia_runner.h"

#include <cstdlib>
#include <cstring>

#include "gguf.h"

static void hip_check(int rc, const char * what) {
    if (rc != 0) TTS_ABORT("%s failed: %s\n", what, tts_hip_last_error());
}

dia_model_loader::dia_model_loader() : tts_model_loader{"dia"} {}
const dia_model_loader dia_loader{};
void dia_register() {}

// dia_model::prep_constants (model.cpp:168-268) + dac_model::prep_constants / prep_layers (dac_model.cpp:15-55): same keys
// and defaults.  The encoder's hidden size has no key in the reference (model.h:68); it is the embedding's row length.
static dia_hparams read_hparams(const gguf_file & m) {
    dia_hparams hp;
    m.get_u32({"dia.decoder.output_heads"}, hp.n_output_heads);
    m.get_u32({"dia.decoder.layers"}, hp.n_decoder_layers);
    m.get_u32({"dia.encoder.layers"}, hp.n_encoder_layers);
    m.get_u32({"dia.decoder.hidden_size"}, hp.decoder_hidden_size);
    m.get_u32({"dia.decoder.attn_heads"}, hp.decoder_attn_heads);
    m.get_u32({"dia.decoder.query_heads"}, hp.decoder_query_heads);
    m.get_u32({"dia.encoder.attn_heads"}, hp.encoder_attn_heads);
    m.get_u32({"dia.attn_head_size"}, hp.head_size);
    m.get_u32({"dia.eos_token_id"}, hp.eos_token_id);
    m.get_u32({"dia.bos_token_id"}, hp.bos_token_id);
    m.get_u32({"dia.pad_token_id"}, hp.pad_token_id);
    m.get_u32({"dia.encoder.max_context_length"}, hp.max_encoder_context_length);
    m.get_u32({"dia.decoder.output_vocab_size"}, hp.output_vocab_size);
    m.get_u32({"dia.decoder.audio_vocab_size"}, hp.audio_vocab_size);
    m.get_u32({"dia.decoder.max_generation_size"}, hp.max_generation_size);
    m.get_u32({"dia.max_delay"}, hp.max_delay);
    if (const gguf_value * v = m.get("dia.cfg_scale")) hp.cfg_scale = (float) v->f;
    for (const gguf_tensor_view & t : m.tensors)
        if (!strcmp(t.name, "dia.encoder.embedding")) hp.encoder_hidden_size = (uint32_t) t.ne[0];
    if (hp.n_output_heads != hp.delay_pattern.size())
        TTS_ABORT("Dia with %u output heads is unsupported: the delay pattern is fixed at %zu heads (dia/model.h:84)\n", hp.n_output_heads, hp.delay_pattern.size());
    if (hp.decoder_query_heads == 0 || hp.decoder_attn_heads % hp.decoder_query_heads)
        TTS_ABORT("dia.decoder.attn_heads must be a multiple of dia.decoder.query_heads\n");
    m.get_u32({"dac.up_sampling_factor", "up_sampling_factor"}, hp.up_sampling_factor);
    uint32_t n_found = 0, up = 1;
    for (uint32_t i = 0; i < TTS_HIP_MAX_DAC_BLOCKS; i++) {   // same rule as the Parler loader: as many blocks as stride keys
        const std::string sk = "dac_layer_stride_" + std::to_string(i), pk = "dac_layer_padding_" + std::to_string(i);
        const std::string dsk = "dac." + sk, dpk = "dac." + pk;
        if (!m.get_u32({dsk.c_str(), sk.c_str()}, hp.dac_stride[i])) {
            if (i == 0) TTS_ABORT("key %s must be specified in gguf file inorder to initialize the DAC audio decoder.\n", dsk.c_str());
            break;
        }
        if (!m.get_u32({dpk.c_str(), pk.c_str()}, hp.dac_padding[i]))
            TTS_ABORT("key %s must be specified in gguf file inorder to initialize the DAC audio decoder.\n", dpk.c_str());
        up *= hp.dac_stride[i];
        n_found++;
    }
    hp.dac_n_layers = n_found;
    hp.up_sampling_factor = up;
    return hp;
}

std::unique_ptr<tts_generation_runner> dia_model_loader::from_file(gguf_file * meta, int, bool, const generation_configuration &) const {
    const dia_hparams hp = read_hparams(*meta);
    const int device = tts_load_device();
    return std::make_unique<dia_runner>(hp, device);
}

dia_runner::dia_runner(const dia_hparams & hp_, int device) : tts_generation_runner{dia_loader}, hp(hp_) {
    tts_hip_dia_desc d{};
    d.struct_size = sizeof(d);
    d.enc_hidden_size = hp.encoder_hidden_size; d.enc_layers = hp.n_encoder_layers; d.enc_attn_heads = hp.encoder_attn_heads;
    d.dec_hidden_size = hp.decoder_hidden_size; d.dec_layers = hp.n_decoder_layers; d.dec_attn_heads = hp.decoder_attn_heads;
    d.dec_kv_heads = hp.decoder_attn_heads / hp.decoder_query_heads;   // model.cpp:463: k/v are projected to attn_heads / query_heads groups
    d.head_dim = hp.head_size; d.n_output_heads = hp.n_output_heads; d.output_vocab_size = hp.output_vocab_size;
    d.max_ctx = hp.max_encoder_context_length; d.max_gen = hp.max_generation_size; d.cfg_scale = hp.cfg_scale;
    max_seqs = tts_load_max_seqs();
    d.max_utterances = max_seqs;
    lm = tts_hip_dia_create(device, &d);
    if (!lm) TTS_ABORT("tts_hip_dia_create failed: %s\n", tts_hip_last_error());
    tts_hip_desc a{};
    a.struct_size = sizeof(a);
    a.dac_n_blocks = hp.dac_n_layers;
    for (uint32_t i = 0; i < hp.dac_n_layers; i++) { a.dac_stride[i] = hp.dac_stride[i]; a.dac_padding[i] = hp.dac_padding[i]; }
    a.dac_max_frames = hp.max_generation_size;
    a.max_seqs = 1;
    a.flags = TTS_HIP_FLAG_NO_PARLER;
    dac = tts_hip_create(device, &a);
    if (!dac) {
        // the destructor does not run for a constructor that throws (TTS_ABORT under g_tts_throw_on_abort): release the model context
        tts_hip_destroy(lm);
        lm = nullptr;
        TTS_ABORT("tts_hip_create (codec) failed: %s\n", tts_hip_last_error());
    }
    sampling_rate = 44100.0f;
    smp.n_output_heads = hp.n_output_heads;
    smp.vocab_size = hp.output_vocab_size;   // model.h:191
    smp.eos_token_id = hp.eos_token_id;
}

dia_runner::~dia_runner() {
    tts_hip_destroy(lm);
    tts_hip_destroy(dac);
}

void dia_runner::assign_weight(const char * name, const gguf_tensor_view & t) {
    // model.cpp:892-898: "audio_encoder." goes to the codec, everything else to the Dia model
    if (!strncmp(name, "audio_encoder.", 14)) hip_check(tts_hip_upload(dac, name, t.type, t.n_dims, t.ne, t.data), name);
    else if (!strncmp(name, "dia.", 4)) hip_check(tts_hip_upload(lm, name, t.type, t.n_dims, t.ne, t.data), name);
    else TTS_ABORT("Unrecognized tensor '%s' when loading Dia from GGUF file.", name);
}

void dia_runner::prepare_post_load() {
    hip_check(tts_hip_finalize(lm, nullptr), "tts_hip_finalize(dia)");
    hip_check(tts_hip_finalize(dac, nullptr), "tts_hip_finalize(dac)");
    logits.resize((size_t) hp.n_output_heads * hp.output_vocab_size);
}

uint32_t dia_tokenize_sentence(const dia_hparams & hp, std::string sentence, std::vector<uint32_t> & tokens) {
    const size_t b = sentence.find_first_not_of(' '), e = sentence.find_last_not_of(' ');   // strip(), util.cpp:273-281
    sentence = b == std::string::npos ? std::string() : sentence.substr(b, e - b + 1);
    const std::string start = sentence.substr(0, 4);
    if (start != "[S1]" && start != "[S2]") sentence = "[S1] " + sentence;
    if (sentence[sentence.size() - 1] != '.') sentence += ".";
    for (const auto & tag : {std::pair<const char *, char>{"[S1]", 1}, {"[S2]", 2}})
        for (size_t p = sentence.find(tag.first); p != std::string::npos; p = sentence.find(tag.first)) sentence.replace(p, 4, std::string(1, tag.second));
    if (sentence.size() > hp.max_encoder_context_length)
        TTS_ABORT("Dia currently only supports a max of %d characters and received an input of %d characters.", (int) hp.max_encoder_context_length, (int) sentence.size());
    tokens.assign(hp.max_encoder_context_length, 0u);
    // bytes as unsigned values: the reference casts a (signed) char, which turns UTF-8 bytes >= 0x80 into row indices far
    // outside the 256-row embedding (:694); the byte value is what the model was trained on
    for (size_t i = 0; i < sentence.size(); i++) tokens[i] = (uint32_t) (unsigned char) sentence[i];
    if (sentence.size() <= 100)
        fprintf(stdout, "Your prompt has fewer than 100 tokens. Please note that Dia's generation with prompts that are fewer than 100 tokens is highly inconsistent.\n");
    return (uint32_t) sentence.size();
}

bool dia_check_stopping(const dia_hparams & hp, std::vector<uint32_t> & audio_tokens, uint32_t current_position, uint32_t max_generation_size, int & delay_steps) {
    if (delay_steps == -1 && (audio_tokens[0] == hp.eos_token_id || current_position >= max_generation_size - hp.max_delay)) delay_steps = (int) hp.max_delay;
    if (delay_steps > 0) {
        const int step_after_eos = (int) hp.max_delay - delay_steps;
        for (size_t i = 0; i < hp.delay_pattern.size(); i++) {
            if (step_after_eos == (int) hp.delay_pattern[i]) audio_tokens[i] = hp.eos_token_id;
            else if (step_after_eos > (int) hp.delay_pattern[i]) audio_tokens[i] = hp.pad_token_id;
        }
        delay_steps -= 1;
    }
    return delay_steps == 0;
}

void dia_adjust_output_tokens(const dia_hparams & hp, const std::vector<uint32_t> & output_tokens, std::vector<uint32_t> & filtered) {
    const size_t size = output_tokens.size(), nh = hp.n_output_heads;
    filtered.clear();
    filtered.reserve(size);
    for (int i = 0; i < (int) (size / nh) - (int) hp.max_delay; i++) {
        bool skip_step = false;
        for (size_t ii = 0; ii < nh; ii++) {
            const size_t next_index = (size_t) i * nh + hp.delay_pattern[ii] * nh + ii;
            if (next_index > size || output_tokens[next_index] >= hp.audio_vocab_size) { skip_step = true; break; }
        }
        if (skip_step) continue;
        for (size_t ii = 0; ii < nh; ii++) filtered.push_back(output_tokens[(size_t) i * nh + hp.delay_pattern[ii] * nh + ii]);
    }
}

// the generation loop on the device (tts_hip_dia_generate).  With a fixed seed every utterance gets the draws a separate generate()
// call would make (each call seeds its own sampler the same way, so call k draws the same U[0,1) values for every utterance); with
// seed == 0 (std::random_device per call, the reference's behaviour) separate calls are independently random, so every utterance of the
// batch draws its own values — as the host loop's per-utterance samplers and the Parler batch path do
static void dia_device_loop(tts_hip_ctx * lm, const dia_hparams & hp, sampler & proto, uint32_t n, uint32_t max_gen, const generation_configuration & config,
                            std::vector<std::vector<uint32_t>> & tokens) {
    const uint32_t nh = hp.n_output_heads;
    tts_hip_dia_codes codes{};
    codes.bos = hp.bos_token_id; codes.eos = hp.eos_token_id; codes.pad = hp.pad_token_id; codes.max_delay = hp.max_delay;
    for (size_t i = 0; i < hp.delay_pattern.size() && i < 16; i++) codes.delay_pattern[i] = hp.delay_pattern[i];
    std::vector<uint32_t> toks((size_t) n * max_gen * nh), steps(n);
    std::vector<float> u;
    tts_hip_sampling sp{(uint32_t) config.top_k, config.top_p, config.temperature, config.repetition_penalty};
    if (config.sample) {
        u.resize((size_t) max_gen * n * nh);
        std::vector<float> draw(nh);
        sampler s = proto;
        s.seed = config.seed; s.n_calls = 0;
        for (uint32_t k = 0; k < max_gen; k++) {
            if (config.seed != 0) {
                s.draw_uniforms(draw.data());
                for (uint32_t i = 0; i < n; i++) std::copy(draw.begin(), draw.end(), u.begin() + ((size_t) k * n + i) * nh);
            } else {
                for (uint32_t i = 0; i < n; i++) s.draw_uniforms(u.data() + ((size_t) k * n + i) * nh);   // a fresh random_device draw per utterance
            }
        }
    }
    if (tts_hip_dia_generate(lm, n, max_gen, &codes, config.sample ? &sp : nullptr, config.sample ? u.data() : nullptr, toks.data(), steps.data()) != 0)
        TTS_ABORT("tts_hip_dia_generate failed: %s\n", tts_hip_last_error());
    tokens.assign(n, {});
    for (uint32_t i = 0; i < n; i++) tokens[i].assign(toks.begin() + (size_t) i * max_gen * nh, toks.begin() + ((size_t) i * max_gen + steps[i]) * nh);
}

void dia_runner::generate(const char * sentence, tts_response & output, const generation_configuration & config) {
    if (!(config.max_tokens == 0 || config.max_tokens > (int) hp.max_delay)) TTS_ABORT("TTS_ASSERT(config.max_tokens == 0 || config.max_tokens > model->max_delay) failed\n");
    smp.temperature = config.temperature;
    smp.repetition_penalty = config.repetition_penalty;
    smp.do_sample = config.sample;
    smp.top_k = (uint32_t) config.top_k;
    smp.top_p = config.top_p;
    smp.seed = config.seed;
    smp.n_calls = 0;
    uint32_t max_gen = config.max_tokens > (int) hp.max_delay ? (uint32_t) config.max_tokens : hp.max_generation_size;
    if (max_gen > hp.max_generation_size) max_gen = hp.max_generation_size;   // the self-attention cache holds max_generation_size positions (:300-301)
    output.data = nullptr;
    output.n_outputs = 0;

    const uint32_t sentence_length = dia_tokenize_sentence(hp, sentence, last_prompt_tokens);
    smp.reset();
    hip_check(tts_hip_dia_encode(lm, last_prompt_tokens.data(), sentence_length, nullptr), "tts_hip_dia_encode");

    // generate_from_batch (:810-833)
    const uint32_t nh = hp.n_output_heads;
    std::vector<uint32_t> & out = last_output_tokens;
    out.clear();
    out.reserve((size_t) max_gen * nh);
    std::vector<uint32_t> audio_tokens(nh, hp.bos_token_id);
    uint32_t current_position = 0;
    int      delay_steps = -1;
    if (!getenv("TTS_HOST_LOOP")) {
        // check_stopping, the step, the sampler and the delay-pattern feedback replay as one captured graph; the host loop below is
        // the reference's shape (logits back every step, sampler::sample here) and stays for TTS_HOST_LOOP=1
        std::vector<std::vector<uint32_t>> t;
        dia_device_loop(lm, hp, smp, 1, max_gen, config, t);
        out = t[0];
        delay_steps = 0;
    }
    while (delay_steps != 0 && !dia_check_stopping(hp, audio_tokens, current_position, max_gen, delay_steps)) {
        hip_check(tts_hip_dia_step(lm, audio_tokens.data(), current_position, logits.data(), nullptr), "tts_hip_dia_step");
        smp.sample(logits.data(), out);
        current_position += 1;
        const uint32_t * last = out.data() + out.size() - nh;
        for (uint32_t i = 0; i < nh; i++) audio_tokens[i] = current_position > i ? last[i] : hp.bos_token_id;
    }

    std::vector<uint32_t> filtered;
    dia_adjust_output_tokens(hp, out, filtered);
    const uint32_t frames = (uint32_t) (filtered.size() / nh);
    if (frames == 0) return;
    pcm.assign((size_t) frames * hp.up_sampling_factor, 0.0f);
    hip_check(tts_hip_dac_decode(dac, filtered.data(), frames, pcm.data()), "tts_hip_dac_decode");
    output.data = pcm.data();
    output.n_outputs = pcm.size();
}

void dia_runner::generate_batch(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs, const generation_configuration & config) {
    const uint32_t n = (uint32_t) sentences.size(), nh = hp.n_output_heads;
    outputs.assign(n, tts_response{});
    if (n == 0) return;
    if (n > max_seqs) TTS_ABORT("generate_batch: %u utterances but the runner was loaded with max_seqs=%u (TTS_HIP_MAX_SEQS)\n", n, max_seqs);
    if (!(config.max_tokens == 0 || config.max_tokens > (int) hp.max_delay)) TTS_ABORT("TTS_ASSERT(config.max_tokens == 0 || config.max_tokens > model->max_delay) failed\n");
    uint32_t max_gen = config.max_tokens > (int) hp.max_delay ? (uint32_t) config.max_tokens : hp.max_generation_size;
    if (max_gen > hp.max_generation_size) max_gen = hp.max_generation_size;

    // per utterance: its own sampler state (n separate generate() calls would each reset and seed theirs), tokens, countdown
    std::vector<sampler> smps(n, smp);
    std::vector<std::vector<uint32_t>> prompts(n);
    for (uint32_t u = 0; u < n; u++) {
        sampler & s = smps[u];
        s.temperature = config.temperature; s.repetition_penalty = config.repetition_penalty; s.do_sample = config.sample;
        s.top_k = (uint32_t) config.top_k; s.top_p = config.top_p; s.seed = config.seed; s.n_calls = 0;
        s.reset();
        const uint32_t len = dia_tokenize_sentence(hp, sentences[u], prompts[u]);
        hip_check(tts_hip_dia_encode_slot(lm, u, prompts[u].data(), len, nullptr), "tts_hip_dia_encode_slot");
    }
    last_batch_tokens.assign(n, {});
    std::vector<std::vector<uint32_t>> audio(n, std::vector<uint32_t>(nh, hp.bos_token_id));
    std::vector<uint32_t> pos(n, 0), ids((size_t) n * nh);
    std::vector<int>      delay(n, -1);
    std::vector<bool>     done(n, false);
    std::vector<float>    lg((size_t) n * nh * hp.output_vocab_size);
    const bool device_loop = !getenv("TTS_HOST_LOOP");
    if (device_loop) dia_device_loop(lm, hp, smp, n, max_gen, config, last_batch_tokens);
    for (; !device_loop;) {
        // check_stopping (:767-785) per utterance before each decode, as generate_from_batch's while condition (:817)
        bool any = false;
        for (uint32_t u = 0; u < n; u++) {
            if (!done[u] && dia_check_stopping(hp, audio[u], pos[u], max_gen, delay[u])) done[u] = true;
            any = any || !done[u];
        }
        if (!any) break;
        for (uint32_t u = 0; u < n; u++) std::copy(audio[u].begin(), audio[u].end(), ids.begin() + (size_t) u * nh);
        // a finished utterance keeps its rows in the step (lock-step shapes stay fixed); its logits are ignored and its position stays
        hip_check(tts_hip_dia_step_batch(lm, n, nullptr, ids.data(), pos.data(), lg.data(), nullptr), "tts_hip_dia_step_batch");
        for (uint32_t u = 0; u < n; u++) {
            if (done[u]) continue;
            std::vector<uint32_t> & out = last_batch_tokens[u];
            smps[u].sample(lg.data() + (size_t) u * nh * hp.output_vocab_size, out);
            pos[u] += 1;
            const uint32_t * last = out.data() + out.size() - nh;
            for (uint32_t i = 0; i < nh; i++) audio[u][i] = pos[u] > i ? last[i] : hp.bos_token_id;
        }
    }

    std::vector<uint32_t> codes, frames(n);
    for (uint32_t u = 0; u < n; u++) {
        std::vector<uint32_t> f;
        dia_adjust_output_tokens(hp, last_batch_tokens[u], f);
        frames[u] = (uint32_t) (f.size() / nh);
        codes.insert(codes.end(), f.begin(), f.end());
    }
    size_t total = 0;
    for (uint32_t f : frames) total += (size_t) f * hp.up_sampling_factor;
    pcm.assign(total, 0.0f);
    if (total) hip_check(tts_hip_dac_decode_batch(dac, codes.data(), frames.data(), n, pcm.data()), "tts_hip_dac_decode_batch");   // one batched codec pass
    size_t off = 0;
    for (uint32_t u = 0; u < n; u++) {
        outputs[u].data = frames[u] ? pcm.data() + off : nullptr;
        outputs[u].n_outputs = (size_t) frames[u] * hp.up_sampling_factor;
        off += outputs[u].n_outputs;
    }
}
