#include "kokoro_runner.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "gguf.h"

static void hip_check(int rc, const char * what) {
    if (rc != 0) TTS_ABORT("%s failed: %s\n", what, tts_hip_last_error());
}

kokoro_model_loader::kokoro_model_loader() : tts_model_loader{"kokoro"} {}
const kokoro_model_loader kokoro_loader{};
void kokoro_register() {}

single_pass_tokenizer::single_pass_tokenizer(std::vector<std::string> tkns) : tokens(std::move(tkns)) {
    for (const auto & t : tokens) max_size = std::max(max_size, t.size());
}

void single_pass_tokenizer::tokenize(const std::string & text, std::vector<uint32_t> & token_ids) const {
    size_t at = 0;
    while (at < text.size()) {
        uint32_t token_id = unknown_id;
        const size_t remaining = text.size() - at;
        for (size_t i = 1; i < std::min(remaining + 1, max_size + 1); i++) {
            const auto pos = std::find(tokens.begin(), tokens.end(), text.substr(at, i));
            if (pos != tokens.end()) {
                token_id = (uint32_t) (pos - tokens.begin());
                at += i;
                break;
            }
        }
        if (token_id == unknown_id) at += 1;
        token_ids.push_back(token_id);
    }
}

static std::string strip_spaces(const std::string & s) {
    const size_t b = s.find_first_not_of(' '), e = s.find_last_not_of(' ');
    return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
}
// replace_any (util.cpp:283-292): every occurrence of any of the characters
static std::string replace_any(std::string target, const std::string & chars, const std::string & with) {
    std::string out;
    for (char ch : target) {
        if (chars.find(ch) != std::string::npos) out += with;
        else out += ch;
    }
    return out;
}
// split(target, split_on) (util.cpp:219-244): pieces between any of the characters, empty pieces dropped
static std::vector<std::string> split_any(const std::string & target, const std::string & on) {
    std::vector<std::string> out;
    size_t last = 0;
    for (size_t i = 0; i < target.size(); i++)
        if (on.find(target[i]) != std::string::npos) {
            if (i > last) out.push_back(target.substr(last, i - last));
            last = i + 1;
        }
    if (last < target.size()) out.push_back(target.substr(last));
    return out;
}

std::vector<std::vector<uint32_t>> kokoro_tokenize_chunks(const kokoro_hparams & hp, const single_pass_tokenizer & tok, std::vector<std::string> clauses) {
    std::vector<std::vector<uint32_t>> chunks;
    for (auto clause : clauses) {
        clause = strip_spaces(clause);
        if (clause.empty()) continue;
        std::vector<uint32_t> tokens;
        tokens.push_back(hp.bos_token_id);
        tok.tokenize(clause, tokens);
        if (tokens.size() > hp.max_context_length - 2) {
            // split at space tokens, mid-word when there is none (model.cpp:1352-1380).  The reference adds the size of the
            // previous chunk to the running length (and reads chunks.back() of an empty list for the first clause); the empty
            // case counts as zero here, the rest is kept as it is.
            size_t last_space_token = 1, last_split = 1;
            for (size_t i = 1; i < tokens.size(); i++) {
                if (tokens[i] == hp.space_token_id) last_space_token = i;
                const size_t prev = chunks.empty() ? 0 : chunks.back().size();
                if ((i - last_split) + prev >= hp.max_context_length - 1) {
                    std::vector<uint32_t> portion = {hp.bos_token_id};
                    if (last_space_token > last_split) {
                        portion.insert(portion.end(), tokens.begin() + (long) last_split, tokens.begin() + (long) last_space_token);
                        last_split = last_space_token;
                    } else {
                        portion.insert(portion.end(), tokens.begin() + (long) last_split, tokens.begin() + (long) i + 1);
                        last_split = i + 1;
                    }
                    portion.push_back(hp.eos_token_id);
                    chunks.push_back(portion);
                }
            }
            if (last_split + 1 < tokens.size()) {
                std::vector<uint32_t> portion = {hp.bos_token_id};
                portion.insert(portion.end(), tokens.begin() + (long) last_split, tokens.end());
                portion.push_back(hp.eos_token_id);
                chunks.push_back(portion);
            }
        } else {
            tokens.push_back(hp.eos_token_id);
            chunks.push_back(tokens);
        }
    }
    return chunks;
}

// kokoro_model::prep_constants (model.cpp:841-930) + the generator geometry of prep_layers (:246-308, :820-836)
static kokoro_hparams read_hparams(const gguf_file & m) {
    kokoro_hparams hp;
    const std::string a = "kokoro.duration_predictor.albert.", g = "kokoro.decoder.generator.";
    m.get_u32({(a + "context_length").c_str()}, hp.max_context_length);
    m.get_u32({(a + "attn_heads").c_str()}, hp.n_attn_heads);
    m.get_u32({(a + "layers").c_str()}, hp.n_layers);
    m.get_u32({(a + "recurrence").c_str()}, hp.n_recurrence);
    m.get_u32({"kokoro.duration_predictor.f0_n_blocks"}, hp.f0_n_blocks);
    m.get_u32({"kokoro.duration_predictor.layers"}, hp.n_duration_prediction_layers);
    m.get_u32({"kokoro.text_encoder.layers"}, hp.n_conv_layers);
    m.get_u32({(g + "up_sampling_factor").c_str()}, hp.up_sampling_factor);
    m.get_u32({(g + "kernels").c_str()}, hp.n_kernels);
    m.get_u32({(g + "upsamples").c_str()}, hp.n_upsamples);
    m.get_u32({(g + "layers").c_str()}, hp.n_decoder_blocks);
    m.get_u32({(g + "padding").c_str()}, hp.out_conv_padding);
    m.get_u32({(g + "n_fft").c_str()}, hp.true_n_fft);
    m.get_u32({(g + "hop").c_str()}, hp.stft_hop);
    if (hp.n_layers != 1) TTS_ABORT("Kokoro with %u ALBERT layers is unsupported (one shared layer, kokoro/model.h:192)\n", hp.n_layers);
    if (hp.n_upsamples == 0 || hp.n_upsamples > 4 || hp.n_upsamples * hp.n_kernels > 16) TTS_ABORT("Kokoro generator geometry out of range\n");
    auto need = [&](const std::string & key, uint32_t & out, const char * what) {
        if (!m.get_u32({key.c_str()}, out)) TTS_ABORT("%s (key '%s')\n", what, key.c_str());
    };
    for (uint32_t i = 0; i < hp.n_upsamples; i++) {   // n_noise_blocks == n_upsamples (model.h:211-212)
        const std::string nb = g + "noise_blocks." + std::to_string(i), ub = g + "up_convs." + std::to_string(i);
        need(nb + ".stride", hp.noise_stride[i], "both padding and stride keys must be assigned in order to initialize a kokoro noise block.");
        need(nb + ".padding", hp.noise_padding[i], "both padding and stride keys must be assigned in order to initialize a kokoro noise block.");
        need(ub + ".stride", hp.up_stride[i], "both padding and stride keys must be assigned in order to initialize a kokoro upsample block.");
        need(ub + ".padding", hp.up_padding[i], "both padding and stride keys must be assigned in order to initialize a kokoro upsample block.");
        for (uint32_t j = 0; j < 3; j++) {
            need(nb + ".res_block." + std::to_string(j) + ".padding", hp.noise_res_padding[i][j], "Could not find dilation and padding for generator residual block");
            need(nb + ".res_block." + std::to_string(j) + ".dilation", hp.noise_res_dilation[i][j], "Could not find dilation and padding for generator residual block");
        }
    }
    for (uint32_t i = 0; i < hp.n_upsamples * hp.n_kernels; i++)
        for (uint32_t j = 0; j < 3; j++) {
            const std::string rb = g + "res_blocks." + std::to_string(i) + "." + std::to_string(j);
            need(rb + ".padding", hp.res_padding[i][j], "Could not find dilation and padding for generator residual block");
            need(rb + ".dilation", hp.res_dilation[i][j], "Could not find dilation and padding for generator residual block");
        }
    if (const gguf_value * v = m.get("kokoro.voices")) hp.voices = v->arr_s;
    return hp;
}

std::unique_ptr<tts_generation_runner> kokoro_model_loader::from_file(gguf_file * meta, int, bool, const generation_configuration & config) const {
    const kokoro_hparams hp = read_hparams(*meta);
    const gguf_value *   toks = meta->get("tokenizer.ggml.tokens");
    if (!toks) TTS_ABORT("The '%s' key must be set in order to support single pass tokenization.", "tokenizer.ggml.tokens");
    const int device = tts_load_device();
    return std::make_unique<kokoro_runner>(hp, new single_pass_tokenizer(toks->arr_s), device, config.voice);
}

kokoro_runner::kokoro_runner(const kokoro_hparams & hp_, single_pass_tokenizer * tok, int device, const std::string & voice_)
    : tts_generation_runner{kokoro_loader}, hp(hp_), tokenizer(tok), voice(voice_) {
    tts_hip_kokoro_desc d{};
    d.struct_size = sizeof(d);
    d.n_attn_heads = hp.n_attn_heads; d.n_recurrence = hp.n_recurrence; d.n_dp_layers = hp.n_duration_prediction_layers; d.f0_n_blocks = hp.f0_n_blocks;
    d.n_conv_layers = hp.n_conv_layers; d.n_decoder_blocks = hp.n_decoder_blocks; d.n_upsamples = hp.n_upsamples; d.n_kernels = hp.n_kernels;
    d.n_fft = hp.true_n_fft; d.hop = hp.stft_hop; d.harmonic_num = hp.harmonic_num; d.up_sampling_factor = hp.up_sampling_factor;
    d.out_conv_padding = hp.out_conv_padding; d.max_ctx = hp.max_context_length;
    d.attn_scale = hp.scale; d.upsample_scale = hp.upsample_scale; d.sample_rate = hp.sample_rate; d.sin_amp = hp.sin_amp; d.noise_std = hp.noise_std;
    d.voice_threshold = hp.voice_threshold;
    memcpy(d.up_stride, hp.up_stride, sizeof(d.up_stride)); memcpy(d.up_padding, hp.up_padding, sizeof(d.up_padding));
    memcpy(d.noise_stride, hp.noise_stride, sizeof(d.noise_stride)); memcpy(d.noise_padding, hp.noise_padding, sizeof(d.noise_padding));
    memcpy(d.res_padding, hp.res_padding, sizeof(d.res_padding)); memcpy(d.res_dilation, hp.res_dilation, sizeof(d.res_dilation));
    memcpy(d.noise_res_padding, hp.noise_res_padding, sizeof(d.noise_res_padding)); memcpy(d.noise_res_dilation, hp.noise_res_dilation, sizeof(d.noise_res_dilation));
    ctx = tts_hip_kokoro_create(device, &d);
    if (!ctx) TTS_ABORT("tts_hip_kokoro_create failed: %s\n", tts_hip_last_error());
    sampling_rate = 24000.0f;      // model.h:424
    supports_voices = true;
}

kokoro_runner::~kokoro_runner() { tts_hip_destroy(ctx); }

void kokoro_runner::assign_weight(const char * name, const gguf_tensor_view & t) {
    if (strncmp(name, "kokoro.", 7) != 0) TTS_ABORT("GGML_ASSERT(name_sv.starts_with(\"kokoro.\")) failed for tensor '%s'\n", name);   // model.cpp:1329
    if (!strncmp(name, "kokoro.voice_tensors.", 21)) uploaded_voices.insert(name + 21);
    if (!strcmp(name, "kokoro.duration_predictor.encode")) duration_hidden = (uint32_t) t.ne[1];
    if (!strcmp(name, "kokoro.duration_predictor.layers.1.gamma_weight")) style_half = (uint32_t) t.ne[0];
    hip_check(tts_hip_upload(ctx, name, t.type, t.n_dims, t.ne, t.data), name);
}

void kokoro_runner::prepare_post_load() {
    hip_check(tts_hip_finalize(ctx, nullptr), "tts_hip_finalize(kokoro)");
    if (duration_hidden == 0 || style_half == 0) TTS_ABORT("the Kokoro duration predictor tensors are missing from the GGUF file\n");
    if (voice.empty()) voice = "af_heart";   // propagate_voice_setting :1390-1396
    if (!uploaded_voices.count(voice)) TTS_ABORT("Failed to find Kokoro voice '%s' aborting.\n", voice.c_str());
}

std::vector<std::string_view> kokoro_runner::list_voices() {
    std::vector<std::string_view> out;
    for (const auto & v : uploaded_voices) out.emplace_back(v);
    std::sort(out.begin(), out.end());
    return out;
}

void kokoro_runner::run(const std::vector<uint32_t> & tokens) {
    const uint32_t n = (uint32_t) tokens.size();
    // the hidden states come back to the host and go in again, as in the reference (:1112-1113, :1261)
    std::vector<float> lens(n), hidden((size_t) n * (duration_hidden + style_half));
    hip_check(tts_hip_kokoro_durations(ctx, tokens.data(), n, voice.c_str(), lens.data(), hidden.data()), "tts_hip_kokoro_durations");
    size_t total = 0;
    for (float l : lens) total += (size_t) l;
    std::vector<float> noise(total * hp.up_sampling_factor * (hp.harmonic_num + 1));   // set_inputs :1255
    for (auto & v : noise) v = noise_dist(noise_engine);
    const size_t at = pcm.size();
    pcm.resize(at + total * hp.up_sampling_factor);
    hip_check(tts_hip_kokoro_generate(ctx, tokens.data(), n, lens.data(), hidden.data(), voice.c_str(), noise.data(), pcm.data() + at, nullptr, nullptr), "tts_hip_kokoro_generate");
    last_prompt_tokens.insert(last_prompt_tokens.end(), tokens.begin(), tokens.end());
    last_lengths.insert(last_lengths.end(), lens.begin(), lens.end());
}

// the chunking of kokoro_runner::generate (:1420-1446): one chunk when the phonemes fit the context (sentence punctuation
// removed), otherwise clause by clause
std::vector<std::vector<uint32_t>> kokoro_clause_chunks(const kokoro_hparams & hp, const single_pass_tokenizer & tok, const std::string & phonemes) {
    std::string p = replace_any(phonemes, "\n", " ");
    if (p.size() < hp.max_context_length - 2) {
        p = strip_spaces(replace_any(p, ".!?", ""));   // :1423
        if (p.empty()) return {};
        std::vector<uint32_t> tokens{hp.bos_token_id};
        tok.tokenize(p, tokens);
        tokens.push_back(hp.eos_token_id);
        return {tokens};
    }
    return kokoro_tokenize_chunks(hp, tok, split_any(p, ".!?"));
}

void kokoro_runner::generate(const char * prompt, tts_response & output, const generation_configuration & config) {
    voice = config.voice.empty() ? std::string("af_heart") : config.voice;
    if (!uploaded_voices.count(voice)) TTS_ABORT("Failed to find Kokoro voice '%s' aborting.\n", voice.c_str());
    output.data = nullptr;
    output.n_outputs = 0;
    pcm.clear();
    last_prompt_tokens.clear();
    last_lengths.clear();
    // the reference phonemizes here (:1415-1417: phonemizer.cpp rule tables or espeak); this runner is handed the phonemes.  A caller
    // written for the reference passes plain text, which would be read as if it were IPA: say so loudly instead of quietly
    // synthesising nonsense — once per runner as a notice (TTS_KOKORO_INPUT_IS_PHONEMES=1 acknowledges it), and on every call that
    // contains characters outside the phoneme vocabulary.
    if (!phoneme_notice_given && !getenv("TTS_KOKORO_INPUT_IS_PHONEMES")) {
        phoneme_notice_given = true;
        fprintf(stderr, "kokoro: NOTE this engine has no phonemizer: the prompt is read as IPA phonemes (the reference phonemizes text first, "
                        "kokoro/model.cpp:1415-1417).  Set TTS_KOKORO_INPUT_IS_PHONEMES=1 to silence this notice.\n");
    }
    const auto chunks = kokoro_clause_chunks(hp, *tokenizer, prompt);
    size_t unknown = 0, total_ids = 0;
    for (const auto & tokens : chunks)
        for (uint32_t id : tokens) { total_ids++; unknown += id == tokenizer->unknown_id; }
    if (unknown)
        fprintf(stderr, "kokoro: WARNING %zu of %zu symbols of the prompt are not in the phoneme vocabulary (plain text instead of phonemes?); "
                        "they are synthesised as the unknown token\n", unknown, total_ids);
    for (const auto & tokens : chunks) {
        // the reference's split lets a chunk reach max_context_length + 2 ids (bos + max_context_length + eos, model.cpp:1359-1372),
        // past what its own graphs are sized for; such a chunk is cut once more here so that every call fits the context
        const size_t inner_max = hp.max_context_length - 2;
        if (tokens.size() <= hp.max_context_length) { run(tokens); continue; }
        for (size_t at = 1; at + 1 < tokens.size(); at += inner_max) {
            std::vector<uint32_t> piece{hp.bos_token_id};
            piece.insert(piece.end(), tokens.begin() + (long) at, tokens.begin() + (long) std::min(at + inner_max, tokens.size() - 1));
            piece.push_back(hp.eos_token_id);
            run(piece);
        }
    }
    if (pcm.empty()) return;
    output.data = pcm.data();
    output.n_outputs = pcm.size();
}
