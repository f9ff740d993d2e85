#include "kokoro_runner.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <chrono>
#include <thread>
#include <type_traits>

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

kokoro_runner::kokoro_runner(const kokoro_hparams & hp_, single_pass_tokenizer * tok, int device_, const std::string & voice_)
    : tts_generation_runner{kokoro_loader}, hp(hp_), tokenizer(tok), voice(voice_), device(device_) {
    {
        const tts_load_options & lo = tts_thread_load_options();
        if (lo.share_with) {
            share_ctx = (tts_hip_ctx *) lo.share_with->device_context();
            if (!share_ctx) TTS_ABORT("load: share_with names a runner that cannot share its weights\n");
        }
        declare_only = lo.declare_only || share_ctx != nullptr;
        if (lo.max_seqs > 0 || getenv("TTS_HIP_MAX_SEQS")) lanes_max = std::min<uint32_t>(tts_load_max_seqs(), 16);   // an explicit 1: one context, as the reference
        if (const char * e = getenv("TTS_KOKORO_LANES")) lanes_max = (uint32_t) std::clamp(atoi(e), 1, 16);
    }
    tts_hip_kokoro_desc & d = desc;
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
    lanes.push_back(ctx);
    sampling_rate = 24000.0f;      // model.h:424
    supports_voices = true;
}

kokoro_runner::~kokoro_runner() {
    for (size_t i = lanes.size(); i-- > 1;) tts_hip_destroy(lanes[i]);   // arenas are reference counted: the order does not matter
    tts_hip_destroy(ctx);
}

// lane i > 0: a further device context (own stream and scratch) declared with the same tensors and finalized on lane 0's arena
tts_hip_ctx * kokoro_runner::lane(size_t i) {
    while (lanes.size() <= i) {
        tts_hip_ctx * c = tts_hip_kokoro_create(device, &desc);
        if (!c) TTS_ABORT("tts_hip_kokoro_create (lane %zu) failed: %s\n", lanes.size(), tts_hip_last_error());
        for (const auto & t : decls) hip_check(tts_hip_upload(c, t.name.c_str(), t.type, t.n_dims, t.ne, nullptr), t.name.c_str());
        if (tts_hip_arena_bytes(c) != tts_hip_arena_bytes(ctx)) TTS_ABORT("kokoro lane: arena layout differs from the loaded runner's\n");
        hip_check(tts_hip_finalize(c, tts_hip_arena_ptr(ctx)), "tts_hip_finalize(kokoro lane)");
        hip_check(tts_hip_arena_filled(c), "tts_hip_arena_filled(kokoro lane)");
        lanes.push_back(c);
    }
    return lanes[i];
}

void kokoro_runner::assign_weight(const char * name, const gguf_tensor_view & t) {
    if (strncmp(name, "kokoro.", 7) != 0) TTS_ABORT("GGML_ASSERT(name_sv.starts_with(\"kokoro.\")) failed for tensor '%s'\n", name);   // model.cpp:1329
    if (!strncmp(name, "kokoro.voice_tensors.", 21)) uploaded_voices.insert(name + 21);
    if (!strcmp(name, "kokoro.duration_predictor.encode")) duration_hidden = (uint32_t) t.ne[1];
    if (!strcmp(name, "kokoro.duration_predictor.layers.1.gamma_weight")) style_half = (uint32_t) t.ne[0];
    tensor_decl dcl{name, t.type, t.n_dims, {t.ne[0], t.ne[1], t.ne[2], t.ne[3]}};
    decls.push_back(std::move(dcl));
    // declare-only: the shape is all the device needs to lay its arena out; the bytes are another runner's (share_with) or arrive by broadcast
    hip_check(tts_hip_upload(ctx, name, t.type, t.n_dims, t.ne, declare_only ? nullptr : t.data), name);
}

void kokoro_runner::prepare_post_load() {
    if (share_ctx) {
        if (tts_hip_arena_bytes(ctx) != tts_hip_arena_bytes(share_ctx)) TTS_ABORT("load: the runner to share weights with holds a different model\n");
        hip_check(tts_hip_finalize(ctx, tts_hip_arena_ptr(share_ctx)), "tts_hip_finalize(kokoro, shared arena)");
        hip_check(tts_hip_arena_filled(ctx), "tts_hip_arena_filled");
    } else
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

// ---- the source noise: ONE minstd stream (random_uniform_gen, util.cpp:65-71), drawn in parallel stretches ---------------------------
uint32_t minstd0_jump(uint32_t state, uint64_t k) {
    const uint64_t m = 2147483647ull;
    uint64_t a = 16807ull, f = 1;
    for (; k; k >>= 1) {
        if (k & 1) f = f * a % m;
        a = a * a % m;
    }
    return (uint32_t) ((uint64_t) state * f % m);
}

static_assert(std::is_same<std::default_random_engine, std::minstd_rand0>::value, "random_uniform_gen's engine is minstd_rand0 here (util.cpp:65-71)");

static uint32_t engine_state(const std::default_random_engine & e) {
    std::default_random_engine c = e;
    const uint32_t next = (uint32_t) c();                                   // x1 = a x0 mod m  ->  x0 = x1 a^-1 mod m  (a^-1 = a^(m - 2))
    return minstd0_jump(next, 2147483647ull - 2);
}

// out[0 .. n) = the draws an engine in `state` would give through uniform_real_distribution<float>(0, 1), by `threads` host threads: stretch j starts
// from the state jumped ahead to its first draw and uses the standard library's own engine and distribution, so the values are the sequential ones
// (a 400-id clause needs 6.5 M draws = 18 ms on one core, a third of a synthesis)
void minstd0_draw_uniform(uint32_t state, size_t n, float * out, unsigned threads) {
    auto stretch = [=](size_t a, size_t b) {
        std::default_random_engine            eng(minstd0_jump(state, a));
        std::uniform_real_distribution<float> dist{0.0f, 1.0f};
        for (size_t i = a; i < b; i++) out[i] = dist(eng);
    };
    threads = (unsigned) std::min<size_t>(std::max(1u, threads), n / 65536 + 1);
    if (threads <= 1) { stretch(0, n); return; }
    const size_t per = (n + threads - 1) / threads;
    std::vector<std::thread> th;
    for (unsigned j = 1; j < threads; j++) th.emplace_back(stretch, std::min(n, j * per), std::min(n, (j + 1) * per));
    stretch(0, std::min(n, per));
    for (auto & t : th) t.join();
}

static unsigned noise_threads(unsigned lanes) {
    if (const char * e = getenv("TTS_KOKORO_NOISE_THREADS")) return (unsigned) std::clamp(atoi(e), 1, 64);
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return std::clamp(hw / std::max(1u, lanes), 1u, 8u);
}

void kokoro_runner::run(const std::vector<uint32_t> & tokens) {
    const uint32_t n = (uint32_t) tokens.size();
    // the hidden states come back to the host and go in again, as in the reference (:1112-1113, :1261)
    std::vector<float> lens(n), hidden((size_t) n * (duration_hidden + style_half));
    hip_check(tts_hip_kokoro_durations(ctx, tokens.data(), n, voice.c_str(), lens.data(), hidden.data()), "tts_hip_kokoro_durations");
    size_t total = 0;
    for (float l : lens) total += (size_t) l;
    std::vector<float> noise(total * hp.up_sampling_factor * (hp.harmonic_num + 1));   // set_inputs :1255
    const uint32_t st = engine_state(noise_engine);
    minstd0_draw_uniform(st, noise.size(), noise.data(), noise_threads(1));
    noise_engine.seed(minstd0_jump(st, noise.size()));                                  // the engine after noise.size() draws
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

// ---- generate_batch: clauses of n utterances through `lanes_max` contexts on one weight arena -------------------------------------

void kokoro_runner::generate_batch(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs, const generation_configuration & config) {
    const size_t n = sentences.size();
    if (n <= 1 || lanes_max <= 1) { tts_generation_runner::generate_batch(sentences, outputs, config); return; }
    voice = config.voice.empty() ? std::string("af_heart") : config.voice;
    if (!uploaded_voices.count(voice)) TTS_ABORT("Failed to find Kokoro voice '%s' aborting.\n", voice.c_str());
    if (!phoneme_notice_given && !getenv("TTS_KOKORO_INPUT_IS_PHONEMES")) {
        phoneme_notice_given = true;
        fprintf(stderr, "kokoro: NOTE this engine has no phonemizer: the prompts are read as IPA phonemes (the reference phonemizes text first, "
                        "kokoro/model.cpp:1415-1417).  Set TTS_KOKORO_INPUT_IS_PHONEMES=1 to silence this notice.\n");
    }
    // clauses in the order sequential generate() calls would run them (the cut of over-long chunks as in generate)
    struct piece { size_t utt = 0; std::vector<uint32_t> tokens; std::vector<float> lens, hidden, pcm; size_t frames = 0; };
    std::vector<piece> work;
    const size_t inner_max = hp.max_context_length - 2;
    for (size_t u = 0; u < n; u++)
        for (const auto & tokens : kokoro_clause_chunks(hp, *tokenizer, sentences[u])) {
            auto add = [&](std::vector<uint32_t> t) { work.emplace_back(); work.back().utt = u; work.back().tokens = std::move(t); };
            if (tokens.size() <= hp.max_context_length) { add(tokens); continue; }
            for (size_t at = 1; at + 1 < tokens.size(); at += inner_max) {
                std::vector<uint32_t> p{hp.bos_token_id};
                p.insert(p.end(), tokens.begin() + (long) at, tokens.begin() + (long) std::min(at + inner_max, tokens.size() - 1));
                p.push_back(hp.eos_token_id);
                add(std::move(p));
            }
        }
    const size_t n_lanes = std::min<size_t>(lanes_max, work.size());
    for (size_t i = 0; i < n_lanes; i++) lane(i);
    // the noise stream: clause i's stretch starts where clause i - 1's ends; a lane takes its start state when every earlier clause's length is known
    std::mutex              mu;
    std::condition_variable cv;
    size_t                  turn = 0;                      // clauses whose stretch has been reserved
    uint32_t                state = engine_state(noise_engine);
    std::atomic<size_t>     next{0};
    std::atomic<bool>       failed{false};
    std::string             error;
    const size_t            per_frame = (size_t) hp.up_sampling_factor * (hp.harmonic_num + 1);
    const unsigned n_noise_threads = noise_threads((unsigned) n_lanes);
    const bool trace = getenv("TTS_KOKORO_BATCH_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    auto worker = [&](size_t li) {
        tts_hip_ctx * c = lanes[li];
        std::vector<float> noise;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= work.size()) return;
            piece & w = work[i];
            const uint32_t nt = (uint32_t) w.tokens.size();
            w.lens.resize(nt);
            w.hidden.resize((size_t) nt * (duration_hidden + style_half));
            const double t0 = now();
            int rc = failed ? 1 : tts_hip_kokoro_durations(c, w.tokens.data(), nt, voice.c_str(), w.lens.data(), w.hidden.data());
            const double t1 = now();
            std::string err = rc ? tts_hip_last_error() : "";
            for (float l : w.lens) w.frames += rc ? 0 : (size_t) l;
            uint32_t start;
            {   // reserve the stretch in clause order, also when this clause failed (the others must not wait forever)
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return turn == i; });
                start = state;
                state = minstd0_jump(state, (uint64_t) w.frames * per_frame);
                turn++;
                if (rc && !failed.exchange(true)) error = "tts_hip_kokoro_durations failed: " + err;
            }
            cv.notify_all();
            if (rc || failed) continue;
            const double t2 = now();
            noise.resize(w.frames * per_frame);
            minstd0_draw_uniform(start, noise.size(), noise.data(), n_noise_threads);
            w.pcm.resize(w.frames * hp.up_sampling_factor);
            const double t3 = now();
            rc = tts_hip_kokoro_generate(c, w.tokens.data(), nt, w.lens.data(), w.hidden.data(), voice.c_str(), noise.data(), w.pcm.data(), nullptr, nullptr);
            if (trace) fprintf(stderr, "kokoro batch: clause %zu lane %zu  start %.1f  durations %.1f  turn %.1f  noise %.1f  generate %.1f ms\n", i, li, t0 - t_begin, t1 - t0, t2 - t1, t3 - t2, now() - t3);
            if (rc) {
                std::lock_guard<std::mutex> lk(mu);
                if (!failed.exchange(true)) error = std::string("tts_hip_kokoro_generate failed: ") + tts_hip_last_error();
            }
        }
    };
    std::vector<std::thread> threads;
    for (size_t li = 1; li < n_lanes; li++) threads.emplace_back(worker, li);
    worker(0);
    for (auto & t : threads) t.join();
    if (failed) TTS_ABORT("%s\n", error.c_str());
    noise_engine.seed(state);                               // as after the same utterances through generate()
    batch_store_.assign(n, {});
    outputs.assign(n, tts_response{});
    last_prompt_tokens.clear();
    last_lengths.clear();
    for (const auto & w : work) {
        batch_store_[w.utt].insert(batch_store_[w.utt].end(), w.pcm.begin(), w.pcm.end());
        last_prompt_tokens.insert(last_prompt_tokens.end(), w.tokens.begin(), w.tokens.end());
        last_lengths.insert(last_lengths.end(), w.lens.begin(), w.lens.end());
    }
    for (size_t u = 0; u < n; u++)
        if (!batch_store_[u].empty()) { outputs[u].data = batch_store_[u].data(); outputs[u].n_outputs = batch_store_[u].size(); }
}
