#include "parler_runner.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "gguf.h"

static void hip_check(int rc, const char * what) {
    if (rc != 0) TTS_ABORT("%s failed: %s\n", what, tts_hip_last_error());
}

parler_model_loader::parler_model_loader() : tts_model_loader{"parler-tts"} {}
const parler_model_loader parler_loader{};
void parler_register() {}

// parler_tts_model::prep_constants (model.cpp:51-108) + dac_model::prep_constants/prep_layers
// (dac_model.cpp:15-55): same keys, same aliases, same defaults.
static parler_hparams read_hparams(const gguf_file & m) {
    parler_hparams hp;
    if (!m.get_u32({"parler-tts.decoder.encode_length", "encode_length"}, hp.n_encode_length))
        TTS_ABORT("key 'parler-tts.decoder.encode_length' must be specified in gguf file.\n");
    m.get_u32({"parler-tts.decoder.hidden_size", "hidden_size"}, hp.hidden_size);
    m.get_u32({"parler-tts.decoder.output_heads", "output_heads"}, hp.n_output_heads);
    m.get_u32({"parler-tts.decoder.context_length", "ctx_length"}, hp.max_ctx_length);
    m.get_u32({"parler-tts.decoder.attention.head_count", "attn_heads"}, hp.n_attn_heads);
    m.get_u32({"parler-tts.decoder.out_vocab_size", "out_vocab_size"}, hp.output_vocab_size);
    m.get_u32({"parler-tts.decoder.audio_vocab_size", "audio_vocab_size"}, hp.audio_vocab_size);
    m.get_u32({"parler-tts.decoder.max_generation", "max_generation"}, hp.max_generation_size);
    m.get_u32({"parler-tts.decoder.num_hidden_layers", "num_hidden_layers"}, hp.n_layers);
    m.get_u32({"audio.bos_token_id", "bos_token_id"}, hp.bos_token_id);
    m.get_u32({"audio.eos_token_id", "eos_token_id"}, hp.eos_token_id);
    // the converter writes dac.up_scaling_factor but the reference reads dac.up_sampling_factor, so the
    // default 512 always applies there (dac_model.cpp:21-24); kept, and cross-checked against the strides below
    m.get_u32({"dac.up_sampling_factor", "up_sampling_factor"}, hp.up_sampling_factor);
    // The reference fixes the codec at 4 decoder blocks (dac_model.h:33) and aborts when one of their
    // stride/padding keys is missing (dac_model.cpp:37-47).  Extension: the block count is taken from how many
    // consecutive dac_layer_stride_i keys the file holds (>= 1), so non-standard codecs load too.
    uint32_t up = 1, n_found = 0;
    for (uint32_t i = 0; i < TTS_HIP_MAX_DAC_BLOCKS; i++) {
        const std::string sk = "dac_layer_stride_" + std::to_string(i), pk = "dac_layer_padding_" + std::to_string(i);
        const std::string dsk = "dac." + sk, dpk = "dac." + pk;
        if (!m.get_u32({dsk.c_str(), sk.c_str()}, hp.dac_stride[i])) {
            if (i == 0) TTS_ABORT("key %s must be specified in gguf file inorder to initialize the DAC audio decoder.\n", sk.c_str());
            break;
        }
        if (!m.get_u32({dpk.c_str(), pk.c_str()}, hp.dac_padding[i]))
            TTS_ABORT("key %s must be specified in gguf file inorder to initialize the DAC audio decoder.\n", pk.c_str());
        up *= hp.dac_stride[i];
        n_found++;
    }
    hp.dac_n_layers = n_found;
    if (up != hp.up_sampling_factor) hp.up_sampling_factor = up;  // non-standard codec: trust the layer strides
    // a file whose metadata cannot describe a model is refused here, not by a division further down
    if (hp.n_output_heads == 0 || hp.output_vocab_size == 0 || hp.hidden_size == 0 || hp.n_attn_heads == 0 || hp.hidden_size % hp.n_attn_heads ||
        hp.n_layers == 0 || hp.max_generation_size < 2 || hp.up_sampling_factor == 0)
        TTS_ABORT("parler-tts metadata out of range: %u output heads, vocabulary %u, hidden %u over %u heads, %u layers, max_generation %u\n", hp.n_output_heads,
                  hp.output_vocab_size, hp.hidden_size, hp.n_attn_heads, hp.n_layers, hp.max_generation_size);
    return hp;
}

std::unique_ptr<tts_generation_runner> parler_model_loader::from_file(gguf_file * meta, int, bool cpu_only,
                                                                      const generation_configuration & config) const {
    const parler_hparams hp = read_hparams(*meta);
    const int device = tts_load_device();
    (void) cpu_only;
    return std::make_unique<parler_runner>(hp, unigram_tokenizer_from_gguf(*meta), device, config.use_cross_attn);
}

parler_runner::parler_runner(const parler_hparams & hp_, unigram_tokenizer * tok, int device, bool cross)
    : tts_generation_runner{parler_loader}, hp(hp_), tokenizer(tok), use_cross_attn(cross), device_id(device) {
    tts_hip_desc d{};
    d.struct_size = sizeof(d);
    d.hidden_size = hp.hidden_size; d.n_layers = hp.n_layers; d.n_attn_heads = hp.n_attn_heads;
    d.n_output_heads = hp.n_output_heads; d.output_vocab_size = hp.output_vocab_size; d.max_ctx_length = hp.max_ctx_length;
    d.n_encode_length = hp.n_encode_length; d.use_cross_attn = cross ? 1 : 0;
    d.dac_n_blocks = hp.dac_n_layers;
    for (uint32_t i = 0; i < hp.dac_n_layers; i++) { d.dac_stride[i] = hp.dac_stride[i]; d.dac_padding[i] = hp.dac_padding[i]; }
    d.dac_max_frames = hp.max_generation_size;
    max_seqs = tts_load_max_seqs();
    st_codec_hold = (uint32_t) std::max(1, tts_thread_load_options().stream_codec_hold);
    d.max_seqs = max_seqs;
    {
        const tts_load_options & lo = tts_thread_load_options();
        if (lo.share_with) {
            share_ctx = (tts_hip_ctx *) lo.share_with->device_context();
            if (!share_ctx) TTS_ABORT("load: share_with names a runner that cannot share its weights\n");
        }
        declare_only = lo.declare_only || share_ctx != nullptr;
    }
    d.kv_type = getenv("TTS_HIP_KV_F16") ? TTS_HIP_F16 : TTS_HIP_F32;
    d.gelu_mode = 1;
    // one utterance: the reference layout (max_ctx_length positions, model.cpp:368-369); lock-step batches keep only
    // the positions generation can reach (check_stopping stops at max_generation, model.cpp:720-722)
    d.kv_positions = max_seqs > 1 ? hp.max_generation_size : 0;
    ctx = tts_hip_create(device, &d);
    if (!ctx) TTS_ABORT("tts_hip_create failed: %s\n", tts_hip_last_error());
    smp.n_output_heads = hp.n_output_heads;
    smp.vocab_size = hp.output_vocab_size;
    smp.eos_token_id = hp.eos_token_id;
    sampling_rate = 44100.0f;
}

parler_runner::~parler_runner() { tts_hip_destroy(ctx); }

void parler_runner::assign_weight(const char * name, const gguf_tensor_view & t) {
    // model.cpp:500-508 routes "audio_encoder." / "decoder." prefixes; the shim does the same by name
    // declare-only: the shape is all the device needs to lay its arena out; the bytes come from another context
    hip_check(tts_hip_upload(ctx, name, t.type, t.n_dims, t.ne, declare_only ? nullptr : t.data), name);
}

void parler_runner::prepare_post_load() {
    // prep_cross_key_values + kv cache init + graph reserve (model.cpp:704-713) all live in finalize
    if (share_ctx) {
        // same model, same device: use the loaded runner's arena (weights + precomputed cross K/V); own KV cache and stream
        if (tts_hip_arena_bytes(ctx) != tts_hip_arena_bytes(share_ctx)) TTS_ABORT("load: the runner to share weights with holds a different model\n");
        hip_check(tts_hip_finalize(ctx, tts_hip_arena_ptr(share_ctx)), "tts_hip_finalize(shared arena)");
        hip_check(tts_hip_arena_filled(ctx), "tts_hip_arena_filled");
    } else
    hip_check(tts_hip_finalize(ctx, nullptr), "tts_hip_finalize");
    logits.resize((size_t) hp.n_output_heads * hp.output_vocab_size);
    pcm.reserve((size_t) hp.max_generation_size * hp.up_sampling_factor);
}

// model.cpp:510-518: text_encoder_from_file (t5/model.cpp:365-402) with THIS runner's tokenizer, t5_runner::generate
// (:359-364: tokenize + EOS, run), then prep_cross_key_values with the response.  The encoder lives in its own device
// context for the duration of the call, as the reference builds and deletes a t5_runner per call.
void parler_runner::update_conditional_prompt(const char * file_path, const char * prompt) {
    std::string err;
    std::shared_ptr<gguf_file> meta = gguf_file::open(file_path, err);
    if (!meta) TTS_ABORT("text_encoder_from_file failed for file %s: %s\n", file_path, err.c_str());
    // t5_encoder::prep_constants (t5/model.cpp:123-164); defaults t5/model.h:43-52
    tts_hip_t5_desc td{};
    td.struct_size = sizeof(td);
    td.n_layers = 24; td.n_attn_heads = 32; td.hidden_size = 2048; td.max_ctx_length = 512; td.n_buckets = 32; td.output_size = 1536;
    uint32_t eos = 1, vocab = 0;
    meta->get_u32({"t5encoder.block_count"}, td.n_layers);
    meta->get_u32({"t5encoder.embedding_length"}, td.hidden_size);
    meta->get_u32({"t5encoder.attention.head_count"}, td.n_attn_heads);
    meta->get_u32({"t5encoder.context_length"}, td.max_ctx_length);
    meta->get_u32({"tokenizer.ggml.eos_token_id"}, eos);
    if (!meta->get_u32({"t5encoder.vocab_size"}, vocab)) TTS_ABORT("key 't5encoder.vocab_size' must be specified in gguf file.\n");
    meta->get_u32({"t5encoder.output_size"}, td.output_size);
    td.gelu_mode = 1;
    if (td.output_size != hp.hidden_size)
        TTS_ABORT("update_conditional_prompt: the encoder's output size %u differs from the decoder's hidden size %u\n", td.output_size, hp.hidden_size);

    tts_hip_ctx * t5 = tts_hip_t5_create(device_id, &td);
    if (!t5) TTS_ABORT("tts_hip_t5_create failed: %s\n", tts_hip_last_error());
    struct guard { tts_hip_ctx * c; ~guard() { tts_hip_destroy(c); } } g{t5};
    for (const gguf_tensor_view & t : meta->tensors) {
        if (!t.data || !*t.name) continue;
        hip_check(tts_hip_upload(t5, t.name, t.type, t.n_dims, t.ne, t.data), t.name);  // assign_to_t5_encoder keeps "t5encoder.*"
    }
    hip_check(tts_hip_finalize(t5, nullptr), "tts_hip_finalize(t5)");

    std::vector<uint32_t> tokens;
    tokenizer->tokenize(prompt, tokens);
    tokens.push_back(eos);
    if (tokens.size() > td.max_ctx_length || tokens.size() > 512)   // max_encode_length, model.h:69
        TTS_ABORT("update_conditional_prompt: %zu prompt tokens exceed the encoder context %u\n", tokens.size(), td.max_ctx_length);
    std::vector<float> enc(tokens.size() * (size_t) td.output_size);
    hip_check(tts_hip_t5_encode(t5, tokens.data(), (uint32_t) tokens.size(), enc.data()), "tts_hip_t5_encode");
    hip_check(tts_hip_parler_set_text_encoding(ctx, enc.data(), (uint32_t) tokens.size()), "tts_hip_parler_set_text_encoding");
    last_conditional_tokens = tokens;
}

// model.cpp:734-760, including the `next_index > size` bound (an index == size would read one past the
// end in the reference; such a frame is dropped here, the only defined outcome).
void parler_runner::adjust_output_tokens(const std::vector<uint32_t> & toks, std::vector<uint32_t> & filtered) const {
    const size_t size = toks.size(), nh = hp.n_output_heads;
    filtered.reserve(size);
    for (size_t i = 0; i < size / nh; i++) {
        bool remove = false;
        for (size_t ii = 0; ii < nh; ii++) {
            const size_t idx = i * nh + ii * nh + ii;
            if (idx >= size || toks[idx] >= hp.audio_vocab_size) { remove = true; break; }
        }
        if (remove) continue;
        for (size_t ii = 0; ii < nh; ii++) filtered.push_back(toks[i * nh + ii * nh + ii]);
    }
}

void parler_runner::generate(const char * sentence, tts_response & output, const generation_configuration & config) {
    smp.temperature = config.temperature;
    smp.repetition_penalty = config.repetition_penalty;
    smp.do_sample = config.sample;
    smp.top_k = (uint32_t) config.top_k;
    smp.top_p = config.top_p;
    smp.seed = config.seed;
    smp.n_calls = 0;
    if (config.use_cross_attn != use_cross_attn)
        TTS_ABORT("generate(): use_cross_attn differs from the value the model was loaded with (the reference only "
                  "loads the encoder_attn tensors when it is set at load time, model.cpp:202-237)\n");

    // batch_from_sentence (model.cpp:473-498)
    std::vector<uint32_t> prompt;
    tokenizer->tokenize(sentence, prompt);
    prompt.push_back(tokenizer->eos_token);
    last_prompt_tokens = prompt;
    smp.reset();
    hip_check(tts_hip_parler_reset(ctx), "tts_hip_parler_reset");
    output.data = nullptr;
    output.n_outputs = 0;
    if (prompt.size() >= hp.max_generation_size || prompt.size() >= hp.max_ctx_length) {
        fprintf(stderr, "prompt of %zu tokens leaves no room for generation\n", prompt.size());
        return;
    }
    hip_check(tts_hip_parler_prefill(ctx, 0, prompt.data(), (uint32_t) prompt.size(), 0), "tts_hip_parler_prefill");

    const uint32_t nh = hp.n_output_heads;
    uint32_t       current_position = (uint32_t) prompt.size();
    std::vector<uint32_t> & out_tokens = last_output_tokens;
    out_tokens.clear();

    // the sampler runs on the device (unless a head has more than 2048 logits).  Greedy never sees the repetition
    // penalty: sampler::max only reads last_token_ids, which stay -1 after reset() (sampler.cpp:71-80,185-204)
    const bool device_loop = !getenv("TTS_HOST_LOOP") && (!config.sample || hp.output_vocab_size <= 2048);
    if (device_loop) {
        // sampler::max / sampler::sample, the delay-pattern feed and the EOS flags run on the device; the host
        // synchronises in chunks only to learn whether check_stopping() would have fired.
        const uint32_t max_steps = hp.max_generation_size - current_position;
        std::vector<uint32_t> toks((size_t) max_steps * nh);
        uint32_t start = current_position, done = 0;
        if (config.sample) {
            // the U[0,1) draws sample() would make, call by call (sampler.cpp:47-50), drawn ahead
            std::vector<float> u((size_t) max_steps * nh);
            for (uint32_t s = 0; s < max_steps; s++) smp.draw_uniforms(u.data() + (size_t) s * nh);
            const tts_hip_sampling sp{smp.top_k, smp.top_p, smp.temperature, smp.repetition_penalty};
            hip_check(tts_hip_parler_generate_sampled(ctx, 1, &start, max_steps, hp.bos_token_id, hp.eos_token_id, &sp, u.data(), toks.data(), &done),
                      "tts_hip_parler_generate_sampled");
        } else
        hip_check(tts_hip_parler_generate_greedy(ctx, 1, &start, max_steps, hp.bos_token_id, hp.eos_token_id, toks.data(), &done),
                  "tts_hip_parler_generate_greedy");
        const uint32_t n = done ? done : max_steps;
        out_tokens.assign(toks.begin(), toks.begin() + (size_t) n * nh);
    } else {
        // generate_from_batch (model.cpp:762-792) with host sampling
        std::vector<uint32_t> ids(nh, hp.bos_token_id);
        std::vector<bool>     eos_seen(nh, false);
        int                   current_step = 0;  // batch.current_step of the decode that just ran
        for (;;) {
            // check_stopping (model.cpp:715-732)
            if (!out_tokens.empty()) {
                if (current_position >= hp.max_generation_size) break;
                bool all = true;
                for (uint32_t i = 0; i < nh; i++) {
                    eos_seen[i] = eos_seen[i] || out_tokens[out_tokens.size() - nh + i] == hp.eos_token_id;
                    all = all && eos_seen[i];
                }
                if (all) break;
            } else if (current_position >= hp.max_generation_size) {
                break;
            }
            current_step++;
            hip_check(tts_hip_parler_step(ctx, 1, ids.data(), &current_position, nullptr, logits.data()), "tts_hip_parler_step");
            smp.sample(logits.data(), out_tokens);
            current_position += 1;
            const uint32_t * last = out_tokens.data() + out_tokens.size() - nh;
            for (uint32_t i = 0; i < nh; i++)
                ids[i] = current_step > (int) i ? (eos_seen[i] ? hp.eos_token_id : last[i]) : hp.bos_token_id;
        }
    }

    std::vector<uint32_t> filtered;
    adjust_output_tokens(out_tokens, filtered);
    const uint32_t frames = (uint32_t) (filtered.size() / nh);
    pcm.assign((size_t) frames * hp.up_sampling_factor, 0.0f);
    if (frames) hip_check(tts_hip_dac_decode(ctx, filtered.data(), frames, pcm.data()), "tts_hip_dac_decode");
    output.data = pcm.data();
    output.n_outputs = pcm.size();
}

void parler_runner::generate_batch(const std::vector<std::string> & sentences, std::vector<tts_response> & outputs,
                                   const generation_configuration & config) {
    const uint32_t n_all = (uint32_t) sentences.size(), nh = hp.n_output_heads;
    outputs.assign(n_all, tts_response{});
    if (n_all == 0) return;
    if (n_all > max_seqs) TTS_ABORT("generate_batch: %u utterances but the runner was loaded with max_seqs=%u (TTS_HIP_MAX_SEQS)\n", n_all, max_seqs);
    if (config.use_cross_attn != use_cross_attn) TTS_ABORT("generate_batch: use_cross_attn differs from load time\n");
    // batch_from_sentence per utterance; an utterance whose prompt leaves no room gets an empty response, exactly as generate() does
    std::vector<uint32_t> ids, lens, start, row_of;
    for (uint32_t i = 0; i < n_all; i++) {
        std::vector<uint32_t> p;
        tokenizer->tokenize(sentences[i], p);
        p.push_back(tokenizer->eos_token);
        if (p.size() >= hp.max_generation_size || p.size() >= hp.max_ctx_length) {
            fprintf(stderr, "prompt %u of %zu tokens leaves no room for generation\n", i, p.size());
            continue;
        }
        row_of.push_back(i);
        lens.push_back((uint32_t) p.size());
        start.push_back((uint32_t) p.size());
        ids.insert(ids.end(), p.begin(), p.end());
    }
    const uint32_t n = (uint32_t) row_of.size();
    last_batch_tokens.assign(n_all, {});
    if (n == 0) return;
    hip_check(tts_hip_parler_reset(ctx), "tts_hip_parler_reset");
    hip_check(tts_hip_parler_prefill_batch(ctx, n, nullptr, ids.data(), lens.data(), nullptr), "tts_hip_parler_prefill_batch");
    // every utterance gets the steps generate() would give it alone (max_generation - its own prompt): the loop runs as long as the
    // shortest prompt needs; the device marks a row finished when its position reaches max_generation and lets it idle there
    const uint32_t shortest = *std::min_element(start.begin(), start.end());
    const uint32_t max_steps = hp.max_generation_size - shortest;
    std::vector<std::vector<uint32_t>> row_tokens(n);

    if (!getenv("TTS_HOST_LOOP") && (!config.sample || hp.output_vocab_size <= 2048)) {
        std::vector<uint32_t> toks((size_t) max_steps * n * nh), done(n);
        if (config.sample) {
            // one sampler state per utterance, seeded like the host loop below: uniforms [step][utterance][head]
            std::vector<float> u((size_t) max_steps * n * nh);
            for (uint32_t i = 0; i < n; i++) {
                sampler si = smp;
                si.seed = config.seed; si.n_calls = 0;   // n separate generate() calls would each seed with config.seed
                for (uint32_t s = 0; s < max_steps; s++) si.draw_uniforms(u.data() + ((size_t) s * n + i) * nh);
            }
            const tts_hip_sampling sp{(uint32_t) config.top_k, config.top_p, config.temperature, config.repetition_penalty};
            hip_check(tts_hip_parler_generate_sampled(ctx, n, start.data(), max_steps, hp.bos_token_id, hp.eos_token_id, &sp, u.data(), toks.data(), done.data()),
                      "tts_hip_parler_generate_sampled");
        } else
        hip_check(tts_hip_parler_generate_greedy(ctx, n, start.data(), max_steps, hp.bos_token_id, hp.eos_token_id, toks.data(), done.data()),
                  "tts_hip_parler_generate_greedy");
        for (uint32_t i = 0; i < n; i++) {
            // check_stopping per sequence: EOS on every head, or position == max_generation
            uint32_t steps = done[i] ? done[i] : max_steps;
            steps = std::min(steps, hp.max_generation_size - start[i]);
            for (uint32_t s = 0; s < steps; s++)
                row_tokens[i].insert(row_tokens[i].end(), toks.begin() + ((size_t) s * n + i) * nh, toks.begin() + ((size_t) s * n + i + 1) * nh);
        }
    } else {
        // host sampling, one sampler state per utterance; finished sequences keep stepping on EOS inputs (their
        // tokens are no longer recorded) until all are done
        std::vector<sampler> smps(n, smp);
        for (uint32_t i = 0; i < n; i++) {
            smps[i].temperature = config.temperature; smps[i].repetition_penalty = config.repetition_penalty;
            smps[i].do_sample = config.sample; smps[i].top_k = (uint32_t) config.top_k; smps[i].top_p = config.top_p;
            smps[i].seed = config.seed; smps[i].n_calls = 0;
            smps[i].reset();
        }
        std::vector<uint32_t> in_ids((size_t) n * nh, hp.bos_token_id), pos(start);
        std::vector<std::vector<bool>> eos_seen(n, std::vector<bool>(nh, false));
        std::vector<bool> finished(n, false);
        std::vector<float> lg((size_t) n * nh * hp.output_vocab_size);
        for (uint32_t step = 1; step <= max_steps; step++) {
            bool all_done = true;
            for (uint32_t i = 0; i < n; i++) {
                if (finished[i]) continue;
                auto & t = row_tokens[i];
                if (!t.empty()) {
                    if (pos[i] >= hp.max_generation_size) { finished[i] = true; continue; }
                    bool all = true;
                    for (uint32_t h = 0; h < nh; h++) {
                        eos_seen[i][h] = eos_seen[i][h] || t[t.size() - nh + h] == hp.eos_token_id;
                        all = all && eos_seen[i][h];
                    }
                    if (all) { finished[i] = true; continue; }
                }
                all_done = false;
            }
            if (all_done) break;
            hip_check(tts_hip_parler_step(ctx, n, in_ids.data(), pos.data(), nullptr, lg.data()), "tts_hip_parler_step");
            for (uint32_t i = 0; i < n; i++) {
                if (pos[i] + 1 < hp.max_generation_size) pos[i] += 1;  // finished rows idle on their last position
                if (finished[i]) continue;
                auto & t = row_tokens[i];
                smps[i].sample(lg.data() + (size_t) i * nh * hp.output_vocab_size, t);
                const uint32_t * last = t.data() + t.size() - nh;
                for (uint32_t h = 0; h < nh; h++)
                    in_ids[(size_t) i * nh + h] = step > h ? (eos_seen[i][h] ? hp.eos_token_id : last[h]) : hp.bos_token_id;
            }
        }
    }

    std::vector<uint32_t> codes, frames(n);
    for (uint32_t i = 0; i < n; i++) {
        std::vector<uint32_t> f;
        adjust_output_tokens(row_tokens[i], f);
        frames[i] = (uint32_t) (f.size() / nh);
        codes.insert(codes.end(), f.begin(), f.end());
        last_batch_tokens[row_of[i]] = std::move(row_tokens[i]);
    }
    size_t total = 0;
    for (uint32_t f : frames) total += (size_t) f * hp.up_sampling_factor;
    pcm.assign(total, 0.0f);
    if (total) hip_check(tts_hip_dac_decode_batch(ctx, codes.data(), frames.data(), n, pcm.data()), "tts_hip_dac_decode_batch");
    size_t off = 0;
    for (uint32_t i = 0; i < n; i++) {
        outputs[row_of[i]].data = pcm.data() + off;
        outputs[row_of[i]].n_outputs = (size_t) frames[i] * hp.up_sampling_factor;
        off += outputs[row_of[i]].n_outputs;
    }
}

// ---- continuous batching (common.h; tts_hip_parler_stream_* underneath) ----------------------------------------------------------------
static constexpr uint32_t STREAM_CHUNK = 32;     // decode steps between two look-in points (what generate_loop's compaction uses)
static constexpr size_t   STREAM_CODEC_GROUP = 64;   // finished utterances per codec pass (the device's pass size); flushed when the session drains

void parler_runner::stream_begin(const generation_configuration & config) {
    if (stream_capacity() == 0) TTS_ABORT("stream_begin: the runner was loaded with max_seqs=%u; a session needs >= 2 (TTS_HIP_MAX_SEQS)\n", max_seqs);
    if (config.use_cross_attn != use_cross_attn) TTS_ABORT("stream_begin: use_cross_attn differs from load time\n");
    if (config.sample && hp.output_vocab_size > 2048) TTS_ABORT("stream_begin: sampling over %u logits per head is host-only\n", hp.output_vocab_size);
    if (st_on) stream_end();
    const uint32_t slots = stream_capacity();
    st_cfg = config;
    st_max_steps = hp.max_generation_size - 1;   // an utterance ends at position max_generation (check_stopping): at most this many steps after a 1-id prompt
    const tts_hip_sampling sp{(uint32_t) config.top_k, config.top_p, config.temperature, config.repetition_penalty};
    hip_check(tts_hip_parler_stream_begin(ctx, slots, st_max_steps, hp.bos_token_id, hp.eos_token_id, config.sample ? &sp : nullptr), "tts_hip_parler_stream_begin");
    st_free.clear();
    for (uint32_t s = slots; s-- > 0;) st_free.push_back(s);   // pop_back hands out slot 0 first
    st_ticket.assign(slots, 0);
    st_start.assign(slots, 0);
    st_wait.clear(); st_codec.clear(); st_pcm.clear();
    st_live = 0;
    st_codec_held = 0;
    st_on = true;
}

void parler_runner::stream_submit(size_t ticket, const std::string & sentence) {
    if (!st_on) TTS_ABORT("stream_submit: no session (stream_begin)\n");
    if (st_free.empty()) TTS_ABORT("stream_submit: no free row (stream_free() == 0)\n");
    pending p;
    p.ticket = ticket;
    tokenizer->tokenize(sentence, p.prompt);
    p.prompt.push_back(tokenizer->eos_token);
    if (p.prompt.size() >= hp.max_generation_size || p.prompt.size() >= hp.max_ctx_length) {
        // generate() answers such a prompt with an empty response: the session does the same at its next step
        fprintf(stderr, "prompt of %zu tokens leaves no room for generation\n", p.prompt.size());
        st_codec.push_back(decoded{ticket, {}});
        return;
    }
    p.slot = st_free.back();
    st_free.pop_back();
    st_wait.push_back(std::move(p));
    st_live++;
}

void parler_runner::stream_step(std::vector<stream_result> & finished) {
    if (!st_on) TTS_ABORT("stream_step: no session (stream_begin)\n");
    finished.clear();
    const uint32_t nh = hp.n_output_heads;
    if (!st_wait.empty()) {   // the newcomers: one prefill side batch, then rows of the lock-step forward
        std::vector<uint32_t> slots, ids, lens;
        std::vector<float> uni;
        for (auto & p : st_wait) {
            slots.push_back(p.slot);
            lens.push_back((uint32_t) p.prompt.size());
            ids.insert(ids.end(), p.prompt.begin(), p.prompt.end());
            st_ticket[p.slot] = p.ticket;
            st_start[p.slot] = (uint32_t) p.prompt.size();
            if (st_cfg.sample) {   // the utterance's own sampler, seeded as a generate() call of its own would be
                sampler si = smp;
                si.seed = st_cfg.seed; si.n_calls = 0;
                const size_t o = uni.size();
                uni.resize(o + (size_t) st_max_steps * nh);
                for (uint32_t s = 0; s < st_max_steps; s++) si.draw_uniforms(uni.data() + o + (size_t) s * nh);
            }
        }
        hip_check(tts_hip_parler_stream_admit(ctx, (uint32_t) slots.size(), slots.data(), ids.data(), lens.data(), st_cfg.sample ? uni.data() : nullptr),
                  "tts_hip_parler_stream_admit");
        st_wait.clear();
    }
    std::vector<uint32_t> fs(stream_capacity()), fn(stream_capacity());
    uint32_t nf = 0;
    hip_check(tts_hip_parler_stream_run(ctx, STREAM_CHUNK, &nf, fs.data(), fn.data()), "tts_hip_parler_stream_run");
    for (uint32_t i = 0; i < nf; i++) {
        const uint32_t slot = fs[i];
        const uint32_t steps = std::min(fn[i], hp.max_generation_size - st_start[slot]);   // what generate() would have run alone
        std::vector<uint32_t> toks((size_t) steps * nh);
        hip_check(tts_hip_parler_stream_collect(ctx, slot, steps, toks.data()), "tts_hip_parler_stream_collect");
        decoded d;
        d.ticket = st_ticket[slot];
        adjust_output_tokens(toks, d.frames);
        st_codec.push_back(std::move(d));
        st_free.push_back(slot);
        st_live--;
    }
    // the codec: whole groups of 64 as they fill (the device's pass size); what is left goes out when nothing is generating any more, and in any case
    // one look-in interval after it finished — a finished utterance waits at most STREAM_CHUNK decode steps for company, never for a group to fill
    // (round 4 held a finished request's audio until 63 more utterances had finished: under steady arrivals with fewer than 64 rows, for ever)
    const bool drain = st_live == 0;
    size_t take = st_codec.size() >= STREAM_CODEC_GROUP ? st_codec.size() / STREAM_CODEC_GROUP * STREAM_CODEC_GROUP : 0;
    if (!take && !st_codec.empty() && (drain || st_codec_held >= st_codec_hold)) take = st_codec.size();
    st_codec_held = (take < st_codec.size()) ? (take ? 0 : st_codec_held + 1) : 0;
    if (take) {
        std::vector<uint32_t> codes, frames(take);
        size_t total = 0;
        for (size_t i = 0; i < take; i++) {
            frames[i] = (uint32_t) (st_codec[i].frames.size() / nh);
            codes.insert(codes.end(), st_codec[i].frames.begin(), st_codec[i].frames.end());
            total += (size_t) frames[i] * hp.up_sampling_factor;
        }
        pcm.assign(total, 0.0f);
        if (total) hip_check(tts_hip_dac_decode_batch(ctx, codes.data(), frames.data(), (uint32_t) take, pcm.data()), "tts_hip_dac_decode_batch");
        size_t off = 0;
        for (size_t i = 0; i < take; i++) {
            stream_result r;
            r.ticket = st_codec[i].ticket;
            r.audio.data = pcm.data() + off;
            r.audio.n_outputs = (size_t) frames[i] * hp.up_sampling_factor;
            off += r.audio.n_outputs;
            finished.push_back(r);
        }
        st_codec.erase(st_codec.begin(), st_codec.begin() + (std::ptrdiff_t) take);
    }
}

void parler_runner::stream_end() {
    if (!st_on) return;
    (void) tts_hip_parler_stream_end(ctx);
    st_on = false;
    st_wait.clear(); st_codec.clear();
    st_live = 0;
}
