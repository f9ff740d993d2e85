// quantize_main.cpp — the `quantize` command line (examples/quantize/quantize.cpp:22-57: same flags, aliases and
// type names).  Links only the GGUF reader and the quantiser: it runs on a machine without a GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>

#include "quantize.h"

static const std::map<std::string, int> TYPES = {{"FP16", TTS_QTYPE_F16}, {"F16", TTS_QTYPE_F16},   {"Q4_0", TTS_QTYPE_Q4_0}, {"Q4", TTS_QTYPE_Q4_0},
                                                 {"Q5_0", TTS_QTYPE_Q5_0}, {"Q5", TTS_QTYPE_Q5_0}, {"Q8_0", TTS_QTYPE_Q8_0}, {"Q8", TTS_QTYPE_Q8_0}};

static void help() {
    puts("--model-path (-mp):\n    (REQUIRED) The local path of the gguf model file to quantize.\n"
         "--quantized-model-path (-qp):\n    (REQUIRED) The path to save the model in a quantized format.\n"
         "--quantized-type (-qt):\n    (OPTIONAL) FP16|F16|Q4_0|Q4|Q5_0|Q5|Q8_0|Q8. Defaults to Q4_0.\n"
         "--n-threads (-nt):\n    (OPTIONAL) The number of cpu threads to run the quantization process with. Defaults to known hardware concurrency.\n"
         "--convert-dac-to-f16 (-df):\n    (OPTIONAL) Whether to convert the DAC audio decoder model to a 16 bit float.\n"
         "--quantize-output-heads (-qh):\n    (OPTIONAL) Whether to quantize the output heads.\n"
         "--quantize-text-embedding (-qe):\n    (OPTIONAL) Whether to quantize the input text embededings (Parler TTS only).\n"
         "--quantize-cross-attn-kv (-qkv):\n    (OPTIONAL) Whether to quantize the cross attention keys and values (Parler TTS only).\n"
         "--convert-non-quantized-to-f16 (-nqf):\n    (OPTIONAL) Whether to convert quantization incompatible tensors to 16 bit precision (Kokoro only).");
}

int main(int argc, const char ** argv) {
    quantization_params qp;
    qp.n_threads = std::max(1u, std::thread::hardware_concurrency());
    std::string in, out, qtype = "Q4_0";
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto is = [&](const char * l, const char * s) { return a == l || a == s; };
        auto value = [&]() -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "ERROR: %s expects a value.\n", a.c_str()); exit(1); }
            return argv[++i];
        };
        if (is("--help", "-h")) { help(); return 0; }
        else if (is("--model-path", "-mp")) in = value();
        else if (is("--quantized-model-path", "-qp")) out = value();
        else if (is("--quantized-type", "-qt")) qtype = value();
        else if (is("--n-threads", "-nt")) qp.n_threads = (uint32_t) std::max(1, atoi(value()));
        else if (is("--convert-dac-to-f16", "-df")) qp.convert_dac_to_f16 = true;
        else if (is("--quantize-output-heads", "-qh")) qp.quantize_output_heads = true;
        else if (is("--quantize-text-embedding", "-qe")) qp.quantize_text_embeddings = true;
        else if (is("--quantize-cross-attn-kv", "-qkv")) qp.quantize_cross_attn_kv = true;
        else if (is("--convert-non-quantized-to-f16", "-nqf")) qp.convert_non_quantizable_to_f16 = true;
        else { fprintf(stderr, "ERROR: unknown argument %s\n", a.c_str()); return 1; }
    }
    if (in.empty() || out.empty()) { fprintf(stderr, "ERROR: --model-path and --quantized-model-path are required.\n"); return 1; }
    const auto t = TYPES.find(qtype);
    if (t == TYPES.end()) { fprintf(stderr, "ERROR: %s is not a valid quantization type.\n", qtype.c_str()); return 1; }
    qp.quantize_type = t->second;
    quantize_gguf(in.c_str(), out.c_str(), qp);
    return 0;
}
