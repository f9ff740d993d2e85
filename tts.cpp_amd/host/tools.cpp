// tools.cpp — two small front ends over the runner API, built as `tts-cli` and `perf_battery`
// (same binary, mode chosen by argv[0] or --perf-battery).  They exist to show that a caller written
// against the reference's API (examples/cli/cli.cpp:79-95, examples/perf_battery/perf_battery.cpp:102-116)
// runs on this engine; the reference's own CLI/server sources are not reproduced here (out of scope).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

static void write_wav16(const char * path, const float * x, size_t n, int rate) {
    FILE * f = fopen(path, "wb");
    if (!f) { perror(path); exit(1); }
    const uint32_t data_bytes = (uint32_t) (n * 2), riff = 36 + data_bytes, fmt_len = 16, byte_rate = (uint32_t) rate * 2, r = (uint32_t) rate;
    const uint16_t pcm = 1, ch = 1, align = 2, bits = 16;
    fwrite("RIFF", 1, 4, f); fwrite(&riff, 4, 1, f); fwrite("WAVEfmt ", 1, 8, f); fwrite(&fmt_len, 4, 1, f);
    fwrite(&pcm, 2, 1, f); fwrite(&ch, 2, 1, f); fwrite(&r, 4, 1, f); fwrite(&byte_rate, 4, 1, f);
    fwrite(&align, 2, 1, f); fwrite(&bits, 2, 1, f); fwrite("data", 1, 4, f); fwrite(&data_bytes, 4, 1, f);
    for (size_t i = 0; i < n; i++) {
        float v = x[i] < -1.0f ? -1.0f : (x[i] > 1.0f ? 1.0f : x[i]);
        const int16_t s = (int16_t) (v * 32767.0f);
        fwrite(&s, 2, 1, f);
    }
    fclose(f);
}

// the Harvard sentences of perf_battery.cpp:25-56 are data the benchmark protocol is defined on; the list
// (including the missing comma that fuses two of them, :39-40) is reproduced so that numbers are comparable
static const std::vector<std::string> SENTENCES = {
    "The birch canoe slid on the smooth planks.", "Glue the sheet to the dark blue background.",
    "It's easy to tell the depth of a well.", "These days a chicken leg is a rare dish.",
    "Rice is often served in round bowls.", "The juice of lemons makes fine punch.",
    "The box was thrown beside the parked truck.", "The hogs were fed chopped corn and garbage.",
    "Four hours of steady work faced us.", "A large size in stockings is hard to sell.",
    "The boy was there when the sun rose.", "A rod is used to catch pink salmon.",
    "The source of the huge river is the clear spring.",
    "Kick the ball straight and follow through." "Help the woman get back to her feet.",
    "A pot of tea helps to pass the evening.", "Smoky fires lack flame and heat.",
    "The soft cushion broke the man's fall.", "The salt breeze came across from the sea.",
    "The girl at the booth sold fifty bonds.", "The small pup gnawed a hole in the sock.",
    "The fish twisted and turned on the bent hook.", "Press the pants and sew a button on the vest.",
    "The swan dive was far short of perfect.", "The beauty of the view stunned the young boy.",
    "Two blue fish swam in the tank.", "Her purse was full of useless trash.",
    "The colt reared and threw the tall rider.", "It snowed, rained, and hailed the same morning.",
    "Read verse out loud for pleasure."};

static const char * arg(int argc, const char ** argv, const char * a, const char * b, const char * def) {
    for (int i = 1; i + 1 < argc; i++)
        if (!strcmp(argv[i], a) || (b && !strcmp(argv[i], b))) return argv[i + 1];
    return def;
}
static bool flag(int argc, const char ** argv, const char * a, const char * b) {
    for (int i = 1; i < argc; i++)
        if (!strcmp(argv[i], a) || (b && !strcmp(argv[i], b))) return true;
    return false;
}

int main(int argc, const char ** argv) {
    const bool perf = strstr(argv[0], "perf_battery") || flag(argc, argv, "--perf-battery", nullptr);
    const char * model = arg(argc, argv, "--model-path", "-mp", nullptr);
    if (!model || flag(argc, argv, "--help", "-h")) {
        fprintf(stderr, "usage: %s --model-path <gguf|test:dummy> [--prompt TEXT] [--save-path out.wav] [--topk N] [--temperature T]\n"
                        "          [--repetition-penalty R] [--top-p P] [--no-cross-attn] [--greedy] [--seed S] [--perf-battery]\n", argv[0]);
        return model ? 0 : 1;
    }
    generation_configuration config{"", atoi(arg(argc, argv, "--topk", "-tk", "50")), (float) atof(arg(argc, argv, "--temperature", "-t", "1.0")),
                                    (float) atof(arg(argc, argv, "--repetition-penalty", "-r", "1.0")), !flag(argc, argv, "--no-cross-attn", "-ca"),
                                    "", 0, (float) atof(arg(argc, argv, "--top-p", "-tp", "1.0")), !flag(argc, argv, "--greedy", nullptr)};
    config.seed = strtoull(arg(argc, argv, "--seed", nullptr, "0"), nullptr, 10);
    const auto t_load = std::chrono::steady_clock::now();
    std::unique_ptr<tts_generation_runner> runner = runner_from_file(model, 1, config, true);
    fprintf(stderr, "loaded %s (%s) in %.1f ms\n", model, runner->loader.get().arch,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_load).count());
    if (!perf) {
        const char * prompt = arg(argc, argv, "--prompt", "-p", nullptr);
        if (!prompt) { fprintf(stderr, "--prompt is required\n"); return 1; }
        tts_response data;
        const auto t0 = std::chrono::steady_clock::now();
        runner->generate(prompt, data, config);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (data.n_outputs == 0) { fprintf(stderr, "Got empty response for prompt, '%s'.\n", prompt); return 1; }
        const char * out = arg(argc, argv, "--save-path", "-sp", "TTS.cpp.wav");
        write_wav16(out, data.data, data.n_outputs, (int) runner->sampling_rate);
        printf("wrote %s: %zu samples (%.2f s) generated in %.1f ms = %.2fx real time\n", out, data.n_outputs,
               data.n_outputs / runner->sampling_rate, ms, data.n_outputs / runner->sampling_rate / (ms / 1e3));
        return 0;
    }
    // perf_battery protocol (perf_battery.cpp:103-116): mean generate ms and mean RTF = gen_ms / audio_ms
    double gen_sum = 0, rtf_sum = 0;
    size_t n = 0;
    for (const std::string & s : SENTENCES) {
        tts_response r;
        const auto t0 = std::chrono::steady_clock::now();
        runner->generate(s.c_str(), r, config);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (r.n_outputs == 0) continue;
        gen_sum += ms;
        rtf_sum += ms / (r.n_outputs / (runner->sampling_rate / 1000.0));
        n++;
    }
    printf("Mean Stats for arch %s:\n\n  Generation Time (ms):             %f\n  Generation Real Time Factor (ms): %f\n",
           runner->loader.get().arch, n ? gen_sum / n : 0.0, n ? rtf_sum / n : 0.0);
    return 0;
}
