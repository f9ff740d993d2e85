// device_pool.h — the reference server's worker pool re-shaped for GPUs: one worker per device, dynamic
// lock-step batching of the queued requests.
//
// Mirrors examples/server/server.cpp (/root/reference): simple_server_task :100-123, simple_task_queue :126-158,
// simple_response_map :160-219, worker :225-307, init_worker :309-314, the pool start-up/terminate :316-330,885-895.
// Same flow — callers push tasks on one queue, workers pull, finished tasks land in a response map keyed by task
// id and a caller blocks on its id — with two changes that the hardware asks for:
//   * a worker owns a DEVICE (worker w -> devices[w mod G]) instead of a set of CPU threads, and
//   * a worker drains up to `max_batch` compatible tasks (same model, same generation_configuration) that are
//     already queued — optionally waiting `batch_window_ms` for more — and decodes them in lock-step with
//     tts_generation_runner::generate_batch (one pass over the weights per step for all of them).  The reference
//     processes one task per worker at a time (:246-262).
// No HTTP here: the transport (cpp-httplib in the reference) stays with the application.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

enum pool_task_type { POOL_TTS = 0, POOL_CONDITIONAL_PROMPT = 1, POOL_VOICES = 2 };  // task_type (server.cpp:94-98)

struct pool_task {  // simple_server_task (:100-123); the audio is copied out of the runner-owned buffer
    pool_task_type           task = POOL_TTS;
    int                      id = 0;
    std::string              model;
    std::string              prompt;
    generation_configuration gen_config;
    std::vector<float>       audio;
    float                    sample_rate = 44100.0f;
    bool                     success = false;
    std::string              message;
    std::chrono::steady_clock::time_point time = std::chrono::steady_clock::now();
    int                      batch_size = 0;  // how many tasks were decoded together with this one
    int                      worker = -1;
    double waited_s() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - time).count(); }
    bool timed_out(int seconds) const { return waited_s() > seconds; }
};

struct pool_options {
    int              n_workers = 1;        // --n-parallelism (:885): here one per device/context
    std::vector<int> devices;              // worker w runs on devices[w % size]; empty = {0}
    int              max_batch = 1;        // lock-step utterances per worker (runners are loaded with this many KV slots)
    int              batch_window_ms = 0;  // after the first task, wait this long for more before decoding
    int              n_threads = 1;
    int              task_timeout_s = 300; // :231, tasks older than this are answered with success == false
    std::string      text_encoder_path;    // T5 GGUF for CONDITIONAL_PROMPT tasks (server --text-encoder-path, :263-271)
    bool             share_weights = true; // parse + upload each model once: RCCL broadcast across devices, one arena per device
    // continuous batching: a worker keeps ONE generation session per run of compatible requests and admits queued requests into rows that
    // free up while the others are still generating (tts_generation_runner::stream_*; every 32 decode steps), instead of forming a batch from
    // what is queued and running it to the end — a request that arrives one step late no longer waits a whole generation, and a ragged
    // batch refills instead of idling.  Runners without a session (stream_capacity() == 0) keep the batch path.
    bool             continuous = false;
    int              continuous_yield_ms = 2000;   // a session stops admitting once a request it cannot take has waited this long at the head of the queue
};

struct pool_stats {
    uint64_t tasks = 0, batches = 0, largest_batch = 0, timed_out = 0;
    uint64_t admitted_in_flight = 0;   // continuous batching: requests that entered a session while other utterances were already generating
};

// generation parameters that must agree for two tasks to share a lock-step batch
bool pool_configs_compatible(const generation_configuration & a, const generation_configuration & b);

class device_pool {
  public:
    // model_paths: id -> GGUF path (or "test:<arch>"), every worker loads every model (init_worker :309-314).
    // Blocks until all workers have loaded (or one of them failed: then ok() is false and error() says why).
    device_pool(const std::map<std::string, std::string> & model_paths, const generation_configuration & load_config,
                const pool_options & opts);
    ~device_pool();
    device_pool(const device_pool &) = delete;
    device_pool & operator=(const device_pool &) = delete;

    bool                ok() const { return error_.empty(); }
    const std::string & error() const { return error_; }

    // enqueue one TTS request; returns its id (the reference uses rand(), :102; ids here are sequential)
    int submit(const std::string & model, const std::string & prompt, const generation_configuration & config);
    // CONDITIONAL_PROMPT (:263-271): every worker re-encodes its runner's voice prompt (the reference hands the task to ONE
    // worker, so only that worker's replica changes voice; here the task is fanned out and the returned id completes when
    // the last worker has applied it).  VOICES (:272-306): "model/voice,voice;model/..." in pool_task::message.
    int submit_conditional_prompt(const std::string & model, const std::string & prompt);
    int submit_voices();
    // block until task `id` is finished (simple_response_map::get :203-218); nullptr after terminate()/timeout
    std::shared_ptr<pool_task> wait(int id, int timeout_ms = -1);
    void       release(int id);  // drop a finished task from the response map (the reference's cleanup thread, :168-189)
    void       terminate();      // :316-330
    pool_stats stats() const;
    int        weight_broadcasts() const { return broadcasts_; }   // RCCL broadcasts performed at load (one per model with > 1 device)
    int        shared_arena_loads() const { return shared_loads_; } // workers that reuse their device's arena instead of uploading

  private:
    struct worker_state;
    void load_all();
    void worker_main(int w);
    void process(int w, std::vector<std::shared_ptr<pool_task>> & batch, worker_state & ws);
    void process_stream(int w, std::vector<std::shared_ptr<pool_task>> & first, worker_state & ws);
    std::vector<std::shared_ptr<pool_task>> poll_compatible(int w, const pool_task & like, size_t cap);
    std::vector<std::shared_ptr<pool_task>> next_batch(int w, int cap);
    void control(int w, pool_task & t, worker_state & ws);
    struct fanout { std::shared_ptr<pool_task> parent; int remaining = 0; bool ok = true; std::string message; };
    std::map<int, fanout>                  fanouts_;     // parent id -> state (guarded by q_mutex_)
    std::vector<std::deque<std::shared_ptr<pool_task>>> per_worker_;  // control tasks addressed to one worker

    std::map<std::string, std::string> model_paths_;
    generation_configuration           load_config_;
    pool_options                       opts_;
    std::string                        error_;

    std::mutex                              q_mutex_;
    std::condition_variable                 q_cv_;
    std::deque<std::shared_ptr<pool_task>>  queue_;
    int                                     idle_workers_ = 0;   // workers waiting in next_batch() (under q_mutex_): one of them serves a stale incompatible request, no session has to yield for it
    std::atomic<bool>                       running_{true};   // read under either mutex (wait() holds r_mutex_, the queue side q_mutex_)

    mutable std::mutex                          r_mutex_;
    std::condition_variable                     r_cv_;
    std::map<int, std::shared_ptr<pool_task>>   completed_;

    std::vector<std::unique_ptr<worker_state>> states_;   // loaded by the constructor, then owned by worker w's thread
    int                     broadcasts_ = 0, shared_loads_ = 0;

    std::atomic<int>         next_id_{1};
    mutable std::mutex       s_mutex_;
    pool_stats               stats_;
    std::vector<std::thread> threads_;
};
