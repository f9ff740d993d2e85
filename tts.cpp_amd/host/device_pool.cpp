#include "device_pool.h"

#include "../../include/tts_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include <atomic>
extern std::atomic<bool> g_tts_throw_on_abort;

bool pool_configs_compatible(const generation_configuration & a, const generation_configuration & b) {
    return a.use_cross_attn == b.use_cross_attn && a.temperature == b.temperature && a.repetition_penalty == b.repetition_penalty &&
           a.top_p == b.top_p && a.top_k == b.top_k && a.max_tokens == b.max_tokens && a.voice == b.voice && a.sample == b.sample &&
           a.espeak_voice_id == b.espeak_voice_id && a.seed == b.seed;
}

struct device_pool::worker_state {
    std::map<std::string, std::unique_ptr<tts_generation_runner>> runners;  // worker::runners (:240)
};

device_pool::device_pool(const std::map<std::string, std::string> & model_paths, const generation_configuration & load_config,
                         const pool_options & opts)
    : model_paths_(model_paths), load_config_(load_config), opts_(opts) {
    if (opts_.n_workers < 1) opts_.n_workers = 1;
    if (opts_.max_batch < 1) opts_.max_batch = 1;
    if (opts_.devices.empty()) opts_.devices.push_back(0);
    per_worker_.resize((size_t) opts_.n_workers);
    for (int w = 0; w < opts_.n_workers; w++) states_.push_back(std::make_unique<worker_state>());
    load_all();
    if (!error_.empty()) { running_ = false; return; }
    for (int w = 0; w < opts_.n_workers; w++) threads_.emplace_back(&device_pool::worker_main, this, w);
}

// init_worker (server.cpp:309-321) for every worker, with the file parsed and uploaded ONCE per model where the runner can hand its
// weights on (device_context() != nullptr): the first worker of the first device loads, the first worker of every other device is
// laid out declare-only and receives the finished arena by RCCL (tts_hip_broadcast_weights, the path's one collective), further
// workers of a device share their device's arena.  Other runners (and the weightless test backend) load per worker as the reference.
// Placement goes through tts_thread_load_options(), not the environment.
void device_pool::load_all() {
    // a failed load must not abort() the whole server; the switch is the embedding application's again once the loads are done
    // (this runs on the constructing caller's thread)
    struct abort_guard {
        bool prev = g_tts_throw_on_abort;
        abort_guard() { g_tts_throw_on_abort = true; }
        ~abort_guard() { g_tts_throw_on_abort = prev; }
    } guard;
    const int nw = opts_.n_workers, nd = (int) opts_.devices.size();
    auto device_of = [&](int w) { return opts_.devices[(size_t) w % (size_t) nd]; };
    auto load = [&](int w, const std::string & path, bool declare, const tts_generation_runner * share) {
        tts_load_options & lo = tts_thread_load_options();
        lo = tts_load_options{};
        lo.device = device_of(w);
        lo.max_seqs = opts_.max_batch;
        lo.declare_only = declare;
        lo.share_with = share;
        std::unique_ptr<tts_generation_runner> r;
        try {
            r = runner_from_file(path.c_str(), opts_.n_threads, load_config_, false);
        } catch (...) {
            lo = tts_load_options{};
            throw;
        }
        lo = tts_load_options{};
        return r;
    };
    try {
        for (const auto & [id, path] : model_paths_) {
            // owners: the first worker on each distinct device
            std::map<int, int> owner;   // device -> worker
            for (int w = 0; w < nw; w++) owner.emplace(device_of(w), w);
            const int root_w = 0;   // worker 0 is the first worker of its device
            states_[(size_t) root_w]->runners[id] = load(root_w, path, false, nullptr);
            tts_generation_runner * root = states_[(size_t) root_w]->runners[id].get();
            const bool can_share = opts_.share_weights && root->device_context() != nullptr;
            std::vector<void *> ctxs{root->device_context()};
            for (const auto & [dev, w] : owner) {
                if (w == root_w) continue;
                states_[(size_t) w]->runners[id] = load(w, path, can_share, nullptr);
                if (can_share) ctxs.push_back(states_[(size_t) w]->runners[id]->device_context());
            }
            if (can_share && ctxs.size() > 1) {
                if (tts_hip_broadcast_weights((tts_hip_ctx **) ctxs.data(), (int) ctxs.size(), 0) != 0) {
                    // no usable RCCL on this host (librccl missing, ncclCommInitAll failed, peer access off): every other device parses and
                    // uploads the file itself, as the reference's server does per worker (server.cpp:316-321)
                    fprintf(stderr, "device_pool: weight broadcast failed (%s); loading the file once per device instead\n", tts_hip_last_error());
                    for (const auto & [dev, w] : owner) {
                        if (w == root_w) continue;
                        states_[(size_t) w]->runners.erase(id);
                        states_[(size_t) w]->runners[id] = load(w, path, false, nullptr);
                    }
                } else
                    broadcasts_++;
            }
            for (int w = 0; w < nw; w++) {
                if (states_[(size_t) w]->runners.count(id)) continue;
                const tts_generation_runner * own = states_[(size_t) owner.at(device_of(w))]->runners[id].get();
                states_[(size_t) w]->runners[id] = load(w, path, false, can_share ? own : nullptr);
                if (can_share) shared_loads_++;
            }
        }
    } catch (const std::exception & e) {
        error_ = std::string("load: ") + e.what();
    }
}

device_pool::~device_pool() {
    terminate();
    for (auto & t : threads_)
        if (t.joinable()) t.join();
}

void device_pool::terminate() {
    {
        std::lock_guard<std::mutex> lock(q_mutex_);
        running_ = false;
    }
    q_cv_.notify_all();
    {
        std::lock_guard<std::mutex> lock(r_mutex_);
    }
    r_cv_.notify_all();
}

int device_pool::submit(const std::string & model, const std::string & prompt, const generation_configuration & config) {
    auto t = std::make_shared<pool_task>();
    t->id = next_id_.fetch_add(1);
    t->model = model;
    t->prompt = prompt;
    t->gen_config = config;
    {
        std::lock_guard<std::mutex> lock(q_mutex_);
        if (!running_) return -1;
        queue_.push_back(t);
    }
    q_cv_.notify_one();
    return t->id;
}

int device_pool::submit_conditional_prompt(const std::string & model, const std::string & prompt) {
    auto parent = std::make_shared<pool_task>();
    parent->task = POOL_CONDITIONAL_PROMPT;
    parent->id = next_id_.fetch_add(1);
    parent->model = model;
    parent->prompt = prompt;
    {
        std::lock_guard<std::mutex> lock(q_mutex_);
        if (!running_) return -1;
        fanouts_[parent->id] = fanout{parent, opts_.n_workers, true, ""};
        for (int w = 0; w < opts_.n_workers; w++) {
            auto t = std::make_shared<pool_task>(*parent);   // same id: the copies report into the fan-out record
            per_worker_[(size_t) w].push_back(t);
        }
    }
    q_cv_.notify_all();
    return parent->id;
}

int device_pool::submit_voices() {
    auto t = std::make_shared<pool_task>();
    t->task = POOL_VOICES;
    t->id = next_id_.fetch_add(1);
    {
        std::lock_guard<std::mutex> lock(q_mutex_);
        if (!running_) return -1;
        queue_.push_back(t);
    }
    q_cv_.notify_one();
    return t->id;
}

std::shared_ptr<pool_task> device_pool::wait(int id, int timeout_ms) {
    std::unique_lock<std::mutex> lock(r_mutex_);
    auto ready = [&] { return completed_.count(id) != 0 || !running_; };
    if (timeout_ms < 0) r_cv_.wait(lock, ready);
    else if (!r_cv_.wait_for(lock, std::chrono::milliseconds(timeout_ms), ready)) return nullptr;
    auto it = completed_.find(id);
    return it == completed_.end() ? nullptr : it->second;
}

void device_pool::release(int id) {
    std::lock_guard<std::mutex> lock(r_mutex_);
    completed_.erase(id);
}

pool_stats device_pool::stats() const {
    std::lock_guard<std::mutex> lock(s_mutex_);
    return stats_;
}

// simple_task_queue::get_next (:132-145) widened to a batch: the oldest task plus every queued task compatible
// with it (queue order preserved among the others), up to `cap`
std::vector<std::shared_ptr<pool_task>> device_pool::next_batch(int w, int cap) {
    std::vector<std::shared_ptr<pool_task>> batch;
    std::unique_lock<std::mutex> lock(q_mutex_);
    auto & mine = per_worker_[(size_t) w];
    idle_workers_++;
    q_cv_.wait(lock, [&] { return !queue_.empty() || !mine.empty() || !running_; });
    idle_workers_--;
    if (!running_) return batch;
    if (!mine.empty()) {  // control tasks addressed to this worker go first, one at a time
        batch.push_back(mine.front());
        mine.pop_front();
        return batch;
    }
    batch.push_back(queue_.front());
    queue_.pop_front();
    if (batch[0]->task != POOL_TTS) return batch;
    auto collect = [&] {
        for (auto it = queue_.begin(); it != queue_.end() && (int) batch.size() < cap;) {
            if ((*it)->task == POOL_TTS && (*it)->model == batch[0]->model && pool_configs_compatible((*it)->gen_config, batch[0]->gen_config)) {
                batch.push_back(*it);
                it = queue_.erase(it);
            } else ++it;
        }
    };
    collect();
    if (opts_.batch_window_ms > 0 && (int) batch.size() < cap) {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(opts_.batch_window_ms);
        while (running_ && (int) batch.size() < cap) {
            if (q_cv_.wait_until(lock, deadline) == std::cv_status::timeout) { collect(); break; }
            collect();
        }
    }
    return batch;
}

void device_pool::worker_main(int w) {
    worker_state & ws = *states_[(size_t) w];
    g_tts_throw_on_abort = true;
    while (true) {
        std::vector<std::shared_ptr<pool_task>> batch = next_batch(w, opts_.max_batch);
        if (batch.empty()) break;
        if (batch[0]->task != POOL_TTS) {
            control(w, *batch[0], ws);
            if (batch[0]->task == POOL_CONDITIONAL_PROMPT) {
                std::shared_ptr<pool_task> done;
                {
                    std::lock_guard<std::mutex> lock(q_mutex_);
                    auto it = fanouts_.find(batch[0]->id);
                    if (it != fanouts_.end()) {
                        it->second.ok = it->second.ok && batch[0]->success;
                        if (!batch[0]->success && it->second.message.empty()) it->second.message = batch[0]->message;
                        if (--it->second.remaining == 0) {
                            done = it->second.parent;
                            done->success = it->second.ok;
                            done->message = it->second.message;
                            fanouts_.erase(it);
                        }
                    }
                }
                if (done) {
                    { std::lock_guard<std::mutex> lock(r_mutex_); completed_[done->id] = done; }
                    r_cv_.notify_all();
                }
            } else {
                { std::lock_guard<std::mutex> lock(r_mutex_); completed_[batch[0]->id] = batch[0]; }
                r_cv_.notify_all();
            }
            continue;
        }
        if (opts_.continuous) process_stream(w, batch, ws);
        else process(w, batch, ws);
    }
}

// queued TTS tasks compatible with `like`, up to cap, without waiting.  None while a control task addressed to this worker waits: the session
// drains first (a CONDITIONAL_PROMPT changes the voice of everything generated after it).
std::vector<std::shared_ptr<pool_task>> device_pool::poll_compatible(int w, const pool_task & like, size_t cap) {
    std::vector<std::shared_ptr<pool_task>> got;
    std::lock_guard<std::mutex> lock(q_mutex_);
    if (!running_ || !per_worker_[(size_t) w].empty()) return got;
    // A request the session cannot take (another model, incompatible sampling parameters) that has waited longer than continuous_yield_ms at the
    // head of the queue ends the admissions: the session drains and next_batch() serves it.  Without the bound a steady stream of compatible
    // requests kept one session alive for ever and starved everything else (task_timeout_s is only looked at on admission).  Only when no worker is
    // idle: an idle worker takes the stale request through next_batch() itself, and with several workers on mixed models every session draining
    // at once for one request cost throughput.
    for (const auto & t : queue_) {
        if (t->task == POOL_TTS && t->model == like.model && pool_configs_compatible(t->gen_config, like.gen_config)) continue;
        if (idle_workers_ == 0 && t->waited_s() * 1e3 > opts_.continuous_yield_ms) return got;
        break;   // the oldest incompatible request is still young: requests behind it are younger
    }
    for (auto it = queue_.begin(); it != queue_.end() && got.size() < cap;) {
        if ((*it)->task == POOL_TTS && (*it)->model == like.model && pool_configs_compatible((*it)->gen_config, like.gen_config)) {
            got.push_back(*it);
            it = queue_.erase(it);
        } else ++it;
    }
    return got;
}

// One generation session for a run of compatible requests: the first batch opens it, later arrivals enter rows that have freed up
// (server.cpp:126-158 is the queue; :236-271 the one-task-at-a-time worker this replaces).
void device_pool::process_stream(int w, std::vector<std::shared_ptr<pool_task>> & first, worker_state & ws) {
    auto found = ws.runners.find(first[0]->model);
    if (found == ws.runners.end() || !found->second || found->second->stream_capacity() == 0) { process(w, first, ws); return; }
    tts_generation_runner & runner = *found->second;
    std::map<size_t, std::shared_ptr<pool_task>> inflight;
    std::vector<std::shared_ptr<pool_task>> done, overflow;
    size_t ticket = 0;
    uint64_t expired = 0, served = 0, joined = 0, peak = 0;
    const pool_task like = *first[0];
    // the counters go into the pool's statistics BEFORE the tasks they count are published (as process() does): whoever has waited for the last
    // request of a run reads complete statistics
    uint64_t booked_served = 0, booked_expired = 0, booked_joined = 0;
    bool session_booked = false;
    auto book = [&] {
        std::lock_guard<std::mutex> lock(s_mutex_);
        stats_.tasks += (served - booked_served) + (expired - booked_expired);
        stats_.timed_out += expired - booked_expired;
        stats_.admitted_in_flight += joined - booked_joined;
        stats_.largest_batch = std::max<uint64_t>(stats_.largest_batch, peak);
        if (!session_booked) { stats_.batches += 1; session_booked = true; }
        booked_served = served; booked_expired = expired; booked_joined = joined;
    };
    auto finish = [&](std::vector<std::shared_ptr<pool_task>> & v) {
        if (v.empty()) return;
        book();
        { std::lock_guard<std::mutex> lock(r_mutex_); for (auto & t : v) completed_[t->id] = t; }
        r_cv_.notify_all();
        v.clear();
    };
    // the batch admit() is working on and how far it got: if stream_begin() or anything inside admit() throws (a cross-attention mismatch, a sampler
    // the device loop does not carry, a failed device allocation — g_tts_throw_on_abort turns every abort into an exception in a worker), the tasks
    // that are in none of inflight / overflow / done yet are failed with the message too.  Round 4 lost them: wait(id) never returned.
    std::vector<std::shared_ptr<pool_task>> admitting = first;
    size_t admitted = 0;
    try {
        runner.stream_begin(like.gen_config);
        auto admit = [&](std::vector<std::shared_ptr<pool_task>> & tasks, bool in_flight) {
            admitting = tasks;
            admitted = 0;
            for (auto & t : tasks) {
                admitted++;
                t->worker = w;
                if (t->timed_out(opts_.task_timeout_s)) { t->message = "timed out in the queue"; expired++; done.push_back(t); continue; }
                if (runner.stream_free() == 0) { overflow.push_back(t); continue; }
                try {
                    runner.stream_submit(ticket, t->prompt);   // host work only (tokenising, taking a row): a request it rejects fails alone
                } catch (const std::exception & e) {
                    t->success = false; t->message = e.what(); served++; done.push_back(t);
                    continue;
                }
                inflight[ticket++] = t;
                served++;
                joined += in_flight;
            }
            const uint32_t rows = runner.stream_capacity() - runner.stream_free();   // utterances sharing the forward from here on
            peak = std::max<uint64_t>(peak, rows);
            for (auto & t : tasks) if (t->batch_size == 0) t->batch_size = (int) rows;
            admitting.clear();
            admitted = 0;
        };
        admit(first, false);
        std::vector<tts_generation_runner::stream_result> fin;
        while (runner.stream_live() > 0) {
            runner.stream_step(fin);
            for (auto & f : fin) {
                auto it = inflight.find(f.ticket);
                if (it == inflight.end()) continue;
                pool_task & t = *it->second;
                t.audio.assign(f.audio.data, f.audio.data + f.audio.n_outputs);   // the runner reuses its buffer on the next step
                t.sample_rate = runner.sampling_rate;
                t.success = f.audio.n_outputs != 0;
                done.push_back(it->second);
                inflight.erase(it);
            }
            finish(done);
            if (!overflow.empty()) {
                std::vector<std::shared_ptr<pool_task>> again;
                again.swap(overflow);
                for (auto & t : again) t->batch_size = 0;
                admit(again, true);
            }
            if (runner.stream_free() > 0) {
                auto more = poll_compatible(w, like, runner.stream_free());
                if (!more.empty()) admit(more, true);
            }
        }
        runner.stream_end();
    } catch (const std::exception & e) {
        for (auto & kv : inflight) { kv.second->success = false; kv.second->message = e.what(); done.push_back(kv.second); }
        for (auto & t : overflow) { t->success = false; t->message = e.what(); done.push_back(t); }
        // the task admit() was working on when the exception left it is in none of the three lists either: it fails with the rest
        for (size_t i = admitted ? admitted - 1 : 0; i < admitting.size(); i++) {
            auto & t = admitting[i];
            bool placed = false;
            for (auto & d : done) placed |= d == t;
            if (placed) continue;
            t->worker = w; t->success = false; t->message = e.what(); served++; done.push_back(t);
        }
        inflight.clear(); overflow.clear(); admitting.clear();
        try { runner.stream_end(); } catch (...) {}
    }
    finish(done);
    book();
}

// worker::process_task, the non-TTS cases (server.cpp:263-306)
void device_pool::control(int w, pool_task & t, worker_state & ws) {
    t.worker = w;
    try {
        if (t.task == POOL_CONDITIONAL_PROMPT) {
            if (opts_.text_encoder_path.empty()) {
                t.message = "A text encoder path must be specified on server initialization in order to support conditional prompting.";  // :264-266
                return;
            }
            auto found = ws.runners.find(t.model);
            if (found == ws.runners.end() || !found->second) { t.message = "unknown model '" + t.model + "'"; return; }
            found->second->update_conditional_prompt(opts_.text_encoder_path.c_str(), t.prompt.c_str());
            t.success = true;
        } else {  // VOICES
            for (const auto & [id, runner] : ws.runners) {
                if (!runner || !runner->supports_voices) continue;
                std::string voices;
                for (const auto v : runner->list_voices()) {
                    if (!voices.empty()) voices += ",";
                    voices += v;
                }
                if (!t.message.empty()) t.message += ";";
                t.message += id + "/" + voices;
            }
            t.success = true;
        }
    } catch (const std::exception & e) {
        t.success = false;
        t.message = e.what();
    }
}

void device_pool::process(int w, std::vector<std::shared_ptr<pool_task>> & batch, worker_state & ws) {
    std::vector<std::shared_ptr<pool_task>> live;
    uint64_t expired = 0;
    for (auto & t : batch) {
        t->worker = w;
        if (t->timed_out(opts_.task_timeout_s)) {  // worker::process_task :247-249 drops it silently; here it is answered
            t->message = "timed out in the queue";
            expired++;
        } else live.push_back(t);
    }
    if (!live.empty()) {
        auto found = ws.runners.find(live[0]->model);
        if (found == ws.runners.end() || !found->second) {
            for (auto & t : live) t->message = "unknown model '" + t->model + "'";
        } else {
            tts_generation_runner & runner = *found->second;
            const size_t cap = std::max<size_t>(1, std::min<size_t>(live.size(), runner.batch_capacity()));
            for (size_t off = 0; off < live.size(); off += cap) {
                const size_t n = std::min(cap, live.size() - off);
                try {
                    std::vector<std::string> prompts;
                    for (size_t i = 0; i < n; i++) prompts.push_back(live[off + i]->prompt);
                    std::vector<tts_response> out;
                    if (n == 1) {
                        out.resize(1);
                        runner.generate(prompts[0].c_str(), out[0], live[off]->gen_config);
                    } else {
                        runner.generate_batch(prompts, out, live[off]->gen_config);
                    }
                    for (size_t i = 0; i < n; i++) {
                        pool_task & t = *live[off + i];
                        t.audio.assign(out[i].data, out[i].data + out[i].n_outputs);  // the runner reuses its buffer on the next call
                        t.sample_rate = runner.sampling_rate;
                        t.success = out[i].n_outputs != 0;                            // :258
                        t.batch_size = (int) n;
                    }
                } catch (const std::exception & e) {
                    for (size_t i = 0; i < n; i++) { live[off + i]->success = false; live[off + i]->message = e.what(); }
                }
            }
        }
    }
    {
        std::lock_guard<std::mutex> lock(s_mutex_);
        stats_.tasks += batch.size();
        stats_.timed_out += expired;
        if (!live.empty()) {
            stats_.batches += 1;
            stats_.largest_batch = std::max<uint64_t>(stats_.largest_batch, live.size());
        }
    }
    {
        std::lock_guard<std::mutex> lock(r_mutex_);
        for (auto & t : batch) completed_[t->id] = t;
    }
    r_cv_.notify_all();
}
