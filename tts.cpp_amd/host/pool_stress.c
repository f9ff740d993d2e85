// pool_stress.c — device_pool under concurrent submitters, on the weightless backend ("test:dummy": no GPU, an utterance of n characters is n seconds of
// audio).  Four threads submit 24 requests each to a pool of three workers and wait for them while reading the statistics; once as lock-step batches,
// once as continuous-batching sessions.  Exit status 0 = every request got its own audio and the statistics were complete when the last wait returned.
// `make sanitize` builds the host library and this driver under ThreadSanitizer and AddressSanitizer and runs both (POOL_STRESS_TIMED=0 there for
// ThreadSanitizer: gcc 11's runtime does not intercept pthread_cond_clockwait, so every timed condition wait reads as a double lock).
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/tts_c.h"

static tts_c_pool *P;
static int fails = 0, timed = 1;

static void *submitter(void *arg) {
    long t = (long) arg;
    tts_c_config cfg;
    tts_c_default_config(&cfg);
    int ids[24];
    for (int i = 0; i < 24; i++) {
        char txt[32];
        int n = 1 + (int) ((t * 7 + i * 3) % 6);
        memset(txt, 'a' + (int) t, (size_t) n);
        txt[n] = 0;
        ids[i] = tts_c_pool_submit(P, txt, &cfg);
        if (ids[i] < 0) __sync_fetch_and_add(&fails, 1);
        if (i % 5 == 0) {
            uint64_t a, b, c, d;
            tts_c_pool_stats(P, &a, &b, &c, &d);
            (void) tts_c_pool_admitted_in_flight(P);
        }
    }
    for (int i = 0; i < 24; i++) {
        const float *data;
        size_t n;
        int bs, wk;
        const int rc = tts_c_pool_wait(P, ids[i], timed ? 20000 : -1, &data, &n, &bs, &wk);
        const int want = 1 + (int) ((t * 7 + i * 3) % 6);
        if (rc != 0 || n != (size_t) want * 44100) {
            __sync_fetch_and_add(&fails, 1);
            fprintf(stderr, "task %d: rc %d, %zu samples, expected %d\n", ids[i], rc, n, want * 44100);
        }
        tts_c_pool_release(P, ids[i]);
    }
    return NULL;
}

int main(void) {
    if (getenv("POOL_STRESS_TIMED")) timed = atoi(getenv("POOL_STRESS_TIMED"));
    for (int mode = 0; mode < 2; mode++) {
        tts_c_config cfg;
        tts_c_default_config(&cfg);
        tts_c_pool_set_continuous(mode);
        P = tts_c_pool_create("test:dummy", 3, NULL, 0, 4, timed ? 2 : 0, &cfg);
        if (!P) { fprintf(stderr, "pool create failed: %s\n", tts_c_last_error()); return 2; }
        pthread_t th[4];
        for (long t = 0; t < 4; t++) pthread_create(&th[t], NULL, submitter, (void *) t);
        for (int t = 0; t < 4; t++) pthread_join(th[t], NULL);
        uint64_t tasks, batches, largest, timed_out;
        tts_c_pool_stats(P, &tasks, &batches, &largest, &timed_out);   // right after the last wait: the counters must already hold every request
        const uint64_t joined = tts_c_pool_admitted_in_flight(P);
        printf("%s: tasks %lu batches %lu largest %lu timed_out %lu admitted_in_flight %lu failures %d\n", mode ? "continuous" : "lock-step",
               (unsigned long) tasks, (unsigned long) batches, (unsigned long) largest, (unsigned long) timed_out, (unsigned long) joined, fails);
        if (tasks != 96 || timed_out != 0 || largest > 4 || (mode == 0 && joined != 0) || (mode == 1 && joined == 0)) fails++;
        tts_c_pool_free(P);
    }
    return fails != 0;
}
