/*
 * oracle/cpu_bench.c — TEST INFRASTRUCTURE (see oracle/README.md).  Not part of the product.
 *
 * Times the reference's UNMODIFIED BitMnistInference (BitNetMCU_MNIST_dll.c:48-121), loaded with
 * dlopen from oracle/_ref/<model>/Bitnet_inf_O3.dll, on T host threads over synthetic images that
 * are already resident in memory (same generator and seeds as the GPU bench, oracle/synth.h).
 * This is bench.py's "cpu_baseline" (kind "reference").  Baseline only — not a target.
 *
 *   cpu_bench <dll> <threads> <seconds> <dist 0|1> [images_per_thread]
 * prints one JSON line.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "synth.h"

typedef uint32_t (*infer_fn)(int8_t *);

typedef struct {
    infer_fn fn;
    int8_t *images;
    uint64_t n_images;
    double seconds;
    uint64_t done;
    uint64_t class_sum;
    double elapsed;
} job_t;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static pthread_barrier_t start_line;

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    pthread_barrier_wait(&start_line);
    double t0 = now_s(), t1 = t0;
    uint64_t done = 0, sum = 0;
    do {
        for (uint64_t i = 0; i < j->n_images; i++) sum += j->fn(j->images + 256 * i);
        done += j->n_images;
        t1 = now_s();
    } while (t1 - t0 < j->seconds);
    j->done = done;
    j->class_sum = sum;
    j->elapsed = t1 - t0;
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <dll> <threads> <seconds> <dist> [images_per_thread]\n", argv[0]);
        return 2;
    }
    const char *dll = argv[1];
    int threads = atoi(argv[2]);
    double seconds = atof(argv[3]);
    int dist = atoi(argv[4]);
    uint64_t per = argc > 5 ? strtoull(argv[5], 0, 10) : 8192;
    void *h = dlopen(dll, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    infer_fn fn = (infer_fn)dlsym(h, "BitMnistInference");
    if (!fn) { fprintf(stderr, "no BitMnistInference in %s\n", dll); return 3; }

    job_t *jobs = calloc(threads, sizeof(job_t));
    pthread_t *tid = calloc(threads, sizeof(pthread_t));
    uint64_t seed = dist ? BNM_SEED_DIST_M : BNM_SEED_DIST_U;
    pthread_barrier_init(&start_line, 0, threads);
    for (int t = 0; t < threads; t++) {
        jobs[t].fn = fn;
        jobs[t].n_images = per;
        jobs[t].seconds = seconds;
        jobs[t].images = malloc(per * 256);
        orc_synth_images(seed, dist, (uint64_t)t * per, per, jobs[t].images);
    }
    for (int t = 0; t < threads; t++) pthread_create(&tid[t], 0, worker, &jobs[t]);
    uint64_t total = 0, sum = 0;
    double wall = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(tid[t], 0);
        total += jobs[t].done;
        sum += jobs[t].class_sum;
        if (jobs[t].elapsed > wall) wall = jobs[t].elapsed;
    }
    /* class_sum of the first pass over thread 0's images lets the caller cross-check ids */
    printf("{\"inferences\": %llu, \"seconds\": %.6f, \"threads\": %d, \"inf_per_s\": %.1f, "
           "\"images_per_thread\": %llu, \"dist\": %d, \"class_sum\": %llu}\n",
           (unsigned long long)total, wall, threads, total / wall, (unsigned long long)per, dist,
           (unsigned long long)sum);
    return 0;
}
