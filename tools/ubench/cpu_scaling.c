/* What the GPU box's "16 cores" are worth when all of them work: N pinned threads each run a
 * fixed amount of (a) dependent floating-point arithmetic (latency-bound, what a hardware thread
 * that shares its core with a busy sibling loses least on), (b) independent integer/FP work at
 * high IPC (what it loses most on), (c) random reads over a 12 MB table (one build's working
 * set).  Prints the time per thread relative to one thread alone.
 *   gcc -O2 -pthread -o tools/ubench/cpu_scaling tools/ubench/cpu_scaling.c && tools/ubench/cpu_scaling */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

static int g_kind;
static uint32_t *g_table;
#define TABLE (3u << 20) /* 12 MB of uint32 */
static pthread_barrier_t g_bar;
static double g_ms[64];
static volatile double g_sink;

static void *work(void *arg) {
    const int t = (int)(intptr_t)arg;
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(t, &set);
    pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    pthread_barrier_wait(&g_bar);
    const double t0 = now();
    double acc = 1.0;
    if (g_kind == 0) {
        for (long i = 0; i < 60000000L; ++i) acc = acc * 1.0000001 + 1e-9;
    } else if (g_kind == 1) {
        double a = 1, b = 2, c = 3, d = 4, e = 5, f = 6, g = 7, h = 8;
        uint64_t x = 1, y = 2, z = 3, w = 4;
        for (long i = 0; i < 40000000L; ++i) {
            a = a * 1.0000001 + 1e-9; b = b * 1.0000001 + 1e-9;
            c = c * 1.0000001 + 1e-9; d = d * 1.0000001 + 1e-9;
            e = e * 1.0000001 + 1e-9; f = f * 1.0000001 + 1e-9;
            g = g * 1.0000001 + 1e-9; h = h * 1.0000001 + 1e-9;
            x = x * 6364136223846793005ull + 1; y ^= x >> 7; z += y; w ^= z << 3;
        }
        acc = a + b + c + d + e + f + g + h + (double)(x ^ y ^ z ^ w);
    } else {
        uint32_t at = (uint32_t)t * 7919u;
        for (long i = 0; i < 20000000L; ++i) at = g_table[at % TABLE] + (uint32_t)i;
        acc = at;
    }
    g_ms[t] = now() - t0;
    g_sink = acc;
    return NULL;
}

int main(void) {
    g_table = malloc(sizeof(uint32_t) * TABLE);
    uint32_t s = 12345;
    for (uint32_t i = 0; i < TABLE; ++i) { s = s * 1664525u + 1013904223u; g_table[i] = s; }
    static const char *name[3] = {"dependent fp chain", "independent fp + int (high IPC)", "random reads, 12 MB"};
    static const int counts[] = {1, 2, 4, 8, 16};
    for (g_kind = 0; g_kind < 3; ++g_kind) {
        double base = 0;
        for (int ci = 0; ci < 5; ++ci) {
            const int n = counts[ci];
            pthread_t th[64];
            pthread_barrier_init(&g_bar, NULL, (unsigned)n);
            for (int t = 0; t < n; ++t) pthread_create(&th[t], NULL, work, (void *)(intptr_t)t);
            double sum = 0, worst = 0;
            for (int t = 0; t < n; ++t) { pthread_join(th[t], NULL); sum += g_ms[t]; if (g_ms[t] > worst) worst = g_ms[t]; }
            pthread_barrier_destroy(&g_bar);
            if (n == 1) base = sum;
            printf("%-34s %2d threads: %7.1f ms per thread (slowest %7.1f) = %.2f x one alone\n",
                   name[g_kind], n, sum / n, worst, sum / n / base);
        }
    }
    return 0;
}
