/* In-process native stack sampler for the host thread (diagnostics; ptrace is not permitted on the GPU boxes).
   sampler_start(period_us, max_samples) is called FROM the thread to sample, after every library is loaded; a helper thread signals
   it every period and the handler stores raw return addresses only.  sampler_dump(path) symbolises them afterwards (dladdr). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define DEPTH 40
typedef struct { double t; int n; void* pc[DEPTH]; } sample_t;
static sample_t* g_buf = NULL;
static volatile int g_count = 0;
static int g_max = 0;
static pthread_t g_target, g_helper;
static volatile int g_running = 0;
static int g_period_us = 1000;

static void handler(int sig) {
    (void)sig;
    int i = g_count;
    if (i >= g_max) return;
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    g_buf[i].t = ts.tv_sec + ts.tv_nsec * 1e-9;
    g_buf[i].n = backtrace(g_buf[i].pc, DEPTH);
    g_count = i + 1;
}

static void* loop(void* arg) {
    (void)arg;
    while (g_running) {
        usleep(g_period_us);
        if (g_running) pthread_kill(g_target, SIGUSR2);
    }
    return NULL;
}

int sampler_start(int period_us, int max_samples) {
    void* warm[4];
    backtrace(warm, 4);                      /* loads libgcc outside the handler */
    g_buf = (sample_t*)calloc((size_t)max_samples, sizeof(sample_t));
    if (!g_buf) return -1;
    g_max = max_samples; g_count = 0;
    g_period_us = period_us;
    g_target = pthread_self();
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGUSR2, &sa, NULL);
    g_running = 1;
    return pthread_create(&g_helper, NULL, loop, NULL);
}

int sampler_stop(void) {
    if (!g_running) return g_count;
    g_running = 0;
    pthread_join(g_helper, NULL);
    signal(SIGUSR2, SIG_IGN);
    return g_count;
}

int sampler_dump(const char* path) {
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    for (int i = 0; i < g_count; ++i) {
        fprintf(f, "--- %.6f\n", g_buf[i].t);
        for (int k = 2; k < g_buf[i].n; ++k) {          /* skip the handler and the signal trampoline */
            Dl_info d;
            if (dladdr(g_buf[i].pc[k], &d) && d.dli_fname) {
                const char* b = strrchr(d.dli_fname, '/');
                fprintf(f, "%s!%s\n", b ? b + 1 : d.dli_fname, d.dli_sname ? d.dli_sname : "?");
            } else {
                fprintf(f, "?!%p\n", g_buf[i].pc[k]);
            }
        }
    }
    fclose(f);
    return g_count;
}
