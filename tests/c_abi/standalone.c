/* standalone.c -- a consumer of include/tetris_piclim.h that is not Python: plain C99, no HIP headers, linked
 * against libtetris_piclim.so and the HIP runtime only for device allocations and copies.  tests/test_c_abi.py
 * compiles it with gcc, runs it on the GPU box and compares what it prints with the CPU oracle fed the same
 * synthetic workload. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tetris_piclim.h"

/* the three HIP runtime entry points this program needs (hipError_t is an int, hipMemcpyKind an enum) */
extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t bytes, int kind);
extern int hipDeviceSynchronize(void);
enum { kHostToDevice = 1, kDeviceToHost = 2 };

#define CHECK(call)                                                                            \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != 0) {                                                                        \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, tpl_last_error());                   \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
    const int32_t L = 5, M = 20;
    const int64_t pool = 512;
    const int steps = argc > 2 ? atoi(argv[2]) : 60;
    const uint64_t seed = 7;

    tpl_env* env = NULL;
    CHECK(tpl_create(&env, n, L, M, 0, 0, seed, NULL, 0));                 /* the library allocates its workspace */
    CHECK(tpl_set_options(env, 1, TPL_ASSIGN_HASH, 1.0f, 5.0f, -1.0f));

    uint16_t* rows = NULL; uint8_t* pieces = NULL; uint8_t* action = NULL; float* reward = NULL; uint8_t* done = NULL;
    uint64_t* stats = NULL;
    CHECK(hipMalloc((void**)&rows, (size_t)pool * TPL_ROWS * sizeof(uint16_t)));
    CHECK(hipMalloc((void**)&pieces, (size_t)pool * (M + 1)));
    CHECK(hipMalloc((void**)&action, (size_t)n));
    CHECK(hipMalloc((void**)&reward, (size_t)n * sizeof(float)));
    CHECK(hipMalloc((void**)&done, (size_t)n));
    CHECK(hipMalloc((void**)&stats, 4 * sizeof(uint64_t)));

    CHECK(tpl_synth_configs(env, seed, 0, pool, rows, pieces, NULL));
    CHECK(tpl_load_configs(env, rows, pieces, pool, NULL, 0, NULL));       /* and its pool */
    CHECK(tpl_reset(env, NULL, NULL));
    double reward_sum = 0.0;
    long done_count = 0;
    float* h_reward = (float*)malloc((size_t)n * sizeof(float));
    uint8_t* h_done = (uint8_t*)malloc((size_t)n);
    for (int t = 0; t < steps; ++t) {
        CHECK(tpl_synth_actions(env, seed, 0, n, (uint64_t)t, action, NULL));
        CHECK(tpl_step(env, action, TPL_U8, reward, done, NULL));
        CHECK(hipMemcpy(h_reward, reward, (size_t)n * sizeof(float), kDeviceToHost));
        CHECK(hipMemcpy(h_done, done, (size_t)n, kDeviceToHost));
        for (int64_t i = 0; i < n; ++i) { reward_sum += h_reward[i]; done_count += h_done[i]; }
    }
    CHECK(tpl_get_stats(env, stats, NULL));
    uint64_t h_stats[4];
    CHECK(hipMemcpy(h_stats, stats, sizeof(h_stats), kDeviceToHost));
    CHECK(hipDeviceSynchronize());
    printf("version %s\n", tpl_version());
    printf("stats %llu %llu %llu %llu\n", (unsigned long long)h_stats[0], (unsigned long long)h_stats[1],
           (unsigned long long)h_stats[2], (unsigned long long)h_stats[3]);
    printf("reward_sum %.1f done_count %ld\n", reward_sum, done_count);

    /* errors come back as status codes with a message, never as a crash */
    if (tpl_step(env, NULL, TPL_U8, reward, done, NULL) != TPL_ERR_ARG || strlen(tpl_last_error()) == 0) {
        fprintf(stderr, "null action was not refused\n");
        return 1;
    }
    /* the supply side from plain C: both generators on the device against their host forms (same seeds, same outputs) */
    {
        const int64_t count = 256;
        const int32_t gL = 5, gM = 20;
        uint16_t *d_rows = NULL, *h_rows = malloc((size_t)count * TPL_ROWS * 2), *g_rows = malloc((size_t)count * TPL_ROWS * 2);
        uint8_t *d_pieces = NULL, *h_pieces = malloc((size_t)count * (gM + 1)), *g_pieces = malloc((size_t)count * (gM + 1));
        int32_t* d_status = NULL;
        void* work = NULL;
        const size_t work_bytes = tpl_generate_configs_device_work_bytes(gM, count);
        CHECK(hipMalloc((void**)&d_rows, (size_t)count * TPL_ROWS * 2));
        CHECK(hipMalloc((void**)&d_pieces, (size_t)count * (gM + 1)));
        CHECK(hipMalloc((void**)&d_status, (size_t)count * sizeof(int32_t)));
        CHECK(hipMalloc(&work, work_bytes));
        CHECK(tpl_generate_configs_device(gL, gM, seed, 0, count, 0, d_rows, d_pieces, NULL, NULL, d_status, work, work_bytes, NULL));
        CHECK(hipMemcpy(g_rows, d_rows, (size_t)count * TPL_ROWS * 2, kDeviceToHost));
        CHECK(hipMemcpy(g_pieces, d_pieces, (size_t)count * (gM + 1), kDeviceToHost));
        CHECK(tpl_generate_configs(gL, gM, seed, 0, count, 2, 0, h_rows, h_pieces, NULL, NULL));
        printf("carve device==host %s\n", memcmp(g_rows, h_rows, (size_t)count * TPL_ROWS * 2) == 0 &&
                                              memcmp(g_pieces, h_pieces, (size_t)count * (gM + 1)) == 0 ? "ok" : "DIFFERS");
        hipFree(work); hipFree(d_status);

        uint64_t h_seeds[100], *d_seeds = NULL;
        uint8_t *d_seq = NULL, *d_win = NULL, h_win[100], g_win[100], *h_seq = malloc(100 * (size_t)gM);
        for (int k = 0; k < 100; ++k) h_seeds[k] = (uint64_t)k;                /* the reference's own seeds (main.py:39-40) */
        const size_t fwork_bytes = tpl_forward_generate_device_work_bytes(gM, 100);
        CHECK(hipMalloc((void**)&d_seeds, sizeof(h_seeds)));
        CHECK(hipMalloc((void**)&d_seq, 100 * (size_t)gM));
        CHECK(hipMalloc((void**)&d_win, 100));
        CHECK(hipMalloc(&work, fwork_bytes));
        CHECK(hipMemcpy(d_seeds, h_seeds, sizeof(h_seeds), kHostToDevice));
        CHECK(tpl_forward_generate_device(gL, gM, 4, 1000, d_seeds, 100, d_rows, d_seq, d_win, NULL, NULL, NULL, NULL, work, fwork_bytes, NULL));
        CHECK(hipMemcpy(g_win, d_win, 100, kDeviceToHost));
        CHECK(hipMemcpy(g_rows, d_rows, 100 * TPL_ROWS * 2, kDeviceToHost));
        CHECK(tpl_forward_generate(gL, gM, 4, 1000, h_seeds, 100, 2, h_rows, h_seq, h_win, NULL, NULL, NULL, NULL));
        int winnable = 0;
        for (int k = 0; k < 100; ++k) winnable += g_win[k];
        printf("forward device==host %s winnable %d\n", memcmp(g_win, h_win, 100) == 0 && memcmp(g_rows, h_rows, 100 * TPL_ROWS * 2) == 0 ? "ok" : "DIFFERS",
               winnable);
        hipFree(work); hipFree(d_seeds); hipFree(d_seq); hipFree(d_win); hipFree(d_rows); hipFree(d_pieces);
        free(h_rows); free(g_rows); free(h_pieces); free(g_pieces); free(h_seq);
    }
    CHECK(tpl_destroy(env));
    hipFree(rows); hipFree(pieces); hipFree(action); hipFree(reward); hipFree(done); hipFree(stats);
    free(h_reward); free(h_done);
    return 0;
}
