/*
 * tstwo_oracle_mt.c — pthread drivers over the single-threaded CPU oracle (tstwo_oracle.c).
 *
 * TEST INFRASTRUCTURE ONLY, like the oracle itself: reached from tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg, never from tstwo_amd/.  Nothing here restates an algorithm: each thread calls the oracle's own
 * orc_cfft_evaluate / orc_merkle_commit / orc_hash_node on its share of the work.
 *
 *   orc_mt_cfft_evaluate : BASELINE config 5's "independent trace columns" — one column per task
 *                          (PolyOps.evaluate per column, backend/cpu/circle.ts:84-134).
 *   orc_mt_merkle_root   : MerkleProver.commit (vcs/prover.ts:13-30) of equal-length columns with the leaf range cut
 *                          into 2^k contiguous shards: every shard is committed to its subtree root by
 *                          orc_merkle_commit, the top k levels are hashNode(left, right, []) (vcs/blake2_merkle.ts:9-24)
 *                          — the same node values as the single tree, so the root is identical (tested).
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "tstwo_oracle.h"

typedef struct {
    uint32_t *const *cols;
    size_t n_cols;
    uint32_t log_size, half_initial, tw_log;
    const uint32_t *tw;
    size_t next;              /* next column to take (guarded by mu) */
    int rc;
    pthread_mutex_t mu;
} cfft_job;

static void *cfft_worker(void *arg) {
    cfft_job *j = (cfft_job *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        size_t c = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (c >= j->n_cols) return NULL;
        int rc = orc_cfft_evaluate(j->cols[c], j->log_size, j->half_initial, j->tw, j->tw_log, 0);
        if (rc) {
            pthread_mutex_lock(&j->mu);
            j->rc = rc;
            pthread_mutex_unlock(&j->mu);
        }
    }
}

int orc_mt_cfft_evaluate(uint32_t *const *cols, size_t n_cols, uint32_t log_size, uint32_t half_initial,
                         const uint32_t *tw, uint32_t tw_log, unsigned threads) {
    if (threads == 0) threads = 1;
    if (threads > 1024) threads = 1024;
    cfft_job j = {cols, n_cols, log_size, half_initial, tw_log, tw, 0, 0, PTHREAD_MUTEX_INITIALIZER};
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    if (!th) return 100;
    unsigned started = 0;
    for (; started < threads; started++)
        if (pthread_create(&th[started], NULL, cfft_worker, &j)) break;
    if (started == 0) cfft_worker(&j);
    for (unsigned t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
    return j.rc;
}

typedef struct {
    const uint32_t *const *cols;
    size_t n_cols;
    uint32_t shard_log;       /* log2(leaves per shard) */
    size_t n_shards;
    uint8_t *roots;           /* n_shards * 32 bytes */
    size_t next;
    int rc;
    pthread_mutex_t mu;
} merkle_job;

static void *merkle_worker(void *arg) {
    merkle_job *j = (merkle_job *)arg;
    const size_t rows = (size_t)1 << j->shard_log;
    const uint32_t **ptrs = (const uint32_t **)malloc(sizeof(uint32_t *) * (j->n_cols ? j->n_cols : 1));
    uint32_t *logs = (uint32_t *)malloc(sizeof(uint32_t) * (j->n_cols ? j->n_cols : 1));
    uint8_t *layers = (uint8_t *)malloc((((size_t)2 << j->shard_log) - 1) * 32);
    if (!ptrs || !logs || !layers) {
        pthread_mutex_lock(&j->mu); j->rc = 100; pthread_mutex_unlock(&j->mu);
        free(ptrs); free(logs); free(layers);
        return NULL;
    }
    for (size_t c = 0; c < j->n_cols; c++) logs[c] = j->shard_log;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        size_t s = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (s >= j->n_shards) break;
        for (size_t c = 0; c < j->n_cols; c++) ptrs[c] = j->cols[c] + s * rows;
        int rc = orc_merkle_commit(ptrs, logs, j->n_cols, layers, j->roots + 32 * s);
        if (rc) { pthread_mutex_lock(&j->mu); j->rc = rc; pthread_mutex_unlock(&j->mu); }
    }
    free(ptrs); free(logs); free(layers);
    return NULL;
}

int orc_mt_merkle_root(const uint32_t *const *cols, size_t n_cols, uint32_t log_size, unsigned threads, uint8_t root[32]) {
    if (threads == 0) threads = 1;
    if (threads > 1024) threads = 1024;
    /* shards: the largest power of two <= 4 * threads that leaves >= 2^4 leaves per shard (dynamic balance) */
    uint32_t k = 0;
    while (((size_t)2 << k) <= (size_t)4 * threads && k + 1 + 4 <= log_size) k++;
    merkle_job j = {cols, n_cols, log_size - k, (size_t)1 << k, NULL, 0, 0, PTHREAD_MUTEX_INITIALIZER};
    j.roots = (uint8_t *)malloc(j.n_shards * 32);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    if (!j.roots || !th) { free(j.roots); free(th); return 100; }
    unsigned started = 0;
    for (; started < threads; started++)
        if (pthread_create(&th[started], NULL, merkle_worker, &j)) break;
    if (started == 0) merkle_worker(&j);
    for (unsigned t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
    if (!j.rc) {
        for (size_t m = j.n_shards; m > 1; m >>= 1)                      /* top k levels: children only, no column values */
            for (size_t i = 0; i < m / 2; i++) orc_hash_node(j.roots + 64 * i, j.roots + 64 * i + 32, NULL, 0, j.roots + 32 * i);
        memcpy(root, j.roots, 32);
    }
    free(j.roots);
    return j.rc;
}
