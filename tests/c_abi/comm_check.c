/* Plain C caller of the library's RCCL exchange (include/mpcx.h, mpcx_comm_*).
 *
 *     comm_check                                one process, one GPU: a one-rank communicator
 *     comm_check WORLD RANK ID_FILE [DEVICE]    rank RANK of WORLD, one process per GPU (DEVICE defaults to RANK modulo
 *                                               the number of visible GPUs).  Start the WORLD processes together, e.g.
 *         for r in 0 1 2 3 4 5 6 7; do ./comm_check 8 $r /tmp/mpcx.id & done; wait
 *                                               Rank 0 obtains the communicator id and publishes it as ID_FILE (written
 *                                               under a temporary name, then renamed); the others wait for the file.
 *
 * Every rank fills N doubles from a generator seeded by its rank, so each rank can reproduce what every other rank
 * holds: the gathered statistics records must equal the records mpcx_stats_f64 gives for those arrays on this rank's
 * own GPU, and the array gathered to rank 0 must equal the concatenation in rank order.  A host in any language does
 * exactly this per rank.  Device memory comes from the HIP runtime the library itself uses.  Exit 3 without a GPU. */
#define _DEFAULT_SOURCE 1 /* usleep, getpid under -std=c99 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "mpcx.h"

#define N 100003

static void fill(double *host, int rank) {
    unsigned long long s = 12345 + 7919ULL * (unsigned long long)rank;
    for (int i = 0; i < N; ++i) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        host[i] = (double)(long long)(s >> 11) / 9007199254740992.0 - 0.5;
    }
    host[17 + rank] = 0.0 / 0.0; /* one non-finite value per rank must be counted */
}

static int publish_id(const char *path, const unsigned char *id) {
    char tmp[4096];
    snprintf(tmp, sizeof tmp, "%s.tmp.%ld", path, (long)getpid());
    FILE *fh = fopen(tmp, "wb");
    if (!fh) return 1;
    size_t w = fwrite(id, 1, MPCX_COMM_ID_BYTES, fh);
    if (fclose(fh) != 0 || w != MPCX_COMM_ID_BYTES) return 1;
    return rename(tmp, path) != 0;
}

static int await_id(const char *path, unsigned char *id) {
    for (int tries = 0; tries < 1200; ++tries) { /* two minutes */
        FILE *fh = fopen(path, "rb");
        if (fh) {
            size_t got = fread(id, 1, MPCX_COMM_ID_BYTES, fh);
            fclose(fh);
            if (got == MPCX_COMM_ID_BYTES) return 0;
        }
        usleep(100000);
    }
    return 1;
}

int main(int argc, char **argv) {
    int world = 1, rank = 0, device = 0;
    const char *id_file = NULL;
    if (argc >= 4) {
        world = atoi(argv[1]);
        rank = atoi(argv[2]);
        id_file = argv[3];
        if (world < 1 || rank < 0 || rank >= world) { printf("usage: comm_check [WORLD RANK ID_FILE [DEVICE]]\n"); return 2; }
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) { printf("create rc=%d\n", MPCX_ERR_NO_DEVICE); return 3; }
        device = argc >= 5 ? atoi(argv[4]) : rank % n_dev;
    } else if (argc != 1) {
        printf("usage: comm_check [WORLD RANK ID_FILE [DEVICE]]\n");
        return 2;
    }
    mpcx_ctx *ctx = NULL;
    int rc = mpcx_create(&ctx, device);
    if (rc != MPCX_OK) { printf("create rc=%d\n", rc); return 3; }
    if (hipSetDevice(device) != hipSuccess) return 1;
    double *host = (double *)malloc(sizeof(double) * N);
    double *back = (double *)malloc(sizeof(double) * N * (size_t)world);
    mpcx_stats *want = (mpcx_stats *)malloc(sizeof(mpcx_stats) * (size_t)world);
    mpcx_stats *all = (mpcx_stats *)malloc(sizeof(mpcx_stats) * (size_t)world);
    double *src = NULL, *dst = NULL, *tmp = NULL;
    if (hipMalloc((void **)&src, sizeof(double) * N) != hipSuccess) return 1;
    if (hipMalloc((void **)&tmp, sizeof(double) * N) != hipSuccess) return 1;
    if (hipMalloc((void **)&dst, sizeof(double) * N * (size_t)world) != hipSuccess) return 1;
    if (hipMemset(dst, 0, sizeof(double) * N * (size_t)world) != hipSuccess) return 1;

    /* what every rank's record has to be, computed here from that rank's (reproducible) array */
    for (int g = 0; g < world; ++g) {
        fill(host, g);
        if (hipMemcpy(g == rank ? src : tmp, host, sizeof(double) * N, hipMemcpyHostToDevice) != hipSuccess) return 1;
        rc = mpcx_stats_f64(ctx, g == rank ? src : tmp, N, &want[g]);
        if (rc != MPCX_OK) { printf("stats rc=%d %s\n", rc, mpcx_last_error(ctx)); return 1; }
    }
    mpcx_stats one = want[rank];

    unsigned char id[MPCX_COMM_ID_BYTES];
    if (rank == 0) {
        rc = mpcx_comm_get_id(id);
        if (rc != MPCX_OK) { printf("get_id rc=%d %s\n", rc, mpcx_comm_last_error(NULL)); return 1; }
        if (id_file && publish_id(id_file, id)) { printf("cannot write %s\n", id_file); return 1; }
    } else if (await_id(id_file, id)) {
        printf("rank %d: no id in %s after two minutes\n", rank, id_file);
        return 1;
    }
    mpcx_comm *comm = NULL;
    rc = mpcx_comm_create(&comm, ctx, id, world, rank);
    if (rc != MPCX_OK) { printf("comm_create rc=%d %s\n", rc, mpcx_comm_last_error(NULL)); return 1; }
    if (mpcx_comm_world(comm) != world || mpcx_comm_rank(comm) != rank) { printf("world/rank wrong\n"); return 1; }

    memset(all, 0, sizeof(mpcx_stats) * (size_t)world);
    rc = mpcx_gather_stats(comm, src, N, NULL, all);
    if (rc != MPCX_OK) { printf("gather_stats rc=%d %s\n", rc, mpcx_comm_last_error(comm)); return 1; }
    for (int g = 0; g < world; ++g)
        if (memcmp(&all[g], &want[g], sizeof one) != 0) { printf("rank %d: gathered record %d differs from mpcx_stats_f64\n", rank, g); return 1; }
    if (one.count != N || one.non_finite != 1) { printf("record wrong: count %lld non_finite %lld\n", (long long)one.count, (long long)one.non_finite); return 1; }

    for (int root = 0; root < (world > 1 ? 2 : 1); ++root) { /* rank 0 and, with more ranks, rank 1 as the root */
        if (hipMemset(dst, 0, sizeof(double) * N * (size_t)world) != hipSuccess) return 1;
        rc = mpcx_gather_f64_dev(comm, src, N, dst, root, NULL);
        if (rc == MPCX_OK) rc = mpcx_comm_synchronize(comm);
        if (rc != MPCX_OK) { printf("gather rc=%d %s\n", rc, mpcx_comm_last_error(comm)); return 1; }
        if (rank == root) {
            if (hipMemcpy(back, dst, sizeof(double) * N * (size_t)world, hipMemcpyDeviceToHost) != hipSuccess) return 1;
            for (int g = 0; g < world; ++g) {
                fill(host, g);
                if (memcmp(back + (size_t)g * N, host, sizeof(double) * N) != 0) { printf("gathered array: block of rank %d differs (root %d)\n", g, root); return 1; }
            }
        }
    }

    rc = mpcx_gather_f64_dev(comm, src, N, dst, world, NULL); /* not a rank */
    printf("bad_root rc=%d\n", rc);
    mpcx_comm_destroy(comm);
    mpcx_destroy(ctx);
    (void)hipFree(src);
    (void)hipFree(tmp);
    (void)hipFree(dst);
    printf("comm ok world=%d rank=%d count=%lld checksum=%016llx\n", world, rank, (long long)one.count, (unsigned long long)one.checksum);
    return 0;
}
