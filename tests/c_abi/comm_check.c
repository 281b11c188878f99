/* Plain C caller of the library's RCCL exchange (include/mpcx.h, mpcx_comm_*): a process that holds one GPU forms a
 * one-rank communicator, and the gathered statistics record / result array must equal what the single-device calls give.
 * A host in any language does exactly this per rank (with world > 1 and the id passed between the ranks).
 * Device memory comes from the HIP runtime the library itself uses.  Exit 3 without a GPU. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpcx.h"

#define N 100003

int main(void) {
    mpcx_ctx *ctx = NULL;
    int rc = mpcx_create(&ctx, 0);
    if (rc != MPCX_OK) { printf("create rc=%d\n", rc); return 3; }
    double *host = (double *)malloc(sizeof(double) * N), *back = (double *)malloc(sizeof(double) * N);
    unsigned long long s = 12345;
    for (int i = 0; i < N; ++i) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        host[i] = (double)(long long)(s >> 11) / 9007199254740992.0 - 0.5;
    }
    host[17] = 0.0 / 0.0; /* one non-finite value must be counted */
    double *src = NULL, *dst = NULL;
    if (hipMalloc((void **)&src, sizeof(double) * N) != hipSuccess || hipMalloc((void **)&dst, sizeof(double) * N) != hipSuccess) return 1;
    if (hipMemcpy(src, host, sizeof(double) * N, hipMemcpyHostToDevice) != hipSuccess) return 1;
    if (hipMemset(dst, 0, sizeof(double) * N) != hipSuccess) return 1;

    mpcx_stats one;
    rc = mpcx_stats_f64(ctx, src, N, &one);
    if (rc != MPCX_OK) { printf("stats rc=%d %s\n", rc, mpcx_last_error(ctx)); return 1; }

    unsigned char id[MPCX_COMM_ID_BYTES];
    rc = mpcx_comm_get_id(id);
    if (rc != MPCX_OK) { printf("get_id rc=%d %s\n", rc, mpcx_comm_last_error(NULL)); return 1; }
    mpcx_comm *comm = NULL;
    rc = mpcx_comm_create(&comm, ctx, id, 1, 0);
    if (rc != MPCX_OK) { printf("comm_create rc=%d %s\n", rc, mpcx_comm_last_error(NULL)); return 1; }
    if (mpcx_comm_world(comm) != 1 || mpcx_comm_rank(comm) != 0) { printf("world/rank wrong\n"); return 1; }

    mpcx_stats all[1];
    memset(all, 0, sizeof all);
    rc = mpcx_gather_stats(comm, src, N, NULL, all);
    if (rc != MPCX_OK) { printf("gather_stats rc=%d %s\n", rc, mpcx_comm_last_error(comm)); return 1; }
    if (memcmp(&all[0], &one, sizeof one) != 0) { printf("gathered record differs from mpcx_stats_f64\n"); return 1; }
    if (one.count != N || one.non_finite != 1) { printf("record wrong: count %lld non_finite %lld\n", (long long)one.count, (long long)one.non_finite); return 1; }

    rc = mpcx_gather_f64_dev(comm, src, N, dst, 0, NULL);
    if (rc == MPCX_OK) rc = mpcx_comm_synchronize(comm);
    if (rc != MPCX_OK) { printf("gather rc=%d %s\n", rc, mpcx_comm_last_error(comm)); return 1; }
    if (hipMemcpy(back, dst, sizeof(double) * N, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (memcmp(back, host, sizeof(double) * N) != 0) { printf("gathered array differs\n"); return 1; }

    rc = mpcx_gather_f64_dev(comm, src, N, dst, 1, NULL); /* not a rank */
    printf("bad_root rc=%d\n", rc);
    mpcx_comm_destroy(comm);
    mpcx_destroy(ctx);
    (void)hipFree(src);
    (void)hipFree(dst);
    printf("comm ok count=%lld checksum=%016llx\n", (long long)one.count, (unsigned long long)one.checksum);
    return 0;
}
