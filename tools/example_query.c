/* example_query.c -- the sybl_* C ABI from plain C99 (what a cgo shim does, minus Go): open a table in
 * compact storage, run `-group <g> -int <v> -op hist -int-filter <v>:gt:<x>`, print the first rows.
 *
 *   gcc -std=c99 -Iinclude tools/example_query.c -Lsybil_amd -lsybilgpu -Wl,-rpath,$PWD/sybil_amd -o example_query
 *   ./example_query db events browser pageload 100
 *
 * Several GPUs, one process each (the last three arguments): rank, number of ranks, and a file through which rank 0 hands the
 * communicator id to the others --
 *   ./example_query db events browser pageload 100 0 2 /tmp/id &  ./example_query db events browser pageload 100 1 2 /tmp/id
 * (HIP_VISIBLE_DEVICES picks each process' GPU).  Everything collective is inside the library: sybl_comm_init,
 * sybl_table_agree, sybl_query_allreduce.
 */
#define _DEFAULT_SOURCE /* usleep under -std=c99 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "sybilgpu.h"

static int die(const char *what) {
    fprintf(stderr, "%s: %s\n", what, sybl_last_error());
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s <dir> <table> <group column> <int column> <greater-than> [<rank> <ranks> <id file>]\n", argv[0]);
        return 2;
    }
    const int rank = argc >= 9 ? atoi(argv[6]) : 0, nranks = argc >= 9 ? atoi(argv[7]) : 1;
    sybl_ctx *ctx = NULL;
    if (sybl_init(0, &ctx)) return die("sybl_init");
    if (nranks > 1) { /* rank 0 makes the 128-byte id and renames it into place; the others wait for the file */
        unsigned char id[128];
        char tmp[4096];
        snprintf(tmp, sizeof(tmp), "%s.tmp", argv[8]);
        if (rank == 0) {
            FILE *f;
            if (sybl_comm_unique_id(id)) return die("sybl_comm_unique_id");
            f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, 128, f) != 128 || fclose(f) || rename(tmp, argv[8])) return die("id file");
        } else {
            size_t got = 0;
            int tries;
            for (tries = 0; got != 128 && tries < 60000; tries++) {
                FILE *f = fopen(argv[8], "rb");
                if (f) {
                    got = fread(id, 1, 128, f);
                    fclose(f);
                }
                if (got != 128) usleep(2000);
            }
            if (got != 128) return die("id file");
        }
        if (sybl_comm_init(ctx, id, nranks, rank)) return die("sybl_comm_init");
    }

    const char *cols[2] = {argv[3], argv[4]}; /* only the referenced columns become resident */
    sybl_table *tab = NULL;
    /* this rank's contiguous share of the block directories */
    if (sybl_table_open_flags(ctx, argv[1], argv[2], cols, 2, rank, nranks, SYBL_OPEN_COMPACT, &tab)) return die("sybl_table_open_flags");
    /* collective: bounds, dictionaries and the group key's dictionary of distinct values become the same on every rank
     * (one rank: the str dictionaries are sorted, so the output does not depend on the number of GPUs) */
    if (sybl_table_agree(tab, &cols[0], 1)) return die("sybl_table_agree");
    if (sybl_table_compact(tab)) return die("sybl_table_compact"); /* re-narrow what the load widened */

    sybl_filter filt;
    memset(&filt, 0, sizeof(filt));
    filt.col = argv[4];
    filt.op = SYBL_OP_GT;
    filt.int_value = atoll(argv[5]);
    const char *groups[1] = {argv[3]}, *aggs[1] = {argv[4]};
    sybl_query_desc d;
    memset(&d, 0, sizeof(d));
    d.n_filters = 1;
    d.filters = &filt;
    d.n_groups = 1;
    d.groups = groups;
    d.n_aggs = 1;
    d.aggs = aggs;
    d.op = SYBL_AGG_HIST;
    d.want_percentiles = 1;
    d.order_by = "$COUNT";
    d.limit = 10;
    d.block_skip = 1;

    sybl_query *q = NULL;
    if (sybl_query_prepare(tab, &d, &q)) return die("sybl_query_prepare");
    if (sybl_query_scan(q)) return die("sybl_query_scan");
    /* the one collective of the step: SUM (+ MAX) all-reduce of the partial group tables over RCCL / xGMI; its first call on
     * a query also checks that the ranks' layouts agree */
    if (nranks > 1 && sybl_query_allreduce(q)) return die("sybl_query_allreduce");
    sybl_result *res = NULL;
    if (sybl_query_finalize(q, &res)) return die("sybl_query_finalize"); /* every rank: it is collective after a reduce-scatter */
    if (rank != 0) { /* rank 0 prints */
        sybl_result_free(res);
        sybl_query_free(q);
        sybl_table_free(tab);
        sybl_comm_free(ctx);
        sybl_shutdown(ctx);
        return 0;
    }

    const sybl_group_row *rows = NULL;
    int64_t n = 0;
    if (sybl_result_rows(res, 0, &rows, &n)) return die("sybl_result_rows");
    printf("matched %lld rows, %lld groups\n", (long long)sybl_result_matched(res), (long long)n);
    for (int64_t i = 0; i < n && i < d.limit; i++) {
        const sybl_agg_out *h = &rows[i].aggs[0];
        printf("%-20s count %-10lld avg %.2f  p50 %lld  p99 %lld\n", rows[i].group_by_key, (long long)rows[i].count, h->avg,
               h->percentiles ? (long long)h->percentiles[50] : 0LL, h->percentiles ? (long long)h->percentiles[99] : 0LL);
    }
    int64_t n_bytes = 0;
    if (sybl_result_encode(res, &n_bytes)) printf("-encode-results: %lld bytes of gob\n", (long long)n_bytes);

    sybl_result_free(res);
    sybl_query_free(q);
    sybl_table_free(tab);
    if (nranks > 1) sybl_comm_free(ctx);
    sybl_shutdown(ctx);
    return 0;
}
