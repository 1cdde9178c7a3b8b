/* sliced_procs_demo.c -- a multi-GPU HOST in plain C, no Python, no torch: one PROCESS per rank of the reference's ONE
 * sorted list (update_idx_leaf, src/indexed_merkle_tree.rs:632-660; insertion i at leaf size + i, :715) through
 * include/imt.h alone: imt_transport_{rccl,ipc}_create, imt_sliced_create / _step / _wait / _flush.
 *
 *   sliced_procs_demo WORLD [rccl|ipc] [STEPS] [SLICE]
 *
 * The parent forks WORLD children BEFORE anything touches the GPU and carries their bootstrap bytes over pipes (what
 * MPI_Bcast / MPI_Allgather would do): the RCCL unique ids from rank 0 to everyone, or every rank's IPC handle blob to
 * everyone.  Rank g uses device IMT_DEMO_DEVICE if set (a one-GPU box: every rank on that device, transport ipc) or
 * device g (one GPU per rank, transport rccl).  Every rank generates the same values, makes the same imt_sliced_step
 * calls, and reports its replica's root; rank 0 also inserts everything into an ordinary tree.  All roots must agree.
 *
 *   gcc -std=c11 -D_POSIX_C_SOURCE=200809L -I include examples/sliced_procs_demo.c -L indexed-merkle-tree-halo2_amd/csrc -limt_hip -o sliced_procs_demo
 */
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>
#include "imt.h"

#define DEPTH 32
#define MAXW 16

static int read_all(int fd, void *buf, size_t n) {
    unsigned char *p = buf;
    while (n) {
        ssize_t k = read(fd, p, n);
        if (k <= 0) return -1;
        p += k;
        n -= (size_t)k;
    }
    return 0;
}
static int write_all(int fd, const void *buf, size_t n) {
    const unsigned char *p = buf;
    while (n) {
        ssize_t k = write(fd, p, n);
        if (k <= 0) return -1;
        p += k;
        n -= (size_t)k;
    }
    return 0;
}

/* the same distinct, non-zero, canonical values in every process */
static void make_values(unsigned char (*v)[32], size_t n) {
    unsigned long long s = 0x494D54ull;
    memset(v, 0, n * 32);
    for (size_t i = 0; i < n; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        unsigned long long a = s;
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        memcpy(v[i], &a, 8);
        memcpy(v[i] + 8, &s, 8);
        unsigned long long idx = i + 1;          /* distinct whatever the generator does */
        memcpy(v[i] + 16, &idx, 8);
        v[i][24] = (unsigned char)(s >> 40);
    }
}

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc__ = (call);                                                                            \
        if (rc__) { fprintf(stderr, "[rank %d] %s: %d %s\n", rank, #call, rc__, imt_last_error(ctx)); return 10; } \
    } while (0)

static int child(int rank, int world, int use_rccl, int steps, size_t slice, int to_parent, int from_parent) {
    const char *dev_env = getenv("IMT_DEMO_DEVICE");
    imt_ctx *ctx = NULL;
    int rc = imt_ctx_create(dev_env ? atoi(dev_env) : rank, &ctx);
    if (rc) { fprintf(stderr, "[rank %d] imt_ctx_create: %d (no GPU?)\n", rank, rc); return 2; }
    imt_itree *tree = NULL;
    const size_t total = (size_t)steps * world * slice;
    uint64_t cap = 2;
    while (cap < total + 1) cap <<= 1;
    CHECK(imt_itree_new(ctx, DEPTH, cap, &tree));
    /* ---- the transport: bootstrap bytes through the parent ---- */
    imt_transport *tp = NULL;
    if (use_rccl) {
        unsigned char ids[IMT_SLICED_ROUNDS * IMT_RCCL_UNIQUE_ID_BYTES];
        if (rank == 0) {
            for (int i = 0; i < IMT_SLICED_ROUNDS; i++) CHECK(imt_rccl_get_unique_id(ids + i * IMT_RCCL_UNIQUE_ID_BYTES));
            if (write_all(to_parent, ids, sizeof ids)) return 3;
        }
        if (read_all(from_parent, ids, sizeof ids)) return 3;
        CHECK(imt_transport_rccl_create(ctx, ids, IMT_SLICED_ROUNDS, world, rank, &tp));
    } else {
        const size_t nb = imt_transport_ipc_blob_bytes();
        unsigned char *all = malloc(nb * (size_t)world);
        CHECK(imt_transport_ipc_create(ctx, world, rank, DEPTH, slice, 0, &tp, all + nb * (size_t)rank));
        if (write_all(to_parent, all + nb * (size_t)rank, nb) || read_all(from_parent, all, nb * (size_t)world)) return 3;
        CHECK(imt_transport_ipc_connect(tp, all));
        free(all);
    }
    imt_sliced *w = NULL;
    CHECK(imt_sliced_create(&tree, 1, world, rank, tp, slice, 0, &w));
    /* ---- values and this rank's witness buffers, in memory the kernels can address ---- */
    unsigned char(*vals)[32];
    unsigned char(*new_root)[32];
    CHECK(imt_host_alloc(ctx, total * 32, (void **)&vals));
    CHECK(imt_host_alloc(ctx, (size_t)steps * slice * 32, (void **)&new_root));
    make_values(vals, total);
    for (int s = 0; s < steps; s++) {
        imt_insert_out out;
        memset(&out, 0, sizeof out);
        out.new_root = new_root[(size_t)s * slice];
        uint64_t round = 0;
        rc = imt_sliced_step(w, vals[(size_t)s * world * slice], slice, &out, 0, &round);
        if (rc) { fprintf(stderr, "[rank %d] imt_sliced_step: %d %s\n", rank, rc, imt_sliced_last_error(w)); return 11; }
        if (s >= 2) {                            /* read a finished step while later ones are in flight */
            rc = imt_sliced_wait(w, 0, round - 2);
            if (rc) { fprintf(stderr, "[rank %d] imt_sliced_wait: %d %s\n", rank, rc, imt_sliced_last_error(w)); return 12; }
        }
    }
    rc = imt_sliced_flush(w);
    if (rc) { fprintf(stderr, "[rank %d] imt_sliced_flush: %d %s\n", rank, rc, imt_sliced_last_error(w)); return 13; }
    imt_sliced_info info;
    CHECK(imt_sliced_get_info(w, &info));
    unsigned char report[64];
    CHECK(imt_itree_root(tree, report, IMT_FMT_CANONICAL));
    /* the last rank's last new root closes the last step: it is the tree's root */
    memcpy(report + 32, rank == world - 1 ? new_root[(size_t)steps * slice - 1] : report, 32);
    if (rank == 0) {                             /* the ordinary tree over the same values */
        imt_ctx *c2 = NULL;
        imt_itree *ref = NULL;
        unsigned char rr[32];
        if (imt_ctx_create(dev_env ? atoi(dev_env) : rank, &c2) || imt_itree_new(c2, DEPTH, cap, &ref) ||
            imt_itree_insert_batch(ref, vals, total, NULL, IMT_DEVICE_PTRS) || imt_ctx_sync(c2) ||
            imt_itree_root(ref, rr, IMT_FMT_CANONICAL))
            return 14;
        if (memcmp(rr, report, 32)) { fprintf(stderr, "[rank 0] replica differs from the one-tree batch\n"); return 15; }
        imt_itree_free(ref);
        imt_ctx_destroy(c2);
        printf("rank 0: lag %d, %llu all-gathers, %.2f ms issuing + %.2f ms waiting per step; root ", info.lag,
               (unsigned long long)info.collectives, info.host_issue_ms / steps, info.host_wait_ms / steps);
        for (int k = 31; k >= 0; k--) printf("%02x", report[k]);
        printf("\n");
        fflush(stdout);
    }
    if (write_all(to_parent, report, 64)) return 3;
    /* nobody frees the buffers it exports while a peer may still read them: wait for the parent's go */
    char go;
    if (read_all(from_parent, &go, 1)) return 3;
    imt_sliced_destroy(w);
    imt_transport_destroy(tp);
    imt_host_free(ctx, vals);
    imt_host_free(ctx, new_root);
    imt_itree_free(tree);
    imt_ctx_destroy(ctx);
    return 0;
}

int main(int argc, char **argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 2;
    const int use_rccl = argc > 2 && !strcmp(argv[2], "rccl");
    const int steps = argc > 3 ? atoi(argv[3]) : 6;
    const size_t slice = argc > 4 ? (size_t)atol(argv[4]) : 256;
    if (world < 1 || world > MAXW || steps < 1 || slice < 1 || (!use_rccl && world < 2)) {
        fprintf(stderr, "usage: %s WORLD(1..%d; ipc: >= 2) [rccl|ipc] [STEPS] [SLICE]\n", argv[0], MAXW);
        return 1;
    }
    signal(SIGPIPE, SIG_IGN);                    /* a rank that died is reported by its exit status, not by a signal here */
    int up[MAXW][2], down[MAXW][2];
    pid_t pid[MAXW];
    for (int r = 0; r < world; r++) {
        if (pipe(up[r]) || pipe(down[r])) return 1;
        pid[r] = fork();                         /* before this process has made any GPU call */
        if (pid[r] < 0) return 1;
        if (pid[r] == 0) {
            for (int q = 0; q <= r; q++) { close(up[q][0]); close(down[q][1]); }
            _exit(child(r, world, use_rccl, steps, slice, up[r][1], down[r][0]));
        }
        close(up[r][1]);
        close(down[r][0]);
    }
    int bad = 0;
    /* bootstrap: broadcast (rccl) or all-gather (ipc) */
    if (use_rccl) {
        unsigned char ids[IMT_SLICED_ROUNDS * IMT_RCCL_UNIQUE_ID_BYTES];
        bad |= read_all(up[0][0], ids, sizeof ids);
        for (int r = 0; r < world && !bad; r++) bad |= write_all(down[r][1], ids, sizeof ids);
    } else {
        const size_t nb = imt_transport_ipc_blob_bytes();      /* arithmetic only: no GPU call in the parent */
        unsigned char *all = malloc(nb * (size_t)world);
        for (int r = 0; r < world && !bad; r++) bad |= read_all(up[r][0], all + nb * (size_t)r, nb);
        for (int r = 0; r < world && !bad; r++) bad |= write_all(down[r][1], all, nb * (size_t)world);
        free(all);
    }
    unsigned char rep[MAXW][64];
    for (int r = 0; r < world && !bad; r++) bad |= read_all(up[r][0], rep[r], 64);
    int same = !bad;
    for (int r = 1; r < world && same; r++) same &= memcmp(rep[r], rep[0], 32) == 0;
    if (same) same &= memcmp(rep[world - 1] + 32, rep[0], 32) == 0;
    for (int r = 0; r < world; r++) (void)!write(down[r][1], "g", 1);
    int status = 0;
    for (int r = 0; r < world; r++) {
        int st = 0;
        waitpid(pid[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st)) status = 1;
    }
    printf("%d processes over %s, %d steps of %d x %zu insertions: replicas %s (the one-tree batch included)\n", world,
           use_rccl ? "RCCL" : "IPC", steps, world, slice, same && !status ? "equal" : "DIFFER / FAILED");
    return same && !status ? 0 : 3;
}
