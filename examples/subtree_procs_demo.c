/* subtree_procs_demo.c -- north_star's multi-GPU layout from a plain-C host, no Python, no torch, no collective library of
 * the host's own: one PROCESS per GPU, GPU g owns the leaf-index range [g << (32 - k), (g + 1) << (32 - k)) of ONE depth-32
 * tree as an indexed subtree (imt_itree_set_placement) and the values with v mod WORLD == g (imt_itree_set_value_partition);
 * per step ONE collective -- the all-gather of the WORLD subtree roots, 32 bytes per rank, through the library's own
 * communicators (imt_transport_all_gather: RCCL's ncclAllGather, or peer reads over HIP IPC) -- one step behind the
 * insertions, then every rank lifts its own witnesses to depth 32 (imt_itree_lift_batch) and checks each of them against
 * every constraint of the reference's insert_leaf (src/indexed_merkle_tree.rs:231-314) with imt_insert_witness_batch.
 * This is what indexed-merkle-tree-halo2_amd/sharded.py does over torch.distributed, restated over include/imt.h alone.
 *
 *   subtree_procs_demo WORLD [ipc|rccl] [STEPS] [BATCH]
 *
 * The parent forks WORLD children BEFORE anything touches the GPU and carries the bootstrap bytes over pipes (RCCL
 * unique id / IPC handle blobs) and, at the end, every rank's chain of roots: within a step rank 0's insertions come
 * first, then rank 1's ..., so rank g's last new root of a step must be rank g + 1's first old root, the last rank's
 * the next step's first -- one continuous sequence of depth-32 roots, as a circuit proving the insertions needs it.
 * IMT_DEMO_DEVICE: every rank on that device (a one-GPU box: transport ipc); else rank g on device g.
 *
 *   gcc -std=c11 -D_POSIX_C_SOURCE=200809L -I include examples/subtree_procs_demo.c -L indexed-merkle-tree-halo2_amd/csrc -limt_hip -o subtree_procs_demo
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
#define MAXSTEPS 64

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

/* distinct, non-zero, canonical values of residue `rank` modulo `world` (a power of two: the low bits of byte 0) */
static void make_values(unsigned char (*v)[32], size_t n, int world, int rank) {
    unsigned long long s = 0x494D5400ull + (unsigned)rank;
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
        v[i][0] = (unsigned char)((v[i][0] & ~(world - 1)) | rank);
    }
}

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc__ = (call);                                                                            \
        if (rc__) { fprintf(stderr, "[rank %d] %s: %d %s / %s\n", rank, #call, rc__, imt_last_error(ctx), tp ? imt_transport_last_error(tp) : ""); return 10; } \
    } while (0)

struct step_report {
    unsigned char first_old[32], last_new[32];
};

static int child(int rank, int world, int use_rccl, int steps, size_t batch, int to_parent, int from_parent) {
    const char *dev_env = getenv("IMT_DEMO_DEVICE");
    imt_ctx *ctx = NULL;
    imt_transport *tp = NULL;
    int rc = imt_ctx_create(dev_env ? atoi(dev_env) : rank, &ctx);
    if (rc) { fprintf(stderr, "[rank %d] imt_ctx_create: %d (no GPU?)\n", rank, rc); return 2; }
    unsigned k = 0;
    while ((1 << k) < world) k++;
    const unsigned sub_height = DEPTH - k;
    uint64_t cap = 2;
    while (cap < (uint64_t)steps * batch + 1) cap <<= 1;
    imt_itree *tree = NULL;
    CHECK(imt_itree_new(ctx, sub_height, cap, &tree));
    CHECK(imt_itree_set_placement(tree, DEPTH, (uint64_t)rank));
    CHECK(imt_itree_set_value_partition(tree, (uint32_t)world, (uint32_t)rank));
    /* ---- the transport: bootstrap bytes through the parent ---- */
    if (use_rccl) {
        unsigned char id[IMT_RCCL_UNIQUE_ID_BYTES];
        if (rank == 0) {
            CHECK(imt_rccl_get_unique_id(id));
            if (write_all(to_parent, id, sizeof id)) return 3;
        }
        if (read_all(from_parent, id, sizeof id)) return 3;
        CHECK(imt_transport_rccl_create(ctx, id, 1, world, rank, &tp));
    } else {
        const size_t nb = imt_transport_ipc_blob_bytes();
        unsigned char *all = malloc(nb * (size_t)world);
        CHECK(imt_transport_ipc_create(ctx, world, rank, DEPTH, 1, 0, &tp, all + nb * (size_t)rank));
        if (write_all(to_parent, all + nb * (size_t)rank, nb) || read_all(from_parent, all, nb * (size_t)world)) return 3;
        CHECK(imt_transport_ipc_connect(tp, all));
        free(all);
    }
    /* ---- buffers the kernels can address (page-locked host memory: the host reads the results in place) ---- */
    const size_t total = (size_t)steps * batch;
    unsigned char(*vals)[32];
    unsigned char *mine, *roots[2], *zero, *fail;
    uint64_t *new_index;
    CHECK(imt_host_alloc(ctx, total * 32, (void **)&vals));
    CHECK(imt_host_alloc(ctx, 32, (void **)&mine));
    CHECK(imt_host_alloc(ctx, (size_t)world * 32, (void **)&roots[0]));
    CHECK(imt_host_alloc(ctx, (size_t)world * 32, (void **)&roots[1]));
    CHECK(imt_host_alloc(ctx, (DEPTH + 1) * 32, (void **)&zero));
    CHECK(imt_host_alloc(ctx, batch, (void **)&fail));
    CHECK(imt_host_alloc(ctx, batch * 8, (void **)&new_index));
    imt_insert_out out[2];
    uint64_t first_new[2] = {0, 0};
    for (int b = 0; b < 2; b++) {
        memset(&out[b], 0, sizeof out[b]);
        CHECK(imt_host_alloc(ctx, batch * 8, (void **)&out[b].low_index));
        CHECK(imt_host_alloc(ctx, batch * 96, &out[b].low_leaf));
        CHECK(imt_host_alloc(ctx, batch, (void **)&out[b].is_largest));
        CHECK(imt_host_alloc(ctx, batch * 32, &out[b].old_root));
        CHECK(imt_host_alloc(ctx, batch * 32, &out[b].interim_root));
        CHECK(imt_host_alloc(ctx, batch * 32, &out[b].new_root));
        CHECK(imt_host_alloc(ctx, batch * 96, &out[b].new_leaf));
        CHECK(imt_host_alloc(ctx, (size_t)DEPTH * batch * 32, &out[b].low_sib));       /* dimensioned for the GLOBAL depth */
        CHECK(imt_host_alloc(ctx, (size_t)DEPTH * batch * 32, &out[b].new_sib));
    }
    make_values(vals, total, world, rank);
    /* before the first step every subtree is empty: its root is the empty subtree of its height */
    CHECK(imt_zero_hashes(ctx, DEPTH, zero, IMT_FMT_CANONICAL));
    for (int g = 0; g < world; g++) memcpy(roots[0] + (size_t)g * 32, zero + (size_t)sub_height * 32, 32);
    static struct step_report rep[MAXSTEPS];
    int cur = 0;                                 /* roots[cur] = every subtree's root BEFORE the step being finished */
    unsigned long long checked = 0;
    /* finish step s (exchange + lift + check); lag = 1 while a younger batch is in flight, 0 at the end */
    for (int s = 0; s <= steps; s++) {
        if (s < steps) {
            first_new[s & 1] = ((uint64_t)rank << sub_height) + imt_itree_size(tree);
            CHECK(imt_itree_insert_batch(tree, vals[(size_t)s * batch], batch, &out[s & 1], IMT_DEVICE_PTRS));
        }
        if (s == 0) continue;
        const int f = s - 1, b = f & 1;
        CHECK(imt_itree_root_lagged(tree, s < steps ? 1 : 0, mine, IMT_DEVICE_PTRS));
        /* THE collective of the layout: every rank's subtree root after step f */
        CHECK(imt_transport_all_gather(tp, mine, roots[cur ^ 1], 32, NULL));
        CHECK(imt_itree_lift_batch(tree, roots[cur], roots[cur ^ 1], (size_t)world, batch, &out[b], IMT_DEVICE_PTRS));
        for (size_t i = 0; i < batch; i++) new_index[i] = first_new[b] + i;
        CHECK(imt_insert_witness_batch(ctx, out[b].old_root, out[b].low_leaf, out[b].low_index, out[b].low_sib, out[b].new_root,
                                       out[b].new_leaf, new_index, NULL, out[b].new_sib, out[b].is_largest, DEPTH, batch, fail, NULL,
                                       IMT_DEVICE_PTRS | IMT_ROOT_PER_ITEM));
        CHECK(imt_ctx_sync(ctx));
        CHECK(imt_transport_poll_error(tp));     /* a peer that never came: the gathered roots were not written */
        for (size_t i = 0; i < batch; i++)
            if (fail[i]) { fprintf(stderr, "[rank %d] step %d insertion %zu: insert_leaf constraints 0x%02x fail at depth 32\n", rank, f, i, fail[i]); return 20; }
        for (size_t i = 1; i < batch; i++)       /* inside the rank's share the roots chain too */
            if (memcmp((unsigned char *)out[b].old_root + i * 32, (unsigned char *)out[b].new_root + (i - 1) * 32, 32)) {
                fprintf(stderr, "[rank %d] step %d: root chain broken at insertion %zu\n", rank, f, i);
                return 21;
            }
        checked += batch;
        memcpy(rep[f].first_old, out[b].old_root, 32);
        memcpy(rep[f].last_new, (unsigned char *)out[b].new_root + (batch - 1) * 32, 32);
        cur ^= 1;
    }
    /* the tree's root from the last exchange: must be the last rank's last new root */
    unsigned char global_root[32];
    CHECK(imt_combine_subtree_roots(ctx, roots[cur], (size_t)world, sub_height, DEPTH, global_root, IMT_FMT_CANONICAL));
    if (write_all(to_parent, rep, sizeof(struct step_report) * (size_t)steps) || write_all(to_parent, global_root, 32)) return 3;
    if (rank == 0) {
        printf("rank 0: %llu witnesses lifted to depth %d and checked against insert_leaf's constraints; global root ", checked, DEPTH);
        for (int j = 31; j >= 0; j--) printf("%02x", global_root[j]);
        printf("\n");
        fflush(stdout);
    }
    char go;                                     /* nobody frees what it exports while a peer may still read it */
    if (read_all(from_parent, &go, 1)) return 3;
    CHECK(imt_transport_destroy(tp));
    imt_itree_free(tree);
    imt_ctx_destroy(ctx);
    return 0;
}

int main(int argc, char **argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 2;
    const int use_rccl = argc > 2 && !strcmp(argv[2], "rccl");
    const int steps = argc > 3 ? atoi(argv[3]) : 5;
    const size_t batch = argc > 4 ? (size_t)atol(argv[4]) : 192;
    if (world < 1 || world > MAXW || (world & (world - 1)) || steps < 1 || steps > MAXSTEPS || batch < 1 || (!use_rccl && world < 2)) {
        fprintf(stderr, "usage: %s WORLD(power of two <= %d; ipc: >= 2) [ipc|rccl] [STEPS <= %d] [BATCH]\n", argv[0], MAXW, MAXSTEPS);
        return 1;
    }
    signal(SIGPIPE, SIG_IGN);
    int up[MAXW][2], down[MAXW][2];
    pid_t pid[MAXW];
    for (int r = 0; r < world; r++) {
        if (pipe(up[r]) || pipe(down[r])) return 1;
        pid[r] = fork();                         /* before this process has made any GPU call */
        if (pid[r] < 0) return 1;
        if (pid[r] == 0) {
            for (int q = 0; q <= r; q++) { close(up[q][0]); close(down[q][1]); }
            _exit(child(r, world, use_rccl, steps, batch, up[r][1], down[r][0]));
        }
        close(up[r][1]);
        close(down[r][0]);
    }
    int bad = 0;
    if (use_rccl) {
        unsigned char id[IMT_RCCL_UNIQUE_ID_BYTES];
        bad |= read_all(up[0][0], id, sizeof id);
        for (int r = 0; r < world && !bad; r++) bad |= write_all(down[r][1], id, sizeof id);
    } else {
        const size_t nb = imt_transport_ipc_blob_bytes();      /* arithmetic only: no GPU call in the parent */
        unsigned char *all = malloc(nb * (size_t)world);
        for (int r = 0; r < world && !bad; r++) bad |= read_all(up[r][0], all + nb * (size_t)r, nb);
        for (int r = 0; r < world && !bad; r++) bad |= write_all(down[r][1], all, nb * (size_t)world);
        free(all);
    }
    static struct step_report rep[MAXW][MAXSTEPS];
    unsigned char root[MAXW][32];
    for (int r = 0; r < world && !bad; r++)
        bad |= read_all(up[r][0], rep[r], sizeof(struct step_report) * (size_t)steps) || read_all(up[r][0], root[r], 32);
    /* one continuous sequence of depth-32 roots: ..., rank g's last new root = rank g + 1's first old root, the last
     * rank's = the next step's first; every rank computed the same global root, the sequence's last element */
    int chain = !bad;
    for (int s = 0; s < steps && chain; s++)
        for (int r = 0; r < world && chain; r++) {
            const unsigned char *next = r + 1 < world ? rep[r + 1][s].first_old : (s + 1 < steps ? rep[0][s + 1].first_old : root[0]);
            chain &= memcmp(rep[r][s].last_new, next, 32) == 0;
        }
    for (int r = 1; r < world && chain; r++) chain &= memcmp(root[r], root[0], 32) == 0;
    for (int r = 0; r < world; r++) (void)!write(down[r][1], "g", 1);
    int status = 0;
    for (int r = 0; r < world; r++) {
        int st = 0;
        waitpid(pid[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st)) status = 1;
    }
    printf("%d processes over %s, %d steps of %d x %zu insertions into %d placed subtrees: %s\n", world, use_rccl ? "RCCL" : "IPC", steps,
           world, batch, world, chain && !status ? "every witness satisfies insert_leaf at depth 32, the roots form one chain" : "FAILED");
    return chain && !status ? 0 : 3;
}
