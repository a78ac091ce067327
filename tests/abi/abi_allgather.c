/*
 * abi_allgather.c -- the multi-rank half of the drop-in boundary, torch-free: what the Go driver of BASELINE
 * configs[2] does through cgo, written as a plain C host.  Links libpolyhip.so and the HIP runtime only.
 *
 *     abi_allgather NRANKS
 *
 * The parent forks NRANKS children BEFORE anything touches HIP; child r binds GPU r (mod the visible devices), child 0
 * asks for the communicator id (polyhip_comm_unique_id) and sends it up a pipe, the parent relays it down to the other
 * children -- the rendezvous is the host's business, include/polyhip.h R1 -- and every child runs
 *   polyhip_comm_init_rank -> polyhip_allgather_sketches_dev                     (every rank's shard, rank order)
 *   polyhip_mash_index_build_part_dev(part = rank) -> polyhip_mash_index_allgather_dev -> ..._shared_counts_reuse_dev
 * and checks: the gathered set equals what every rank is known to hold (shards are a pure function of the rank), the
 * row block from the assembled index equals the one from an index built locally in one shot, and sampled cells equal
 * a merge count written out here after search/mash/mash.go:107-135.
 * RCCL refuses two ranks on one device, so NRANKS > visible GPUs exits 77 (skipped).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include "polyhip.h"

#define NFAM 10
#define COPIES 12
#define NLOCAL (NFAM * COPIES)
#define S 192

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int _rc = (call);                                                                        \
        if (_rc != 0) {                                                                          \
            fprintf(stderr, "rank %d: %s -> %d (%s)\n", g_rank, #call, _rc, polyhip_last_error()); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)
#define HIPCHECK(call)                                                                  \
    do {                                                                                \
        hipError_t _e = (call);                                                         \
        if (_e != hipSuccess) {                                                         \
            fprintf(stderr, "rank %d: %s -> %s\n", g_rank, #call, hipGetErrorString(_e)); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static int g_rank = -1;

static uint64_t splitmix(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int cmp_u32(const void *a, const void *b)
{
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* shard of `rank`: families of related ascending sketches (a pure function of the rank) */
static void shard(int rank, uint32_t *out)
{
    uint64_t st = 0xABCD0000ull + (uint64_t)rank;
    for (int f = 0; f < NFAM; ++f) {
        uint32_t base[S];
        for (int e = 0; e < S; ++e)
            base[e] = (uint32_t)(splitmix(&st) >> 37);
        for (int c = 0; c < COPIES; ++c) {
            uint32_t *row = out + ((size_t)f * COPIES + c) * S;
            for (int e = 0; e < S; ++e)
                row[e] = (splitmix(&st) % 100) < 15 ? (uint32_t)(splitmix(&st) >> 37) : base[e];
            qsort(row, S, sizeof(uint32_t), cmp_u32);
        }
    }
}

/* sameHashes of two ascending sketches of one size: the two-pointer merge of mash.go:117-132 */
static uint32_t merge_count(const uint32_t *a, const uint32_t *b, uint32_t s)
{
    if (a[s - 1] < b[0] || b[s - 1] < a[0])
        return 0;
    uint32_t same = 0, i = 0, j = 0;
    while (i < s && j < s) {
        if (a[i] == b[j]) {
            ++same;
            ++i;
            ++j;
        } else if (b[j] < a[i]) {
            ++j;
        } else {
            ++i;
        }
    }
    return same;
}

static int child(int rank, int nranks, int up_fd, int down_fd)
{
    g_rank = rank;
    const int ndev = polyhip_device_count();
    if (ndev < 1) {
        fprintf(stderr, "rank %d: no HIP device (%s)\n", rank, polyhip_last_error());
        return 1;
    }
    CHECK(polyhip_set_device(rank % ndev));
    uint8_t id[128];
    if (rank == 0) {
        CHECK(polyhip_comm_unique_id(id));
        if (write(up_fd, id, 128) != 128)
            return 1;
    }
    if (read(down_fd, id, 128) != 128) {
        fprintf(stderr, "rank %d: no communicator id from the parent\n", rank);
        return 1;
    }
    polyhip_comm *comm = NULL;
    CHECK(polyhip_comm_init_rank(id, rank, nranks, &comm));
    if (polyhip_comm_rank(comm) != rank || polyhip_comm_size(comm) != nranks)
        return 1;

    const size_t N = (size_t)nranks * NLOCAL;
    uint32_t *h_all = (uint32_t *)malloc(N * S * 4), *h_got = (uint32_t *)malloc(N * S * 4);
    for (int r = 0; r < nranks; ++r)
        shard(r, h_all + (size_t)r * NLOCAL * S);
    uint32_t *d_local, *d_all;
    HIPCHECK(hipMalloc((void **)&d_local, (size_t)NLOCAL * S * 4));
    HIPCHECK(hipMalloc((void **)&d_all, N * S * 4));
    HIPCHECK(hipMemset(d_all, 0, N * S * 4));
    HIPCHECK(hipMemcpy(d_local, h_all + (size_t)rank * NLOCAL * S, (size_t)NLOCAL * S * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));
    CHECK(polyhip_allgather_sketches_dev(comm, d_local, NLOCAL, S, d_all, st));
    HIPCHECK(hipStreamSynchronize(st));
    HIPCHECK(hipMemcpy(h_got, d_all, N * S * 4, hipMemcpyDeviceToHost));
    if (memcmp(h_got, h_all, N * S * 4) != 0) {
        fprintf(stderr, "rank %d: gathered sketches differ from the shards\n", rank);
        return 1;
    }
    /* the index in parts, exchanged, then this rank's row block */
    const size_t wb = polyhip_mash_shared_counts_workspace_bytes(NLOCAL, S, N, S);
    void *d_work, *d_work1;
    uint16_t *d_counts, *d_counts1;
    HIPCHECK(hipMalloc(&d_work, wb));
    HIPCHECK(hipMalloc(&d_work1, wb));
    HIPCHECK(hipMalloc((void **)&d_counts, (size_t)NLOCAL * N * 2));
    HIPCHECK(hipMalloc((void **)&d_counts1, (size_t)NLOCAL * N * 2));
    CHECK(polyhip_mash_index_build_part_dev(d_all, N, S, (uint32_t)rank, (uint32_t)nranks, d_work, wb, st));
    CHECK(polyhip_mash_index_allgather_dev(comm, N, S, d_work, wb, st));
    const uint32_t *d_X = d_all + (size_t)rank * NLOCAL * S;
    CHECK(polyhip_mash_shared_counts_reuse_dev(d_X, NLOCAL, S, d_all, N, S, d_counts, N, d_work, wb, st));
    CHECK(polyhip_mash_shared_counts_dev(d_X, NLOCAL, S, d_all, N, S, d_counts1, N, d_work1, wb, st));
    HIPCHECK(hipStreamSynchronize(st));
    uint16_t *h_c = (uint16_t *)malloc((size_t)NLOCAL * N * 2), *h_c1 = (uint16_t *)malloc((size_t)NLOCAL * N * 2);
    HIPCHECK(hipMemcpy(h_c, d_counts, (size_t)NLOCAL * N * 2, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(h_c1, d_counts1, (size_t)NLOCAL * N * 2, hipMemcpyDeviceToHost));
    if (memcmp(h_c, h_c1, (size_t)NLOCAL * N * 2) != 0) {
        fprintf(stderr, "rank %d: row block from the assembled index differs from the locally built one\n", rank);
        return 1;
    }
    for (size_t i = 0; i < NLOCAL; i += 7)
        for (size_t j = 0; j < N; ++j) {
            const uint32_t want = merge_count(h_all + ((size_t)rank * NLOCAL + i) * S, h_all + j * S, S);
            if (h_c[i * N + j] != want) {
                fprintf(stderr, "rank %d: counts[%zu][%zu] = %u, merge says %u\n", rank, i, j, h_c[i * N + j], want);
                return 1;
            }
        }
    CHECK(polyhip_comm_destroy(comm));
    return 0;
}

int main(int argc, char **argv)
{
    const int nranks = argc > 1 ? atoi(argv[1]) : 1;
    if (nranks < 1 || nranks > 64)
        return 2;
    int up[2], down[64][2];
    if (pipe(up) != 0)
        return 2;
    pid_t pids[64];
    for (int r = 0; r < nranks; ++r) {
        if (pipe(down[r]) != 0)
            return 2;
        pids[r] = fork();
        if (pids[r] == 0) { /* child: HIP is first touched here */
            /* keep only my own ends: a sibling's copy of a write end would hide an EOF */
            close(up[0]);
            if (r != 0)
                close(up[1]);
            for (int q = 0; q <= r; ++q)
                close(down[q][1]);
            const int ndev = polyhip_device_count();
            if (nranks > ndev && nranks > 1) {
                if (r == 0)
                    fprintf(stderr, "abi_allgather: %d ranks but %d GPU(s): RCCL needs one device per rank\n", nranks, ndev);
                _exit(77);
            }
            _exit(child(r, nranks, up[1], down[r][0]));
        }
    }
    close(up[1]);
    uint8_t id[128];
    int have = read(up[0], id, 128) == 128;
    for (int r = 0; r < nranks; ++r) {
        if (have && write(down[r][1], id, 128) != 128)
            have = 0;
        close(down[r][1]); /* a child that waits for an id that never comes sees EOF */
    }
    int bad = 0, skipped = 0;
    for (int r = 0; r < nranks; ++r) {
        int stt = 0;
        waitpid(pids[r], &stt, 0);
        if (WIFEXITED(stt) && WEXITSTATUS(stt) == 77)
            skipped = 1;
        else if (!WIFEXITED(stt) || WEXITSTATUS(stt) != 0)
            bad = 1;
    }
    if (skipped && !bad) {
        printf("abi_allgather skipped\n");
        return 77;
    }
    if (bad) {
        printf("abi_allgather FAILED\n");
        return 1;
    }
    printf("abi_allgather ok: %d rank(s): all-gather, index parts + ragged all-gather, row blocks = local index = merge\n", nranks);
    return 0;
}
