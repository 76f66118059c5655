/* How fast does one process put N GB into a new file on /dev/shm?  (s5view's ordered write phase is one write() per chunk: 4-5 GB/s.)
 *   write      one thread, write() of 32 MB chunks
 *   pwriteT    T threads, pwrite() of disjoint parts of each chunk (the inode lock serialises them?)
 *   mmapT      ftruncate + mmap of the chunk's range, T threads memcpy into it
 *   fallocT    fallocate of the chunk's range first, then as mmapT
 * gcc -O2 -pthread tools/hw_probe/shm_write_probe.c -o /tmp/shm_write_probe && /tmp/shm_write_probe /dev/shm/probe.bin 4 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
typedef struct { int fd; const char *src; char *dst; size_t len; off_t off; int mode; } job_t;
static void *job(void *a) {
    job_t *j = (job_t *)a;
    if (j->mode == 0) { size_t d = 0; while (d < j->len) { ssize_t w = pwrite(j->fd, j->src + d, j->len - d, j->off + (off_t)d); if (w <= 0) break; d += (size_t)w; } }
    else memcpy(j->dst, j->src, j->len);
    return NULL;
}
int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "/dev/shm/probe.bin";
    const size_t gb = argc > 2 ? (size_t)atoi(argv[2]) : 4, chunk = 32u << 20, n = gb * (1u << 30) / chunk;
    char *src = (char *)malloc(chunk);
    for (size_t i = 0; i < chunk; i++) src[i] = (char)(i * 131 + (i >> 7));
    for (int mode = 0; mode < 8; mode++) {
        const int T = mode == 0 ? 1 : (mode & 1) ? 4 : 8, kind = mode == 0 ? 0 : mode <= 2 ? 1 : mode <= 4 ? 2 : mode <= 6 ? 3 : 4;
        if (kind == 4) break;
        unlink(path);
        int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
        const double t0 = now();
        for (size_t c = 0; c < n; c++) {
            const off_t fo = (off_t)(c * chunk) + 77;        /* (not page aligned, as a record stream behind a header is not) */
            if (kind == 0) { size_t d = 0; while (d < chunk) { ssize_t w = write(fd, src + d, chunk - d); if (w <= 0) return 1; d += (size_t)w; } continue; }
            char *map = NULL;
            off_t base = fo & ~(off_t)4095;
            if (kind >= 2) {
                if (kind == 3 && fallocate(fd, 0, fo, (off_t)chunk) != 0) { perror("fallocate"); return 1; }
                if (kind == 2 && ftruncate(fd, fo + (off_t)chunk) != 0) return 1;
                map = (char *)mmap(NULL, (size_t)(fo - base) + chunk, PROT_READ | PROT_WRITE, MAP_SHARED, fd, base);
                if (map == MAP_FAILED) { perror("mmap"); return 1; }
            }
            pthread_t th[8]; job_t jb[8];
            const size_t part = ((chunk / T) + 4095) & ~(size_t)4095;
            int used = 0;
            for (int t = 0; t < T; t++) {
                const size_t lo = (size_t)t * part; if (lo >= chunk) break;
                jb[t] = (job_t){fd, src + lo, map ? map + (fo - base) + lo : NULL, chunk - lo < part ? chunk - lo : part, fo + (off_t)lo, kind == 1 ? 0 : 1};
                used++;
            }
            for (int t = 1; t < used; t++) pthread_create(&th[t], NULL, job, &jb[t]);
            job(&jb[0]);
            for (int t = 1; t < used; t++) pthread_join(th[t], NULL);
            if (map) munmap(map, (size_t)(fo - base) + chunk);
        }
        close(fd);
        const double dt = now() - t0;
        printf("%-8s T=%d: %zu GB in %.3f s = %.2f GB/s\n", kind == 0 ? "write" : kind == 1 ? "pwrite" : kind == 2 ? "mmap" : "falloc+mmap", T, gb, dt, (double)gb * 1.073741824 / dt);
        fflush(stdout);
    }
    unlink(path);
    return 0;
}
