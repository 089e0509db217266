// sink_probe.cpp -- how fast can 1.25 GB of IQ get from a (pinned-like) host buffer into a file?  Variants:
//   write   : one thread, write() in 8 MB pieces                     (what the round-2 CLI did through fwrite)
//   pwrite  : T threads, pwrite() on disjoint ranges of the same file (buffered writes take the inode lock)
//   mmap    : ftruncate + mmap(MAP_SHARED), T threads memcpy disjoint ranges (page faults run in parallel)
//   falloc  : fallocate() of the whole file alone (page allocation without the copy), then
//   fa+pwr  : fallocate, then T threads pwrite() into the allocated pages
//   fa+map  : fallocate, then mmap + T threads memcpy (minor faults only)
//   fa|map  : a thread fallocates 64 MB ahead while T threads memcpy through the mapping behind it (pipelined, as a sink
//             would do it: the file size is known from -d before the first sample exists)
// usage: sink_probe <file> [MB=1189] [threads...]
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    const char *path = argv[1];
    const size_t bytes = (size_t)(argc > 2 ? atol(argv[2]) : 1189) << 20;
    std::vector<int> ts;
    for (int i = 3; i < argc; ++i) ts.push_back(atoi(argv[i]));
    if (ts.empty()) ts = {1, 2, 4, 8, 16};
    char *src = (char *)aligned_alloc(4096, bytes);
    for (size_t i = 0; i < bytes; i += 8) *(size_t *)(src + i) = i * 0x9E3779B97F4A7C15ull;
    auto run = [&](const char *name, int T, auto fn) {
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); exit(1); }
        const double t0 = now();
        fn(fd, T);
        close(fd);
        const double dt = now() - t0;
        printf("%-8s T=%2d  %7.1f ms  %6.2f GB/s\n", name, T, dt * 1e3, bytes / dt / 1e9);
        fflush(stdout);
    };
    run("write", 1, [&](int fd, int) {
        for (size_t o = 0; o < bytes;) {
            ssize_t n = write(fd, src + o, std::min(bytes - o, (size_t)8 << 20));
            if (n <= 0) { perror("write"); exit(1); }
            o += n;
        }
    });
    for (int T : ts) {
        run("pwrite", T, [&](int fd, int T) {
            if (ftruncate(fd, bytes)) perror("ftruncate");
            std::vector<std::thread> th;
            const size_t piece = (size_t)4 << 20;
            const size_t np = (bytes + piece - 1) / piece;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    for (size_t p = t; p < np; p += T) {
                        const size_t o = p * piece, n = std::min(piece, bytes - o);
                        if (pwrite(fd, src + o, n, o) != (ssize_t)n) { perror("pwrite"); exit(1); }
                    }
                });
            for (auto &x : th) x.join();
        });
        run("mmap", T, [&](int fd, int T) {
            if (ftruncate(fd, bytes)) perror("ftruncate");
            char *dst = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (dst == MAP_FAILED) { perror("mmap"); exit(1); }
            std::vector<std::thread> th;
            const size_t piece = (size_t)4 << 20;
            const size_t np = (bytes + piece - 1) / piece;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    for (size_t p = t; p < np; p += T) {
                        const size_t o = p * piece, n = std::min(piece, bytes - o);
                        memcpy(dst + o, src + o, n);
                    }
                });
            for (auto &x : th) x.join();
            munmap(dst, bytes);
        });
    }
    run("falloc", 1, [&](int fd, int) {
        if (fallocate(fd, 0, 0, bytes)) perror("fallocate");
    });
    for (int T : ts) {
        run("fa+pwr", T, [&](int fd, int T) {
            const double t0 = now();
            if (fallocate(fd, 0, 0, bytes)) perror("fallocate");
            const double t1 = now();
            std::vector<std::thread> th;
            const size_t piece = (size_t)4 << 20;
            const size_t np = (bytes + piece - 1) / piece;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    for (size_t p = t; p < np; p += T) {
                        const size_t o = p * piece, n = std::min(piece, bytes - o);
                        if (pwrite(fd, src + o, n, o) != (ssize_t)n) { perror("pwrite"); exit(1); }
                    }
                });
            for (auto &x : th) x.join();
            printf("   (fallocate %.1f ms, copy %.1f ms) ", (t1 - t0) * 1e3, (now() - t1) * 1e3);
        });
        run("fa+map", T, [&](int fd, int T) {
            const double t0 = now();
            if (fallocate(fd, 0, 0, bytes)) perror("fallocate");
            const double t1 = now();
            char *dst = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (dst == MAP_FAILED) { perror("mmap"); exit(1); }
            std::vector<std::thread> th;
            const size_t piece = (size_t)4 << 20;
            const size_t np = (bytes + piece - 1) / piece;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    for (size_t p = t; p < np; p += T) {
                        const size_t o = p * piece, n = std::min(piece, bytes - o);
                        memcpy(dst + o, src + o, n);
                    }
                });
            for (auto &x : th) x.join();
            const double t2 = now();
            munmap(dst, bytes);
            printf("   (fallocate %.1f ms, copy %.1f ms, munmap %.1f ms) ", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (now() - t2) * 1e3);
        });
        run("fa|map", T, [&](int fd, int T) {
            if (ftruncate(fd, bytes)) perror("ftruncate");
            char *dst = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (dst == MAP_FAILED) { perror("mmap"); exit(1); }
            const size_t piece = (size_t)4 << 20, ahead = (size_t)64 << 20;
            const size_t np = (bytes + piece - 1) / piece;
            std::atomic<size_t> allocated{0}, next{0};
            std::thread fa([&] {
                for (size_t o = 0; o < bytes; o += ahead) {
                    const size_t n = std::min(ahead, bytes - o);
                    if (fallocate(fd, 0, o, n)) perror("fallocate");
                    allocated.store(o + n, std::memory_order_release);
                }
            });
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&] {
                    for (;;) {
                        const size_t p = next.fetch_add(1);
                        if (p >= np) break;
                        const size_t o = p * piece, n = std::min(piece, bytes - o);
                        while (allocated.load(std::memory_order_acquire) < o + n) std::this_thread::yield();
                        memcpy(dst + o, src + o, n);
                    }
                });
            fa.join();
            for (auto &x : th) x.join();
            munmap(dst, bytes);
        });
    }
    unlink(path);
    return 0;
}
