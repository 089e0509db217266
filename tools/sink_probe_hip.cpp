// sink_probe_hip.cpp -- can the device->host copy land in the page cache directly?  ftruncate + mmap(MAP_SHARED) of the
// output file, hipHostRegister of the mapping (whole, or window by window), hipMemcpyAsync D2H into it, unregister.
// usage: sink_probe_hip <file> [MB=1189] [window_MB=0 (whole)] [prefault_threads=0]
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    const char *path = argv[1];
    const size_t bytes = (size_t)(argc > 2 ? atol(argv[2]) : 1189) << 20;
    size_t win = (size_t)(argc > 3 ? atol(argv[3]) : 0) << 20;
    const int pf = argc > 4 ? atoi(argv[4]) : 0;
    if (win == 0) win = bytes;
    char *dev;
    CK(hipMalloc((void **)&dev, bytes));
    CK(hipMemset(dev, 0x5a, bytes));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    // reference: pinned buffer
    char *pin;
    CK(hipHostMalloc((void **)&pin, bytes, hipHostMallocDefault));
    CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    double t0 = now();
    CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    printf("pinned D2H          %7.1f ms  %6.2f GB/s\n", (now() - t0) * 1e3, bytes / (now() - t0) / 1e9);
    for (int rep = 0; rep < 2; ++rep) {
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); return 1; }
        const double ta = now();
        if (ftruncate(fd, bytes)) perror("ftruncate");
        char *dst = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (dst == MAP_FAILED) { perror("mmap"); return 1; }
        double t_pf = 0, t_reg = 0, t_cp = 0, t_un = 0;
        if (pf > 0) {
            const double t = now();
            std::vector<std::thread> th;
            const size_t piece = (bytes / pf + 4095) & ~(size_t)4095;
            for (int k = 0; k < pf; ++k)
                th.emplace_back([&, k] {
                    const size_t o = k * piece;
                    if (o >= bytes) return;
                    const size_t n = std::min(piece, bytes - o);
#ifdef MADV_POPULATE_WRITE
                    if (madvise(dst + o, n, MADV_POPULATE_WRITE) == 0) return;
#endif
                    for (size_t i = 0; i < n; i += 4096) dst[o + i] = 0;
                });
            for (auto &x : th) x.join();
            t_pf = now() - t;
        }
        for (size_t o = 0; o < bytes; o += win) {
            const size_t n = std::min(win, bytes - o);
            double t = now();
            hipError_t e = hipHostRegister(dst + o, n, hipHostRegisterDefault);
            if (e != hipSuccess) { printf("hipHostRegister failed: %s\n", hipGetErrorString(e)); return 1; }
            t_reg += now() - t;
            t = now();
            CK(hipMemcpyAsync(dst + o, dev + o, n, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            t_cp += now() - t;
            t = now();
            CK(hipHostUnregister(dst + o));
            t_un += now() - t;
        }
        munmap(dst, bytes);
        close(fd);
        const double dt = now() - ta;
        printf("mmap+register win %4zu MB pf %2d: total %7.1f ms %6.2f GB/s  (prefault %.1f, register %.1f, copy %.1f, unregister %.1f ms)\n",
               win >> 20, pf, dt * 1e3, bytes / dt / 1e9, t_pf * 1e3, t_reg * 1e3, t_cp * 1e3, t_un * 1e3);
        // verify
        fd = open(path, O_RDONLY);
        unsigned char buf[4096];
        if (pread(fd, buf, 4096, bytes - 4096) != 4096 || buf[100] != 0x5a) printf("VERIFY FAILED\n");
        close(fd);
    }
    unlink(path);
    return 0;
}
