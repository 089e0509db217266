// galileo-sdr-sim -- command-line front door with the reference's option surface and file format, so
// that reference command lines (README.md:63) run unchanged against the MI355X engine:
//
//   galileo-sdr-sim -e <rinex> [-l lat,lon,hgt | -u motion.csv] [-t YYYY/MM/DD,hh:mm:ss] [-d sec]
//                   [-o file|-] [-I x] [-U x] [-b x] [-v]
//
// Options follow src/main.cpp:216-326 (getopt string "e:n:o:u:g:l:T:t:d:G:a:p:iI:U:b:v"); USRP / bit
// streamer / UDP options are accepted and ignored (file sink only; -U/-b are therefore implied).  Output is
// the reference's `ishort` stream: headerless little-endian interleaved int16 I,Q at 2.6 MS/s, exactly
// ((int)(10 d + 0.5) - 1) * 260000 * 4 bytes (src/galileo-sdr.cpp:438,536-542); default name
// galileosim.ishort, "-" = stdout.  Errors print a message and exit(1); success exits 0.
//
// Pipeline: a producer thread runs the host front-end (libgalscen: orbits, ranges, I/NAV pages) up to two batches
// ahead -> the main thread plans and executes each batch on the GPU and, after gal_synth_finish(), enqueues the copy
// into one of two pinned buffers on two copy streams -> the sink moves full buffers into the output with sequential
// write()s.  (--writers n > 0: a regular file is sized up front and mapped instead, and n threads copy disjoint pieces of
// the pinned buffer into the mapping -- built to get past one core's copy speed, measured no faster once the unmap is
// counted: the kernel's page-cache insertion for ONE file does not scale with threads; kept as an option.)  In steady
// state the run time is that of the slowest stage: the device->host link for /dev/null, the page cache for a file.
//
// --sites <file>: BASELINE config 5 as a product entry point -- one line `lat,lon,hgt[,outfile]` per receiver site, one
// child process of this executable per site, spread over the GPUs of the node (GAL_DEVICE), every site's ishort file
// written on its own; the parent prints the aggregate.  This is the reference's way of doing it too: N independent
// usrp_galileo processes (src/main.cpp:168-408), nothing is exchanged between sites.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <getopt.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/galscen.h"
#include "../../include/galsynth.h"

namespace {

std::atomic<bool> g_stop{false};
void on_sigint(int) { g_stop = true; }

void usage(const char *prog)
{
    printf("Usage: %s [options]\n"
           "Options:\n"
           "  -e <Ephemeris>   RINEX navigation file for Galileo ephemerides (required)\n"
           "  -o <File sink>   File to store IQ samples (default galileosim.ishort, - = stdout)\n"
           "  -l <location>    Lat,Lon,Hgt (static mode) e.g. 35.274,137.014,100\n"
           "  -u <user_motion> ECEF user motion file t,x,y,z at 10 Hz (dynamic mode)\n"
           "  -t <date,time>   Scenario start time YYYY/MM/DD,hh:mm:ss\n"
           "  -d <duration>    Duration [sec]\n"
           "  -I <x>           Disable ionospheric delay\n"
           "  -T <date,time>   As the reference runs it: -t without its range check, UTC reference time overwritten (use `now` for\n"
           "                   the current time).  The same bytes as the reference program for the same command line: its loop that\n"
           "                   would shift TOC / TOE never runs, so a start outside the file's span gives an empty sky (warned about)\n"
           "  --shift-toe      with -T: what the option sets out to do -- TOC and TOE of every record shifted to the start time, the\n"
           "                   navigation file becomes valid at any start (include/galscen.h: time_overwrite 2; not the reference's bytes)\n"
           "  --ref-T          with -T: the explicit spelling of the default (kept for round 5's command lines)\n"
           "  -P <port>        UDP port for run-time position updates lat,lon,hgt as 3 doubles (0 = off; default: 7533 on\n"
           "                   the loopback interface; a port given here is bound on all interfaces, as the reference's)\n"
           "  -r               Pace the output to real time (one 0.1 s epoch per 0.1 s)\n"
           "  -C               CBOC(6,1,1/11) sub-carrier of the E1 OS ICD instead of the reference's BOC(1,1) (opt-in)\n"
           "  --exact-replay   Synthesise with the exact-replay kernel only (every sample of both NCO recurrences stepped; the same bytes, slower)\n"
           "  --strict         Stop with an error where a satellite in view runs out of ephemeris (default: its channel\n"
           "                   keeps the last valid record; the reference indexes out of bounds there)\n"
           "  --sites <file>   One line lat,lon,hgt[,outfile] per receiver site: one process per site over the GPUs of the\n"
           "                   node (--gpus N, default all; --per-gpu K processes per GPU, default 1); -o is the name stem;\n"
           "                   -P <port>: site k listens on port + k (default: no position listener)\n"
           "  --writers <n>    Threads that move finished batches into a regular output file (default 0: sequential write(); > 0: mapped file, n copy threads)\n"
           "  -v               Verbose\n"
           "  -U/-b/-a/-G/-p/-n/-g/-i     accepted for compatibility (file sink only)\n",
           prog);
}

struct Slot {  // one pinned host buffer of the double-buffered sink
    int16_t *host = nullptr;
    size_t bytes = 0;
    hipEvent_t copied[2] = {nullptr, nullptr};  // the batch leaves the GPU in two halves on two copy streams
    bool full = false;
};

// ---- the output ------------------------------------------------------------------------------------------------
// Regular file: ftruncate to the final size + one shared mapping; put() has `n_workers` threads copy disjoint pieces.
// Anything else (stdout, pipe, device): sequential write().
class Sink {
public:
    ~Sink()
    {
        close_workers();
        if (prealloc_.joinable()) {
            stop_prealloc_ = true;
            prealloc_.join();
        }
    }

    bool open(const char *path, size_t total_bytes, int n_workers)
    {
        total_ = total_bytes;
        if (strcmp(path, "-") == 0) {
            fd_ = STDOUT_FILENO;
            own_fd_ = false;
        } else {
            fd_ = ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
            if (fd_ < 0) fd_ = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);  // write-only sinks (devices, FIFOs)
            if (fd_ < 0) return false;
            own_fd_ = true;
        }
        struct stat sb;
        const char *force = getenv("GAL_SINK");  // "stream": never map (A/B measurements)
        if (own_fd_ && fstat(fd_, &sb) == 0 && S_ISREG(sb.st_mode) && total_ > 0 && n_workers > 0 &&
            !(force && strcmp(force, "stream") == 0) && ftruncate(fd_, (off_t)total_) == 0) {
            void *m = mmap(nullptr, total_, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);
            if (m != MAP_FAILED) {
                map_ = (char *)m;
                for (int i = 0; i < n_workers; ++i) workers_.emplace_back([this, i] { worker(i); });
            } else if (ftruncate(fd_, 0) != 0) {
                return false;
            }
        }
        // Sequential sink into a regular file of known size: a helper thread allocates the file's pages ahead of the writes
        // (fallocate, size kept: the file grows as it is written), 64 MB at a time, WHILE the device starts up -- the run's
        // write()s then copy into pages that exist (tmpfs on the MI355X host: 1.19 GB allocate 70 ms, write() into allocated
        // pages 130 ms, write() that allocates as it goes 190 ms; profiles/archive/r03j_sink_probe.log).  Failure is not an error:
        // the writes allocate for themselves then.  GAL_SINK=noprealloc turns it off.
        if (!map_ && own_fd_ && fstat(fd_, &sb) == 0 && S_ISREG(sb.st_mode) && total_ > 0 &&
            !(force && (strcmp(force, "noprealloc") == 0))) {
            prealloc_ = std::thread([this] {
                const size_t step = (size_t)64 << 20;
                for (size_t o = 0; o < total_ && !stop_prealloc_.load(std::memory_order_relaxed); o += step) {
                    const size_t n = total_ - o < step ? total_ - o : step;
                    if (fallocate(fd_, FALLOC_FL_KEEP_SIZE, (off_t)o, (off_t)n) != 0) break;
                }
            });
        }
        return true;
    }
    bool mapped() const { return map_ != nullptr; }
    int workers() const { return (int)workers_.size(); }

    // appends `bytes` from `src`; returns false on an I/O error
    bool put(const char *src, size_t bytes)
    {
        if (map_) {
            if (pos_ + bytes > total_) return false;
            {
                std::lock_guard<std::mutex> lk(mu_);
                job_src_ = src;
                job_dst_ = map_ + pos_;
                job_bytes_ = bytes;
                job_next_ = 0;
                job_left_ = (int)workers_.size();
                ++job_id_;
            }
            cv_.notify_all();
            std::unique_lock<std::mutex> lk(mu_);
            done_cv_.wait(lk, [&] { return job_left_ == 0; });
            pos_ += bytes;
            return true;
        }
        while (bytes) {
            const ssize_t n = ::write(fd_, src, bytes < ((size_t)1 << 30) ? bytes : ((size_t)1 << 30));
            if (n < 0) {
                if (errno == EINTR) continue;
                return false;
            }
            src += n;
            bytes -= (size_t)n;
            pos_ += (size_t)n;
        }
        return true;
    }

    // all data are in the file (mapped sink: in its page cache); a run that stopped early is cut to what was written
    bool finish()
    {
        close_workers();
        bool ok = true;
        if (prealloc_.joinable()) {
            stop_prealloc_ = true;
            prealloc_.join();
            // a run that stopped early (SIGINT, error): give back the pages allocated beyond what was written
            if (pos_ < total_ && ftruncate(fd_, (off_t)pos_) != 0) ok = false;
        }
        if (map_) {
            munmap(map_, total_);
            map_ = nullptr;
            if (pos_ < total_ && ftruncate(fd_, (off_t)pos_) != 0) ok = false;
        }
        if (own_fd_ && fd_ >= 0 && ::close(fd_) != 0) ok = false;
        fd_ = -1;
        return ok;
    }

private:
    static constexpr size_t kPiece = (size_t)2 << 20;  // 2 MiB: whole huge pages where the file system has them

    void worker(int)
    {
        unsigned long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return quit_ || job_id_ != seen; });
            if (quit_) return;
            seen = job_id_;
            for (;;) {
                const size_t o = job_next_;
                if (o >= job_bytes_) break;
                const size_t n = job_bytes_ - o < kPiece ? job_bytes_ - o : kPiece;
                job_next_ = o + n;
                lk.unlock();
                memcpy(job_dst_ + o, job_src_ + o, n);
                lk.lock();
            }
            if (--job_left_ == 0) done_cv_.notify_all();
        }
    }
    void close_workers()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
        workers_.clear();
    }

    int fd_ = -1;
    bool own_fd_ = false;
    std::thread prealloc_;
    std::atomic<bool> stop_prealloc_{false};
    char *map_ = nullptr;
    size_t total_ = 0, pos_ = 0;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    bool quit_ = false;
    unsigned long job_id_ = 0;
    const char *job_src_ = nullptr;
    char *job_dst_ = nullptr;
    size_t job_bytes_ = 0, job_next_ = 0;
    int job_left_ = 0;
};

// ---- --sites: one child process per receiver site ---------------------------------------------------------------
struct Site {
    std::string llh, out;
};

int run_sites(const char *self, const std::vector<std::string> &base_args, const char *sites_file, const char *out_stem,
              int n_gpus, int per_gpu, int udp_base)
{
    std::vector<Site> sites;
    FILE *fp = fopen(sites_file, "r");
    if (!fp) {
        fprintf(stderr, "ERROR: cannot read the site list %s\n", sites_file);
        return 1;
    }
    char line[4096];
    while (fgets(line, sizeof(line), fp)) {
        double a, b, c;
        int used = 0;
        if (line[0] == '#' || sscanf(line, " %lf , %lf , %lf%n", &a, &b, &c, &used) != 3) continue;
        Site s;
        char buf[128];
        snprintf(buf, sizeof(buf), "%.10g,%.10g,%.10g", a, b, c);
        s.llh = buf;
        const char *rest = line + used;
        while (*rest == ' ' || *rest == ',' || *rest == '\t') ++rest;
        std::string name(rest);
        while (!name.empty() && (name.back() == '\n' || name.back() == '\r' || name.back() == ' ')) name.pop_back();
        if (name.empty()) {
            snprintf(buf, sizeof(buf), ".site%zu.ishort", sites.size());
            std::string stem(out_stem[0] ? out_stem : "galileosim");
            const std::string ext = ".ishort";
            if (stem.size() > ext.size() && stem.compare(stem.size() - ext.size(), ext.size(), ext) == 0)
                stem.resize(stem.size() - ext.size());
            name = stem + buf;
        }
        s.out = name;
        sites.push_back(s);
    }
    fclose(fp);
    if (sites.empty()) {
        fprintf(stderr, "ERROR: no site (lat,lon,hgt) in %s\n", sites_file);
        return 1;
    }
    if (udp_base > 0 && (size_t)udp_base + sites.size() - 1 > 65535) {  // site k listens on port + k
        fprintf(stderr, "ERROR: -P %d with %zu sites runs past port 65535\n", udp_base, sites.size());
        return 1;
    }
    if (n_gpus <= 0) n_gpus = gal_synth_device_count();
    if (n_gpus <= 0) {
        fprintf(stderr, "ERROR: no usable gfx950 device (there is no CPU fallback)\n");
        return 1;
    }
    if (per_gpu < 1) per_gpu = 1;
    const int lanes = n_gpus * per_gpu;
    fprintf(stderr, "%zu sites over %d GPU%s (%d process%s per GPU)\n", sites.size(), n_gpus, n_gpus > 1 ? "s" : "", per_gpu,
            per_gpu > 1 ? "es" : "");
    std::vector<pid_t> lane_pid(lanes, 0);
    std::vector<int> lane_site(lanes, -1);
    std::vector<std::chrono::steady_clock::time_point> lane_t0(lanes);
    const auto t0 = std::chrono::steady_clock::now();
    size_t next = 0, done = 0;
    int failed = 0;
    long long total_bytes = 0;
    signal(SIGINT, on_sigint);
    while (done < sites.size()) {
        for (int l = 0; l < lanes && next < sites.size() && !g_stop; ++l) {
            if (lane_pid[l] != 0) continue;
            const Site &s = sites[next];
            std::vector<std::string> args(base_args);
            args.push_back("-l");
            args.push_back(s.llh);
            args.push_back("-o");
            args.push_back(s.out);
            args.push_back("-P");
            args.push_back(std::to_string(udp_base > 0 ? udp_base + (int)next : 0));
            const pid_t pid = fork();
            if (pid < 0) {
                perror("fork");
                return 1;
            }
            if (pid == 0) {
                char dev[16];
                snprintf(dev, sizeof(dev), "%d", l % n_gpus);
                setenv("GAL_DEVICE", dev, 1);
                std::vector<char *> av;
                av.push_back(const_cast<char *>(self));
                for (auto &a : args) av.push_back(const_cast<char *>(a.c_str()));
                av.push_back(nullptr);
                execv(self, av.data());
                perror("execv");
                _exit(127);
            }
            lane_pid[l] = pid;
            lane_site[l] = (int)next;
            lane_t0[l] = std::chrono::steady_clock::now();
            ++next;
        }
        int status = 0;
        const pid_t pid = wait(&status);
        if (pid < 0) {
            if (errno == EINTR) continue;
            break;
        }
        for (int l = 0; l < lanes; ++l) {
            if (lane_pid[l] != pid) continue;
            const Site &s = sites[lane_site[l]];
            struct stat sb;
            const long long bytes = stat(s.out.c_str(), &sb) == 0 ? (long long)sb.st_size : 0;
            const bool ok = WIFEXITED(status) && WEXITSTATUS(status) == 0;
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - lane_t0[l]).count();
            // (the child's wall time: HIP start-up, front-end, synthesis, device->host copy, file -- the sink, not the engine, sets it)
            fprintf(stderr, "site %d (%s) on GPU %d -> %s: %s, %lld bytes in %.2f s = %.0f Msamples/s = %.2f GB/s into its file\n", lane_site[l],
                    s.llh.c_str(), l % n_gpus, s.out.c_str(), ok ? "ok" : "FAILED", bytes, dt, bytes / 4 / dt / 1e6, bytes / dt / 1e9);
            if (!ok) ++failed;
            total_bytes += bytes;
            lane_pid[l] = 0;
            ++done;
        }
        if (g_stop && next < sites.size()) {  // interrupted: the sites not started yet are dropped
            done += sites.size() - next;
            next = sites.size();
        }
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "\nDone!\nSites = %zu  failed = %d  Process time = %.2f [sec]  (%.1f Msamples/s aggregate over %d GPU%s, incl. process "
                    "start-up; per site the device->host link and the file system bound this figure, not the synthesis engine: see the "
                    "per-site lines)\n", sites.size(), failed, el, total_bytes / 4 / el / 1e6, n_gpus, n_gpus > 1 ? "s" : "");
    return failed ? 1 : 0;
}

}  // namespace

// GAL_CLI_TIMING=1: where the start-up goes (stage times on stderr)
static void stage(const char *what)
{
    static const bool on = getenv("GAL_CLI_TIMING") != nullptr;
    static auto t0 = std::chrono::steady_clock::now();
    static auto tl = t0;
    if (!on) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[timing] %-28s +%7.1f ms  (%7.1f ms since start)\n", what, std::chrono::duration<double, std::milli>(t - tl).count(),
            std::chrono::duration<double, std::milli>(t - t0).count());
    tl = t;
}

int main(int argc, char *argv[])
{
    stage("process start");
    if (argc < 3) {
        usage(argv[0]);
        exit(1);
    }
    gal_scen_cfg_t sc;
    memset(&sc, 0, sizeof(sc));
    char navfile[4096] = "", outfile[4096] = "", umfile[4096] = "", sitesfile[4096] = "";
    sc.llh[0] = 42.3601;  // defaults of src/main.cpp:179-196
    sc.llh[1] = -71.0589;
    sc.llh[2] = 2;
    sc.duration_s = 300.0;
    sc.iono_enable = 1;
    sc.n_slots = GAL_MAX_CHAN;
    bool verbose = false, have_batch = false, udp_given = false, realtime = false, cboc = false, exact_replay = false, shift_toe = false, ref_T = false;
    int batch_epochs = 128, n_writers = -1, sites_gpus = 0, sites_per_gpu = 1;
    sc.udp_port = GAL_SCEN_UDP_PORT;  // the reference always listens for position updates (src/galileo-sdr.cpp:185)
    sc.udp_loopback = 1;              // ... on every interface; the default listener here takes local datagrams only

    enum { OPT_STRICT = 1000, OPT_SITES, OPT_WRITERS, OPT_GPUS, OPT_PER_GPU, OPT_EXACT, OPT_SHIFT_TOE, OPT_REF_T };
    static const struct option long_opts[] = {{"strict", no_argument, nullptr, OPT_STRICT},
                                              {"exact-replay", no_argument, nullptr, OPT_EXACT},
                                              {"shift-toe", no_argument, nullptr, OPT_SHIFT_TOE},
                                             {"ref-T", no_argument, nullptr, OPT_REF_T},
                                              {"sites", required_argument, nullptr, OPT_SITES},
                                              {"writers", required_argument, nullptr, OPT_WRITERS},
                                              {"gpus", required_argument, nullptr, OPT_GPUS},
                                              {"per-gpu", required_argument, nullptr, OPT_PER_GPU},
                                              {nullptr, 0, nullptr, 0}};
    std::vector<std::string> child_args;  // --sites: everything but -l / -o / --sites / --gpus / --per-gpu goes to the children
    int opt;
    while ((opt = getopt_long(argc, argv, "e:n:o:u:g:l:T:t:d:G:a:p:iI:U:b:vB:P:rC", long_opts, nullptr)) != -1) {
        if (opt != 'l' && opt != 'o' && opt != 'P' && opt != OPT_SITES && opt != OPT_GPUS && opt != OPT_PER_GPU && opt != '?' && opt != ':') {
            if (opt >= 1000) {
                child_args.push_back(opt == OPT_STRICT ? "--strict" : opt == OPT_EXACT ? "--exact-replay" : opt == OPT_SHIFT_TOE ? "--shift-toe" : opt == OPT_REF_T ? "--ref-T" : "--writers");
            } else {
                char name[3] = {'-', (char)opt, 0};
                child_args.push_back(name);
            }
            if (optarg) child_args.push_back(optarg);
        }
        switch (opt) {
        case 'e': snprintf(navfile, sizeof(navfile), "%s", optarg); break;
        case 'o': snprintf(outfile, sizeof(outfile), "%s", optarg); break;
        case 'u': snprintf(umfile, sizeof(umfile), "%s", optarg); break;
        case 'l': sscanf(optarg, "%lf,%lf,%lf", &sc.llh[0], &sc.llh[1], &sc.llh[2]); break;
        case 'T':  // -t without the range check, UTC reference time overwritten (src/main.cpp:237-257; galscen.h: time_overwrite)
            if (!sc.time_overwrite) sc.time_overwrite = 1;
            if (strncmp(optarg, "now", 3) == 0) {
                time_t timer;
                time(&timer);
                const struct tm *gmt = gmtime(&timer);
                sc.start[0] = gmt->tm_year + 1900; sc.start[1] = gmt->tm_mon + 1; sc.start[2] = gmt->tm_mday;
                sc.start[3] = gmt->tm_hour; sc.start[4] = gmt->tm_min;
                sc.start_sec = (double)gmt->tm_sec;
                sc.have_start = 1;
                break;
            }
            /* fall through */
        case 't':
            if (sscanf(optarg, "%d/%d/%d,%d:%d:%lf", &sc.start[0], &sc.start[1], &sc.start[2], &sc.start[3],
                       &sc.start[4], &sc.start_sec) != 6) {
                printf("ERROR: Invalid date and time.\n");
                exit(1);
            }
            sc.have_start = 1;
            break;
        case 'd': sc.duration_s = atof(optarg); break;
        case 'I': sc.iono_enable = 0; break;
        case 'v': verbose = true; break;
        case 'B': batch_epochs = atoi(optarg); have_batch = true; break;
        case 'P':
            sc.udp_port = atoi(optarg);
            sc.udp_loopback = 0;
            udp_given = true;
            break;
        case 'r': realtime = true; break;
        case 'C': cboc = true; break;
        case OPT_STRICT: sc.strict_eph = 1; break;
        case OPT_EXACT: exact_replay = true; break;
        case OPT_SHIFT_TOE: shift_toe = true; break;
        case OPT_REF_T: ref_T = true; break;
        case OPT_SITES: snprintf(sitesfile, sizeof(sitesfile), "%s", optarg); break;
        case OPT_WRITERS: n_writers = atoi(optarg); break;
        case OPT_GPUS: sites_gpus = atoi(optarg); break;
        case OPT_PER_GPU: sites_per_gpu = atoi(optarg); break;
        case 'n': case 'g': case 'G': case 'a': case 'p': case 'i': case 'U': case 'b': break;
        case ':':
        case '?':
            usage(argv[0]);
            exit(1);
        default: break;
        }
    }
    if (navfile[0] == 0) {
        printf("ERROR: Galileo ephemeris/nav_msg file is not specified.\n");
        exit(1);
    }
    // plain -T is the reference as built (time_overwrite 1: the same command line gives the same bytes; ADVICE r5 -- round 5 had it
    // shift, and every replay of a reference command line had to add --ref-T); --shift-toe asks for the shift of TOC / TOE (2)
    if ((shift_toe || ref_T) && !sc.time_overwrite) {
        printf("ERROR: %s needs -T <date,time>.\n", ref_T ? "--ref-T" : "--shift-toe");
        exit(1);
    }
    if (shift_toe && ref_T) {
        printf("ERROR: --shift-toe and --ref-T exclude each other.\n");
        exit(1);
    }
    if (sc.time_overwrite) sc.time_overwrite = shift_toe ? 2 : 1;
    if (sitesfile[0]) {
        // several listeners cannot share a port: the sites run without the position listener unless -P names a base port, in
        // which case site k (in file order) listens on port + k
        char self[4096];
        const ssize_t n = readlink("/proc/self/exe", self, sizeof(self) - 1);
        if (n <= 0) snprintf(self, sizeof(self), "%s", argv[0]);
        else self[n] = 0;
        return run_sites(self, child_args, sitesfile, outfile, sites_gpus, sites_per_gpu, udp_given ? sc.udp_port : 0);
    }
    if (outfile[0] == 0) {
        printf("[+] File sink not specified. Using galileosim.ishort\n");
        snprintf(outfile, sizeof(outfile), "galileosim.ishort");
    }
    if (realtime && !have_batch) batch_epochs = 1;  // paced output: position updates take effect within 0.1 s
    if (batch_epochs < 1) batch_epochs = 1;
    sc.nav_file = navfile;
    sc.motion_file = umfile[0] ? umfile : nullptr;
    sc.verbose = 1;

    gal_scen_t *scen = nullptr;
    int orc = gal_scen_open(&sc, &scen);
    if (orc == GAL_E_BUSY && sc.udp_port > 0 && !udp_given) {
        // the default port is taken (another instance): the reference would exit; carry on without the listener
        fprintf(stderr, "WARNING: %s; continuing without run-time position updates\n", gal_scen_last_error());
        sc.udp_port = 0;
        orc = gal_scen_open(&sc, &scen);
    }
    if (orc != GAL_OK) {
        fprintf(stderr, "%s\n", gal_scen_last_error());
        if (orc == GAL_E_EMPTY && strcmp(outfile, "-") != 0) {
            // (int)(10 d + 0.5) < 2: the reference opens its sink, finds no epoch to generate and closes it (src/galileo-sdr.cpp:438)
            FILE *f = fopen(outfile, "wb");
            if (f) fclose(f);
            exit(0);
        }
        exit(1);
    }
    stage("scenario opened (RINEX)");
    const int total = gal_scen_total_epochs(scen);
    int32_t wk;
    double ws;
    gal_scen_start_time(scen, &wk, &ws);
    fprintf(stderr, "\n%s\nStart = %d:%.0f  Duration = %.1f [sec]  (%d epochs of 0.1 s)\n",
            sc.motion_file ? "Using user motion file." : "Using static location mode.", wk, ws, (total + 1) / 10.0, total);

    gal_synth_cfg_t cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.sample_rate = 2.6e6;
    cfg.samples_per_epoch = 260000;
    cfg.n_slots = sc.n_slots;
    cfg.device = getenv("GAL_DEVICE") ? atoi(getenv("GAL_DEVICE")) : -1;
    if (cboc) cfg.flags |= GAL_CFG_CBOC;
    if (exact_replay) cfg.flags |= GAL_CFG_EXACT_REPLAY;
    const size_t epoch_bytes = (size_t)cfg.samples_per_epoch * 4;

    if (n_writers < 0) {
        // Default: the sequential sink.  Measured on the MI355X host (256 cores, tmpfs, 1.25 GB): write() stream 0.20 s;
        // mapped sink with 4 / 8 / 16 copy threads 0.24 / 0.22 / 0.24 s INCLUDING the unmap (0.10-0.21 s without it, which is
        // what round 3's first measurements showed): the kernel inserts pages into ONE file's page cache at ~6 GB/s whoever
        // asks -- write(), page faults of many threads, pwrite()s of many threads (inode lock) -- profiles/archive/r03a_sink_probe.log.
        n_writers = 0;
    }
    Sink sink;
    if (!sink.open(outfile, realtime ? 0 : (size_t)total * epoch_bytes, n_writers)) {  // (paced runs stream: they are slow by design)
        fprintf(stderr, "ERROR: Failed to open output file.\n");
        exit(1);
    }

    stage("sink opened");
    // (Engine creation and buffer allocation one after the other: running them in two threads was measured -- 6 alternations
    // on one box, 157-318 ms against 250-328 ms for this stage, no difference beyond the run-to-run spread of the HIP
    // start-up itself; the runtime serialises code-object loading and page pinning.)
    gal_synth_t *eng = nullptr;
    if (gal_synth_create(&cfg, &eng) != GAL_OK) {
        fprintf(stderr, "ERROR: %s\n", gal_synth_last_error());
        exit(1);
    }
    stage("gal_synth_create");
    if (batch_epochs > total) batch_epochs = total > 0 ? total : 1;
    const size_t batch_bytes = epoch_bytes * batch_epochs;
    int16_t *d_iq[2] = {nullptr, nullptr};
    Slot slot[2];
    for (int i = 0; i < 2; ++i) {
        if (hipMalloc((void **)&d_iq[i], batch_bytes) != hipSuccess ||
            hipHostMalloc((void **)&slot[i].host, batch_bytes, hipHostMallocDefault) != hipSuccess) {
            fprintf(stderr, "ERROR: buffer allocation failed\n");
            exit(1);
        }
        hipEventCreate(&slot[i].copied[0]);
        hipEventCreate(&slot[i].copied[1]);
    }
    stage("device + pinned buffers");
    // (a stream for the engine, made here: left to the engine it would be made inside the first gal_synth_plan -- 7-17 ms
    // of the run instead of the start-up)
    hipStream_t stream;
    hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    gal_synth_set_stream(eng, stream);
    // device -> host on two streams of their own: two DMA engines share the link, and the copy of batch k runs
    // beside the front-end and the synthesis of batch k+1
    hipStream_t copy_stream[2];
    hipStreamCreateWithFlags(&copy_stream[0], hipStreamNonBlocking);
    hipStreamCreateWithFlags(&copy_stream[1], hipStreamNonBlocking);
    // First use of the copy path belongs to the start-up, not to the run: the first device -> host copy of a process
    // took 8 ms on its own (profiles/archive/r03p_cli_batches.log: batch 2 started 10.4 ms into a run whose batches take 2.35 ms)
    if (!getenv("GAL_CLI_COLD_COPY")) {
        for (int i = 0; i < 2; ++i)
            for (int k = 0; k < 2; ++k) {
                const size_t off = k ? batch_bytes / 2 : 0, nb = batch_bytes / 2 < ((size_t)1 << 20) ? batch_bytes / 2 : ((size_t)1 << 20);
                if (nb) hipMemcpyAsync((char *)slot[i].host + off, (const char *)d_iq[i] + off, nb, hipMemcpyDeviceToHost, copy_stream[k]);
            }
        hipStreamSynchronize(copy_stream[0]);
        hipStreamSynchronize(copy_stream[1]);
    }
    stage("copy path warmed");

    // writer thread: drains full slots in order
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    std::atomic<bool> io_error{false};  // written by the writer thread, read by the producer loop
    int next_write = 0;
    std::thread writer([&]() {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return slot[next_write].full || done; });
            if (!slot[next_write].full) return;
            Slot &s = slot[next_write];
            lk.unlock();
            if (hipEventSynchronize(s.copied[0]) != hipSuccess || hipEventSynchronize(s.copied[1]) != hipSuccess ||
                !sink.put((const char *)s.host, s.bytes))
                io_error = true;
            lk.lock();
            s.full = false;
            next_write ^= 1;
            lk.unlock();
            cv.notify_all();
        }
    });

    signal(SIGINT, on_sigint);
    const auto t_start = std::chrono::steady_clock::now();
    // producer: front-end rows, kRowBufs batches deep
    constexpr int kRowBufs = 3;
    struct RowBuf {
        std::vector<gal_chan_epoch_t> rows;
        int n = 0;  // epochs in it; < 0: front-end error, 0 with `last`: end of the scenario
        bool ready = false;
    };
    RowBuf rb[kRowBufs];
    for (auto &b : rb) b.rows.resize((size_t)batch_epochs * sc.n_slots);
    std::mutex rmu;
    std::condition_variable rcv;
    bool consumer_gone = false;
    std::string scen_error;
    long produced_epochs = 0;
    std::thread producer([&]() {
        int w = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(rmu);
                rcv.wait(lk, [&] { return !rb[w].ready || consumer_gone; });
                if (consumer_gone) return;
            }
            if (realtime) {  // FIFO-style pacing (the reference: src/fifo.cpp + src/galileo-sdr.cpp:570-595): epoch k is
                             // produced no earlier than k * 0.1 s after the start, so that position updates are current
                std::this_thread::sleep_until(t_start + std::chrono::milliseconds(100) * produced_epochs);
            }
            const int n = g_stop ? 0 : gal_scen_next(scen, batch_epochs, rb[w].rows.data());
            if (n > 0) produced_epochs += n;
            {
                std::lock_guard<std::mutex> lk(rmu);
                if (n < 0) scen_error = gal_scen_last_error();
                rb[w].n = n;
                rb[w].ready = true;
            }
            rcv.notify_all();
            if (n <= 0) return;
            w = (w + 1) % kRowBufs;
        }
    });
    std::vector<gal_chan_state_t> state(sc.n_slots);
    memset(state.data(), 0, sizeof(gal_chan_state_t) * sc.n_slots);
    bool have_state = false;
    int emitted = 0, cur = 0, rc = 0, r = 0;
    // (SIGINT: the batch in flight is finished and written, batches the producer has queued behind it are dropped)
    const bool batch_timing = getenv("GAL_CLI_TIMING") != nullptr;
    auto ms_since = [&](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    while (emitted < total && !io_error && !g_stop) {
        const auto tb0 = std::chrono::steady_clock::now();
        {
            std::unique_lock<std::mutex> lk(rmu);
            rcv.wait(lk, [&] { return rb[r].ready; });
        }
        const double tb_rows = ms_since(tb0);
        const int n = rb[r].n;
        const gal_chan_epoch_t *rows_ptr = rb[r].rows.data();
        if (n < 0) {
            fprintf(stderr, "\nERROR: %s\n", scen_error.c_str());
            rc = 1;
            break;
        }
        if (n == 0) break;
        {  // wait until the slot we are about to fill has been written out
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !slot[cur].full; });
        }
        const double tb_slot = ms_since(tb0);
        if (gal_synth_plan(eng, rows_ptr, n, have_state ? state.data() : nullptr) != GAL_OK ||
            gal_synth_execute(eng, d_iq[cur]) != GAL_OK) {
            fprintf(stderr, "\nERROR: %s\n", gal_synth_last_error());
            rc = 1;
            break;
        }
        {  // plan() has uploaded the rows: the producer may refill this buffer
            std::lock_guard<std::mutex> lk(rmu);
            rb[r].ready = false;
        }
        rcv.notify_all();
        r = (r + 1) % kRowBufs;
        // The IQ in d_iq[cur] is final only once gal_synth_finish() has returned: finish() may find the speculative
        // carrier chain unverified (or the replay check unhappy) and synthesise the batch again.  The copies are
        // therefore enqueued after it; they still run beside the front-end and the synthesis of the next batch.
        if (gal_synth_finish(eng, state.data(), nullptr) != GAL_OK) {
            fprintf(stderr, "\nERROR: %s\n", gal_synth_last_error());
            rc = 1;
            break;
        }
        const double tb_synth = ms_since(tb0);
        if (batch_timing)
            fprintf(stderr, "[timing] batch at %7.2f ms: %3d epochs, waited %.2f ms for rows, %.2f for a free slot, plan + execute + finish %.2f\n",
                    std::chrono::duration<double, std::milli>(tb0 - t_start).count(), n, tb_rows, tb_slot - tb_rows, tb_synth - tb_slot);
        slot[cur].bytes = epoch_bytes * n;
        {
            const size_t half = epoch_bytes * (size_t)((n + 1) / 2);
            const size_t part[2] = {half, slot[cur].bytes - half};
            size_t off = 0;
            hipError_t cerr = hipSuccess;
            for (int k = 0; k < 2; ++k) {
                if (part[k] && cerr == hipSuccess)
                    cerr = hipMemcpyAsync((char *)slot[cur].host + off, (const char *)d_iq[cur] + off, part[k],
                                          hipMemcpyDeviceToHost, copy_stream[k]);
                if (cerr == hipSuccess) cerr = hipEventRecord(slot[cur].copied[k], copy_stream[k]);
                off += part[k];
            }
            if (cerr != hipSuccess) {
                fprintf(stderr, "\nERROR: device -> host copy failed: %s\n", hipGetErrorString(cerr));
                rc = 1;
                break;
            }
        }
        have_state = true;
        {
            std::lock_guard<std::mutex> lk(mu);
            slot[cur].full = true;
        }
        cv.notify_all();
        cur ^= 1;
        emitted += n;
        if (verbose || isatty(fileno(stderr))) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
            fprintf(stderr, "\rTime into run = %4.1f - %4.1f", emitted / 10.0, el);
        }
    }
    {
        std::lock_guard<std::mutex> lk(rmu);
        consumer_gone = true;
    }
    rcv.notify_all();
    producer.join();
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !slot[0].full && !slot[1].full; });
        done = true;
    }
    cv.notify_all();
    writer.join();
    // (the sink is closed inside the reported time: unmapping a 1.2 GB file mapping is not free, and the file is only
    // the caller's once it is closed)
    if (!sink.finish()) io_error = true;
    stage("run (= Process time)");
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    fprintf(stderr, "\nDone!\nProcess time = %.2f [sec]  (%.1f Msamples/s, %.0fx real time)\n", el,
            emitted * 0.26 / el, emitted * 0.1 / el);
    if (gal_scen_eph_gaps(scen) > 0)
        fprintf(stderr, "NOTE: %d (satellite, refresh) pairs ran on a stale ephemeris record (see the warning above)\n",
                gal_scen_eph_gaps(scen));
    gal_synth_destroy(eng);
    gal_scen_close(scen);
    stage("teardown");
    if (io_error) {
        fprintf(stderr, "ERROR: short write on the output sink\n");
        rc = 1;
    }
    return rc;
}
