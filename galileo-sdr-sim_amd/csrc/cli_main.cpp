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
// Pipeline of three threads: a producer runs the host front-end (libgalscen: orbits, ranges, I/NAV pages) up to two
// batches ahead -> the main thread plans and executes each batch on the GPU and, after gal_synth_finish(), enqueues
// the copy into one of two pinned buffers -> a writer thread streams full buffers to the sink.  In steady state the
// run time is that of the slowest stage: the device->host copy for /dev/null, the write for a file.
#include <hip/hip_runtime.h>
#include <getopt.h>
#include <signal.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
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
           "  -T <date,time>   Overwrite TOC and TOE to scenario start time (use `now` for the current time)\n"
           "  -P <port>        UDP port for run-time position updates lat,lon,hgt as 3 doubles (default 7533, 0 = off)\n"
           "  -r               Pace the output to real time (one 0.1 s epoch per 0.1 s)\n"
           "  -C               CBOC(6,1,1/11) sub-carrier of the E1 OS ICD instead of the reference's BOC(1,1) (opt-in)\n"
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

}  // namespace

int main(int argc, char *argv[])
{
    if (argc < 3) {
        usage(argv[0]);
        exit(1);
    }
    gal_scen_cfg_t sc;
    memset(&sc, 0, sizeof(sc));
    char navfile[4096] = "", outfile[4096] = "", umfile[4096] = "";
    sc.llh[0] = 42.3601;  // defaults of src/main.cpp:179-196
    sc.llh[1] = -71.0589;
    sc.llh[2] = 2;
    sc.duration_s = 300.0;
    sc.iono_enable = 1;
    sc.n_slots = GAL_MAX_CHAN;
    bool verbose = false, have_batch = false, udp_given = false, realtime = false, cboc = false;
    int batch_epochs = 128;
    sc.udp_port = GAL_SCEN_UDP_PORT;  // the reference always listens for position updates (src/galileo-sdr.cpp:185)

    int opt;
    while ((opt = getopt(argc, argv, "e:n:o:u:g:l:T:t:d:G:a:p:iI:U:b:vB:P:rC")) != -1) {
        switch (opt) {
        case 'e': snprintf(navfile, sizeof(navfile), "%s", optarg); break;
        case 'o': snprintf(outfile, sizeof(outfile), "%s", optarg); break;
        case 'u': snprintf(umfile, sizeof(umfile), "%s", optarg); break;
        case 'l': sscanf(optarg, "%lf,%lf,%lf", &sc.llh[0], &sc.llh[1], &sc.llh[2]); break;
        case 'T':  // -t plus: overwrite TOC / TOE so that the file is valid at that time (src/main.cpp:237-257)
            sc.time_overwrite = 1;
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
        case 'P': sc.udp_port = atoi(optarg); udp_given = true; break;
        case 'r': realtime = true; break;
        case 'C': cboc = true; break;
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
    if (orc == GAL_E_IO && sc.udp_port > 0 && !udp_given && strstr(gal_scen_last_error(), "UDP")) {
        // the default port is taken (another instance): the reference would exit; carry on without the listener
        fprintf(stderr, "WARNING: %s; continuing without run-time position updates\n", gal_scen_last_error());
        sc.udp_port = 0;
        orc = gal_scen_open(&sc, &scen);
    }
    if (orc != GAL_OK) {
        fprintf(stderr, "%s\n", gal_scen_last_error());
        exit(1);
    }
    const int total = gal_scen_total_epochs(scen);
    int32_t wk;
    double ws;
    gal_scen_start_time(scen, &wk, &ws);
    fprintf(stderr, "\n%s\nStart = %d:%.0f  Duration = %.1f [sec]  (%d epochs of 0.1 s)\n",
            sc.motion_file ? "Using user motion file." : "Using static location mode.", wk, ws, (total + 1) / 10.0, total);

    FILE *fp = stdout;
    if (strcmp("-", outfile)) {
        fp = fopen(outfile, "wb");
        if (!fp) {
            fprintf(stderr, "ERROR: Failed to open output file.\n");
            exit(1);
        }
    }

    gal_synth_cfg_t cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.sample_rate = 2.6e6;
    cfg.samples_per_epoch = 260000;
    cfg.n_slots = sc.n_slots;
    cfg.device = getenv("GAL_DEVICE") ? atoi(getenv("GAL_DEVICE")) : -1;
    if (cboc) cfg.flags |= GAL_CFG_CBOC;
    gal_synth_t *eng = nullptr;
    if (gal_synth_create(&cfg, &eng) != GAL_OK) {
        fprintf(stderr, "ERROR: %s\n", gal_synth_last_error());
        exit(1);
    }
    hipStream_t stream;
    hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    gal_synth_set_stream(eng, stream);

    const size_t epoch_bytes = (size_t)cfg.samples_per_epoch * 4;
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
    // device -> host on two streams of their own: two DMA engines share the link, and the copy of batch k runs
    // beside the front-end and the synthesis of batch k+1
    hipStream_t copy_stream[2];
    hipStreamCreateWithFlags(&copy_stream[0], hipStreamNonBlocking);
    hipStreamCreateWithFlags(&copy_stream[1], hipStreamNonBlocking);

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
            hipEventSynchronize(s.copied[0]);
            hipEventSynchronize(s.copied[1]);
            if (fwrite(s.host, 1, s.bytes, fp) != s.bytes) io_error = true;
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
    while (emitted < total && !io_error) {
        {
            std::unique_lock<std::mutex> lk(rmu);
            rcv.wait(lk, [&] { return rb[r].ready; });
        }
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
        slot[cur].bytes = epoch_bytes * n;
        {
            const size_t half = epoch_bytes * (size_t)((n + 1) / 2);
            const size_t part[2] = {half, slot[cur].bytes - half};
            size_t off = 0;
            for (int k = 0; k < 2; ++k) {
                if (part[k])
                    hipMemcpyAsync((char *)slot[cur].host + off, (const char *)d_iq[cur] + off, part[k],
                                   hipMemcpyDeviceToHost, copy_stream[k]);
                hipEventRecord(slot[cur].copied[k], copy_stream[k]);
                off += part[k];
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
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    fprintf(stderr, "\nDone!\nProcess time = %.2f [sec]  (%.1f Msamples/s, %.0fx real time)\n", el,
            emitted * 0.26 / el, emitted * 0.1 / el);
    if (fp != stdout) fclose(fp);
    else fflush(stdout);
    gal_synth_destroy(eng);
    gal_scen_close(scen);
    if (io_error) {
        fprintf(stderr, "ERROR: short write on the output sink\n");
        rc = 1;
    }
    return rc;
}
