// bf_motion_compensator -- command-line front end of the motion-compensation path.
//
// Drop-in for the reference's tool of the same name (better_flow_core/src/bf_motion_compensator.cpp:61-216): the same
// flags with the same meaning, the same text input ("t x y p" per line) and the same "-o" output ("t x y 1 v u"); the
// optimizer behind it runs on the MI355X.  The flag SURFACE is the reference's; the parser is a table (one row per
// flag: spelling, kind of argument, destination, help text), so adding a flag is one row.
//
// Flags the reference fixes at compile time (common.h:39-40, bf_motion_compensator.cpp:6-7) are run-time options here:
// --res-x= --res-y= (sensor rows / columns), --scale=, --max-iter=, --device=, --max-events= --span= (the event ring);
// --to-bin= converts a text recording to the binary structure-of-arrays format (better_flow/event_reader.h);
// --slice-log= writes one CSV record per slice.
//
// Two slice managers sit behind the same flags and give the same results (tests/test_host_cli.py holds one to the
// other): bf::StreamEngine (better_flow/stream_flow.h) -- events enter a pinned structure-of-arrays ring in bulk, a
// binary input is read straight into that ring, slices are solved on a worker thread while the next block is read,
// --devices= spreads independent slices (--stm-disable) over several GPUs, per-event flow is fetched only for -o -- and
// DVS_flow (better_flow/dvs_flow.h), the reference's array-of-Event ring, which --img / --video / -i need (the frame
// renderer works on its optimizer object) and --engine=ring selects.
#include <better_flow/common.h>
#include <better_flow/dvs_flow.h>
#include <better_flow/stream_flow.h>

#include <chrono>
#include <functional>
#include <thread>

namespace {

constexpr size_t kMaxEvents = 50000;       // ring capacity (the reference's EVENT_WIDTH)
constexpr double kMaxSpanSec = 0.2;        // ring time span (TIME_WIDTH)

struct Options {
    double refresh_time = 0.033;           // seconds between two slices ...
    unsigned long long refresh_events = 20000;   // ... or new events, whichever comes first
    bool interactive = false, quiet = false, gpu_flag = false, stm_disable = false, bufferize = false;
    bool frames = false, video = false;
    std::string frame_prefix = "./", video_name = "./out.avi";
    int video_fps = 60;
    int scale = 3, max_iter = -1;
    std::string input, output, to_bin, slice_log;
    std::vector<std::string> more_inputs;  // further recordings: each is a stream of its own
    bool have_input = false, have_output = false;
    // the event ring (EVENT_WIDTH / TIME_WIDTH of the reference, bf_motion_compensator.cpp:6-7)
    unsigned long long max_events = kMaxEvents;
    double span_sec = kMaxSpanSec;
    bool ring_flags = false;               // --max-events / --span given
    std::string engine;                    // "", "stream" or "ring"
    std::vector<int> devices;              // --devices
    int contexts = 1;                      // slice contexts per device
    bool sync = false, timing = false;
    int parse_threads = 0;                 // threads of the text parser
    int threads = 0;                       // reader / writer threads (0: one per core, at most 4 -- measured: 2.4 / 2.3 / 2.1 / 1.9
                                           // Gevents/s from a binary file with 2 / 4 / 8 / 16 reader threads)
};

// "0-3", "0,2,5", "1": the HIP devices of --devices
std::vector<int> parse_device_list(const char *v) {
    std::vector<int> out;
    const char *p = v;
    while (*p) {
        char *e = nullptr;
        const long a = std::strtol(p, &e, 10);
        if (e == p || a < 0) return {};
        long b = a;
        p = e;
        if (*p == '-') {
            b = std::strtol(p + 1, &e, 10);
            if (e == p + 1 || b < a) return {};
            p = e;
        }
        for (long d = a; d <= b; ++d) out.push_back((int)d);
        if (*p == ',') ++p;
        else if (*p) return {};
    }
    return out;
}

enum class Arg { None, Inline, Next };     // "--flag", "--flag=value", "--flag value"

struct Flag {
    const char *name;                      // spelling, without the '=' of an inline value
    Arg arg;
    std::function<void(Options &, const char *)> set;
    const char *value_hint, *help;
};

const std::vector<Flag> &flag_table() {
    static const std::vector<Flag> t = {
        {"--refresh-time", Arg::Inline, [](Options &o, const char *v) { o.refresh_time = atof(v); }, "<seconds>",
         "start a slice once this much time has passed since the previous one"},
        {"--refresh-event-count", Arg::Inline, [](Options &o, const char *v) { o.refresh_events = (unsigned long long)atoll(v); },
         "<n>", "... or once this many new events have arrived"},
        {"-i", Arg::None, [](Options &o, const char *) { o.interactive = true; }, "", "interactive mode (not available without a display)"},
        {"--interactive", Arg::None, [](Options &o, const char *) { o.interactive = true; }, "", "same as -i"},
        {"-G", Arg::None, [](Options &o, const char *) { o.gpu_flag = true; }, "", "accepted for compatibility: the GPU path is the only one"},
        {"--stm-disable", Arg::None, [](Options &o, const char *) { o.stm_disable = true; }, "",
         "start every slice from the zero model instead of the previous slice's estimate"},
        {"--img", Arg::None, [](Options &o, const char *) { o.frames = true; }, "", "write one frame (PPM + text side-car) per slice"},
        {"--img-prefix", Arg::Next, [](Options &o, const char *v) { o.frame_prefix = v; }, "<dir>", "directory of those frames"},
        {"--video", Arg::None, [](Options &o, const char *) { o.video = true; }, "", "write the frames to a video (uncompressed AVI)"},
        {"--video-name", Arg::Next, [](Options &o, const char *v) { o.video_name = v; }, "<file>", "name of that video"},
        {"--video-fps", Arg::Inline, [](Options &o, const char *v) { o.video_fps = atoi(v); }, "<n>", "its frame rate"},
        {"--bufferize-file", Arg::None, [](Options &o, const char *) { o.bufferize = true; }, "",
         "read the whole input first, then process (timing runs)"},
        {"--quiet", Arg::None, [](Options &o, const char *) { o.quiet = true; }, "", "print nothing but errors"},
        {"-o", Arg::Next, [](Options &o, const char *v) { o.output = v; o.have_output = true; }, "<file>",
         "write every event with its flow: \"t x y 1 v u\""},
        {"--outfile", Arg::Inline, [](Options &o, const char *v) { o.output = v; o.have_output = true; }, "<file>", "same as -o"},
        {"--res-x", Arg::Inline, [](Options &, const char *v) { bf::sensor().res_x = atoi(v); }, "<rows>", "sensor rows"},
        {"--res-y", Arg::Inline, [](Options &, const char *v) { bf::sensor().res_y = atoi(v); }, "<columns>", "sensor columns"},
        {"--scale", Arg::Inline, [](Options &o, const char *v) { o.scale = atoi(v); }, "<odd>", "image scale of the minimizer"},
        {"--max-iter", Arg::Inline, [](Options &o, const char *v) { o.max_iter = atoi(v); }, "<n>", "cap on minimizer iterations per slice"},
        {"--device", Arg::Inline, [](Options &, const char *v) { bf::DeviceContext::device() = atoi(v); }, "<n>", "HIP device"},
        {"--to-bin", Arg::Inline, [](Options &o, const char *v) { o.to_bin = v; }, "<file>",
         "convert the text input to the binary event format and exit"},
        {"--slice-log", Arg::Inline, [](Options &o, const char *v) { o.slice_log = v; }, "<file>",
         "one CSV record per slice: slice,events,new_events,rc,iterations,ms,mevents_per_s"},
        {"--max-events", Arg::Inline, [](Options &o, const char *v) { o.max_events = (unsigned long long)atoll(v); o.ring_flags = true; }, "<n>",
         "event ring capacity = most events in a slice (the reference compiles in 50000)"},
        {"--span", Arg::Inline, [](Options &o, const char *v) { o.span_sec = atof(v); o.ring_flags = true; }, "<seconds>",
         "time span of the event ring = longest slice (the reference compiles in 0.2)"},
        {"--engine", Arg::Inline, [](Options &o, const char *v) { o.engine = v; }, "stream|ring",
         "slice manager: structure-of-arrays stream engine (default) or the reference's array-of-Event ring"},
        {"--devices", Arg::Inline, [](Options &o, const char *v) { o.devices = parse_device_list(v); if (o.devices.empty()) o.devices.push_back(-1); },
         "<list>", "HIP devices for independent slices (needs --stm-disable), e.g. 0-7 or 0,2"},
        {"--contexts", Arg::Inline, [](Options &o, const char *v) { o.contexts = atoi(v); }, "<n>", "slice contexts (worker threads) per device"},
        {"--sync", Arg::None, [](Options &o, const char *) { o.sync = true; }, "", "solve every slice before reading on (no pipelining)"},
        {"--threads", Arg::Inline, [](Options &o, const char *v) { o.threads = atoi(v); }, "<n>", "threads for reading the input and formatting -o"},
        {"--timing", Arg::None, [](Options &o, const char *) { o.timing = true; }, "", "print a JSON line with the run's wall-clock phases to stderr"},
    };
    return t;
}

void print_version() {
    std::printf("bf_motion_compensator %s, built %s %s\n", BF_VERSION, __DATE__, __TIME__);
    std::printf("  event ring: %zu events, %.3f s; optimizer: %s\n", kMaxEvents, kMaxSpanSec, bf_version());
}

void print_usage(const Options &defaults) {
    print_version();
    std::printf("\nusage: bf_motion_compensator [flags] <events file, or \"-\">\n\n");
    for (const Flag &f : flag_table()) {
        std::string lhs = f.name;
        if (f.arg == Arg::Inline) lhs += std::string("=") + f.value_hint;
        if (f.arg == Arg::Next) lhs += std::string(" ") + f.value_hint;
        std::printf("  %-34s %s\n", lhs.c_str(), f.help);
    }
    std::printf("  %-34s %s\n  %-34s %s\n", "--version, -v", "print the version", "--help", "print this text");
    std::printf("\ndefaults: --refresh-time=%g --refresh-event-count=%llu --scale=%d --res-x=%d --res-y=%d --video-fps=%d\n",
                defaults.refresh_time, defaults.refresh_events, defaults.scale, RES_X, RES_Y, defaults.video_fps);
}

// Returns -1 to go on, otherwise the process exit code.
int parse(int argc, char **argv, Options &o) {
    const Options defaults;
    if (argc == 1) { print_usage(defaults); return 1; }
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--help") { print_usage(defaults); return 0; }
        if (a == "-v" || a == "--version") { print_version(); return 0; }
        if (a == "-") continue;   // (the reference accepts and ignores a lone dash)
        if (a[0] != '-') {
            if (o.have_input) { o.more_inputs.push_back(a); continue; }   // several recordings: independent streams (see run)
            o.input = a; o.have_input = true;
            continue;
        }
        const Flag *hit = nullptr;
        const char *value = nullptr;
        for (const Flag &f : flag_table()) {
            const size_t n = std::strlen(f.name);
            if (f.arg == Arg::Inline) {
                if (a.compare(0, n, f.name) == 0 && a.size() > n && a[n] == '=') { hit = &f; value = argv[i] + n + 1; }
            } else if (a == f.name) {
                hit = &f;
                if (f.arg == Arg::Next) {
                    if (i + 1 >= argc) {
                        std::fprintf(stderr, "%s needs an argument\n", f.name);
                        return 1;
                    }
                    value = argv[++i];
                }
            }
            if (hit) break;
        }
        if (!hit) {
            std::fprintf(stderr, "unknown flag \"%s\" (--help lists them)\n", argv[i]);
            return 1;
        }
        hit->set(o, value);
    }
    if (!o.have_input) { std::fprintf(stderr, "no input file\n"); return 1; }
    if (o.scale < 1 || o.scale % 2 == 0) { std::fprintf(stderr, "--scale must be odd\n"); return 1; }
    if (!o.engine.empty() && o.engine != "stream" && o.engine != "ring") { std::fprintf(stderr, "--engine must be stream or ring\n"); return 1; }
    const bool needs_ring = o.frames || o.video || o.interactive;
    if (o.engine.empty()) o.engine = needs_ring ? "ring" : "stream";
    if (o.engine == "stream" && needs_ring) { std::fprintf(stderr, "--img / --video / -i work on the reference ring: drop --engine=stream\n"); return 1; }
    if (o.engine == "ring" && (o.ring_flags || !o.devices.empty() || o.contexts != 1)) {
        std::fprintf(stderr, "--max-events / --span / --devices / --contexts belong to the stream engine (the reference ring is compiled for %zu events, %g s, one device)\n",
                     kMaxEvents, kMaxSpanSec);
        return 1;
    }
    if (!o.more_inputs.empty() && (o.engine != "stream" || !o.to_bin.empty())) {
        std::fprintf(stderr, "several input files are independent streams of the stream engine: not with --engine=ring / --img / --video / -i / --to-bin\n");
        return 1;
    }
    if (o.max_events < 1 || o.span_sec <= 0 || o.contexts < 1) { std::fprintf(stderr, "--max-events, --span and --contexts must be positive\n"); return 1; }
    for (int d : o.devices) if (d < 0) { std::fprintf(stderr, "--devices: a list such as 0-7 or 0,2\n"); return 1; }
    if ((o.devices.size() > 1 || o.contexts > 1) && !o.stm_disable && o.more_inputs.empty()) {
        std::fprintf(stderr, "--devices / --contexts spread INDEPENDENT slices: they need --stm-disable (a warm-start chain is sequential)\n");
        return 1;
    }
    if (o.threads <= 0) {
        const unsigned hc = std::thread::hardware_concurrency();
        o.threads = hc == 0 ? 1 : (hc > 4 ? 4 : (int)hc);
        o.parse_threads = hc == 0 ? 1 : (hc > 8 ? 8 : (int)hc);   // the text parser scales further than the block reads: 32 / 58 / 91 Mevents/s on 1 / 4 / 8 threads
    } else {
        o.parse_threads = o.threads;
    }
    return -1;
}

int convert_to_binary(const Options &o) {   // text -> binary structure-of-arrays; no GPU involved
    bf::EventReader reader(o.input.c_str());
    if (!reader.good()) { std::fprintf(stderr, "cannot read '%s'\n", o.input.c_str()); return 1; }
    std::vector<uint64_t> t;
    std::vector<uint16_t> x, y;
    std::vector<uint8_t> p;
    bool too_large = false;
    reader.for_each_event([&](unsigned row, unsigned col, unsigned long long t_ns) {
        too_large |= row > 65535u || col > 65535u;
        t.push_back(t_ns); x.push_back((uint16_t)col); y.push_back((uint16_t)row); p.push_back(1);
    });
    if (too_large) { std::fprintf(stderr, "'%s': a coordinate above 65535 does not fit the binary format\n", o.input.c_str()); return 1; }
    if (!bf::EventReader::write_binary(o.to_bin.c_str(), t, x, y, p)) {
        std::fprintf(stderr, "cannot write '%s'\n", o.to_bin.c_str());
        return 1;
    }
    if (!o.quiet) std::cout << "Converted " << t.size() << " events to " << o.to_bin << std::endl;
    return 0;
}

int run_ring(const Options &o) {
    typedef DVS_flow<kMaxEvents, (sll)FROM_SEC(kMaxSpanSec)> Estimator;
    Estimator estimator(o.refresh_events, FROM_SEC(o.refresh_time));
    estimator.set_quiet(o.quiet);
    estimator.set_scale(o.scale);
    estimator.set_max_iter(o.max_iter);
    if (o.have_output) estimator.set_accumulate();           // the -o file needs every processed event
    if (o.interactive) estimator.set_manual_mode(true);
    if (o.frames) estimator.set_generate_pictures(true, o.frame_prefix);
    if (o.video) estimator.set_generate_video(true, o.video_name, o.video_fps);
    if (o.stm_disable) estimator.set_stm_disable(true);
    if (!o.slice_log.empty() && !estimator.open_slice_log(o.slice_log)) {
        std::fprintf(stderr, "cannot write '%s'\n", o.slice_log.c_str());
        return 1;
    }

    if (o.bufferize) {
        LinearEventCloud cloud;
        EventFile::from_file(&cloud, o.input);
        const auto t_all = std::chrono::steady_clock::now();
        auto t_slice = t_all;
        ull seen = 0;
        for (auto &e : cloud) {
            ++seen;
            if (!estimator.add_event(e)) continue;
            const auto now = std::chrono::steady_clock::now();
            if (!o.quiet)
                std::cout << 100.0f * float(seen) / float(cloud.size()) << " %\t" << seen << "\t"
                          << std::chrono::duration<double>(now - t_slice).count() << " sec\t" << estimator.get_buf_size()
                          << " events\t" << double(estimator.get_time_diff()) * 1e-9 << " slice_td\t"
                          << double(estimator.get_buf_time_diff()) * 1e-9 << " buffer_td\n";
            t_slice = now;
        }
        std::cout << "Total flow elapsed: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count()
                  << " sec." << std::endl;
    } else {
        if (!o.quiet) std::cout << "Reading " << o.input << " ..." << std::endl;
        bf::EventReader reader(o.input.c_str());   // text "t x y p" or binary SoA (better_flow/event_reader.h)
        if (!reader.good()) { std::fprintf(stderr, "cannot read '%s'\n", o.input.c_str()); return 1; }
        const ull n = reader.for_each_event([&](unsigned row, unsigned col, unsigned long long t_ns) {
            Event e(row, col, (ull)t_ns);
            estimator.add_event(e);
        });
        if (!o.quiet) std::cout << "Read and processed " << n << " events" << std::endl;
    }
    estimator.recompute();   // the tail of the stream: every event must have been in a slice

    if (o.have_output) {
        LinearEventCloudTemplate<Event> all = estimator.get_accumulated();
        EventFile::to_file_uv(&all, o.output);
    }
    if (!o.quiet)
        std::cout << "slices: " << estimator.get_slices_done() << " (skipped " << estimator.get_slices_skipped()
                  << "), minimizer iterations: " << estimator.get_iterations_total() << std::endl;
    bf::DeviceContext::release();
    return 0;
}

double seconds_since(std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
}

// The same job on the stream engine (better_flow/stream_flow.h).
int run_stream(const Options &o) {
    const auto t_start = std::chrono::steady_clock::now();
    bf::StreamEngine engine((size_t)o.max_events, (sll)FROM_SEC(o.span_sec), o.refresh_events, FROM_SEC(o.refresh_time));
    engine.set_scale(o.scale);
    engine.set_max_iter(o.max_iter);
    engine.set_stm_disable(o.stm_disable);
    engine.set_want_flow(false);                       // per-event flow only travels back for -o
    if (o.have_output) engine.set_accumulate();
    engine.set_pipelined(!o.sync && !o.bufferize);
    if (!o.devices.empty() || o.contexts > 1)
        engine.set_devices(o.devices.empty() ? std::vector<int>{bf::DeviceContext::device()} : o.devices, o.contexts);
    FILE *slice_log = nullptr;
    if (!o.slice_log.empty()) {
        slice_log = std::fopen(o.slice_log.c_str(), "w");
        if (!slice_log) { std::fprintf(stderr, "cannot write '%s'\n", o.slice_log.c_str()); return 1; }
        std::fprintf(slice_log, "slice,events,new_events,rc,iterations,ms,mevents_per_s\n");
    }
    struct Past { ObjectModel model; size_t size; ull first_ts, last_ts; };   // what dvs_flow.h:245-252 prints per past slice
    std::vector<Past> memory;
    unsigned long long total_events = 0;
    std::chrono::steady_clock::time_point t_first_slice;          // --timing: when the first (cold) slice was delivered
    unsigned long long events_first_slice = 0;
    engine.on_slice([&](const bf::SliceRecord &r) {   // in slice order, on the worker's thread
        if (r.index == 0) { t_first_slice = std::chrono::steady_clock::now(); events_first_slice = r.events_seen; }
        if (!o.quiet) {
            memory.push_back(Past{r.model, (size_t)r.events, r.events ? r.trigger_time : 0, r.oldest_time});
            std::cout << "\n\n------------------------\n";
            for (const Past &p : memory) std::cout << p.model << "\n" << p.size << "\t" << p.first_ts << "\t" << p.last_ts << "\n";
            if (o.bufferize)
                std::cout << 100.0f * float(r.events_seen) / float(total_events ? total_events : 1) << " %\t" << r.events_seen << "\t"
                          << r.ms * 1e-3 << " sec\t" << r.ring_size << " events\t" << double(r.time_diff) * 1e-9 << " slice_td\t"
                          << double((sll)(r.trigger_time - r.start_time)) * 1e-9 << " buffer_td\n";
        }
        if (slice_log) {
            std::fprintf(slice_log, "%llu,%llu,%llu,%d,%d,%.3f,%.3f\n", (unsigned long long)r.index, (unsigned long long)r.events,
                         (unsigned long long)r.new_events, r.rc, (int)r.info.iterations, r.ms, r.ms > 0 ? r.events / r.ms * 1e-3 : 0.0);
            std::fflush(slice_log);
        }
    });
    engine.warm_up();                                  // device contexts, worker threads, pinned ring
    const double s_init = seconds_since(t_start);

    const auto t_read = std::chrono::steady_clock::now();
    double s_flow = 0, s_read = 0;
    unsigned long long n_events = 0;
    if (!o.quiet && !o.bufferize) std::cout << "Reading " << o.input << " ..." << std::endl;
    if (bf::SoaFile::is_soa(o.input) && !o.bufferize) {
        // binary structure-of-arrays input: column blocks are read straight into the engine's pinned ring
        bf::SoaFile file(o.input);
        bool fine = file.good();
        if (fine) n_events = bf::feed_soa_file(file, engine, o.threads, &fine, &s_read);
        if (!fine) { std::fprintf(stderr, "cannot read '%s'\n", o.input.c_str()); return 1; }
    } else {
        // text "t x y p" (the reference's format), or --bufferize-file: the whole input first, then the flow
        if (o.bufferize) std::cout << "Reading from file... (" << o.input << ")" << std::endl;
        bf::EventReader reader(o.input.c_str());
        if (!reader.good()) { std::fprintf(stderr, "cannot read '%s'\n", o.input.c_str()); return 1; }
        std::vector<unsigned long long> t_ns;
        std::vector<uint32_t> row, col;
        const auto t_parse = std::chrono::steady_clock::now();
        if (reader.is_binary() || !reader.parse_text_parallel(o.parse_threads, t_ns, row, col)) {
            t_ns.clear(); row.clear(); col.clear();
            reader.for_each_event([&](unsigned r, unsigned c, unsigned long long t) { row.push_back(r); col.push_back(c); t_ns.push_back(t); });
        }
        n_events = total_events = t_ns.size();
        s_read = seconds_since(t_parse);   // (text: the parse; the file itself was read by the EventReader's constructor)
        if (o.bufferize) std::cout << "Read " << n_events << " events, finished" << std::endl;
        const auto t_flow = std::chrono::steady_clock::now();
        engine.add_events(row.data(), col.data(), t_ns.data(), t_ns.size());
        if (o.bufferize) {
            engine.recompute();
            engine.drain();
            s_flow = seconds_since(t_flow);
            std::cout << "Total flow elapsed: " << s_flow << " sec." << std::endl;
        }
    }
    if (!o.bufferize) {
        if (!o.quiet) std::cout << "Read and processed " << n_events << " events" << std::endl;
        engine.recompute();   // the tail of the stream: every event must have been in a slice
        engine.drain();
    }
    const double s_stream = seconds_since(t_read);
    // the stream behind its first slice: a cold start (hundreds of iterations) followed by warm-started slices
    const double s_steady = engine.get_slices_done() > 1 ? seconds_since(t_first_slice) : 0.0;

    const auto t_out = std::chrono::steady_clock::now();
    if (o.have_output) {
        if (!o.quiet) std::cout << "Aggregating events into one cloud...\n";
        bf::FlowTable all = engine.get_accumulated();
        if (!o.quiet) std::cout << "Final buffer contains " << all.size() << " events." << std::endl;
        std::cout << "Writing events and flow to file... (" << o.output << ")" << std::endl;
        if (!bf::write_flow_text(o.output, all.timestamp, all.row, all.col, all.u, all.v, o.parse_threads)) {
            std::fprintf(stderr, "cannot write '%s'\n", o.output.c_str());
            return 1;
        }
        std::cout << "Written " << all.size() << " events, finished" << std::endl;
    }
    const double s_output = seconds_since(t_out);
    if (!o.quiet)
        std::cout << "slices: " << engine.get_slices_done() << " (skipped " << engine.get_slices_skipped()
                  << "), minimizer iterations: " << engine.get_iterations_total() << std::endl;
    if (o.timing)
        std::fprintf(stderr, "{\"engine\": \"stream\", \"events\": %llu, \"slices\": %llu, \"iterations\": %llu, \"init_s\": %.6f, "
                             "\"stream_s\": %.6f, \"read_s\": %.6f, \"blocked_s\": %.6f, \"output_s\": %.6f, \"total_s\": %.6f, \"mevents_per_s\": %.3f, "
                             "\"steady_s\": %.6f, \"steady_mevents_per_s\": %.3f}\n",
                     n_events, (unsigned long long)engine.get_slices_done(), (unsigned long long)engine.get_iterations_total(), s_init,
                     s_stream, s_read, engine.seconds_blocked(), s_output, seconds_since(t_start), s_stream > 0 ? n_events / s_stream * 1e-6 : 0.0,
                     s_steady, s_steady > 0 ? (n_events - events_first_slice) / s_steady * 1e-6 : 0.0);
    if (slice_log) std::fclose(slice_log);
    return 0;
}

// Several recordings: every file is a stream of its own -- its own ring, its own warm-start chain, its own worker -- and the
// streams run side by side, stream i on device devices[i mod #devices] (--devices=0-7; default: --device).  Independent
// chains are the other way, next to independent slices, to use the GPUs of a node ("farming applies to
// independent streams / slices").  -o, --slice-log: one file per stream, "<name>.<i>".
int run_streams(const Options &o) {
    std::vector<std::string> inputs;
    inputs.push_back(o.input);
    inputs.insert(inputs.end(), o.more_inputs.begin(), o.more_inputs.end());
    const std::vector<int> devs = o.devices.empty() ? std::vector<int>{bf::DeviceContext::device()} : o.devices;
    std::vector<int> rc(inputs.size(), 0);
    std::vector<std::string> errors(inputs.size());
    std::vector<std::thread> th;
    for (size_t i = 0; i < inputs.size(); ++i)
        th.emplace_back([&, i] {
            Options oi = o;
            oi.more_inputs.clear();
            oi.input = inputs[i];
            oi.devices = std::vector<int>{devs[i % devs.size()]};
            oi.contexts = 1;
            if (o.have_output) oi.output = o.output + "." + std::to_string(i);
            if (!o.slice_log.empty()) oi.slice_log = o.slice_log + "." + std::to_string(i);
            try {
                rc[i] = run_stream(oi);
            } catch (const bf::AccelError &e) {
                rc[i] = 2;
                errors[i] = e.what();
            }
        });
    for (auto &t : th) t.join();
    int worst = 0;
    for (size_t i = 0; i < inputs.size(); ++i) {
        if (rc[i] != 0) std::fprintf(stderr, "bf_motion_compensator: stream %zu (%s): %s\n", i, inputs[i].c_str(), errors[i].empty() ? "failed" : errors[i].c_str());
        worst = rc[i] > worst ? rc[i] : worst;
    }
    return worst;
}

int run(const Options &o) {
    if (o.engine == "ring") return run_ring(o);
    return o.more_inputs.empty() ? run_stream(o) : run_streams(o);
}

}  // namespace

int main(int argc, char *argv[]) {
    Options o;
    const int rc = parse(argc, argv, o);
    if (rc >= 0) return rc;
    try {
        return o.to_bin.empty() ? run(o) : convert_to_binary(o);
    } catch (const bf::AccelError &e) {   // a failed call of the C-ABI: reported, not fatal to a host application
        std::fprintf(stderr, "bf_motion_compensator: %s\n", e.what());
        bf::DeviceContext::release();
        return 2;
    }
}
